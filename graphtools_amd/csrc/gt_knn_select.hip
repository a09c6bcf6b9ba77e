// Candidate generation for brute-force kNN / radius search on gfx950.
//
// One workgroup (4 waves) owns BQ = 128*QT query rows and streams the whole padded float32 database
// through LDS in tiles of BN rows.  Each wave keeps its 32*QT query rows as the B operand of
// v_mfma_f32_32x32x2_f32 in registers for the whole kernel; the database tile is the A operand, so the
// 32x32 result block lands with ONE QUERY PER LANE (column = lane & 31) and 16 database rows per lane
// (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).  The accumulator is pre-loaded with -|y|^2/2, so after the
// K loop it holds the score  s = x.y - |y|^2/2  (d^2 = |x|^2 - 2 s: larger score = closer), and the
// per-query running threshold is a single VGPR compare per lane.
//
// MODE 0 (top-M' selection): survivors (s > thr) are appended to the query's candidate list in global
//   memory (slot from an LDS counter; the wave owns its queries, so no cross-wave traffic).  When a list
//   passes its trigger level the owning wave sorts it in registers (bitonic, 64*NT keys), keeps the best
//   M' = 16*NT and raises thr to the M'-th score.  At the end every list is sorted; list[0..M') are the
//   candidates in descending score order.  Exact ordering is established afterwards in fp64 (gt_rerank.hip);
//   the scores here only have to be within a known error bound of the true ones.
// MODE 1 (radius collect): thr is a fixed per-query score bound; every survivor is appended (up to `cap`
//   entries per query, the true count is always reported).
//
// Roofline: MFMA-bound (fp32 matrix, 157.3 TF peak): 2*DP flop per (query, database row) pair.
#include "gt_common.h"
#include "gt_device.h"
#include "gt_knn_select.h"

namespace {

template <int DP>
struct SelCfg {
    static constexpr int KS = DP / 2;                   // k-steps (2 k per MFMA), per-lane fragment length
    static constexpr int QT = (DP <= 64) ? 2 : 1;       // 32-row query tiles per wave
    static constexpr int BQ = 4 * QT * 32;              // query rows per workgroup
    static constexpr int BN = (DP <= 64) ? 128 : 64;    // database rows per LDS tile
    static constexpr int LDP = DP + 4;                  // LDS row stride (floats): conflict-free ds_read_b128
    static constexpr int NF4 = BN * DP / 4;             // float4 per tile
    static constexpr int F4_PER_THREAD = (NF4 + 255) / 256;
    static constexpr int TILE_FLOATS = BN * LDP;
    static constexpr size_t LDS_BYTES =
        size_t(2) * TILE_FLOATS * 4 + size_t(2) * BN * 4 + size_t(BQ) * 4 * 2;
};

template <int DP, int NT, int MODE>
__global__ __launch_bounds__(256, 2) void knn_select_kernel(
    const float* __restrict__ Yp, const float* __restrict__ hneg, const float* __restrict__ Qp,
    const int32_t* __restrict__ qrows, const int64_t q0, const int32_t nq, const int32_t ntiles,
    uint64_t* __restrict__ lists, uint32_t* __restrict__ counts, const float* __restrict__ thr_in,
    const int32_t cap) {
    using C = SelCfg<DP>;
    constexpr int KS = C::KS, QT = C::QT, BQ = C::BQ, BN = C::BN, LDP = C::LDP;
    constexpr int LCAP = 64 * NT;        // list capacity in selection mode
    constexpr int MKEEP = 16 * NT;       // M'
    constexpr int TRIG = LCAP - BN;      // compaction trigger (a tile can add at most BN entries per query)
    static_assert(TRIG >= MKEEP, "list too small for the tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);                    // [2][BN][LDP]
    float* hn = tile + 2 * C::TILE_FLOATS;                               // [2][BN]
    uint32_t* cnt = reinterpret_cast<uint32_t*>(hn + 2 * BN);            // [BQ]
    float* thr_lds = reinterpret_cast<float*>(cnt + BQ);                 // [BQ]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int li = lane & 31;
    const int h = lane >> 5;
    const int64_t qblock = int64_t(blockIdx.x) * BQ;
    const size_t lstride = (MODE == 0) ? size_t(LCAP) : size_t(cap);

    // ---- query fragments: lane (li, h) holds query (w*QT + qt)*32 + li, features [h*KS, h*KS + KS) ----
    float bq[QT][KS];
    float thr[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int ql = (w * QT + qt) * 32 + li;
        const int64_t qg = qblock + ql;
        const int64_t qc = qg < nq ? qg : int64_t(nq) - 1;   // clamp pad queries onto a real row
        const int64_t row = qrows ? int64_t(qrows[qc]) : q0 + qc;
        const float4* src = reinterpret_cast<const float4*>(Qp + row * DP + h * KS);
#pragma unroll
        for (int c = 0; c < KS / 4; ++c) {
            const float4 v = src[c];
            bq[qt][4 * c + 0] = v.x;
            bq[qt][4 * c + 1] = v.y;
            bq[qt][4 * c + 2] = v.z;
            bq[qt][4 * c + 3] = v.w;
        }
        thr[qt] = (MODE == 0) ? -INFINITY : thr_in[qc];
    }
    if (tid < BQ) {
        cnt[tid] = 0u;
        thr_lds[tid] = -INFINITY;
    }

    // ---- tile staging (global -> registers -> LDS, padded rows) ----
    float4 stage[C::F4_PER_THREAD];
    float stage_h = 0.f;
#define GT_STAGE_LOAD(T_)                                                                     \
    {                                                                                         \
        const float4* src_ = reinterpret_cast<const float4*>(Yp + size_t(T_) * BN * DP);      \
        _Pragma("unroll") for (int u = 0; u < C::F4_PER_THREAD; ++u) {                        \
            const int f = tid + u * 256;                                                      \
            stage[u] = (f < C::NF4) ? src_[f] : make_float4(0.f, 0.f, 0.f, 0.f);              \
        }                                                                                     \
        stage_h = (tid < BN) ? hneg[size_t(T_) * BN + tid] : 0.f;                             \
    }
#define GT_STAGE_STORE(BUF_)                                                                  \
    {                                                                                         \
        float* tb_ = tile + (BUF_) * C::TILE_FLOATS;                                          \
        _Pragma("unroll") for (int u = 0; u < C::F4_PER_THREAD; ++u) {                        \
            const int f = tid + u * 256;                                                      \
            if (f < C::NF4) {                                                                 \
                const int r = (f * 4) / DP;                                                   \
                const int c = (f * 4) % DP;                                                   \
                *reinterpret_cast<float4*>(tb_ + r * LDP + c) = stage[u];                     \
            }                                                                                 \
        }                                                                                     \
        if (tid < BN) hn[(BUF_) * BN + tid] = stage_h;                                        \
    }

    GT_STAGE_LOAD(0);
    GT_STAGE_STORE(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) GT_STAGE_LOAD(t + 1);
        const float* tb = tile + buf * C::TILE_FLOATS;
        const float* hb = hn + buf * BN;
        const uint32_t tbase = uint32_t(t) * BN;

#pragma unroll 1
        for (int sb = 0; sb < BN / 32; ++sb) {
            // A fragment: database row sb*32 + li, features [h*KS, h*KS+KS)
            float a[KS];
            const float4* ap = reinterpret_cast<const float4*>(tb + (sb * 32 + li) * LDP + h * KS);
#pragma unroll
            for (int c = 0; c < KS / 4; ++c) {
                const float4 v = ap[c];
                a[4 * c + 0] = v.x;
                a[4 * c + 1] = v.y;
                a[4 * c + 2] = v.z;
                a[4 * c + 3] = v.w;
            }
            // accumulator init: acc[4g+e] <-> database row 8g + 4h + e  => -|y|^2/2 of that row
            f32x16 acc[QT];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 hv = *reinterpret_cast<const float4*>(hb + sb * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    acc[qt][4 * g + 0] = hv.x;
                    acc[qt][4 * g + 1] = hv.y;
                    acc[qt][4 * g + 2] = hv.z;
                    acc[qt][4 * g + 3] = hv.w;
                }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    acc[qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bq[qt][s], acc[qt], 0, 0, 0);
            }
            // ---- epilogue: one query per lane, 16 database rows in registers ----
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float m = acc[qt][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[qt][r]);
                if (m > thr[qt]) {
                    const int ql = (w * QT + qt) * 32 + li;
                    uint64_t* lp = lists + size_t(qblock + ql) * lstride;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[qt][r];
                        if (v > thr[qt]) {
                            const uint32_t slot = atomicAdd(&cnt[ql], 1u);
                            const uint32_t j = tbase + uint32_t(sb * 32 + 8 * (r >> 2) + 4 * h + (r & 3));
                            if (MODE == 0 || slot < uint32_t(cap)) st_agent_u64(lp + slot, cand_pack(v, j));
                        }
                    }
                }
            }
        }

        if (MODE == 0) {
            // ---- list maintenance: lane L <-> query w*QT*32 + L of this wave ----
            const uint32_t c = (lane < QT * 32) ? cnt[w * QT * 32 + lane] : 0u;
            unsigned long long need = __ballot(c > uint32_t(TRIG));
            if (need) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's list stores have reached L2
                while (need) {
                    const int L = __ffsll((long long)need) - 1;
                    need &= need - 1;
                    const int ql = w * QT * 32 + L;
                    const uint32_t n = cnt[ql];
                    uint64_t* lp = lists + size_t(qblock + ql) * lstride;
                    uint64_t key[NT];
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        const uint32_t e = uint32_t(u * 64 + lane);
                        key[u] = e < n ? ld_agent_u64(lp + e) : 0ull;
                    }
                    wave_bitonic_desc<NT>(key, lane);
                    // survivors: positions [0, MKEEP) = registers [0, NT/4)
#pragma unroll
                    for (int u = 0; u < NT / 4; ++u) st_agent_u64(lp + u * 64 + lane, key[u]);
                    const uint64_t last = __shfl((unsigned long long)key[NT / 4 - 1], 63);
                    if (lane == 0) {
                        cnt[ql] = n < uint32_t(MKEEP) ? n : uint32_t(MKEEP);
                        thr_lds[ql] = n >= uint32_t(MKEEP) ? cand_score(last) : -INFINITY;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) thr[qt] = thr_lds[(w * QT + qt) * 32 + li];
            }
        }

        if (t + 1 < ntiles) GT_STAGE_STORE(buf ^ 1);
        __syncthreads();
    }

    // ---- finalisation ----
    if (MODE == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int L = 0; L < QT * 32; ++L) {
            const int ql = w * QT * 32 + L;
            const uint32_t n = cnt[ql];
            uint64_t* lp = lists + size_t(qblock + ql) * lstride;
            uint64_t key[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const uint32_t e = uint32_t(u * 64 + lane);
                key[u] = e < n ? ld_agent_u64(lp + e) : 0ull;
            }
            wave_bitonic_desc<NT>(key, lane);
#pragma unroll
            for (int u = 0; u < NT / 4; ++u) st_agent_u64(lp + u * 64 + lane, key[u]);
            if (lane == 0) counts[qblock + ql] = n;   // >= MKEEP means "list was truncated to MKEEP"
        }
    } else {
        if (tid < BQ) counts[qblock + tid] = cnt[tid];
    }
}

template <int DP, int NT, int MODE>
int launch_one(gt_ctx* ctx, const SelectArgs& a) {
    using C = SelCfg<DP>;
    const int64_t nblocks = ceil_div64(a.nq, C::BQ);
    auto kern = knn_select_kernel<DP, NT, MODE>;
    GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    int(C::LDS_BYTES)));
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), C::LDS_BYTES, ctx->stream, a.Yp, a.hneg, a.Qp,
                       a.qrows, a.q0, a.nq, int32_t(a.n_pad / C::BN), a.lists, a.counts, a.thr_in, a.cap);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

template <int DP>
int launch_dp(gt_ctx* ctx, const SelectArgs& a) {
    if (a.mode == 1) return launch_one<DP, 8, 1>(ctx, a);
    switch (a.nt) {
        case 8: return launch_one<DP, 8, 0>(ctx, a);
        case 32: return launch_one<DP, 32, 0>(ctx, a);
    }
    GT_FAIL(ctx, GT_E_ARG, "knn_select: unsupported list size");
}

}  // namespace

// This file is compiled once per padded feature count (-DGT_SEL_DP=<dp>) so the instantiations build in
// parallel; gt_knn_select_dispatch.cpp routes to the right one.
#ifndef GT_SEL_DP
#error "compile with -DGT_SEL_DP=<dp>"
#endif
#define GT_CAT2(a, b) a##b
#define GT_CAT(a, b) GT_CAT2(a, b)
int GT_CAT(gt_launch_select_dp, GT_SEL_DP)(gt_ctx* ctx, const SelectArgs& a) { return launch_dp<GT_SEL_DP>(ctx, a); }
