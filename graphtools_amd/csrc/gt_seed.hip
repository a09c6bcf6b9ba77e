// Threshold seeding of the symmetric candidate pass (launch A of gt_sym.hip) as DENSE cell blocks.
//
// What the launch has to deliver (reference semantics: graphtools/graphs.py:881-911 - kNN, then bandwidth = distance to
// the knn-th neighbour, then the radius): for every row `need` DISTINCT rows that are close to it.  sym_thresholds_kernel
// evaluates them exactly (float64); the largest of their distances bounds the need-th neighbour's, which fixes the row's
// threshold for the collect launch.  Any `need` distinct rows are correct - the closer they are, the shorter the
// candidate lists of launch B.
//
// The points are in cell-sorted order, so the neighbours of a row are in the tiles of the few cells around its own
// (sym_schedule_kernel lists them per query block, plus a strided sample of the rest).  The streaming-selection kernel
// (knn_select_kernel<MODE 0>) kept candidate lists in global memory and spent its time maintaining them (13.5 % of the
// MFMA peak, 35 scalar + vector instructions per MFMA).  Here nothing leaves the registers until the end:
//
//   * one workgroup = 4 waves = 128 query rows (one 32-row query tile per wave, the B operand of a 32x32x16 f16 MFMA,
//     resident in registers); the tiles of the block's list stream through LDS (global_load_lds, double buffered);
//   * the 32 x 32 score block lands with one query per lane and 16 database rows per lane; every lane keeps, for each of
//     its 16 accumulator slots and each of the NSUB sub-tiles of a tile, the BEST score seen in that slot:
//     64 (DP <= 64) keys per lane, 128 per query (two lanes share a query).  The low 10 mantissa bits of a score give
//     way to the position of the tile in the walk, so a key is ONE register and the update is v_and_or_b32 + v_max_f32
//     per score - 32 vector instructions per 4 MFMAs, no branch, no LDS, no memory;
//   * a slot sees a fixed residue class of the rows, so the 128 keys are 128 distinct rows.  The `need` best of them are
//     found by a bitwise search per lane pair and written out (score, position) - the only global stores of the kernel.
//     Two of the true `need` nearest rows that share a slot cost one of them its place (the expected loss is
//     need^2 / 256 rows: the 16 kept rows are the 16 best of ~17) - which only moves the threshold, never the result.
//
// Roofline: MFMA-bound; executed flop = 2 * DP * 128 * BN per (workgroup, tile).
#include "gt_common.h"
#include "gt_device.h"
#include "gt_knn.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int DP>
struct SeedCfg {
    static constexpr int BQ = 128;                      // query rows per workgroup
    static constexpr int BN = (DP <= 64) ? 128 : 64;    // database rows per LDS tile (as gt_select_bn)
    static constexpr int NSUB = BN / 32;
    static constexpr int NS = DP / 16;                  // MFMA k-steps
    static constexpr int RW = DP / 2;                   // row width in dwords (float16 hi plane)
    static constexpr int RB = 2 * DP;                   // row bytes
    static constexpr bool GLDS = (RB & (RB - 1)) == 0 && (BN * RB) % 4096 == 0;
    static constexpr int LDP = GLDS ? RW : RW + 4;      // LDS row stride (dwords); padded rows: conflict-free ds_read_b128
    static constexpr int TILE_FLOATS = BN * LDP;
    static constexpr size_t LDS_BYTES = size_t(2) * TILE_FLOATS * 4 + size_t(2) * BN * 4;
    // swizzle geometry of the direct global -> LDS copy (same scheme as the candidate kernels, gt_knn_select.hip)
    static constexpr int CPR = RB / 16;
    static constexpr int RDIV = (RB >= 256) ? 1 : 256 / RB;
    static constexpr int SMASK = (CPR < 16 ? CPR : 16) - 1;
    static constexpr int RPP = 1024 / RB > 0 ? 1024 / RB : 1;
    static constexpr int NPW = (BN * RB / 1024) / 4;
    static constexpr int NF4 = BN * RW / 4;             // 16-byte units per tile (register staging)
    static constexpr int F4_PER_THREAD = (NF4 + 255) / 256;
};

constexpr uint32_t kWalkBits = 10;                       // tiles per walk: at most 1024
constexpr uint32_t kWalkMask = (1u << kWalkBits) - 1u;

template <int DP>
__global__ __launch_bounds__(256, 3) void sym_seed_dense_kernel(
    const float* __restrict__ Ys, const float* __restrict__ hs, const int64_t n, const int32_t* __restrict__ tile_list,
    const int32_t* __restrict__ tile_cnt, const int32_t tile_stride, const int32_t list_shift, const int32_t block0,
    const int32_t need, uint64_t* __restrict__ lists, const int32_t lstride, uint32_t* __restrict__ counts) {
    using C = SeedCfg<DP>;
    constexpr int BN = C::BN, NSUB = C::NSUB, NS = C::NS, LDP = C::LDP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);   // [2][BN][LDP]
    float* hn = tile + 2 * C::TILE_FLOATS;              // [2][BN]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, h = lane >> 5;
    // workgroups are dealt to the 8 XCDs round robin: each XCD takes a contiguous eighth of the blocks, whose
    // neighbourhoods overlap (the tiles come out of its L2)
    int64_t bidx = blockIdx.x;
    {
        const int64_t nb = gridDim.x, xcd = bidx & 7, base = nb >> 3, rem = nb & 7;
        bidx = xcd * base + (xcd < rem ? xcd : rem) + (bidx >> 3) + block0;
    }
    const int64_t qg = bidx * C::BQ + w * 32 + li;       // this lane's query (sorted position)
    const int64_t qc = qg < n ? qg : n - 1;              // pad queries ride along on a real row, nothing is written for them
    const int32_t* tl = tile_list + size_t(bidx >> list_shift) * size_t(tile_stride);
    int T = tile_cnt[bidx >> list_shift];
    if (T > int(kWalkMask) + 1) T = int(kWalkMask) + 1;

    // query fragments (B operand): lane (li, h) holds features [16 s + 8 h, +8) of its row per k-step s
    f16x8 bq[NS];
    {
        const f16x8* p = reinterpret_cast<const f16x8*>(Ys + qc * C::RW);
#pragma unroll
        for (int s = 0; s < NS; ++s) bq[s] = p[2 * s + h];
    }
    // best score per (sub-tile, accumulator slot), tile position in the low mantissa bits
    float key[NSUB][16];
#pragma unroll
    for (int g = 0; g < NSUB; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) key[g][r] = -INFINITY;

    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    auto glds_issue = [&](const int t_, const int buf_) {
        const char* gt_ = reinterpret_cast<const char*>(Ys) + size_t(t_) * BN * C::RB;
        char* lt_ = reinterpret_cast<char*>(tile + buf_ * C::TILE_FLOATS);
        const uint32_t lv_ = uint32_t(lane);
#pragma unroll
        for (int i_ = 0; i_ < C::NPW; ++i_) {
            const uint32_t p_ = uint32_t(wu * C::NPW + i_);
            const uint32_t r_ = p_ * C::RPP + lv_ / C::CPR;
            const uint32_t c_ = (lv_ % C::CPR) ^ ((r_ / C::RDIV) & C::SMASK);
            __builtin_amdgcn_global_load_lds((glb_void*)(gt_ + (r_ * C::RB + c_ * 16u)), (lds_void*)(lt_ + p_ * 1024u), 16, 0, 0);
        }
        if (wu < BN / 64)
            __builtin_amdgcn_global_load_lds((glb_void*)(hs + size_t(t_) * BN + uint32_t(wu * 64) + lv_),
                                             (lds_void*)(hn + buf_ * BN + wu * 64), 4, 0, 0);
    };
    float4 stage[C::GLDS ? 1 : C::F4_PER_THREAD];
    float stage_h = 0.f;
    auto stage_load = [&](const int t_) {
        const float4* src_ = reinterpret_cast<const float4*>(Ys + size_t(t_) * BN * C::RW);
#pragma unroll
        for (int u_ = 0; u_ < C::F4_PER_THREAD; ++u_) {
            const int f = tid + u_ * 256;
            stage[C::GLDS ? 0 : u_] = (f < C::NF4) ? src_[f] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        stage_h = (tid < BN) ? hs[size_t(t_) * BN + tid] : 0.f;
    };
    auto stage_store = [&](const int buf_) {
        float* tb_ = tile + buf_ * C::TILE_FLOATS;
#pragma unroll
        for (int u_ = 0; u_ < C::F4_PER_THREAD; ++u_) {
            const int f = tid + u_ * 256;
            if (f < C::NF4) {
                const int r = (f * 4) / C::RW, c = (f * 4) % C::RW;
                *reinterpret_cast<float4*>(tb_ + r * LDP + c) = stage[C::GLDS ? 0 : u_];
            }
        }
        if (tid < BN) hn[buf_ * BN + tid] = stage_h;
    };

    // the tile list, 64 entries at a time (one per lane), entries come out with a readlane
    int32_t tl_cache = tl[lane < T ? lane : 0];
    int t = __builtin_amdgcn_readlane(tl_cache, 0);
    if constexpr (C::GLDS) {
        glds_issue(t, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        stage_load(t);
        stage_store(0);
    }
    __syncthreads();
    const int aswz = C::GLDS ? ((li / C::RDIV) & C::SMASK) : 0;   // sub-tiles start at multiples of 32 rows

    for (int it = 0; it < T; ++it) {
        const int buf = it & 1;
        int t_next = t;
        if (it + 1 < T) {
            const int nx = it + 1;
            if ((nx & 63) == 0) tl_cache = tl[nx + lane < T ? nx + lane : nx];
            t_next = __builtin_amdgcn_readlane(tl_cache, nx & 63);
            if constexpr (C::GLDS) glds_issue(t_next, buf ^ 1);   // every wave left buf^1 at the barrier that ended the previous tile
            else stage_load(t_next);
        }
        const float* tb = tile + buf * C::TILE_FLOATS;
        const float* hb = hn + buf * BN;
        const uint32_t itag = uint32_t(it);
        // pad rows (behind the last real row, seeds -inf) score -inf: with the tag in its mantissa that would be a NaN
        // pattern - tiles that hold pad rows clamp their scores first (wave-uniform, the last tile of the order only)
        const bool clamp = int64_t(t) * BN + BN > n;
        auto unit_loop = [&](auto clamp_c) {
            constexpr bool CL = decltype(clamp_c)::value;
            f16x8 afr[2][NS];
            f32x16 acc[2];
            auto load_a = [&](const int sb, f16x8 (&a)[NS]) {
                const f16x8* p = reinterpret_cast<const f16x8*>(tb + (sb * 32 + li) * LDP);
#pragma unroll
                for (int s = 0; s < NS; ++s) a[s] = p[(2 * s + h) ^ aswz];
            };
            auto seed = [&](const int sb, f32x16& a) {
#pragma unroll
                for (int g_ = 0; g_ < 4; ++g_) {
                    const float4 hv = *reinterpret_cast<const float4*>(hb + sb * 32 + 8 * g_ + 4 * h);
                    a[4 * g_ + 0] = hv.x;
                    a[4 * g_ + 1] = hv.y;
                    a[4 * g_ + 2] = hv.z;
                    a[4 * g_ + 3] = hv.w;
                }
            };
            load_a(0, afr[0]);
            seed(0, acc[0]);
#pragma unroll
            for (int u = 0; u <= NSUB; ++u) {
                if (u < NSUB) {
                    if (u + 1 < NSUB) load_a(u + 1, afr[(u + 1) & 1]);
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[u & 1][s], bq[s], acc[u & 1], 0, 0, 0);
                }
                if (u > 0) {
                    const f32x16& pa = acc[(u - 1) & 1];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = pa[r];
                        if (CL) v = fmaxf(v, -3.0e38f);
                        const uint32_t tagged = (__float_as_uint(v) & ~kWalkMask) | itag;
                        key[u - 1][r] = fmaxf(key[u - 1][r], __uint_as_float(tagged));
                    }
                    if (u + 1 < NSUB) seed(u + 1, acc[(u + 1) & 1]);   // (its last reader was the selection above)
                }
                if (u == 0 && NSUB > 1) seed(1, acc[1]);
                if (u > 0 && u < NSUB) {
                    // the selection of block u-1 in the issue gaps of block u's MFMAs
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        if (i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);   // DS reads
                        __builtin_amdgcn_sched_group_barrier(0x002, (32 + NS - 1) / NS, 0);   // VALU
                    }
                }
            }
        };
        if (__builtin_expect(clamp, 0)) unit_loop(std::true_type{});
        else unit_loop(std::false_type{});
        if constexpr (C::GLDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next tile are in LDS
        } else {
            if (it + 1 < T) stage_store(buf ^ 1);
        }
        __syncthreads();
        t = t_next;
    }

    // ---- the `need` best of the 128 (64) keys of every query: bitwise search per lane pair (li, li + 32) ----
    // ord = order-preserving unsigned image of the key; unseen slots (-inf) -> 0, below every real key (>= 0x00800000)
    uint32_t ord[NSUB][16];
#pragma unroll
    for (int g = 0; g < NSUB; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) ord[g][r] = (key[g][r] == -INFINITY) ? 0u : f32_ord(key[g][r]);
    uint32_t Tp = 0u;                         // prefix (shifted into place) of the need-th largest key
#pragma unroll 1
    for (int b = 31; b >= int(kWalkBits); --b) {
        const uint32_t trial = Tp | (1u << b);
        uint32_t c = 0;
#pragma unroll
        for (int g = 0; g < NSUB; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) c += (ord[g][r] >= trial) ? 1u : 0u;
        c += uint32_t(__shfl_xor(int(c), 32));
        Tp = (c >= uint32_t(need)) ? trial : Tp;
    }
    // entries above the prefix all go out, entries on it up to the quota (lane h = 0 first), unseen slots never
    uint32_t n_gt = 0, n_eq = 0;
#pragma unroll
    for (int g = 0; g < NSUB; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t pfx = ord[g][r] & ~kWalkMask;
            n_gt += (pfx > Tp) ? 1u : 0u;
            n_eq += (pfx == Tp && ord[g][r] != 0u) ? 1u : 0u;
        }
    const uint32_t o_gt = uint32_t(__shfl_xor(int(n_gt), 32)), o_eq = uint32_t(__shfl_xor(int(n_eq), 32));
    const uint32_t tot_gt = n_gt + o_gt;                        // < need by construction of the prefix
    const uint32_t quota = tot_gt < uint32_t(need) ? uint32_t(need) - tot_gt : 0u;
    const uint32_t eq0 = h ? o_eq : n_eq, eq1 = h ? n_eq : o_eq;   // ties held by the h = 0 / h = 1 lane of the pair
    const uint32_t take0 = eq0 < quota ? eq0 : quota;              // ... and how many of them go out
    const uint32_t take1 = eq1 < quota - take0 ? eq1 : quota - take0;
    uint32_t slot_gt = h ? o_gt : 0u;                              // the h = 0 lane writes its entries first
    uint32_t slot_eq = tot_gt + (h ? take0 : 0u);
    uint32_t eq_left = h ? take1 : take0;
    if (qg < n) {
        uint64_t* lp = lists + size_t(qg) * size_t(lstride);
#pragma unroll
        for (int g = 0; g < NSUB; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t o = ord[g][r];
                const uint32_t pfx = o & ~kWalkMask;
                const bool gt = pfx > Tp;
                const bool eq = pfx == Tp && o != 0u && eq_left > 0u;
                if (gt || eq) {
                    const uint32_t bits = __float_as_uint(ord_f32(o));
                    const uint32_t itw = bits & kWalkMask;
                    const uint32_t pos = uint32_t(tl[itw]) * uint32_t(BN) + uint32_t(g * 32 + 8 * (r >> 2) + 4 * h + (r & 3));
                    const uint32_t slot = gt ? slot_gt : slot_eq;
                    if (slot < uint32_t(lstride)) lp[slot] = cand_pack(__uint_as_float(bits & ~kWalkMask), pos);
                    slot_gt += gt ? 1u : 0u;
                    slot_eq += gt ? 0u : 1u;
                    eq_left -= gt ? 0u : 1u;
                }
            }
        const uint32_t kept = tot_gt + take0 + take1;
        if (h == 0) counts[qg] = kept < uint32_t(lstride) ? kept : uint32_t(lstride);
    }
}

template <int DP>
int launch_seed(gt_ctx* ctx, const float* Ys, const float* hs, int64_t n, int64_t n_pad, const int32_t* tile_list,
                const int32_t* tile_cnt, int tile_stride, int list_shift, int64_t block0, int64_t nblk, int need,
                uint64_t* lists, int lstride, uint32_t* counts) {
    using C = SeedCfg<DP>;
    auto kern = sym_seed_dense_kernel<DP>;
    GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    int(C::LDS_BYTES)));
    const int64_t grid = nblk > 0 ? nblk : n_pad / C::BQ;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::LDS_BYTES, ctx->stream, Ys, hs, n, tile_list, tile_cnt,
                       tile_stride, list_shift, int32_t(block0), need, lists, lstride, counts);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

}  // namespace

// Rows [block0 * 128, (block0 + nblk) * 128) of the sorted order (nblk = 0: all of n_pad) against the tile lists of
// gt_sym_schedule (made for blocks of 128 << list_shift rows, tiles of gt_select_bn(dp) rows).  lists [n_pad][lstride]
// keys (score, sorted position), counts [n_pad]: `need` entries per real row (fewer only when the walk saw fewer rows).
int gt_sym_seed_dense(gt_ctx* ctx, int dp, const void* Ys, const float* hs, int64_t n, int64_t n_pad, const int32_t* tile_list,
                      const int32_t* tile_cnt, int tile_stride, int list_shift, int64_t block0, int64_t nblk, int need,
                      uint64_t* lists, int lstride, uint32_t* counts) {
    if (need < 1 || need > 64 || lstride < need || n_pad % 128 != 0)
        GT_FAIL(ctx, GT_E_ARG, "gt_sym_seed_dense: 1 <= need <= 64 rows per point, whole 128-row blocks");
    const float* Y = static_cast<const float*>(Ys);
#define GT_SEED_CASE(DP_)                                                                                                  \
    case DP_:                                                                                                              \
        return launch_seed<DP_>(ctx, Y, hs, n, n_pad, tile_list, tile_cnt, tile_stride, list_shift, block0, nblk, need, lists, \
                                lstride, counts);
    switch (dp) {
        GT_SEED_CASE(16)
        GT_SEED_CASE(32)
        GT_SEED_CASE(48)
        GT_SEED_CASE(64)
        GT_SEED_CASE(80)
        GT_SEED_CASE(96)
        GT_SEED_CASE(112)
        GT_SEED_CASE(128)
    }
#undef GT_SEED_CASE
    GT_FAIL(ctx, GT_E_ARG, "gt_sym_seed_dense: unsupported padded feature count");
}
