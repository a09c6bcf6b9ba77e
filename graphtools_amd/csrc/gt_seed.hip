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

#ifndef GT_SEED_QT
#define GT_SEED_QT 1   // 2: two query tiles per wave for DP <= 64 (needs ~290 registers: spills)
#endif
#ifndef GT_SEED_NBUF
#define GT_SEED_NBUF 3   // tile buffers of the direct copy (NBUF - 1 tiles in flight)
#endif
#ifndef GT_SEED_WAVES
#define GT_SEED_WAVES 3  // waves per SIMD the register budget is cut for (166 VGPRs at DP = 64 with one fragment set)
#endif
#ifndef GT_SEED_AFR
#define GT_SEED_AFR 1    // fragment sets (2: the next sub-tile's fragments are read while this one's chains run)
#endif
#ifndef GT_SEED_SGB
#define GT_SEED_SGB 1   // schedule hints: the key updates of a block between the MFMAs of the next one
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int DP>
struct SeedCfg {
    static constexpr int BN = (DP <= 64) ? 128 : 64;    // database rows per LDS tile (as gt_select_bn)
    static constexpr int NSUB = BN / 32;
    static constexpr int NS = DP / 16;                  // MFMA k-steps
    static constexpr int RW = DP / 2;                   // row width in dwords (float16 hi plane)
    static constexpr int RB = 2 * DP;                   // row bytes
    static constexpr bool GLDS = (RB & (RB - 1)) == 0 && (BN * RB) % 4096 == 0;
    static constexpr int LDP = GLDS ? RW : RW + 4;      // LDS row stride (dwords); padded rows: conflict-free ds_read_b128
    static constexpr int TILE_FLOATS = BN * LDP;
    // 32-row query tiles per wave.  One tile per wave reads as many LDS bytes as its MFMAs can absorb (per 32 x 32 block:
    // 4 KiB of fragments + the seeds = 32 LDS cycles per wave, 8 waves per CU, against 4 MFMAs = 128 cycles per SIMD:
    // the LDS pipe is as busy as the matrix pipes, 0.65 PF at N = 1e6) - with two, a fragment feeds two chains
    static constexpr int QT = (GT_SEED_QT == 2 && DP <= 64) ? 2 : 1;
    static constexpr int G = (QT == 2 && NSUB >= 2) ? 2 : NSUB;   // key groups per query tile (sub-tile sb -> group sb % G)
    static constexpr int NW = 4;                        // waves per workgroup
    static constexpr int BQ = 32 * QT * NW;             // query rows per workgroup
    // tile buffers: the direct copy keeps NBUF - 1 tiles in flight (a tile's scores take less time than its trip from
    // the L2 / Infinity Cache: with one tile ahead the launch was bound by that latency - 4.1 ms at N = 1e6)
    static constexpr int NBUF = GLDS ? GT_SEED_NBUF : 2;
    static constexpr size_t LDS_BYTES = size_t(NBUF) * TILE_FLOATS * 4 + size_t(NBUF) * BN * 4;
    // swizzle geometry of the direct global -> LDS copy (same scheme as the candidate kernels, gt_knn_select.hip)
    static constexpr int CPR = RB / 16;
    static constexpr int RDIV = (RB >= 256) ? 1 : 256 / RB;
    static constexpr int SMASK = (CPR < 16 ? CPR : 16) - 1;
    static constexpr int RPP = 1024 / RB > 0 ? 1024 / RB : 1;
    static constexpr int NPW = (BN * RB / 1024) / NW > 0 ? (BN * RB / 1024) / NW : 1;   // 1 KiB pieces per wave and tile
    static constexpr int NF4 = BN * RW / 4;             // 16-byte units per tile (register staging)
    static constexpr int F4_PER_THREAD = (NF4 + 32 * NW * 2 - 1) / (64 * NW);
};

constexpr uint32_t kWalkBits = 10;                       // tiles per walk: at most 1024
constexpr uint32_t kWalkMask = (1u << kWalkBits) - 1u;

template <int DP>
__global__ __launch_bounds__(64 * SeedCfg<DP>::NW, GT_SEED_WAVES) void sym_seed_dense_kernel(
    const float* __restrict__ Ys, const float* __restrict__ hs, const int64_t n, const int32_t* __restrict__ tile_list,
    const int32_t* __restrict__ tile_cnt, const int32_t tile_stride, const int32_t list_shift, const int32_t block0,
    const int32_t need, uint64_t* __restrict__ lists, const int32_t lstride, uint32_t* __restrict__ counts) {
    using C = SeedCfg<DP>;
    constexpr int BN = C::BN, NSUB = C::NSUB, NS = C::NS, LDP = C::LDP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NBUF = C::NBUF;
    float* tile = reinterpret_cast<float*>(smem_raw);   // [NBUF][BN][LDP]
    float* hn = tile + NBUF * C::TILE_FLOATS;           // [NBUF][BN]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, h = lane >> 5;
    // workgroups are dealt to the 8 XCDs round robin: each XCD takes a contiguous eighth of the blocks, whose
    // neighbourhoods overlap (the tiles come out of its L2)
    int64_t bidx = blockIdx.x;
    {
        const int64_t nb = gridDim.x, xcd = bidx & 7, base = nb >> 3, rem = nb & 7;
        bidx = xcd * base + (xcd < rem ? xcd : rem) + (bidx >> 3) + block0;
    }
    constexpr int QT = C::QT, G = C::G;
    const int32_t* tl = tile_list + size_t(bidx >> list_shift) * size_t(tile_stride);
    int T = tile_cnt[bidx >> list_shift];
    if (T > (int(kWalkMask) + 1) / (C::NSUB / G)) T = (int(kWalkMask) + 1) / (C::NSUB / G);   // (the tag field; dropping tiles is safe)

    // query fragments (B operand): lane (li, h) holds features [16 s + 8 h, +8) of its row per k-step s
    // (query tile qt of wave w = sorted positions bidx * BQ + (w * QT + qt) * 32 + li; pad queries ride along on a real
    //  row, nothing is written for them)
    f16x8 bq[QT][NS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int64_t qg_ = bidx * C::BQ + (w * QT + qt) * 32 + li;
        const f16x8* p = reinterpret_cast<const f16x8*>(Ys + (qg_ < n ? qg_ : n - 1) * C::RW);
#pragma unroll
        for (int s = 0; s < NS; ++s) bq[qt][s] = p[2 * s + h];
    }
    // The fragments must be in their registers before the first copy is issued: the copies are inline asm (below), the
    // waits for them too, and a load hipcc still believes to be in flight would make it wait - by its own count, which
    // knows nothing of the copies - in front of every MFMA of the loop.  (s_waitcnt vmcnt(0), the other counters open)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // best score per (query tile, group of sub-tiles, accumulator slot), tile position in the low mantissa bits
    float key[QT][G][16];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) key[qt][g][r] = -INFINITY;

    typedef __attribute__((address_space(3))) void lds_void;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    // The copies are issued as inline asm: hipcc orders every LDS read behind ALL LDS-DMA copies it knows of (s_waitcnt
    // vmcnt(0) in front of the first ds_read of an iteration), which would serialise the tiles in flight.  Unknown to the
    // compiler, the copies are ordered by hand: counted vmcnt waits before the barrier that publishes a tile, and the
    // "memory" clobbers keep compiler-generated LDS reads on their side of those barriers.
    auto lds_copy16 = [&](const void* g_, const uint32_t lds_) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g_), "s"(lds_) : "memory");
    };
    auto lds_copy4 = [&](const void* g_, const uint32_t lds_) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g_), "s"(lds_) : "memory");
    };
    const uint32_t lds_tile_base = uint32_t(size_t((lds_void*)tile)), lds_hn_base = uint32_t(size_t((lds_void*)hn));
    auto glds_issue = [&](const int t_, const int buf_) {
        const char* gt_ = reinterpret_cast<const char*>(Ys) + size_t(t_) * BN * C::RB;
        const uint32_t lt_ = lds_tile_base + uint32_t(buf_) * uint32_t(C::TILE_FLOATS * 4);
        const uint32_t lv_ = uint32_t(lane);
#pragma unroll
        for (int i_ = 0; i_ < C::NPW; ++i_) {
            const uint32_t p_ = uint32_t(wu * C::NPW + i_);
            const uint32_t r_ = p_ * C::RPP + lv_ / C::CPR;
            const uint32_t c_ = (lv_ % C::CPR) ^ ((r_ / C::RDIV) & C::SMASK);
            lds_copy16(gt_ + (r_ * C::RB + c_ * 16u), uint32_t(__builtin_amdgcn_readfirstlane(int(lt_ + p_ * 1024u))));
        }
        // seeds: every wave copies BN / NW of them (the same number of loads per wave and tile: the counted waits below)
        if (lv_ < uint32_t(BN / C::NW))
            lds_copy4(hs + size_t(t_) * BN + uint32_t(wu * (BN / C::NW)) + lv_,
                      uint32_t(__builtin_amdgcn_readfirstlane(int(lds_hn_base + uint32_t(buf_ * BN + wu * (BN / C::NW)) * 4u))));
    };
    float4 stage[C::GLDS ? 1 : C::F4_PER_THREAD];
    float stage_h = 0.f;
    auto stage_load = [&](const int t_) {
        const float4* src_ = reinterpret_cast<const float4*>(Ys + size_t(t_) * BN * C::RW);
#pragma unroll
        for (int u_ = 0; u_ < C::F4_PER_THREAD; ++u_) {
            const int f = tid + u_ * (64 * C::NW);
            stage[C::GLDS ? 0 : u_] = (f < C::NF4) ? src_[f] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        stage_h = (tid < BN) ? hs[size_t(t_) * BN + tid] : 0.f;
    };
    auto stage_store = [&](const int buf_) {
        float* tb_ = tile + buf_ * C::TILE_FLOATS;
#pragma unroll
        for (int u_ = 0; u_ < C::F4_PER_THREAD; ++u_) {
            const int f = tid + u_ * (64 * C::NW);
            if (f < C::NF4) {
                const int r = (f * 4) / C::RW, c = (f * 4) % C::RW;
                *reinterpret_cast<float4*>(tb_ + r * LDP + c) = stage[C::GLDS ? 0 : u_];
            }
        }
        if (tid < BN) hn[buf_ * BN + tid] = stage_h;
    };

    constexpr int D = NBUF - 1;          // tiles in flight (direct copy)
    constexpr int LPT = C::NPW + 1;      // loads per wave and tile (direct copy)
    const int aswz = C::GLDS ? ((li / C::RDIV) & C::SMASK) : 0;   // sub-tiles start at multiples of 32 rows
    // ---- one tile: NSUB x QT units of 32 x 32 scores, every score folded into its slot's key ----
    constexpr int NU = NSUB * QT;
    // (the tag = position of the tile in the walk x NSUB / G + which of the group's sub-tiles: slot + tag name the row)
    uint32_t vmask = ~kWalkMask;   // in a VGPR: with the mask as a literal and the tag in an SGPR the update is two instructions
    asm volatile("" : "+v"(vmask));
    auto fold = [&](const f32x16& pa, const int pu, const uint32_t it_, const int r0, const int r1) {
        const int pqt = pu % QT, pg = (pu / QT) % G;
        const uint32_t itag = it_ * uint32_t(NSUB / G) + uint32_t((pu / QT) / G);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r >= r0 && r < r1) {
                const uint32_t tagged = (__float_as_uint(pa[r]) & vmask) | itag;
                key[pqt][pg][r] = fmaxf(key[pqt][pg][r], __uint_as_float(tagged));
            }
    };
    auto compute_tile = [&](const int it, const int buf) {
        const float* tb = tile + buf * C::TILE_FLOATS;
        const float* hb = hn + buf * BN;
        constexpr int AM = GT_SEED_AFR - 1;   // fragment set of sub-tile sb: sb & AM
        f16x8 afr[GT_SEED_AFR][NS];
        f32x16 acc[2];
        auto load_a = [&](const int sb) {
            const f16x8* p = reinterpret_cast<const f16x8*>(tb + (sb * 32 + li) * LDP);
#pragma unroll
            for (int s = 0; s < NS; ++s) afr[sb & AM][s] = p[(2 * s + h) ^ aswz];
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
        // Program order IS issue order (the wave issues in order, and a chain's MFMAs depend on each other): the key
        // updates of unit u-1 are placed between the MFMAs of unit u, behind full scheduling barriers.  (The loops
        // around this body are single basic blocks on purpose: with a branch behind it hipcc sinks all the updates
        // into the loop latch, behind the barrier, where they overlap nothing.)
        constexpr int UPM = (16 + NS - 1) / NS;   // key updates per MFMA
        load_a(0);
        seed(0, acc[0]);
#pragma unroll
        for (int u = 0; u <= NU; ++u) {
            const int sb = u / QT, qt = u % QT;
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                if (u < NU) {
                    acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[sb & AM][s_], bq[qt][s_], acc[u & 1], 0, 0, 0);
                    // the next sub-tile's fragments travel while these chains run and the previous unit is examined (one
                    // fragment set: they can only be fetched behind the last MFMA that reads the current ones)
                    if (s_ == (AM ? 0 : NS - 1) && qt == (AM ? 0 : QT - 1) && sb + 1 < NSUB) load_a(sb + 1);
                }
#if GT_SEED_SGB
                __builtin_amdgcn_sched_barrier(0);
#endif
                if (u > 0) fold(acc[(u - 1) & 1], u - 1, uint32_t(it), s_ * UPM, (s_ + 1) * UPM);
#if GT_SEED_SGB
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            // the seeds of the next unit go into the set the fold above has just finished with
            if (u >= 1 && u + 1 < NU) seed((u + 1) / QT, acc[(u + 1) & 1]);
            if (u == 0 && NU > 1) seed(1 / QT, acc[1]);
        }
    };

    if constexpr (C::GLDS) {
        // (a raw s_barrier everywhere: __syncthreads() would drain the copies in flight - its fence waits for vmcnt(0)
        //  while an LDS-DMA is pending)
        if (T <= D) {
            // short walk: every tile fits the buffers
            for (int k = 0; k < T; ++k) glds_issue(tl[k], k);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int it = 0; it < T; ++it) compute_tile(it, it);
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) glds_issue(tl[k], k);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * LPT) : "memory");   // the first tile has landed
            __builtin_amdgcn_s_barrier();
            int it = 0;
            int t_issue = tl[D];   // the tile whose copy starts in the next iteration
            // main part: a copy is started in every iteration, D - 1 tiles stay in flight across the barrier
            for (; it + D < T; ++it) {
                // buffer (it + D) % NBUF held tile it - 1: every wave left it at the barrier that ended the previous iteration
                glds_issue(t_issue, (it + D) % NBUF);
                t_issue = tl[it + D + 1 < T ? it + D + 1 : T - 1];
                compute_tile(it, it % NBUF);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * LPT) : "memory");   // tile it + 1 has landed
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // the last D tiles are all on their way: one wait, no more barriers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (; it < T; ++it) compute_tile(it, it % NBUF);
        }
    } else {
        stage_load(tl[0]);
        stage_store(0);
        __syncthreads();
        for (int it = 0; it < T; ++it) {
            if (it + 1 < T) stage_load(tl[it + 1]);
            compute_tile(it, it & 1);
            if (it + 1 < T) stage_store((it & 1) ^ 1);
            __syncthreads();
        }
    }

    // ---- the `need` best of the 2 x 16 G keys of every query: bitwise search per lane pair (li, li + 32) ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int64_t qg = bidx * C::BQ + (w * QT + qt) * 32 + li;   // this lane's query (sorted position)
        // ord = order-preserving unsigned image of the key; unseen slots (-inf) -> 0, below every real key (>= 0x00800000)
        uint32_t ord[G][16];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) ord[g][r] = (key[qt][g][r] == -INFINITY) ? 0u : f32_ord(key[qt][g][r]);
        uint32_t Tp = 0u;   // prefix (the bits above the tag, in place) of the need-th largest key
#pragma unroll 1
        for (int b = 31; b >= int(kWalkBits); --b) {
            const uint32_t trial = Tp | (1u << b);
            uint32_t c = 0;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) c += (ord[g][r] >= trial) ? 1u : 0u;
            c += uint32_t(__shfl_xor(int(c), 32));
            Tp = (c >= uint32_t(need)) ? trial : Tp;
        }
        // entries above the prefix all go out, entries on it up to the quota (lane h = 0 first), unseen slots never
        uint32_t n_gt = 0, n_eq = 0;
#pragma unroll
        for (int g = 0; g < G; ++g)
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
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t o = ord[g][r];
                    const uint32_t pfx = o & ~kWalkMask;
                    const bool gt = pfx > Tp;
                    const bool eq = pfx == Tp && o != 0u && eq_left > 0u;
                    if (gt || eq) {
                        // slot (g, r, h) + the walk position in the tag -> the row: sub-tile sb = g (mod G) of tile tl[tag]
                        // is not recoverable from g alone when G < NSUB: the sub-tile index rides in the tag's top bits
                        const uint32_t bits = __float_as_uint(ord_f32(o));
                        const uint32_t tag = bits & kWalkMask;
                        const uint32_t itw = tag / uint32_t(NSUB / G), sbh = tag % uint32_t(NSUB / G);
                        const uint32_t sb = sbh * uint32_t(G) + uint32_t(g);
                        const uint32_t pos = uint32_t(tl[itw]) * uint32_t(BN) + sb * 32u + uint32_t(8 * (r >> 2) + 4 * h + (r & 3));
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
}

template <int DP>
int launch_seed(gt_ctx* ctx, const float* Ys, const float* hs, int64_t n, int64_t n_pad, const int32_t* tile_list,
                const int32_t* tile_cnt, int tile_stride, int list_rows, int64_t block0, int64_t nblk, int need,
                uint64_t* lists, int lstride, uint32_t* counts) {
    using C = SeedCfg<DP>;
    auto kern = sym_seed_dense_kernel<DP>;
    GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    int(C::LDS_BYTES)));
    // block0 / nblk count 128-row blocks; the tile lists are made for blocks of list_rows rows
    if (n_pad % C::BQ != 0 || list_rows % C::BQ != 0 || (block0 * 128) % C::BQ != 0 || (nblk * 128) % C::BQ != 0)
        GT_FAIL(ctx, GT_E_ARG, "gt_sym_seed_dense: the row range must be whole workgroups");
    int list_shift = 0;
    while ((C::BQ << list_shift) < list_rows) ++list_shift;
    const int64_t grid = nblk > 0 ? nblk * 128 / C::BQ : n_pad / C::BQ;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * C::NW), C::LDS_BYTES, ctx->stream, Ys, hs, n, tile_list, tile_cnt,
                       tile_stride, list_shift, int32_t(block0 * 128 / C::BQ), need, lists, lstride, counts);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

}  // namespace

// hs: the seeds with FINITE values on the pad rows (-3e38 instead of -inf: a score of -inf with a tag in its mantissa
// would be a NaN pattern; pad rows then simply score below every real row).
// Rows [block0 * 128, (block0 + nblk) * 128) of the sorted order (nblk = 0: all of n_pad) against the tile lists of
// gt_sym_schedule (made for blocks of list_rows rows, tiles of gt_select_bn(dp) rows).  lists [n_pad][lstride]
// keys (score, sorted position), counts [n_pad]: `need` entries per real row (fewer only when the walk saw fewer rows).
int gt_sym_seed_dense(gt_ctx* ctx, int dp, const void* Ys, const float* hs, int64_t n, int64_t n_pad, const int32_t* tile_list,
                      const int32_t* tile_cnt, int tile_stride, int list_rows, int64_t block0, int64_t nblk, int need,
                      uint64_t* lists, int lstride, uint32_t* counts) {
    if (need < 1 || need > 64 || lstride < need || n_pad % 128 != 0)
        GT_FAIL(ctx, GT_E_ARG, "gt_sym_seed_dense: 1 <= need <= 64 rows per point, whole 128-row blocks");
    const float* Y = static_cast<const float*>(Ys);
#define GT_SEED_CASE(DP_)                                                                                                  \
    case DP_:                                                                                                              \
        return launch_seed<DP_>(ctx, Y, hs, n, n_pad, tile_list, tile_cnt, tile_stride, list_rows, block0, nblk, need, lists, \
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
