// Candidate generation for brute-force kNN / radius search on gfx950.
//
// One workgroup (4 waves) owns BQ = 128*QT query rows and streams the padded working copy of the database
// through LDS in tiles of BN rows.  Each wave keeps its 32*QT query rows as the B operand of a 32x32 MFMA in
// registers for the whole kernel; the database tile is the A operand, so the 32x32 result block lands with
// ONE QUERY PER LANE (column = lane & 31) and 16 database rows per lane
// (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).  The accumulator is pre-loaded with -|y|^2/2, so after the K
// loop it holds the score  s = x.y - |y|^2/2  (d^2 = |x|^2 - 2 s: larger score = closer), and the per-query
// running threshold is a single VGPR compare per lane.
//
// Three arithmetic back ends produce the scores (PREC):
//   PREC 0  float32 operands, v_mfma_f32_32x32x2_f32 (157 TF peak): exact products, fp32 accumulation.
//   PREC 1  every (power-of-two scaled) value is split into two float16 planes x = hi + lo (22 significant
//           bits); x.y ~ hi.hi + hi.lo + lo.hi with three v_mfma_f32_32x32x16_f16 chains (2.5 PF peak, fp32
//           accumulation).  5.3x fewer matrix-pipe cycles than PREC 0 at fp32-class accuracy.
//   PREC 2  a compact copy of the hi planes alone (rows of 2*DP bytes): one MFMA chain per product (11 significant
//           bits; the score error is bounded with the measured float16 residual norms, gt_knn.cpp).  3x fewer
//           matrix-pipe cycles again, half the LDS and fabric traffic, 3 workgroups per CU; the host uses it for the
//           main pass when the data tolerate the wider error bound - repairs always run in PREC 1.
// Either way the scores only have to be within a KNOWN error bound of the true ones: exact ordering is
// established afterwards in float64 (gt_rerank.hip), which also proves the candidate table complete or sends
// the row to an exhaustive fallback.
//
// MODE 0 (top-M' selection): survivors (s > thr) are appended to the query's candidate list in global
//   memory (each half-wave owns one half of the list and keeps its fill count in a register: no atomics, no
//   cross-wave traffic).  When a half passes its trigger level the owning wave selects the best 16*NT of its
//   64*NT keys (bitwise search for the 16*NT-th largest score with wave ballots, then an order-free filter) and
//   raises thr to that score.
//   At the end list[0..count) holds the candidates (unordered) and thr_out the last admission threshold.
// MODE 1 (radius collect): thr is a fixed per-query score bound; every survivor is appended (up to `cap`
//   entries per query, the true count is always reported).  Slots come from a global counter so the
//   database can be split over gridDim.y workgroups when there are few query rows.
//
// MODE 2 (symmetric collect, self queries over the whole point set in cell-sorted order, gt_knn.cpp): thr is a fixed
//   per-row score bound (from a MODE 0 launch with sched = 1 over the row's own neighbourhood + a strided sample).  The
//   score matrix is symmetric up to the seeds - x_i.x_j is the same MFMA chain whichever of the two is the query - so
//   workgroup I (query block I) streams only the blocks I .. I + (NB-1)/2 (mod NB) and tests every result twice: once
//   for its query (the lane's own threshold) and once for the database row as a query of its own (per-row thresholds
//   staged with the tile; a per-sub-tile minimum keeps the hot path at one compare).  Survivors of either test go to
//   the list of the row they are candidates OF (tlists, one per row of the point set) through a global slot counter:
//   one atomic per lane and unit for the forward hits, one per transposed hit, all issued before the first store.
//   A row's list is fed by many workgroups, so a block's walk can be cut into nseg independent work items (grid =
//   blocks x nseg) - the host picks nseg so that the last round of workgroups is full.  Every unordered pair of rows is
//   scored exactly once: N^2 d executed flop for the 2 N^2 d the problem asks for.
//
// Roofline: MFMA-bound.  Algorithmic work 2*d flop per (query, database row) pair.
#include "gt_common.h"
#include "gt_device.h"
#include "gt_knn_select.h"

#ifndef GT_SEL_PIPE
#define GT_SEL_PIPE 1
#endif
// -DGT_SEL_QT1=1 builds the narrow variant of a (precision, DP <= 64) unit: one query tile per wave
#ifndef GT_SEL_QT1
#define GT_SEL_QT1 0
#endif
#ifndef GT_SEL_DSFIRST
#define GT_SEL_DSFIRST 1
#endif
// 1: (float16 back ends, two query tiles per wave) the -|y|^2/2 seeds of a sub-tile are read from LDS once, into registers
// that enter the first MFMA of both query tiles' chains as the C operand, instead of once per accumulator
#ifndef GT_SEL_NBUF3
#define GT_SEL_NBUF3 1   // symmetric collect: three tile buffers (two tiles in flight) instead of two
#endif
#ifndef GT_SEL_TWO_VPM
#define GT_SEL_TWO_VPM 5   // two-stage collect: VALU instructions scheduled behind each MFMA of the unit loop
#endif
#ifndef GT_SEL_TWO_NS1
#define GT_SEL_TWO_NS1(DP_) 1   // k-steps (16 features each) of stage one of the two-stage symmetric collect
#endif
#ifndef GT_SEL_PAIRCOLD
#define GT_SEL_PAIRCOLD 1
#endif
#ifndef GT_SEL_SEEDREG
#define GT_SEL_SEEDREG 1
#endif
// development ablations (tools/build_variant.py; results are invalid when set):
//   1 seeds not read from LDS   2 A fragments read once per tile   4 no staging / barrier after the first tile
//   8 no admission test   16 staging but no barrier   32 barrier but no staging   64 stream 16 L2-resident tiles
#ifndef GT_SEL_EXP
#define GT_SEL_EXP 0
#endif
// -DGT_SEL_EXP_MODE=<m>: the ablation applies to the kernels of that MODE only (e.g. 2: the symmetric collect keeps a
// valid threshold-seeding launch in front of it)
#ifndef GT_SEL_EXP_MODE
#define GT_SEL_EXP_MODE -1
#endif
#define GT_EXP ((GT_SEL_EXP_MODE < 0 || MODE == GT_SEL_EXP_MODE) ? GT_SEL_EXP : 0)


namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// GLDS: the database tile goes global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction, no
// staging registers, no ds_write pass).  The LDS image is then lane-linear, so rows cannot be padded; bank
// conflicts are avoided by an XOR swizzle of the 16-byte chunk index with a function of the row, applied to the
// per-lane SOURCE address of the load and to the ds_read address alike (power-of-two row sizes only).
#ifndef GT_SEL_GLDS
#define GT_SEL_GLDS 1
#endif
template <int DP, int PREC>
struct SelCfg {
    static constexpr int QT = (DP <= 64 && !GT_SEL_QT1) ? 2 : 1;       // 32-row query tiles per wave
    static constexpr int BQ = 4 * QT * 32;              // query rows per workgroup
    static constexpr int BN = (DP <= 64) ? 128 : 64;    // database rows per LDS tile
    static constexpr int RW = (PREC == 2) ? DP / 2 : DP;   // row width in dwords: float32 | hi,lo float16 planes | hi plane
    static constexpr int RB = 4 * RW;                   // row bytes
    static constexpr bool GLDS = GT_SEL_GLDS && PREC >= 1 && (RB & (RB - 1)) == 0 && RB <= 512 && (BN * RB) % 4096 == 0;
    static constexpr int LDP = GLDS ? RW : RW + 4;      // LDS row stride (dwords); padded: conflict-free ds_read_b128
    static constexpr int NF4 = BN * RW / 4;             // 16-byte units per tile
    static constexpr int F4_PER_THREAD = (NF4 + 255) / 256;
    static constexpr int TILE_FLOATS = BN * LDP;
    static constexpr size_t LDS_BYTES =
        size_t(2) * TILE_FLOATS * 4 + size_t(2) * BN * 4;
    // MODE 2 streams the tiles through NBUF_SYM buffers (two tiles in flight per workgroup: the collect launch is bound
    // by the bytes in flight, cdna_hip_programming.md "latency x bandwidth") and also stages the per-sub-tile minima
    // of the row thresholds [NBUF][8]
    static constexpr int NBUF_SYM = (GLDS && GT_SEL_NBUF3) ? 3 : 2;
    // (+ the balls of the unit skipping of the two-stage collect: per buffer 4 sub-tile centres [16] and {radius, need} pairs,
    //  per wave the same for its GT_SEL_TWO_QT query tiles)
    static constexpr size_t LDS_BYTES_SYM =
        size_t(NBUF_SYM) * TILE_FLOATS * 4 + size_t(NBUF_SYM) * BN * 4 + size_t(NBUF_SYM) * 8 * 4 +
        size_t(NBUF_SYM) * (64 + 8) * 4 + size_t(4) * GT_SEL_TWO_QT * 18 * 4;
    static constexpr int TPB = BQ / BN;                 // database tiles per query block
    // swizzle geometry (GLDS)
    static constexpr int CPR = RB / 16;                 // 16-byte chunks per row
    static constexpr int RDIV = (RB >= 256) ? 1 : 256 / RB;   // rows sharing one 256-byte bank row
    static constexpr int SMASK = (CPR < 16 ? CPR : 16) - 1;
    static constexpr int RPP = 1024 / RB > 0 ? 1024 / RB : 1; // rows per 1 KiB piece (RB <= 512)
    static constexpr int NPW = (BN * RB / 1024) / 4;    // pieces per wave and tile
    // waves per SIMD the register budget is cut for: the single-chain kernel needs < 128 VGPRs and half the LDS
    static constexpr int WAVES = (PREC == 2) ? 4 : 2;
};
__device__ __forceinline__ int swz_of_row(int row, int rdiv, int smask) { return (row / rdiv) & smask; }

// ---- operand fragments -------------------------------------------------------------------------
// PREC 0: row = DP floats; lane (li, h) holds features [h*DP/2, (h+1)*DP/2)  (one per MFMA k-step)
// PREC 1: row = hi plane (DP halves) | lo plane (DP halves); per k-step s of 16 features lane (li, h) holds
//         features [16 s + 8 h, 16 s + 8 h + 8) of a plane as one f16x8
template <int DP, int PREC>
struct Frag;

template <int DP>
struct Frag<DP, 0> {
    static constexpr int KS = DP / 2;
    float v[KS];
    __device__ __forceinline__ void load(const float* row, int h, int = 0) {
        const float4* p = reinterpret_cast<const float4*>(row + h * KS);
#pragma unroll
        for (int c = 0; c < KS / 4; ++c) {
            const float4 q = p[c];
            v[4 * c + 0] = q.x;
            v[4 * c + 1] = q.y;
            v[4 * c + 2] = q.z;
            v[4 * c + 3] = q.w;
        }
    }
};

template <int DP>
struct Frag<DP, 1> {
    static constexpr int NS = DP / 16;
    f16x8 hi[NS], lo[NS];
    __device__ __forceinline__ void load(const float* row, int h, int swz = 0) {
        const f16x8* p = reinterpret_cast<const f16x8*>(row);   // 16-byte units: hi plane = units [0, DP/8)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            hi[s] = p[(2 * s + h) ^ swz];
            lo[s] = p[(DP / 8 + 2 * s + h) ^ swz];
        }
    }
};

template <int DP>
struct Frag<DP, 2> {
    static constexpr int NS = DP / 16;
    f16x8 hi[NS];
    __device__ __forceinline__ void load(const float* row, int h, int swz = 0) {
        const f16x8* p = reinterpret_cast<const f16x8*>(row);   // hi plane of the split layout
#pragma unroll
        for (int s = 0; s < NS; ++s) hi[s] = p[(2 * s + h) ^ swz];
    }
};

// one query tile's K chain (the two query tiles of a wave are issued back to back so that the epilogue of the
// first can overlap the matrix work of the second)
template <int DP>
__device__ __forceinline__ void mma_chain(const Frag<DP, 0>& a, const Frag<DP, 0>& b, f32x16& acc) {
#pragma unroll
    for (int s = 0; s < DP / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s], b.v[s], acc, 0, 0, 0);
}

template <int DP>
__device__ __forceinline__ void mma_chain(const Frag<DP, 1>& a, const Frag<DP, 1>& b, f32x16& acc) {
    // small cross terms first, the dominant hi.hi chain last
#pragma unroll
    for (int s = 0; s < DP / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo[s], b.hi[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < DP / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.lo[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < DP / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.hi[s], acc, 0, 0, 0);
}

template <int DP>
__device__ __forceinline__ void mma_chain(const Frag<DP, 2>& a, const Frag<DP, 2>& b, f32x16& acc) {
#pragma unroll
    for (int s = 0; s < DP / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.hi[s], acc, 0, 0, 0);
}

// split-float16 chain started from a seed that stays intact (D != C on the first instruction)
template <int DP>
__device__ __forceinline__ void mma_chain_seeded(const Frag<DP, 1>& a, const Frag<DP, 1>& b, const f32x16& seed,
                                                 f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo[0], b.hi[0], seed, 0, 0, 0);
#pragma unroll
    for (int s = 1; s < DP / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo[s], b.hi[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < DP / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.lo[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < DP / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.hi[s], acc, 0, 0, 0);
}

// same chain, started from a seed that stays intact (D != C on the first instruction)
template <int DP>
__device__ __forceinline__ void mma_chain_seeded(const Frag<DP, 2>& a, const Frag<DP, 2>& b, const f32x16& seed,
                                                 f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[0], b.hi[0], seed, 0, 0, 0);
#pragma unroll
    for (int s = 1; s < DP / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.hi[s], acc, 0, 0, 0);
}

// first NS1 k-steps only (MODE 2, two-stage scoring)
template <int DP, int NS1>
__device__ __forceinline__ void frag_load_part(Frag<DP, 2>& a, const float* row, int h, int swz) {
    const f16x8* p = reinterpret_cast<const f16x8*>(row);
#pragma unroll
    for (int s = 0; s < NS1; ++s) a.hi[s] = p[(2 * s + h) ^ swz];
}
template <int DP, int NS1, int PR>
__device__ __forceinline__ void frag_load_part(Frag<DP, PR>& a, const float* row, int h, int swz) {
    a.load(row, h, swz);
}
template <int DP, int NS1>
__device__ __forceinline__ void mma_chain_seeded_part(const Frag<DP, 2>& a, const Frag<DP, 2>& b, const f32x16& seed,
                                                      f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[0], b.hi[0], seed, 0, 0, 0);
#pragma unroll
    for (int s = 1; s < NS1; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.hi[s], acc, 0, 0, 0);
}
template <int DP, int NS1, int PR>
__device__ __forceinline__ void mma_chain_seeded_part(const Frag<DP, PR>& a, const Frag<DP, PR>& b, const f32x16& seed,
                                                      f32x16& acc) {
    mma_chain_seeded<DP>(a, b, seed, acc);
}

// Append store that hipcc does not track: a compiler-visible store makes hipcc park the wave on
// `s_waitcnt vmcnt(0)` at the top of every tile (it guards the reuse of the store's registers), i.e. on the full
// L2 write-acknowledge latency.  The data registers are protected by the s_nop inside the string; the untracked
// entries in the VMEM queue can only make hipcc's own counted waits stricter (cdna_hip_programming.md 5.7);
// readers of the list wait with an explicit `s_waitcnt vmcnt(0)` first.
__device__ __forceinline__ void list_store(uint64_t* p, uint64_t v) {
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// ---- candidate-list compaction ----------------------------------------------------------------------
// A query's list has two halves of HALF = 32*NT slots, one per half-wave (lanes li and li+32 see disjoint
// database rows of every tile), each with its own fill count kept in a REGISTER of the owning lane - the hot
// path needs no atomics.  Compaction keeps the MKEEP = 16*NT best scores of both halves together.  The list
// does not have to be ordered while streaming (the float64 re-rank sorts the final table), so this is a
// selection, not a sort: a 32-step bitwise search for the MKEEP-th largest score (wave ballots + scalar
// popcounts, no cross-lane shuffles), then an order-free filter.  Entries tied with the threshold are kept
// only up to the quota; dropping the rest is consistent with the strict `score > thr` admission test.
// CONTIG: survivors go to slots [0, kept) (final table); else they are dealt evenly to the two halves.
// Returns the threshold implied by the selection (-inf when nothing had to be dropped).
// NTE = key slots per lane actually populated (entries per half <= 32*NTE): the ballots of the bitwise search scale
// with it, and most compactions (threshold-seeding phase, final tables) see short lists.
template <int NT, int NTE, bool CONTIG>
__device__ __forceinline__ float compact_impl(uint64_t* __restrict__ lp, const uint32_t n0, const uint32_t n1,
                                              const int lane, uint32_t& kept, const uint32_t MKEEP) {
    constexpr uint32_t HALF = 32 * NT;
    uint64_t key[NTE];
    uint32_t ord[NTE];
#pragma unroll
    for (int u = 0; u < NTE; ++u) {
        const bool second = u >= NTE / 2;
        const uint32_t e = uint32_t((second ? u - NTE / 2 : u) * 64 + lane);
        const bool valid = e < (second ? n1 : n0);
        key[u] = valid ? ld_agent_u64(lp + (second ? HALF : 0u) + e) : 0ull;
        ord[u] = uint32_t(key[u] >> 32);   // a valid key has ord > 0 (ord(-inf) = 0x007fffff)
    }
    const uint32_t n = n0 + n1;
    uint32_t T = 0u, c_gt = n, quota_eq = 0u;
    if (n > MKEEP) {
#pragma unroll 1
        for (int b = 31; b >= 0; --b) {
            const uint32_t trial = T | (1u << b);
            uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < NTE; ++u) c += uint32_t(__popcll(__ballot(ord[u] >= trial)));
            if (c >= MKEEP) T = trial;
        }
        c_gt = 0;
#pragma unroll
        for (int u = 0; u < NTE; ++u) c_gt += uint32_t(__popcll(__ballot(ord[u] > T)));
        quota_eq = MKEEP - c_gt;
    }
    kept = n > MKEEP ? MKEEP : n;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t base_gt = 0, base_eq = 0;
#pragma unroll
    for (int u = 0; u < NTE; ++u) {
        const bool gt = ord[u] > T;
        const bool eq = (ord[u] == T) && (T != 0u);
        const unsigned long long mg = __ballot(gt), me = __ballot(eq);
        const uint32_t kg = base_gt + uint32_t(__popcll(mg & lt_mask));
        const uint32_t ke = base_eq + uint32_t(__popcll(me & lt_mask));
        const bool take = gt || (eq && ke < quota_eq);
        const uint32_t k = gt ? kg : c_gt + ke;
        const uint32_t slot = CONTIG ? k : ((k < MKEEP / 2) ? k : HALF + (k - MKEEP / 2));
        if (take) st_agent_u64(lp + slot, key[u]);
        base_gt += uint32_t(__popcll(mg));
        base_eq += uint32_t(__popcll(me));
    }
    return n > MKEEP ? ord_f32(T) : -INFINITY;
}

template <int NT, bool CONTIG>
__device__ __forceinline__ float compact_list(uint64_t* __restrict__ lp, const uint32_t n0, const uint32_t n1,
                                              const int lane, uint32_t& kept, const uint32_t MKEEP = 16 * NT) {
    const uint32_t nm = n0 > n1 ? n0 : n1;   // wave-uniform
    if (nm <= 64u) return compact_impl<NT, 2, CONTIG>(lp, n0, n1, lane, kept, MKEEP);
    if (nm <= 128u) return compact_impl<NT, 4, CONTIG>(lp, n0, n1, lane, kept, MKEEP);
    if constexpr (NT > 8) {
        if (nm <= 256u) return compact_impl<NT, 8, CONTIG>(lp, n0, n1, lane, kept, MKEEP);
        if (nm <= 512u) return compact_impl<NT, 16, CONTIG>(lp, n0, n1, lane, kept, MKEEP);
    }
    return compact_impl<NT, NT, CONTIG>(lp, n0, n1, lane, kept, MKEEP);
}

#ifndef GT_SEL_P2_WAVES
#define GT_SEL_P2_WAVES 3
#endif
template <int DP, int NT, int MODEX, int PREC>
__global__ __launch_bounds__(256, (PREC == 2 && NT == 8) ? GT_SEL_P2_WAVES : 2) void knn_select_kernel(
    const float* __restrict__ Yp, const float* __restrict__ hneg, const float* __restrict__ Qp,
    const int32_t* __restrict__ qrows, const int64_t q0, const int32_t nq, const int32_t ntiles,
    uint64_t* __restrict__ lists, uint32_t* __restrict__ counts, const float* __restrict__ thr_in,
    float* __restrict__ thr_out, const int32_t cap, const int32_t dbg, unsigned long long* __restrict__ prof,
    const int32_t samp_stride, const int32_t samp_keep, const int32_t samp_end, const int32_t samp2_level,
    const int32_t samp2_keep, const int32_t samp_trig, const int32_t final_keep, const SymDev sy) {
    using C = SelCfg<DP, PREC>;
    // MODEX 3 = MODE 2 with two-stage scoring: the unit loop runs the first NS1 k-steps against partial-distance
    // thresholds, the cold path recomputes the survivors in full (SymDev::half_steps)
    constexpr int MODE = MODEX == 3 ? 2 : MODEX;
    constexpr bool TWO = MODEX == 3;
    constexpr int NS1 = TWO ? GT_SEL_TWO_NS1(DP) : DP / 16;   // k-steps of the unit loop (PREC 2)
    unsigned long long t_adm = 0, t_cmp = 0, t_bar = 0, n_cmp = 0, n_adm = 0, t_lvl0 = 0, n_adm_lvl0 = 0;
    const unsigned long long t_start = prof ? __builtin_readcyclecounter() : 0ull;
    // (two-stage collect: the unit loop keeps only the first NS1 k-steps of the query fragments in registers, which buys
    //  more query tiles per wave - every streamed tile then meets 128 x QT rows, the stream shrinks accordingly)
    constexpr int QT = TWO ? GT_SEL_TWO_QT : C::QT, BQ = 128 * QT, BN = C::BN, LDP = C::LDP, TPB = BQ / BN;
    static_assert(TWO || BQ == C::BQ, "query block size");
    constexpr int LCAP = 64 * NT;        // list capacity in selection mode (two halves of HALF slots)
    constexpr int HALF = 32 * NT;
    constexpr int MKEEP = 16 * NT;       // M'
    constexpr int TRIGH = HALF - BN / 2; // per-half trigger: a tile adds at most BN/2 entries to one half
    static_assert(TRIGH >= MKEEP / 2, "list too small for the tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NBUF = (MODE == 2) ? C::NBUF_SYM : 2;                  // tile buffers (MODE 2: two tiles in flight)
    float* tile = reinterpret_cast<float*>(smem_raw);                    // [NBUF][BN][LDP]
    float* hn = tile + NBUF * C::TILE_FLOATS;                            // [NBUF][BN]
    float* gm = hn + NBUF * BN;                                          // MODE 2: [NBUF][8] sub-tile minima of the row thresholds
    // two-stage collect, unit skipping (SymDev::zc): balls of the tile's sub-tiles per buffer, of the wave's query tiles
    float* tcz = gm + NBUF * 8;                                          // [NBUF][4][16] centres
    float* trn = tcz + NBUF * 64;                                        // [NBUF][4][2]  {radius, need}
    float* qcz = trn + NBUF * 8;                                         // [4 waves][QT][16]
    float* qrn = qcz + 4 * GT_SEL_TWO_QT * 16;                           // [4 waves][QT][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int li = lane & 31;
    const int h = lane >> 5;
    // MODE 2 / sched 1: the queries ARE the database rows (cell-sorted order), query block I = rows [I*BQ, (I+1)*BQ).
    // Workgroups are dealt to the 8 XCDs round robin: each XCD takes a contiguous eighth of the blocks, so that the
    // workgroups sharing an L2 walk overlapping windows of the database.
    const bool own_sched = (MODE == 2) || (MODE == 0 && sy.sched == 1);
    int64_t bidx = blockIdx.x;
    int seg = 0;
    if (own_sched) {
        const int64_t nb = gridDim.x, xcd = bidx & 7, base = nb >> 3, rem = nb & 7;
        bidx = xcd * base + (xcd < rem ? xcd : rem) + (bidx >> 3);
        if (MODE == 2) {   // work items are block-major (item = block * nseg + seg): every XCD gets its share of the
                           // hit-rich first segments (the blocks' own neighbourhoods)
            const int nseg = sy.nseg > 0 ? sy.nseg : 1;
            seg = int(bidx % nseg);
            bidx /= nseg;
            if (sy.shard_world > 1) seg += int((int64_t(sy.shard_rank) + bidx / sy.shard_group) % sy.shard_world) * nseg;
            // own-only collect (a rank's own query blocks against EVERY tile, forward test only): the launch covers the
            // blocks [block0, block0 + nblk)
            if (sy.own_only != 0) bidx += sy.block0;
        } else {
            bidx += sy.block0;
        }
    }
    const int64_t qblock = bidx * BQ;
    const size_t lstride = (MODE == 0) ? size_t(LCAP) : size_t(cap);
    // database tile range of this workgroup (MODE 1 may split the database over gridDim.y)
    int t_begin = 0, t_end = ntiles;
    if (MODE == 1) {
        const int per = (ntiles + int(gridDim.y) - 1) / int(gridDim.y);
        t_begin = int(blockIdx.y) * per;
        t_end = t_begin + per < ntiles ? t_begin + per : ntiles;
        if (t_begin >= t_end) return;
    }
    // own_sched tile walks (T tiles = NB blocks of TPB tiles):
    //   MODE 2:   the own block, then H = (NB-1)/2 blocks (mod NB) with both directions tested, then (NB even) the
    //             antipodal block forward only - its owner does the same for the other direction
    //   sched 1:  the explicit tile list of the query block (gt_sym.hip: the tiles of the cells nearest to the block's
    //             own cells, then a strided sample of the rest; every tile at most once)
    int n_tr_end = 0;
    const int32_t* tl_base = nullptr;
    int32_t tl_cache = 0;
    // MODE 2, listed walk (SymDev::walk_list): `it` runs over the entries of the block's list, an entry is a walk position
    const int32_t* wl_base = nullptr;
    if (own_sched) {
        const int T = ntiles, NB = T / TPB;
        t_begin = 0;
        if (MODE == 2) {
            const int H = (NB - 1) / 2;
            n_tr_end = TPB * (1 + H);
            int walk = n_tr_end + ((NB & 1) ? 0 : (NB > 1 ? TPB : 0));
            if (sy.own_only != 0) {   // every tile once, starting with the own block; no tile takes the results as its queries
                walk = T;
                n_tr_end = 0;
            }
            if (!TWO && sy.walk_list != nullptr) {
                const int wc_ = sy.walk_cnt[bidx];
                if (wc_ >= 0) {   // (< 0: the list did not fit - the whole walk)
                    walk = wc_;
                    wl_base = sy.walk_list + size_t(bidx) * size_t(sy.walk_stride);
                }
            }
            const int nseg = (sy.nseg > 0 ? sy.nseg : 1) * (sy.shard_world > 1 ? sy.shard_world : 1);
            t_begin = int(int64_t(walk) * seg / nseg);          // this item's part of the block's walk
            t_end = int(int64_t(walk) * (seg + 1) / nseg);
            if (t_begin >= t_end) return;
        } else {
            t_end = sy.tile_cnt[bidx >> sy.list_shift];
            tl_base = sy.tile_list + size_t(bidx >> sy.list_shift) * size_t(sy.tile_stride);
            tl_cache = tl_base[lane < t_end ? lane : 0];
        }
    }

    // ---- query fragments (B operand), resident for the whole kernel ----
    Frag<DP, PREC> bq[QT];
    float thr[QT];
    float hnq[QT];       // MODE 2: the query's own seed (-inf on pad queries: never admitted anywhere)
    uint32_t fill[QT];   // MODE 0: entries in this lane's half of the query's list
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        fill[qt] = 0u;
        const int ql = (w * QT + qt) * 32 + li;
        const int64_t qg = qblock + ql;
        const int64_t qc = qg < nq ? qg : int64_t(nq) - 1;   // clamp pad queries onto a real row
        const int64_t row = qrows ? int64_t(qrows[qc]) : q0 + qc;
        if constexpr (TWO) frag_load_part<DP, NS1>(bq[qt], Qp + row * C::RW, h, 0);
        else bq[qt].load(Qp + row * C::RW, h);
        // MODE 0: -inf, or a proven lower bound of the wanted scores indexed by row (gt_query_order); MODE 1: the radius
        thr[qt] = (MODE == 0) ? ((dbg & 1) ? INFINITY : (thr_in ? thr_in[row - q0] : -INFINITY))
                              : ((qg < nq) ? thr_in[qc] : INFINITY);
        hnq[qt] = (MODE == 2) ? ((qg < nq) ? hneg[row] : -INFINITY) : 0.f;
        if constexpr (TWO) {
            // the unit loop tests partial scores: half thresholds / half seeds here, the full ones come back from
            // memory on the cold path
            thr[qt] = (qg < nq) ? sy.thrh[qc] : INFINITY;
            hnq[qt] = (qg < nq) ? sy.hh[row] : -INFINITY;
        }
    }

    // ---- tile staging (global -> registers -> LDS, padded rows), in two halves to halve the staging registers ----
    constexpr int HF4 = C::NF4 / 2;                      // 16-byte units per half tile
    constexpr int HF4_PER_THREAD = (HF4 + 255) / 256;
    float4 stage[HF4_PER_THREAD];
    float stage_h = 0.f, stage_gm = 0.f;
#define GT_STAGE_LOAD(T_, HALF_)                                                                          \
    {                                                                                                     \
        const float4* src_ = reinterpret_cast<const float4*>(Yp + size_t((GT_EXP & 64) ? ((T_) & 15) : (T_)) * BN * C::RW) + (HALF_) * HF4;   \
        _Pragma("unroll") for (int u_ = 0; u_ < HF4_PER_THREAD; ++u_) {                                    \
            const int f = tid + u_ * 256;                                                                 \
            stage[u_] = (f < HF4) ? src_[f] : make_float4(0.f, 0.f, 0.f, 0.f);                            \
        }                                                                                                 \
        if ((HALF_) == 0) stage_h = (tid < BN) ? (TWO ? sy.hh : hneg)[size_t(T_) * BN + tid] : 0.f;       \
        if (MODE == 2 && (HALF_) == 0) {                                                                  \
            stage_gm = (tid < BN / 32) ? (TWO ? sy.gminh : sy.gmin)[size_t(T_) * (BN / 32) + tid] : 0.f;  \
        }                                                                                                 \
    }
#define GT_STAGE_STORE(BUF_, HALF_)                                                                       \
    {                                                                                                     \
        float* tb_ = tile + (BUF_) * C::TILE_FLOATS;                                                      \
        _Pragma("unroll") for (int u_ = 0; u_ < HF4_PER_THREAD; ++u_) {                                    \
            const int f = tid + u_ * 256;                                                                 \
            if (f < HF4) {                                                                                \
                const int g_ = f + (HALF_) * HF4;                                                         \
                const int r = (g_ * 4) / C::RW;                                                           \
                const int c = (g_ * 4) % C::RW;                                                           \
                *reinterpret_cast<float4*>(tb_ + r * LDP + c) = stage[u_];                                \
            }                                                                                             \
        }                                                                                                 \
        if ((HALF_) == 0 && tid < BN) hn[(BUF_) * BN + tid] = stage_h;                                    \
        if (MODE == 2 && (HALF_) == 0 && tid < BN / 32) gm[(BUF_) * 8 + tid] = stage_gm;                  \
    }

    // direct global -> LDS staging of one tile (GLDS): wave wu copies pieces [wu*NPW, (wu+1)*NPW) of 1 KiB; lane L of
    // piece p fills LDS bytes [p*1024 + 16 L, +16) = (row p*RPP + L/CPR, physical chunk L%CPR), i.e. it fetches
    // the logical chunk (L%CPR) ^ swz(row) of that row.  Seeds: waves [0, BN/64) copy 64 floats each.
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int wu = __builtin_amdgcn_readfirstlane(w);
#define GT_GLDS_ISSUE(T_, BUF_)                                                                            \
    {                                                                                                      \
        const char* gt_ = reinterpret_cast<const char*>(Yp) + size_t((GT_EXP & 64) ? ((T_) & 15) : (T_)) * BN * C::RB; \
        char* lt_ = reinterpret_cast<char*>(tile + (BUF_) * C::TILE_FLOATS);                               \
        uint32_t lv_ = uint32_t(lane);                                                                     \
        /* many pieces: recompute the lane offsets per tile instead of pinning a register each */          \
        if (C::NPW > 2) asm volatile("" : "+v"(lv_));                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < C::NPW; ++i_) {                                            \
            const uint32_t p_ = uint32_t(wu * C::NPW + i_);                                                \
            const uint32_t r_ = p_ * C::RPP + lv_ / C::CPR;                                                \
            const uint32_t c_ = (lv_ % C::CPR) ^ ((r_ / C::RDIV) & C::SMASK);                              \
            __builtin_amdgcn_global_load_lds((glb_void*)(gt_ + (r_ * C::RB + c_ * 16u)),                   \
                                             (lds_void*)(lt_ + p_ * 1024u), 16, 0, 0);                     \
        }                                                                                                  \
        if (wu < BN / 64)                                                                                  \
            __builtin_amdgcn_global_load_lds((glb_void*)((TWO ? sy.hh : hneg) + size_t(T_) * BN + uint32_t(wu * 64) + lv_), \
                                             (lds_void*)(hn + (BUF_) * BN + wu * 64), 4, 0, 0);            \
        /* MODE 2: only the sub-tile minima of the row thresholds are staged (the cold path reads the rows' own values  \
           from global memory), by a wave that carries no seed piece */                                    \
        if (MODE == 2 && wu == 3 && lv_ < BN / 32)                                                         \
            __builtin_amdgcn_global_load_lds((glb_void*)((TWO ? sy.gminh : sy.gmin) + size_t(T_) * (BN / 32) + lv_), \
                                             (lds_void*)(gm + (BUF_) * 8), 4, 0, 0);                       \
        /* two-stage collect: the balls of the tile's four sub-tiles (64 centre floats by wave 2, 8 {radius, need} by wave 3) */ \
        if (TWO && sy.zc != nullptr && wu == 2)                                                            \
            __builtin_amdgcn_global_load_lds((glb_void*)(sy.zc + size_t(T_) * 64 + lv_), (lds_void*)(tcz + (BUF_) * 64), 4, 0, 0); \
        if (TWO && sy.zc != nullptr && wu == 3 && lv_ < 8)                                                 \
            __builtin_amdgcn_global_load_lds((glb_void*)(sy.zrn + size_t(T_) * 8 + lv_), (lds_void*)(trn + (BUF_) * 8), 4, 0, 0); \
    }

    // Tile order (MODE 0 with samp_stride = S > 1, a power of two): level 0 visits the tiles 0, S, 2S, ... with a small
    // list budget (keep the samp_keep best), which gives every query a tight admission threshold after 1/S of the
    // stream; level l >= 1 visits the odd multiples of S >> l, so that the tiles seen at the end of any level are an
    // evenly strided sample of the database whatever order its rows are stored in.  At the end of level samp2_level
    // (2^samp2_level / S of the stream seen) every list is cut once more, to its samp2_keep best.  Any order and
    // any budget is correct - thresholds only ever rise, and every entry dropped or rejected scored <= the final
    // threshold, which the float64 stage turns into a distance bound per row - the levels only cut the number of
    // admissions.
    const int n_a = (MODE == 0 && samp_stride > 1 && !own_sched) ? (ntiles + samp_stride - 1) / samp_stride : 0;
    int t = t_begin, t_step = n_a ? samp_stride : 1, level = 0;
    // MODE 2: walk positions of the tiles at `it`, it + 1, it + 2 (the position itself, or the entries of the block's list - a
    // scalar load each, issued an iteration before the value is used), and the tile a position stands for
    const int tbase_blk = (MODE == 2) ? int((int64_t(bidx) * TPB) % ntiles) : 0;
    // (list entries come 64 at a time into one register, lane l <- entry wl_c0 + l, and out of it with a readlane: one
    //  compiler-tracked vector load - and the drain of the memory queue it implies - per ~60 tiles)
    int32_t wl_cache = 0;
    int wl_c0 = -(1 << 30);
    auto walk_rel = [&](const int i_) -> int {
        if (wl_base == nullptr || i_ >= t_end) return i_;   // wave-uniform
        if (i_ >= wl_c0 + 64) {
            wl_c0 = i_;
            wl_cache = wl_base[i_ + lane < t_end ? i_ + lane : t_end - 1];
        }
        return __builtin_amdgcn_readlane(wl_cache, i_ - wl_c0);
    };
    auto walk_tile = [&](const int rel_) -> int {
        const int t_ = tbase_blk + rel_;
        return t_ >= ntiles ? t_ - ntiles : t_;
    };
    int rel_c = 0, rel_n1 = 0, rel_n2 = 0, rel_n3 = 0;
    if (own_sched) {
        if (MODE == 2) {
            rel_c = walk_rel(t_begin);
            rel_n1 = walk_rel(t_begin + 1);
            rel_n2 = walk_rel(t_begin + 2);
            t = walk_tile(rel_c);                                          // MODE 2: the own block first (segment 0)
        }
        if (MODE == 0) t = __builtin_amdgcn_readlane(tl_cache, 0);         // sched 1: first entry of the list
    }
    // two-stage collect, unit skipping: the balls of this wave's query tiles (groups of 32 sorted rows), once
    const bool skip_on = TWO && sy.zc != nullptr;
    if constexpr (TWO) {
        if (skip_on) {
            const int64_t g0_ = qblock / 32 + w * QT;
            for (int f_ = lane; f_ < QT * 16; f_ += 64) qcz[w * QT * 16 + f_] = sy.zc[g0_ * 16 + f_];
            if (lane < QT * 2) qrn[w * QT * 2 + lane] = sy.zrn[g0_ * 2 + lane];
        }
    }
    if constexpr (C::GLDS) {
        GT_GLDS_ISSUE(t, 0);
        if constexpr (NBUF == 3) {
            if (t_begin + 1 < t_end) {
                const int t1_ = (MODE == 2) ? walk_tile(rel_n1) : ((t + 1 >= ntiles) ? t + 1 - ntiles : t + 1);
                GT_GLDS_ISSUE(t1_, 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        GT_STAGE_LOAD(t, 0);
        GT_STAGE_STORE(0, 0);
        GT_STAGE_LOAD(t, 1);
        GT_STAGE_STORE(0, 1);
    }
    __syncthreads();
    const int aswz = C::GLDS ? swz_of_row(li, C::RDIV, C::SMASK) : 0;   // sub-tiles start at multiples of 32 rows
    uint32_t n_scored = 0u;                                               // two-stage collect: units this wave scored (not skipped)
    uint32_t qfill = 0u;                                                  // two-stage collect: entries this wave has queued
    uint32_t pend_pm = 0u, pend_t = 0u;                                   //   pairs of the previous tile not yet written
// The wave owns its region of the queue: the slots come from a register, the entries leave as stores nobody waits for.
// They must sit in the memory queue BEHIND the loads of the tile that is waited for next and IN FRONT of the loads issued
// after them: a counted wait then passes them on the way to the loads it is about, and never has to drain younger loads
// to reach them (loads return in order among themselves, stores only ever make a counted wait stricter).
#define GT_QUEUE_FLUSH()                                                                                   \
    if (TWO && pend_pm != 0u) {   /* wave-uniform */                                                       \
        const uint32_t n_ = uint32_t(__popc(pend_pm));                                                     \
        uint2* dst_ = qreg + qfill;                                                                        \
        bool room_ = qfill + n_ <= uint32_t(sy.qcap);                                                      \
        if (!room_) {   /* region full (wave-uniform, rare): slots in the shared spill area, from an atomic */ \
            uint32_t sb0_ = 0u;                                                                            \
            if (lane == 0) sb0_ = atomicAdd(sy.qspill_count, n_);                                          \
            sb0_ = uint32_t(__builtin_amdgcn_readfirstlane(int(sb0_)));                                    \
            room_ = sb0_ + n_ <= uint32_t(sy.qspill_cap);   /* (beyond it: the count says so, the caller starts over) */ \
            dst_ = sy.qspill + sb0_;                                                                       \
        } else {                                                                                           \
            qfill += n_;                                                                                   \
        }                                                                                                  \
        if (uint32_t(lane) < n_ && room_) {                                                                \
            uint32_t m_ = pend_pm;                                                                         \
            for (int i_ = 0; i_ < lane; ++i_) m_ &= m_ - 1u;   /* the lane-th set bit (lane < 16) */       \
            const uint32_t p_ = uint32_t(__ffs(int(m_)) - 1);                                              \
            const uint32_t sb_ = p_ / uint32_t(QT / 2), pr_ = p_ % uint32_t(QT / 2);                       \
            const uint64_t ent_ = uint64_t(uint32_t((qblock + (w * QT + 2 * pr_) * 32) / 64)) |            \
                                  (uint64_t(pend_t * (BN / 32) + sb_) << 32);                              \
            list_store(reinterpret_cast<uint64_t*>(dst_ + uint32_t(lane)), ent_);                          \
        }                                                                                                  \
        pend_pm = 0u;                                                                                      \
    }
    uint2* qreg = TWO ? sy.queue + (size_t(blockIdx.x) * 4 + w) * size_t(sy.qcap) : nullptr;

    for (int it = t_begin; it < t_end; ++it) {
        const int buf = (GT_EXP & (4 | 32)) ? 0 : (NBUF == 3 ? (it - t_begin) % 3 : ((it - t_begin) & 1));
        int t_next = t + t_step, level_next = level, t_step_next = t_step;
        if (own_sched) {
            if (MODE == 2) {
                t_next = walk_tile(rel_n1);
            } else if (it + 1 < t_end) {
                // the list is read 64 entries at a time (one per lane), entries come out with a readlane
                const int nx = it + 1;
                if ((nx & 63) == 0) tl_cache = tl_base[nx + lane < t_end ? nx + lane : nx];
                t_next = __builtin_amdgcn_readlane(tl_cache, nx & 63);
            }
        }
        const bool level_end = n_a && t_next >= ntiles;
        if (level_end) {   // next level: the odd multiples of samp_stride >> level_next
            level_next = level + 1;
            t_step_next = samp_stride >> level;
            t_next = samp_stride >> level_next;
        }
        GT_QUEUE_FLUSH();
        if (NBUF == 3 && !(GT_EXP & (4 | 32))) {
            // three buffers: the tile after the next goes into the buffer every wave left at the barrier that ended the
            // previous tile; the next tile has been on its way since the previous iteration
            if (it + 2 < t_end) {
                int t2_ = t_next + 1;
                if (t2_ >= ntiles) t2_ -= ntiles;
                if (MODE == 2) t2_ = walk_tile(rel_n2);
                GT_GLDS_ISSUE(t2_, (it + 2 - t_begin) % 3);
            }
            if (MODE == 2) rel_n3 = walk_rel(it + 3);   // (used from the next iteration on)
        } else if (!(GT_EXP & (4 | 32)) && it + 1 < t_end) {
            if constexpr (C::GLDS) {
                GT_GLDS_ISSUE(t_next, buf ^ 1);   // every wave left buf^1 at the barrier that ended the previous tile
            } else {
                GT_STAGE_LOAD(t_next, 0);
            }
        }
        const float* tb = tile + buf * C::TILE_FLOATS;
        const float* hb = hn + buf * BN;
        const uint32_t tbase = uint32_t(t) * BN;
        // MODE 2: does this tile's block also take the results as ITS queries?  (wave-uniform; +inf switches the test off)
        const bool tr_on = MODE == 2 && rel_c >= TPB && rel_c < n_tr_end;
        const float* gglob = (MODE == 2) ? sy.g + size_t(tbase) : nullptr;   // row thresholds of this tile (cold path only)
        float gms[BN / 32];
#pragma unroll
        for (int sb_ = 0; sb_ < BN / 32; ++sb_)
            gms[sb_] = tr_on ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(gm[buf * 8 + sb_]))) : INFINITY;
        // two-stage collect: which of the tile's (sub-tile, query tile) units can hold a pair that passes stage one at all?
        // Lane l < NU judges unit l (sub-tile l / QT, query tile l % QT) by the balls of the two groups of 32 rows in the
        // stage-one space: every pair is at least |c_q - c_t| - R_q - R_t apart, and passes only below the larger of the two
        // groups' needs (z_balls_kernel; float32 here, radii and needs rounded up there, a relative margin of 1e-4 on the
        // distance of the centres).  ~50 vector instructions per tile for up to 32 MFMAs and their tests.
        uint32_t act = 0xFFFFFFFFu;
        if constexpr (TWO) {
            if (skip_on) {   // (wave-uniform)
                const int ul_ = lane < (BN / 32) * QT ? lane : 0;
                const int sbl_ = ul_ / QT, qtl_ = ul_ % QT;
                const float* cq_ = qcz + (w * QT + qtl_) * 16;
                const float* ct_ = tcz + buf * 64 + sbl_ * 16;
                float d2_ = 0.f;
#pragma unroll
                for (int k_ = 0; k_ < 4; ++k_) {
                    const float4 a_ = *reinterpret_cast<const float4*>(cq_ + 4 * k_);
                    const float4 b_ = *reinterpret_cast<const float4*>(ct_ + 4 * k_);
                    d2_ = fmaf(a_.x - b_.x, a_.x - b_.x, d2_);
                    d2_ = fmaf(a_.y - b_.y, a_.y - b_.y, d2_);
                    d2_ = fmaf(a_.z - b_.z, a_.z - b_.z, d2_);
                    d2_ = fmaf(a_.w - b_.w, a_.w - b_.w, d2_);
                }
                const float rsum_ = qrn[(w * QT + qtl_) * 2] + trn[buf * 8 + sbl_ * 2];
                const float need_ = fmaxf(qrn[(w * QT + qtl_) * 2 + 1], trn[buf * 8 + sbl_ * 2 + 1]);
                const bool far_ = sqrtf(d2_) * (1.f - 1e-4f) - rsum_ * (1.f + 1e-6f) > need_ * (1.f + 1e-6f);
                act = uint32_t(__ballot(!far_ && lane < (BN / 32) * QT));
                if (GT_EXP & 512) act = 0xFFFFFFFFu;   // (development: the mask is formed, nothing is skipped)
            }
            n_scored += uint32_t(__builtin_popcount(act));
        }

        // Software pipeline over the NU = (BN/32)*QT units (sub-tile, query tile) of this tile, fully unrolled:
        //   unit u:  [ MFMA chain of u   ||   admission predicates of u-1 (VALU/SALU in the MFMA issue gaps) ]
        //            admission path of u-1 (taken only when a lane beat its threshold)
        // Two accumulator sets alternate, A fragments are prefetched one sub-tile ahead (ping-pong registers),
        // so neither the LDS latency nor the epilogue sits between two matrix bursts of the wave.
        constexpr int NSUB = BN / 32;
        constexpr int NU = NSUB * QT;
        Frag<DP, PREC> afr[2];
        frag_load_part<DP, NS1>(afr[0], tb + li * LDP, h, aswz);
        // three accumulator sets rotate: unit u accumulates into accp[u%3] while the predicates of u-1 read
        // accp[(u-1)%3] and the seeds (-|y|^2/2) of u+1 are fetched from LDS into accp[(u+1)%3]
        f32x16 accp[3];
        bool any_hit = false;
        bool hit_now = false;    // MODE 2: the compare of the unit just examined fired in some lane (wave-uniform)
        uint32_t hitmask = 0u;   // MODE 2: units of this tile whose compare fired
        bool cold_tile = false;  // MODE 2: this wave ran the cold path in this tile
        float mx[5];
#define GT_SEED(U_)                                                                                        \
    {                                                                                                      \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                 \
            const float4 hv_ = (GT_EXP & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) :                        \
                *reinterpret_cast<const float4*>(hb + ((U_) / QT) * 32 + 8 * g_ + 4 * h);                  \
            accp[(U_) % 3][4 * g_ + 0] = hv_.x;                                                            \
            accp[(U_) % 3][4 * g_ + 1] = hv_.y;                                                            \
            accp[(U_) % 3][4 * g_ + 2] = hv_.z;                                                            \
            accp[(U_) % 3][4 * g_ + 3] = hv_.w;                                                            \
        }                                                                                                  \
    }
// cold path: the partial maxima of the hot path's v_max3 tree (MX_[t] covers elements 3t .. 3t+2, element 15 stands
// alone) say which triples hold a hit, only those are looked at element by element
#define GT_ADMIT_ONE(PA_, E_, PSB_, PQT_)                                                                  \
    {                                                                                                      \
        const float v = (PA_)[E_];                                                                         \
        if (v > tq_) {                                                                                     \
            const uint32_t j = tbase + uint32_t((PSB_) * 32 + 8 * ((E_) >> 2) + 4 * h + ((E_) & 3));       \
            if (MODE == 0) {                                                                               \
                list_store(lp + fill[PQT_], cand_pack(v, j));                                              \
                fill[PQT_] += 1u;                                                                          \
            } else {                                                                                       \
                const uint32_t slot = atomicAdd(&counts[qblock + ql], 1u);                                 \
                if (slot < uint32_t(cap)) lp[slot] = cand_pack(v, j);                                      \
            }                                                                                              \
        }                                                                                                  \
    }
// MODE 2 cold path, run after the unit loop of a tile for the units whose hot-path compare fired (their 32 x 32 block is
// recomputed - four MFMAs - so that nothing of it has to stay live in the unrolled loop).  Forward: the lane's query
// takes every row of the unit that beats its threshold - the hits of the lane are counted first, one atomic reserves
// their slots in the query's list.  Transposed: database row r_ takes query (PQT_, li) when the score seen from its
// side, (x.y - |y_j|^2/2) + |y_j|^2/2 - |x_q|^2/2, beats its own threshold: (acc + hneg_q) > g_j; one atomic per hit
// on the row's counter.  All atomics of a unit are in flight together (one memory round trip per unit).
#define GT_ADMIT2(PA_, SD_, PSB_, PQT_)                                                                    \
    {                                                                                                      \
        const float tq_ = thrF[PQT_];                                                                      \
        const int ql = (w * QT + (PQT_)) * 32 + li;                                                        \
        const uint32_t qpos_ = uint32_t(qblock + ql);                                                      \
        const float hq_ = hnqF[PQT_];                                                                      \
        uint32_t fmask_ = 0u, tmask_ = 0u;                                                                 \
        _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) fmask_ |= ((PA_)[e_] > tq_) ? (1u << e_) : 0u;   \
        if (tr_on) {                                                                                       \
            _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                             \
                const float4 gv_ = *reinterpret_cast<const float4*>(gglob + (PSB_) * 32 + 8 * g_ + 4 * h); \
                tmask_ |= (((PA_)[4 * g_ + 0] + hq_) > gv_.x) ? (1u << (4 * g_ + 0)) : 0u;                 \
                tmask_ |= (((PA_)[4 * g_ + 1] + hq_) > gv_.y) ? (1u << (4 * g_ + 1)) : 0u;                 \
                tmask_ |= (((PA_)[4 * g_ + 2] + hq_) > gv_.z) ? (1u << (4 * g_ + 2)) : 0u;                 \
                tmask_ |= (((PA_)[4 * g_ + 3] + hq_) > gv_.w) ? (1u << (4 * g_ + 3)) : 0u;                 \
            }                                                                                              \
        }                                                                                                  \
        /* every atomic of the unit is issued before the first result is looked at: one round trip */      \
        const uint32_t nf_ = uint32_t(__popc(fmask_));                                                     \
        uint32_t k_ = 0u;                                                                                  \
        if (nf_) k_ = atomicAdd(&sy.tcounts[qpos_], nf_);                                                  \
        /* transposed: the 32 lanes of a half-wave (32 queries) aim at the SAME row's counter - the first hitting lane  \
           of the half reserves the slots of all of them (same-address atomics of one instruction would serialise) */ \
        uint32_t tslot_[16];                                                                               \
        _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                                \
            const uint32_t j = tbase + uint32_t((PSB_) * 32 + 8 * (e_ >> 2) + 4 * h + (e_ & 3));           \
            const unsigned long long bm_ = __ballot((tmask_ >> e_) & 1u);                                  \
            const uint32_t mh_ = h ? uint32_t(bm_ >> 32) : uint32_t(bm_);                                  \
            tslot_[e_] = 0u;                                                                               \
            if (mh_ != 0u && uint32_t(li) == uint32_t(__ffs(int(mh_)) - 1))                                \
                tslot_[e_] = atomicAdd(&sy.tcounts[j], uint32_t(__popc(mh_)));                             \
        }                                                                                                  \
        if (nf_) {                                                                                         \
            uint64_t* lp_ = sy.tlists + size_t(qpos_) * size_t(sy.tcap);                                   \
            _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                            \
                if (fmask_ & (1u << e_)) {                                                                 \
                    const uint32_t j = tbase + uint32_t((PSB_) * 32 + 8 * (e_ >> 2) + 4 * h + (e_ & 3));   \
                    if (k_ < uint32_t(sy.tcap)) list_store(lp_ + k_, cand_pack((PA_)[e_], j));             \
                    ++k_;                                                                                  \
                }                                                                                          \
            }                                                                                              \
            /* a row whose list has overflowed (count > capacity) is repaired later anyway: stop collecting */ \
            if (k_ > uint32_t(sy.tcap)) thr[PQT_] = INFINITY;                                              \
        }                                                                                                  \
        if (__ballot(tmask_ != 0u)) {                                                                      \
            _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                            \
                const unsigned long long bm_ = __ballot((tmask_ >> e_) & 1u);                              \
                if (bm_) {   /* wave-uniform */                                                            \
                    const uint32_t mh_ = h ? uint32_t(bm_ >> 32) : uint32_t(bm_);                          \
                    const int lead_ = (mh_ ? __ffs(int(mh_)) - 1 : 0) + 32 * h;                            \
                    const uint32_t slot_ = uint32_t(__shfl(int(tslot_[e_]), lead_)) +                      \
                                           uint32_t(__popc(mh_ & ((1u << li) - 1u)));                      \
                    if (((tmask_ >> e_) & 1u) && slot_ < uint32_t(sy.tcap)) {                              \
                        const uint32_t j = tbase + uint32_t((PSB_) * 32 + 8 * (e_ >> 2) + 4 * h + (e_ & 3)); \
                        list_store(sy.tlists + size_t(j) * size_t(sy.tcap) + slot_,                        \
                                   cand_pack(((PA_)[e_] + hq_) - (SD_)[e_], qpos_));                       \
                    }                                                                                      \
                }                                                                                          \
            }                                                                                              \
        }                                                                                                  \
    }
// Both query tiles of a wave against one sub-tile in ONE memory round trip (QT == 2): the forward atomics of the two
// queries a lane owns and the transposed atomics - aggregated per database row over the 64 queries of the two half-wave
// tiles - are all issued before the first result is looked at.
#define GT_ADMIT2P(A0_, A1_, SD_, PSB_)                                                                    \
    {                                                                                                      \
        const float tq0_ = thrF[0], tq1_ = thrF[QT - 1];                                                   \
        const uint32_t qpos0_ = uint32_t(qblock + (w * QT) * 32 + li);                                     \
        const uint32_t qpos1_ = uint32_t(qblock + (w * QT + QT - 1) * 32 + li);                            \
        const float hq0_ = hnqF[0], hq1_ = hnqF[QT - 1];                                                   \
        uint32_t f0_ = 0u, f1_ = 0u, t0_ = 0u, t1_ = 0u;                                                   \
        _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                                \
            f0_ |= ((A0_)[e_] > tq0_) ? (1u << e_) : 0u;                                                   \
            f1_ |= ((A1_)[e_] > tq1_) ? (1u << e_) : 0u;                                                   \
        }                                                                                                  \
        if (tr_on) {                                                                                       \
            _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                             \
                const float4 gv_ = *reinterpret_cast<const float4*>(gglob + (PSB_) * 32 + 8 * g_ + 4 * h); \
                const float ge_[4] = {gv_.x, gv_.y, gv_.z, gv_.w};                                         \
                _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                         \
                    t0_ |= (((A0_)[4 * g_ + c_] + hq0_) > ge_[c_]) ? (1u << (4 * g_ + c_)) : 0u;           \
                    t1_ |= (((A1_)[4 * g_ + c_] + hq1_) > ge_[c_]) ? (1u << (4 * g_ + c_)) : 0u;           \
                }                                                                                          \
            }                                                                                              \
        }                                                                                                  \
        const uint32_t nf0_ = uint32_t(__popc(f0_)), nf1_ = uint32_t(__popc(f1_));                         \
        uint32_t k0_ = 0u, k1_ = 0u;                                                                       \
        if (nf0_) k0_ = atomicAdd(&sy.tcounts[qpos0_], nf0_);                                              \
        if (nf1_) k1_ = atomicAdd(&sy.tcounts[qpos1_], nf1_);                                              \
        uint32_t tslot_[16];                                                                               \
        _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                                \
            const uint32_t j = tbase + uint32_t((PSB_) * 32 + 8 * (e_ >> 2) + 4 * h + (e_ & 3));           \
            const unsigned long long b0_ = __ballot((t0_ >> e_) & 1u), b1_ = __ballot((t1_ >> e_) & 1u);   \
            const uint32_t m0_ = h ? uint32_t(b0_ >> 32) : uint32_t(b0_);                                  \
            const uint32_t m1_ = h ? uint32_t(b1_ >> 32) : uint32_t(b1_);                                  \
            tslot_[e_] = 0u;                                                                               \
            if ((m0_ | m1_) != 0u && uint32_t(li) == uint32_t(__ffs(int(m0_ ? m0_ : m1_)) - 1))            \
                tslot_[e_] = atomicAdd(&sy.tcounts[j], uint32_t(__popc(m0_) + __popc(m1_)));               \
        }                                                                                                  \
        if (nf0_) {                                                                                        \
            uint64_t* lp_ = sy.tlists + size_t(qpos0_) * size_t(sy.tcap);                                  \
            _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                            \
                if (f0_ & (1u << e_)) {                                                                    \
                    const uint32_t j = tbase + uint32_t((PSB_) * 32 + 8 * (e_ >> 2) + 4 * h + (e_ & 3));   \
                    if (k0_ < uint32_t(sy.tcap)) list_store(lp_ + k0_, cand_pack((A0_)[e_], j));           \
                    ++k0_;                                                                                 \
                }                                                                                          \
            }                                                                                              \
            if (k0_ > uint32_t(sy.tcap)) thr[0] = INFINITY;                                                \
        }                                                                                                  \
        if (nf1_) {                                                                                        \
            uint64_t* lp_ = sy.tlists + size_t(qpos1_) * size_t(sy.tcap);                                  \
            _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                            \
                if (f1_ & (1u << e_)) {                                                                    \
                    const uint32_t j = tbase + uint32_t((PSB_) * 32 + 8 * (e_ >> 2) + 4 * h + (e_ & 3));   \
                    if (k1_ < uint32_t(sy.tcap)) list_store(lp_ + k1_, cand_pack((A1_)[e_], j));           \
                    ++k1_;                                                                                 \
                }                                                                                          \
            }                                                                                              \
            if (k1_ > uint32_t(sy.tcap)) thr[QT - 1] = INFINITY;                                           \
        }                                                                                                  \
        if (__ballot((t0_ | t1_) != 0u)) {                                                                 \
            _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                            \
                const unsigned long long b0_ = __ballot((t0_ >> e_) & 1u), b1_ = __ballot((t1_ >> e_) & 1u); \
                if (b0_ | b1_) {   /* wave-uniform */                                                      \
                    const uint32_t m0_ = h ? uint32_t(b0_ >> 32) : uint32_t(b0_);                          \
                    const uint32_t m1_ = h ? uint32_t(b1_ >> 32) : uint32_t(b1_);                          \
                    const uint32_t mm_ = m0_ ? m0_ : m1_;                                                  \
                    const int lead_ = (mm_ ? __ffs(int(mm_)) - 1 : 0) + 32 * h;                            \
                    const uint32_t base_ = uint32_t(__shfl(int(tslot_[e_]), lead_));                       \
                    const uint32_t below_ = (1u << li) - 1u;                                               \
                    const uint32_t j = tbase + uint32_t((PSB_) * 32 + 8 * (e_ >> 2) + 4 * h + (e_ & 3));   \
                    uint64_t* lj_ = sy.tlists + size_t(j) * size_t(sy.tcap);                               \
                    const uint32_t s0_ = base_ + uint32_t(__popc(m0_ & below_));                           \
                    const uint32_t s1_ = base_ + uint32_t(__popc(m0_)) + uint32_t(__popc(m1_ & below_));   \
                    if (((t0_ >> e_) & 1u) && s0_ < uint32_t(sy.tcap))                                     \
                        list_store(lj_ + s0_, cand_pack(((A0_)[e_] + hq0_) - (SD_)[e_], qpos0_));          \
                    if (((t1_ >> e_) & 1u) && s1_ < uint32_t(sy.tcap))                                     \
                        list_store(lj_ + s1_, cand_pack(((A1_)[e_] + hq1_) - (SD_)[e_], qpos1_));          \
                }                                                                                          \
            }                                                                                              \
        }                                                                                                  \
    }
#define GT_ADMIT(PA_, ANY_, MX_, PSB_, PQT_)                                                               \
    if (__builtin_expect(__ballot(ANY_) != 0ull, 0)) {   /* wave-uniform and cold: most units admit nothing */  \
        const float tq_ = thr[PQT_];                                                                       \
        const unsigned long long ts_ = prof ? __builtin_readcyclecounter() : 0ull;                         \
        const int ql = (w * QT + (PQT_)) * 32 + li;                                                        \
        uint64_t* lp = lists + size_t(qblock + ql) * lstride + (MODE == 0 ? size_t(h) * HALF : size_t(0)); \
        _Pragma("unroll") for (int t_ = 0; t_ < 5; ++t_) {                                                 \
            if (__ballot((MX_)[t_] > tq_)) {                                                               \
                GT_ADMIT_ONE(PA_, 3 * t_ + 0, PSB_, PQT_);                                                 \
                GT_ADMIT_ONE(PA_, 3 * t_ + 1, PSB_, PQT_);                                                 \
                GT_ADMIT_ONE(PA_, 3 * t_ + 2, PSB_, PQT_);                                                 \
            }                                                                                              \
        }                                                                                                  \
        if (__ballot((PA_)[15] > tq_)) GT_ADMIT_ONE(PA_, 15, PSB_, PQT_);                                  \
        if (prof) { t_adm += __builtin_readcyclecounter() - ts_; n_adm += 1; }                             \
    }
        constexpr bool SEEDREG = GT_SEL_SEEDREG && PREC >= 1 && QT >= 2;
        constexpr int NACC = SEEDREG ? 2 : 3;
        f32x16 seedr;
#define GT_SEEDR(SB_)                                                                                      \
    {                                                                                                      \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                 \
            const float4 hv_ = *reinterpret_cast<const float4*>(hb + (SB_) * 32 + 8 * g_ + 4 * h);         \
            seedr[4 * g_ + 0] = hv_.x;                                                                     \
            seedr[4 * g_ + 1] = hv_.y;                                                                     \
            seedr[4 * g_ + 2] = hv_.z;                                                                     \
            seedr[4 * g_ + 3] = hv_.w;                                                                     \
        }                                                                                                  \
    }
        if constexpr (SEEDREG) {
            GT_SEEDR(0);
        } else {
            GT_SEED(0);
        }
        if constexpr (TWO) {
            // ---- the unit loop of the two-stage collect (round 6) ----
            // The SQ counters of the generic loop on this kernel (manifold set, profiles/r6_pmc_sq_two_stage.txt): 26 scalar
            // instructions and 6 branches per MFMA - the compare masks of every unit were OR-ed, tested and folded into the
            // wave's hit mask on the scalar unit, ONE per compute unit, which the twelve waves of a CU kept ~80 % busy while
            // the matrix pipes ran at 18 %.  Here a unit is one guarded block - skipped as a whole when its balls are too far
            // apart (`act`) - and what it finds stays in vector registers: bit u of `hitv` in the lanes whose test fired.  The
            // scalar side looks at `hitv` once per tile, and at its bits only when there is one (3 % of the units pass).
            // The MFMA's latency is no longer covered by the previous unit's tests of the same wave - at a fifth of the
            // matrix peak the other waves of the SIMD cover it.
            uint32_t hitv = 0u;   // (afr[0]: the first sub-tile's fragment, loaded above)
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) {
                if (sb + 1 < NSUB) frag_load_part<DP, NS1>(afr[(sb + 1) & 1], tb + ((sb + 1) * 32 + li) * LDP, h, aswz);
                if (((act >> (sb * QT)) & ((1u << QT) - 1u)) != 0u) {   // (wave-uniform: some unit of the sub-tile is scored)
                    GT_SEEDR(sb);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        if ((act >> (sb * QT + qt)) & 1u) {
                            f32x16 pa;
                            mma_chain_seeded_part<DP, NS1>(afr[sb & 1], bq[qt], seedr, pa);
#pragma unroll
                            for (int t3 = 0; t3 < 5; ++t3) mx[t3] = fmaxf(fmaxf(pa[3 * t3], pa[3 * t3 + 1]), pa[3 * t3 + 2]);
                            const float m5 = fmaxf(fmaxf(mx[0], mx[1]), mx[2]);
                            const float m6 = fmaxf(fmaxf(mx[3], mx[4]), pa[15]);
                            const float m16 = fmaxf(m5, m6);
                            // (the same two tests as the generic loop: the lane's query, or some row of the sub-tile)
                            const bool fired = (m16 > thr[qt]) || (m16 + hnq[qt] > gms[sb]);
                            hitv |= fired ? (1u << (sb * QT + qt)) : 0u;
                        }
                    }
                }
            }
            if (__ballot(hitv != 0u) != 0ull) {   // (rare)
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (__ballot((hitv >> u) & 1u) != 0ull) hitmask |= 1u << u;
            }
            (void)any_hit;
            (void)hit_now;
        } else {
#pragma unroll
        for (int u = 0; u <= NU; ++u) {
            const int sb = u / QT, qt = u % QT;              // the unit whose chain is issued now (u < NU)
            const int psb = (u - 1) / QT, pqt = (u - 1) % QT;   // the unit whose results are examined now (u > 0)
            if (!C::GLDS && !(GT_EXP & (4 | 32)) && u == NU / 2 && it + 1 < t_end) {
                // first half of the next tile has landed: park it in the other LDS buffer, fetch the second half
                GT_STAGE_STORE(buf ^ 1, 0);
                GT_STAGE_LOAD(t_next, 1);
            }
            if (u < NU) {
                if (!(GT_EXP & 2) && qt == 0 && sb + 1 < NSUB)
                    frag_load_part<DP, NS1>(afr[(sb + 1) & 1], tb + ((sb + 1) * 32 + li) * LDP, h, aswz);
                if constexpr (SEEDREG) {
                    if (!TWO || ((act >> u) & 1u))   // (two-stage collect: a unit whose balls are too far apart is not scored)
                        mma_chain_seeded_part<DP, NS1>(afr[sb & 1], bq[qt], seedr, accp[u % NACC]);
                    if (!(GT_EXP & 1) && qt == QT - 1 && sb + 1 < NSUB) GT_SEEDR(sb + 1);   // behind the last reader of this sub-tile's seeds
                } else {
                    if (u + 1 < NU) GT_SEED(u + 1);
                    mma_chain<DP>(afr[(GT_EXP & 2) ? 0 : (sb & 1)], bq[qt], accp[u % 3]);
                }
            }
            hit_now = false;
            if (u > 0 && (!TWO || ((act >> (u - 1)) & 1u))) {
                // one predicate per lane: the largest of its 16 scores against the query's threshold (a v_max3 tree
                // and one compare in the MFMA issue gaps; the per-element compares are redone on the cold admission path)
                const float tq = (GT_EXP & 8) ? INFINITY : thr[pqt];
                const f32x16& pa = accp[(u - 1) % NACC];
                if (GT_EXP & 8) asm volatile("" ::"v"(pa));   // keep the matrix work alive
#pragma unroll
                for (int t3 = 0; t3 < 5; ++t3) mx[t3] = fmaxf(fmaxf(pa[3 * t3], pa[3 * t3 + 1]), pa[3 * t3 + 2]);
                const float m5 = fmaxf(fmaxf(mx[0], mx[1]), mx[2]);
                const float m6 = fmaxf(fmaxf(mx[3], mx[4]), pa[15]);
                const float m16 = fmaxf(m5, m6);
                any_hit = m16 > tq;
                // MODE 2: ... or some row of the sub-tile may want this query (same arithmetic as the cold path:
                // rounding is monotone, so max(acc) + hq > min(g) whenever one acc_i + hq > g_i)
                if constexpr (MODE == 2) {
                    // the two compares write lane masks: their union is tested on the scalar side, no per-lane flag
                    unsigned long long hm_ = __ballot(m16 > tq);
                    if (!(GT_EXP & 8)) hm_ |= __ballot(m16 + hnq[pqt] > gms[psb]);
                    hit_now = hm_ != 0ull;
                }
            }
#if GT_SEL_PIPE
            if (u > 0 && u < NU) {
                // The LDS reads of this step (seeds of unit u+1, A fragments of the next sub-tile) go right behind the
                // first MFMA: the wait that MFMA needs (its seeds, read one unit ago) then has nothing young in front
                // of it, and the new reads have the rest of the chain to land.
                // Then: one MFMA, a few of the max / compare instructions of the previous unit, ...
#pragma unroll
                for (int i = 0; i < (PREC == 1 ? 3 * DP / 16 : PREC == 2 ? NS1 : DP / 2); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
#if GT_SEL_DSFIRST
                    if (i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);   // DS read (as many as there are)
#endif
                    __builtin_amdgcn_sched_group_barrier(0x002, PREC == 1 ? 1 : PREC == 2 ? (TWO ? GT_SEL_TWO_VPM : 3) : 1, 0);   // VALU
                    __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);   // SALU
                }
            }
#endif
            if (u > 0) {
                if constexpr (MODE == 2) {
                    // no branch in the unit loop: the units that need the cold path are only noted (bit = unit)
                    hitmask |= hit_now ? (1u << (u - 1)) : 0u;
                } else {
                    GT_ADMIT(accp[(u - 1) % NACC], any_hit, mx, psb, pqt);
                }
            }
        }
        }
        if constexpr (TWO) {
            // two-stage scoring: the pairs that passed stage one go to the queue of the cold launch (sym_cold_kernel)
            if (GT_EXP & 256) hitmask = 0u;
            cold_tile = hitmask != 0u;
            if (__builtin_expect(hitmask != 0u, 0)) {   // wave-uniform
                // one entry per (sub-tile, PAIR of query tiles): bit sb * QT/2 + pair.  They are written at the top of the
                // next iteration, in front of the tile loads issued there (GT_QUEUE_FLUSH)
#pragma unroll
                for (int p_ = 0; p_ < (BN / 32) * (QT / 2); ++p_) pend_pm |= ((hitmask >> (2 * p_)) & 3u) ? (1u << p_) : 0u;
                pend_t = uint32_t(t);
            }
        } else if constexpr (MODE == 2) {
            if (GT_EXP & 256) hitmask = 0u;   // experiment: the unit loop with its compares, no cold path
            cold_tile = hitmask != 0u;
            if (__builtin_expect(hitmask != 0u, 0)) {
                const unsigned long long ts_ = prof ? __builtin_readcyclecounter() : 0ull;
                // the full thresholds / seeds of the lane's queries (two-stage scoring keeps the half ones in registers)
                float thrF[QT], hnqF[QT];
#pragma unroll
                for (int qt_ = 0; qt_ < QT; ++qt_) {
                    if constexpr (TWO) {
                        const int64_t qp_ = qblock + (w * QT + qt_) * 32 + li;
                        const bool live_ = qp_ < nq && thr[qt_] != INFINITY;   // (INFINITY: pad query, or list overflowed)
                        thrF[qt_] = live_ ? thr_in[qp_] : INFINITY;
                        hnqF[qt_] = qp_ < nq ? hneg[qp_] : -INFINITY;
                    } else {
                        thrF[qt_] = thr[qt_];
                        hnqF[qt_] = hnq[qt_];
                    }
                }
                while (hitmask) {   // wave-uniform
                    const int pu = __ffs(int(hitmask)) - 1;
                    const int csb = pu / QT, cqt = pu % QT;
                    if constexpr (QT == 2 && GT_SEL_PAIRCOLD) hitmask &= ~(3u << (csb * QT));
                    else hitmask &= hitmask - 1u;
                    Frag<DP, PREC> ca;
                    ca.load(tb + (csb * 32 + li) * LDP, h, aswz);
                    f32x16 cs, cacc;
#pragma unroll
                    for (int g_ = 0; g_ < 4; ++g_) {
                        // (two-stage scoring staged the half seeds: the full ones come from memory)
                        const float4 hv_ = TWO ? *reinterpret_cast<const float4*>(hneg + size_t(tbase) + csb * 32 + 8 * g_ + 4 * h)
                                               : *reinterpret_cast<const float4*>(hb + csb * 32 + 8 * g_ + 4 * h);
                        cs[4 * g_ + 0] = hv_.x;
                        cs[4 * g_ + 1] = hv_.y;
                        cs[4 * g_ + 2] = hv_.z;
                        cs[4 * g_ + 3] = hv_.w;
                    }
                    if constexpr (QT == 2 && GT_SEL_PAIRCOLD) {
                        // an unflagged tile of the pair has no passing element (its unit test is implied by every
                        // element test), so scoring both unconditionally adds nothing but four MFMAs
                        f32x16 cacc1 = cs;
                        cacc = cs;
                        mma_chain<DP>(ca, bq[0], cacc);
                        mma_chain<DP>(ca, bq[QT - 1], cacc1);
                        GT_ADMIT2P(cacc, cacc1, cs, csb);
                    } else if (QT == 1 || cqt == 0) {
                        cacc = cs;
                        mma_chain<DP>(ca, bq[0], cacc);
                        GT_ADMIT2(cacc, cs, csb, 0);
                    } else {
                        cacc = cs;
                        mma_chain<DP>(ca, bq[QT - 1], cacc);
                        GT_ADMIT2(cacc, cs, csb, QT - 1);
                    }
                    if (prof) n_adm += 1;
                }
                if (prof) t_adm += __builtin_readcyclecounter() - ts_;
            }
        }
        if (MODE == 0) {
            // ---- list maintenance: lane (li, h) owns half h of query (qt, li) ----
            const bool phase_a = (n_a && level == 0) || own_sched;
            // forced cuts: after the last tile of level 0 (settles the seed threshold) and of level samp2_level
            const bool end_a = level_end && ((level == 0 && samp_end > 0) || (level > 0 && level == samp2_level));
            const uint32_t mkeep = end_a ? uint32_t(level == 0 ? samp_end : samp2_keep)
                                         : phase_a ? uint32_t(samp_keep) : uint32_t(MKEEP);
            const uint32_t trig = phase_a ? uint32_t(samp_trig) : uint32_t(TRIGH);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const unsigned long long full =
                    end_a ? __ballot(fill[qt] + __shfl_xor(fill[qt], 32) > mkeep) : __ballot(fill[qt] > trig);
                uint32_t need = uint32_t(full) | uint32_t(full >> 32);
                if ((dbg & 18) && need) {   // experiment: no selection, just pretend the lists were compacted
                    if (need & (1u << li)) fill[qt] = mkeep / 2u;
                    need = 0;
                }
                if (need) {
                    const unsigned long long ts_ = prof ? __builtin_readcyclecounter() : 0ull;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's list stores have reached L2
                    while (need) {
                        n_cmp += 1;
                        const int L = __ffs(int(need)) - 1;
                        need &= need - 1;
                        const uint32_t n0 = __shfl(fill[qt], L);
                        const uint32_t n1 = __shfl(fill[qt], L + 32);
                        uint64_t* lp = lists + size_t(qblock + (w * QT + qt) * 32 + L) * lstride;
                        uint32_t kept;
                        const float tc = compact_list<NT, false>(lp, n0, n1, lane, kept, mkeep);
                        if (li == L) {
                            const uint32_t k0 = kept < mkeep / 2u ? kept : mkeep / 2u;
                            fill[qt] = h ? kept - k0 : k0;
                            thr[qt] = fmaxf(thr[qt], tc);
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (prof) t_cmp += __builtin_readcyclecounter() - ts_;
                }
            }
        }

        if constexpr (C::GLDS && NBUF == 3) {
            // the next tile must have landed, the one after it may stay in flight: a wave issued NPW (+1) loads for it,
            // all younger than the next tile's.  A tile that went through the cold path has younger memory operations
            // of other kinds in the queue - it simply drains everything.
            // (the two-stage kernel only issues stores nobody reads back: they can make a counted wait stricter, never
            //  let it pass early - loads return in order among themselves)
            if (it + 2 < t_end && (TWO || !cold_tile))
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NPW) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (C::GLDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next tile are in LDS
        } else {
            if (!(GT_EXP & (4 | 32)) && it + 1 < t_end) GT_STAGE_STORE(buf ^ 1, 1);
        }
        if (!(GT_EXP & (4 | 16))) {
            const unsigned long long ts_ = prof ? __builtin_readcyclecounter() : 0ull;
            if constexpr (C::GLDS && NBUF == 3) {
                // a tile is still in flight: __syncthreads() would drain it (its fence waits vmcnt(0) while an LDS-DMA is
                // pending) - the LDS reads of this tile have to be back, nothing else
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
            if (prof) t_bar += __builtin_readcyclecounter() - ts_;
        }
        if (prof && level_end && level == 0) {
            t_lvl0 = __builtin_readcyclecounter() - t_start;
            n_adm_lvl0 = n_adm;
        }
        t = t_next;
        t_step = t_step_next;
        level = level_next;
        if (MODE == 2) {
            if (NBUF != 3) rel_n3 = walk_rel(it + 3);
            rel_c = rel_n1;
            rel_n1 = rel_n2;
            rel_n2 = rel_n3;
        }
    }
    if (prof && lane == 0) {
        unsigned long long* o = prof + (size_t(blockIdx.x) * 4 + w) * 8;
        o[0] = t_adm; o[1] = t_cmp; o[2] = t_bar; o[3] = n_cmp; o[4] = n_adm;
        o[5] = t_lvl0; o[6] = n_adm_lvl0; o[7] = __builtin_readcyclecounter() - t_start;
    }

    GT_QUEUE_FLUSH();
    if (TWO && lane == 0) sy.qcount[size_t(blockIdx.x) * 4 + w] = qfill;
    if (TWO && lane == 0 && n_scored != 0u) atomicAdd(sy.qspill_count + 1, n_scored);   // (statistics: units scored by stage one)
    // ---- finalisation: gather every list into slots [0, count), publish count and the last admission threshold ----
    if (MODE == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll 1
            for (int L = 0; L < 32; ++L) {
                const uint32_t n0 = __shfl(fill[qt], L);
                const uint32_t n1 = __shfl(fill[qt], L + 32);
                const float t_prev = __shfl(thr[qt], L);
                const int ql = (w * QT + qt) * 32 + L;
                uint64_t* lp = lists + size_t(qblock + ql) * lstride;
                uint32_t kept;
                const float t = compact_list<NT, true>(lp, n0, n1, lane, kept, uint32_t(final_keep));
                if (lane == 0) {
                    counts[qblock + ql] = kept;
                    // every rejected database row scored <= thr_out (-inf: nothing was ever rejected)
                    thr_out[qblock + ql] = (dbg & 1) ? -INFINITY : fmaxf(t_prev, t);
                }
            }
        }
    }
}

// ---- query ordering: nearest of L landmark rows (single float16 chain, approximate by design) --------------------
// The candidate pass is fastest when the 32 queries of a wave share their neighbourhood: a database row that beats
// one query's threshold then beats many of them in the same unit, and the (wave-wide) admission path is entered
// once instead of once per query.  gt_query_order (gt_order.hip) therefore processes the queries grouped by the
// nearest of L sample rows.  This kernel finds that row: Yl = the L landmark rows of the compact hi-plane copy, hl
// their score seeds; one query per lane exactly as in the candidate kernel, the landmark stream shared through LDS.
template <int DP>
__global__ __launch_bounds__(256) void assign_cells_kernel(const float* __restrict__ Yc, const float* __restrict__ Yl,
                                                           const float* __restrict__ hl, const int64_t q0,
                                                           const int32_t nq, const int32_t L, const int32_t need,
                                                           uint32_t* __restrict__ cell, float* __restrict__ thr0,
                                                           float* __restrict__ best_out) {
    // best_out (optional): the score of the row's nearest landmark (how far the row is from every cell: gt_order.hip's
    // outlier cell)
    constexpr int RW = DP / 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 31, h = lane >> 5;
    const int64_t q = int64_t(blockIdx.x) * 128 + w * 32 + li;
    const int64_t qc = q < nq ? q : int64_t(nq) - 1;
    Frag<DP, 2> bq;
    bq.load(Yc + (q0 + qc) * RW, h);
    // gm[e] / gl[e]: best score among the landmarks that land in accumulator slot e of this lane, and the unit that
    // held it - 16 disjoint sets of database rows, so at least 16 rows score >= min(gm) (32 with the other half-wave's
    // sets): a valid (if loose, ~rank n/100) starting threshold for a selection of the `need` <= 32 best over the same
    // scores (single-chain arithmetic, same seeds, same chain order).  Independent per slot: no serial max chain.
    float gm[16];
    int gl[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        gm[e] = -INFINITY;
        gl[e] = 0;
    }
    // the four waves share the landmark stream through LDS: chunks of 32 landmarks (padded rows, two buffers)
    constexpr int LDPA = RW + 4;
    __shared__ __attribute__((aligned(16))) float lmk[2][32 * LDPA];
    __shared__ __attribute__((aligned(16))) float lsd[2][32];
    auto stage = [&](int l0, int buf) {
        for (int f = threadIdx.x; f < 32 * (RW / 4); f += 256) {
            const int r = f / (RW / 4), c4 = f % (RW / 4);
            *reinterpret_cast<float4*>(&lmk[buf][r * LDPA + 4 * c4]) =
                *reinterpret_cast<const float4*>(Yl + size_t(l0 + r) * RW + 4 * c4);
        }
        if (threadIdx.x < 32) lsd[buf][threadIdx.x] = hl[l0 + threadIdx.x];
    };
    stage(0, 0);
    __syncthreads();
    for (int l0 = 0, buf = 0; l0 < L; l0 += 32, buf ^= 1) {
        if (l0 + 32 < L) stage(l0 + 32, buf ^ 1);   // the other buffer was released by the barrier that ended the last step
        Frag<DP, 2> a;
        a.load(&lmk[buf][li * LDPA], h);
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 hv = *reinterpret_cast<const float4*>(&lsd[buf][8 * g + 4 * h]);
            acc[4 * g + 0] = hv.x;
            acc[4 * g + 1] = hv.y;
            acc[4 * g + 2] = hv.z;
            acc[4 * g + 3] = hv.w;
        }
        mma_chain<DP>(a, bq, acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool better = acc[e] > gm[e];
            gm[e] = better ? acc[e] : gm[e];
            gl[e] = better ? l0 : gl[e];
        }
        __syncthreads();
    }
    float best = gm[0];
    uint32_t bidx = uint32_t(gl[0] + 4 * h);
#pragma unroll
    for (int e = 1; e < 16; ++e) {
        const uint32_t idx = uint32_t(gl[e] + 8 * (e >> 2) + 4 * h + (e & 3));
        const bool better = gm[e] > best || (gm[e] == best && idx < bidx);
        best = better ? gm[e] : best;
        bidx = better ? idx : bidx;
    }
    float gmin = gm[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) gmin = fminf(gmin, gm[e]);
    // need <= 16: either half-wave's 16 sets will do (take the tighter); need <= 32: all 32 sets of the two halves
    const float gother = __shfl_xor(gmin, 32);
    gmin = need <= 16 ? fmaxf(gmin, gother) : fminf(gmin, gother);
    const float ob = __shfl_xor(best, 32);
    const uint32_t oi = __shfl_xor(bidx, 32);
    if (ob > best || (ob == best && oi < bidx)) bidx = oi;
    if (h == 0 && q < nq) {
        cell[q] = bidx;
        thr0[q] = gmin;
        if (best_out) best_out[q] = fmaxf(best, ob);
    }
}

// The same assignment with two query tiles per wave and 64 landmarks per barrier (launches of many rows: C3).  assign_cells_kernel
// reads every landmark fragment from the LDS once per 32 queries and meets at a barrier every 32 landmarks - at L = n / 512
// landmarks the launch ran at 13 % of the float16 peak, waiting on both.  Here a fragment serves 64 queries and a barrier 64
// landmarks.  Scores, ties and therefore cells / thr0 / best are those of assign_cells_kernel bit for bit (the same chain per
// (query, landmark) pair, the same first-wins rule per accumulator slot in the same landmark order).
template <int DP>
__global__ __launch_bounds__(256) void assign_cells2_kernel(const float* __restrict__ Yc, const float* __restrict__ Yl,
                                                            const float* __restrict__ hl, const int64_t q0,
                                                            const int32_t nq, const int32_t L, const int32_t need,
                                                            uint32_t* __restrict__ cell, float* __restrict__ thr0,
                                                            float* __restrict__ best_out) {
    constexpr int RW = DP / 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 31, h = lane >> 5;
    const int64_t qa = int64_t(blockIdx.x) * 256 + w * 64 + li;   // this lane's queries: qa and qa + 32
    Frag<DP, 2> bq[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t q = qa + 32 * t;
        const int64_t qc = q < nq ? q : int64_t(nq) - 1;
        bq[t].load(Yc + (q0 + qc) * RW, h);
    }
    float gm[2][16];
    int gl[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            gm[t][e] = -INFINITY;
            gl[t][e] = 0;
        }
    constexpr int LDPA = RW + 4;
    __shared__ __attribute__((aligned(16))) float lmk[2][64 * LDPA];
    __shared__ __attribute__((aligned(16))) float lsd[2][64];
    auto stage = [&](int l0, int buf) {
        const int rows = L - l0 < 64 ? L - l0 : 64;   // (L is a multiple of 32)
        for (int f = threadIdx.x; f < rows * (RW / 4); f += 256) {
            const int r = f / (RW / 4), c4 = f % (RW / 4);
            *reinterpret_cast<float4*>(&lmk[buf][r * LDPA + 4 * c4]) =
                *reinterpret_cast<const float4*>(Yl + size_t(l0 + r) * RW + 4 * c4);
        }
        if (threadIdx.x < rows) lsd[buf][threadIdx.x] = hl[l0 + threadIdx.x];
    };
    stage(0, 0);
    __syncthreads();
    for (int l0 = 0, buf = 0; l0 < L; l0 += 64, buf ^= 1) {
        if (l0 + 64 < L) stage(l0 + 64, buf ^ 1);   // the other buffer was released by the barrier that ended the last step
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (l0 + 32 * s < L) {   // (uniform)
                Frag<DP, 2> a;
                a.load(&lmk[buf][(32 * s + li) * LDPA], h);
                f32x16 seed;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 hv = *reinterpret_cast<const float4*>(&lsd[buf][32 * s + 8 * g + 4 * h]);
                    seed[4 * g + 0] = hv.x;
                    seed[4 * g + 1] = hv.y;
                    seed[4 * g + 2] = hv.z;
                    seed[4 * g + 3] = hv.w;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x16 acc = seed;
                    mma_chain<DP>(a, bq[t], acc);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const bool better = acc[e] > gm[t][e];
                        gm[t][e] = better ? acc[e] : gm[t][e];
                        gl[t][e] = better ? l0 + 32 * s : gl[t][e];
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t q = qa + 32 * t;
        float best = gm[t][0];
        uint32_t bidx = uint32_t(gl[t][0] + 4 * h);
#pragma unroll
        for (int e = 1; e < 16; ++e) {
            const uint32_t idx = uint32_t(gl[t][e] + 8 * (e >> 2) + 4 * h + (e & 3));
            const bool better = gm[t][e] > best || (gm[t][e] == best && idx < bidx);
            best = better ? gm[t][e] : best;
            bidx = better ? idx : bidx;
        }
        float gmin = gm[t][0];
#pragma unroll
        for (int e = 1; e < 16; ++e) gmin = fminf(gmin, gm[t][e]);
        const float gother = __shfl_xor(gmin, 32);
        gmin = need <= 16 ? fmaxf(gmin, gother) : fminf(gmin, gother);
        const float ob = __shfl_xor(best, 32);
        const uint32_t oi = __shfl_xor(bidx, 32);
        if (ob > best || (ob == best && oi < bidx)) bidx = oi;
        if (h == 0 && q < nq) {
            cell[q] = bidx;
            thr0[q] = gmin;
            if (best_out) best_out[q] = fmaxf(best, ob);
        }
    }
}

// The wave regions of the collect launch -> one dense queue (order does not matter): a workgroup takes 16 regions, their slots
// in the dense queue come from ONE atomic (round 6: one per region - 27 000 returning atomics on one address, ~90 per
// microsecond - was half of the launch's 0.63 ms on the manifold set).  total (pre-zeroed): entries copied; flag
// (pre-zeroed): the largest region count.
__global__ __launch_bounds__(256) void sym_queue_compact_kernel(const uint2* __restrict__ regions,
                                                                const uint32_t* __restrict__ cnts, const int64_t nwaves,
                                                                const int rcap, uint2* __restrict__ dense,
                                                                const uint32_t dense_cap, uint32_t* __restrict__ total,
                                                                uint32_t* __restrict__ flag) {
    __shared__ uint32_t off_s[17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r0 = int64_t(blockIdx.x) * 16;
    if (threadIdx.x < 64) {   // wave 0: the 16 counts, their exclusive scan, the reservation
        const uint32_t c = (lane < 16 && r0 + lane < nwaves) ? cnts[r0 + lane] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t up = uint32_t(__shfl_up(int(inc), o));
            if (lane >= o) inc += up;
        }
        const uint32_t sum = uint32_t(__shfl(int(inc), 15));
        uint32_t mx = c;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = max(mx, uint32_t(__shfl_xor(int(mx), o)));
        uint32_t base = 0u;
        if (lane == 0 && sum != 0u) {
            base = atomicAdd(total, sum);
            atomicMax(flag, mx);   // the fullest region (statistics)
        }
        base = uint32_t(__shfl(int(base), 0));
        if (lane < 16) off_s[lane] = base + inc - c;
        if (lane == 0) off_s[16] = sum;
    }
    __syncthreads();
    if (off_s[16] == 0u) return;
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + w * 4 + i;
        if (r >= nwaves) break;
        const uint32_t c = cnts[r];
        const uint32_t base = off_s[w * 4 + i];
        const uint2* src = regions + size_t(r) * size_t(rcap);
        for (uint32_t e = uint32_t(lane); e < c; e += 64u)
            if (base + e < dense_cap) dense[base + e] = src[e];
    }
}

// ... and the spill area behind them (spill_n entries, already dense): appended with one atomic per workgroup
__global__ __launch_bounds__(256) void sym_queue_spill_kernel(const uint2* __restrict__ spill, const uint32_t* __restrict__ spill_count,
                                                              const uint32_t spill_cap, uint2* __restrict__ dense,
                                                              const uint32_t dense_cap, uint32_t* __restrict__ total,
                                                              uint32_t* __restrict__ flag) {
    __shared__ uint32_t base_s;
    const uint32_t n = *spill_count;
    if (n > spill_cap) {   // the spill area overflowed: entries are missing - the caller must start over
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flag + 1, 1u);
        return;
    }
    const uint32_t i0 = blockIdx.x * 256u;
    if (i0 >= n) return;
    const uint32_t c = n - i0 < 256u ? n - i0 : 256u;
    if (threadIdx.x == 0) base_s = atomicAdd(total, c);
    __syncthreads();
    if (threadIdx.x < c && base_s + threadIdx.x < dense_cap) dense[base_s + threadIdx.x] = spill[i0 + threadIdx.x];
}

// Admission of one (64 queries x 32 rows) unit of the cold launch: the tests GT_ADMIT2P makes, filed with fewer
// instructions.  The compare results are used where they are born - as 64-bit lane masks in scalar registers: counts per
// database row are scalar popcounts, a lane's slot inside a row's batch is its prefix count in the mask (v_mbcnt) - instead of
// being packed into per-lane bit words and unpacked again.  One returning atomic instruction reserves the batches of all
// 32 database rows (the count of row (e, h) sits in lane e + 16 h), one more per query tile the forward lists.
// A0 / A1: scores of the wave's two query tiles (lane (li, h): query li of the tile, database rows 8 (e >> 2) + 4 h + (e & 3)),
// SD: the rows' seeds (removed again from what is filed under the database rows).  List order differs from GT_ADMIT2P's;
// the re-rank does not depend on it.
// Returns (wave-uniform) whether the unit issued - and waited for - a returning atomic: every memory operation of the wave older
// than it (loads and LDS-DMA copies return in order with returning atomics) has then completed too.
template <typename SYM>
__device__ __forceinline__ bool cold_admit(const f32x16& A0, const f32x16& A1, const f32x16& SD, const float tq0, const float tq1,
                                           const float hq0, const float hq1, const uint32_t qpos0, const uint32_t qpos1,
                                           const uint32_t tbase, const float* __restrict__ gglob, const bool tr_on,
                                           const int lane, const int li, const int h, const SYM& sy) {
    const uint32_t tcap = uint32_t(sy.tcap);
    // ---- forward: how many of the lane's 16 rows pass each query's threshold ----
    uint32_t nf0 = 0u, nf1 = 0u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        nf0 += (A0[e] > tq0) ? 1u : 0u;
        nf1 += (A1[e] > tq1) ? 1u : 0u;
    }
    uint32_t k0 = 0u, k1 = 0u;
    if (nf0) k0 = atomicAdd(&sy.tcounts[qpos0], nf0);
    if (nf1) k1 = atomicAdd(&sy.tcounts[qpos1], nf1);
    // ---- transposed: entries per database row, both query tiles together ----
    uint32_t vcnt = 0u;
    if (tr_on) {   // wave-uniform
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) {
            const float4 gv = *reinterpret_cast<const float4*>(gglob + 8 * g_ + 4 * h);
            const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) {
                const int e = 4 * g_ + c_;
                const unsigned long long c0 = __ballot((A0[e] + hq0) > ge[c_]), c1 = __ballot((A1[e] + hq1) > ge[c_]);
                const uint32_t lo = uint32_t(__popcll(c0 & 0xFFFFFFFFull) + __popcll(c1 & 0xFFFFFFFFull));
                const uint32_t hi = uint32_t(__popcll(c0 >> 32) + __popcll(c1 >> 32));
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vcnt) : "s"(lo), "n"(e));        // lane e      <- lo
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vcnt) : "s"(hi), "n"(e + 16));   // lane e + 16 <- hi
            }
        }
    }
    uint32_t vbase = 0u;
    if (vcnt != 0u) {   // (lanes 0 ... 31 only: the others never received a count)
        const uint32_t e = uint32_t(lane) & 15u, hh = uint32_t(lane) >> 4;
        vbase = atomicAdd(&sy.tcounts[tbase + 8u * (e >> 2) + 4u * hh + (e & 3u)], vcnt);
    }
    // All three reservations are waited for HERE, in front of the first list store: the compiler then knows of no memory
    // operation in flight behind this point (the list stores are inline asm it does not track), and inserts no wait of its own
    // further down - at the top of the caller's next unit it would be a `vmcnt(0)` that also sits out the stores just issued.
    asm volatile("" ::"v"(k0), "v"(k1), "v"(vbase));
    // ---- forward stores ----
    if (nf0) {
        uint64_t* lp = sy.tlists + size_t(qpos0) * size_t(tcap);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (A0[e] > tq0) {
                if (k0 < tcap) list_store(lp + k0, cand_pack(A0[e], tbase + uint32_t(8 * (e >> 2) + 4 * h + (e & 3))));
                ++k0;
            }
    }
    if (nf1) {
        uint64_t* lp = sy.tlists + size_t(qpos1) * size_t(tcap);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (A1[e] > tq1) {
                if (k1 < tcap) list_store(lp + k1, cand_pack(A1[e], tbase + uint32_t(8 * (e >> 2) + 4 * h + (e & 3))));
                ++k1;
            }
    }
    // ---- transposed stores: row (e, h)'s batch starts at the value its atomic returned; inside it the entries of query
    //      tile 0 come first, each in the order of the lanes ----
    if (tr_on && __ballot(vcnt != 0u) != 0ull) {
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) {
            const float4 gv = *reinterpret_cast<const float4*>(gglob + 8 * g_ + 4 * h);   // (again: not kept across the atomics)
            const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) {
                const int e = 4 * g_ + c_;
                const float s0 = A0[e] + hq0, s1 = A1[e] + hq1;
                const bool p0 = s0 > ge[c_], p1 = s1 > ge[c_];
                const unsigned long long c0 = __ballot(p0), c1 = __ballot(p1);
                if ((c0 | c1) != 0ull) {   // wave-uniform
                    const uint32_t bl = uint32_t(__builtin_amdgcn_readlane(int(vbase), e));
                    const uint32_t bh = uint32_t(__builtin_amdgcn_readlane(int(vbase), e + 16));
                    const uint32_t n0l = uint32_t(__popcll(c0 & 0xFFFFFFFFull)), n0h = uint32_t(__popcll(c0 >> 32));
                    const uint32_t n1l = uint32_t(__popcll(c1 & 0xFFFFFFFFull));
                    // prefix counts over the whole wave: the upper half sees the lower half's entries too - taken out of its
                    // base
                    const uint32_t pre0 = __builtin_amdgcn_mbcnt_hi(uint32_t(c0 >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(c0), 0u));
                    const uint32_t pre1 = __builtin_amdgcn_mbcnt_hi(uint32_t(c1 >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(c1), 0u));
                    const uint32_t b0 = h ? bh - n0l : bl;                       // tile 0 entries of my half's row
                    const uint32_t b1 = h ? bh + n0h - n1l : bl + n0l;           // tile 1 entries behind them
                    const uint32_t j = tbase + uint32_t(8 * (e >> 2) + (e & 3)) + 4u * uint32_t(h);
                    uint64_t* lj = sy.tlists + size_t(j) * size_t(tcap);
                    const uint32_t t0 = b0 + pre0, t1 = b1 + pre1;
                    if (p0 && t0 < tcap) list_store(lj + t0, cand_pack(s0 - SD[e], qpos0));
                    if (p1 && t1 < tcap) list_store(lj + t1, cand_pack(s1 - SD[e], qpos1));
                }
            }
        }
    }
    return __ballot((nf0 | nf1 | vcnt) != 0u) != 0ull;
}

// ---- deferred cold pass of the two-stage symmetric collect: one wave per queue entry -------------------------------
// Entry {q64, d32}: the 64 queries [64 q64, 64 q64 + 64) (the two query tiles a wave of the collect launch owns)
// against the 32 database rows [32 d32, 32 d32 + 32).  Operands come straight from the sorted compact copy; the block
// is scored exactly as the collect kernel's own cold path scores it (same chain, same seeds), then tested and filed
// by the same code (GT_ADMIT2P).
#ifndef GT_SEL_COLD_ADMIT
#define GT_SEL_COLD_ADMIT 1   // 1: cold_admit (scalar-mask filing), 0: the collect kernel's GT_ADMIT2P
#endif
#ifndef GT_SEL_COLD_WAVES
#define GT_SEL_COLD_WAVES 3   // waves per SIMD the cold kernel's registers are cut for (it fits 4 at 122 VGPRs)
#endif
#ifndef GT_SEL_COLD_EPW
#define GT_SEL_COLD_EPW 16   // queue entries per wave of the cold launch (consecutive entries mostly share their queries)
#endif
// Round 6: the operands of unit i + 1 travel while unit i is scored and filed.  A unit used to be a chain of three dependent
// round trips - its queue entry, its operand rows (32 rows of the compact copy + their seeds and thresholds), the returning
// atomics that reserve its list slots - at four waves per SIMD: 1.8 ns per unit on the chip, ~14 000 cycles per unit and wave,
// all of it latency (13.6 of the manifold set's 43 ms for 7.6 M units).  Now the wave's 16 entries arrive in ONE load, and the
// rows, seeds and thresholds of the NEXT unit are copied global -> LDS (LDS-DMA: no registers - the version that prefetched into
// a second set of fragment registers lost more to occupancy than the round trip it hid, round 5) right after this unit's
// fragments have been read out of the same buffer: the copy is in flight behind the matrix instructions and the atomics.  The
// copies are inline asm (hipcc would fence every LDS read behind every copy it knows of); they are older than the unit's
// atomics, so a unit that filed anything has waited for them by the time its atomics returned (cold_admit's return value) and
// the list stores behind - which nobody waits for - stay out of the wait; a unit that filed nothing waits for the copies alone.
#ifndef GT_SEL_COLD_LDS
#define GT_SEL_COLD_LDS 1
#endif
template <int DP>
__global__ __launch_bounds__(256, GT_SEL_COLD_WAVES) void sym_cold_kernel(const float* __restrict__ Yp, const float* __restrict__ hneg,
                                                       const float* __restrict__ thr_in, const int32_t nq,
                                                       const int32_t ntiles, const SymDev sy) {
    using C = SelCfg<DP, 2>;
    constexpr int RWC = C::RW;   // row width of the compact copy
    constexpr int QT = 2;
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int64_t en0 = (int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * GT_SEL_COLD_EPW;
    if (en0 >= int64_t(sy.qn)) return;
    const int w = 0;
    constexpr int BQ = 128 * GT_SEL_TWO_QT, TPB = BQ / C::BN;   // query blocks of the collect launch
    const int T = ntiles, NB = T / TPB, H = (NB - 1) / 2;
    Frag<DP, 2> bq[QT], ca;
    float thrF[QT], hnqF[QT], thr[QT];
    uint32_t have_q = 0xFFFFFFFFu;
#if GT_SEL_COLD_LDS
    // (one wave per workgroup: launch_sym_cold) 32 rows of the compact copy | 2 x {32 seeds | 32 row thresholds}
    constexpr int TILE_B = 32 * C::RB;
    static_assert(TILE_B % 1024 == 0, "whole 1 KiB pieces");
    __shared__ __attribute__((aligned(16))) unsigned char cold_lds[TILE_B + 2 * 256];
    typedef __attribute__((address_space(3))) void lds_void_;
    const uint32_t lds_tile = uint32_t(size_t((lds_void_*)cold_lds)), lds_sg = lds_tile + uint32_t(TILE_B);
    const float* tileA = reinterpret_cast<const float*>(cold_lds);
    const float* sgbuf = reinterpret_cast<const float*>(cold_lds + TILE_B);
    static_assert(GT_SEL_COLD_EPW <= 64, "one queue entry per lane");
    uint2 ent_mine = make_uint2(0u, 0u);
    if (lane < GT_SEL_COLD_EPW && en0 + lane < int64_t(sy.qn)) ent_mine = sy.queue[en0 + lane];
    const int n_ent = int(int64_t(sy.qn) - en0 < int64_t(GT_SEL_COLD_EPW) ? int64_t(sy.qn) - en0 : int64_t(GT_SEL_COLD_EPW));
    const bool with_g = sy.own_only == 0;   // (own-only launches read no row thresholds: sy.g may be null)
    auto issue = [&](const int i_) {
        const uint32_t tb_ = uint32_t(__builtin_amdgcn_readlane(int(ent_mine.y), i_)) * 32u;
        const char* src_ = reinterpret_cast<const char*>(Yp) + size_t(tb_) * size_t(C::RB) + size_t(lane) * 16;
#pragma unroll
        for (int k_ = 0; k_ < TILE_B / 1024; ++k_)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_ + k_ * 1024), "s"(lds_tile + uint32_t(k_ * 1024)) : "memory");
        // lanes 0 .. 31: the rows' seeds; lanes 32 .. 63: their thresholds (g) - one 256-byte piece, double buffered (the
        // thresholds are read again behind the atomics, when the next unit's copy is already under way)
        const float* sp_ = lane < 32 ? hneg + size_t(tb_) + lane : (with_g ? sy.g + size_t(tb_) + (lane - 32) : hneg + size_t(tb_) + (lane - 32));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(sp_), "s"(lds_sg + uint32_t((i_ & 1) * 256)) : "memory");
    };
    issue(0);
    bool had = false;
    for (int ce_ = 0; ce_ < n_ent; ++ce_) {
        const uint2 ent = make_uint2(uint32_t(__builtin_amdgcn_readlane(int(ent_mine.x), ce_)),
                                     uint32_t(__builtin_amdgcn_readlane(int(ent_mine.y), ce_)));
#else
    // a wave takes a few consecutive entries: the bound pass files the units of a group of 64 queries together (and the
    // collect launch most of a wave's), so the query fragments, thresholds and seeds are loaded once per run
    for (int ce_ = 0; ce_ < GT_SEL_COLD_EPW; ++ce_) {
        const int64_t en_ = en0 + ce_;
        if (en_ >= int64_t(sy.qn)) break;   // wave-uniform
        const uint2 ent = sy.queue[en_];
#endif
        const int64_t qblock = int64_t(ent.x) * (QT * 32);
        const uint32_t tbase = ent.y * 32u;
        // does the sub-tile's block take these queries as ITS candidates too?  (position of its tile in the walk of the
        // queries' block, as in the collect kernel)
        int rel = int(ent.y / uint32_t(C::BN / 32)) - int(qblock / BQ) * TPB;
        if (rel < 0) rel += T;
        const bool tr_on = sy.own_only == 0 && rel >= TPB && rel < TPB * (1 + H);
        if (ent.x != have_q) {   // wave-uniform
            have_q = ent.x;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int64_t qg = qblock + qt * 32 + li;
                const int64_t qc = qg < nq ? qg : int64_t(nq) - 1;
                bq[qt].load(Yp + qc * RWC, h);
                thrF[qt] = qg < nq ? thr_in[qc] : INFINITY;
                hnqF[qt] = qg < nq ? hneg[qc] : -INFINITY;
                thr[qt] = thrF[qt];
            }
#if GT_SEL_COLD_LDS
            // (the queries' loads are waited for here, once per group: a compiler-inserted wait for them further down would be a
            //  `vmcnt(0)` behind the next unit's copy - the very round trip the copy is meant to hide)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                for (int s_ = 0; s_ < DP / 16; ++s_) asm volatile("" ::"v"(bq[qt].hi[s_]));
                asm volatile("" ::"v"(thrF[qt]), "v"(hnqF[qt]));
            }
#endif
        }
        f32x16 cs;
#if GT_SEL_COLD_LDS
        // the unit's operands have landed: a unit that filed something waited for its atomics, which are younger than the copy
        if (ce_ == 0 || !had) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ca.load(tileA + li * RWC, h);
        const float* sg_ = sgbuf + (ce_ & 1) * 64;
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) {
            const float4 hv_ = *reinterpret_cast<const float4*>(sg_ + 8 * g_ + 4 * h);
            cs[4 * g_ + 0] = hv_.x;
            cs[4 * g_ + 1] = hv_.y;
            cs[4 * g_ + 2] = hv_.z;
            cs[4 * g_ + 3] = hv_.w;
        }
        // fragments and seeds are in registers: the tile and the other {seeds | thresholds} buffer may be overwritten
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ce_ + 1 < n_ent) issue(ce_ + 1);
        const float* gglob = sg_ + 32;
#else
        ca.load(Yp + (size_t(tbase) + li) * RWC, h);
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) {
            const float4 hv_ = *reinterpret_cast<const float4*>(hneg + size_t(tbase) + 8 * g_ + 4 * h);
            cs[4 * g_ + 0] = hv_.x;
            cs[4 * g_ + 1] = hv_.y;
            cs[4 * g_ + 2] = hv_.z;
            cs[4 * g_ + 3] = hv_.w;
        }
        const float* gglob = sy.g + size_t(tbase);
#endif
        f32x16 cacc = cs, cacc1 = cs;
        mma_chain<DP>(ca, bq[0], cacc);
        mma_chain<DP>(ca, bq[QT - 1], cacc1);
#if GT_SEL_COLD_ADMIT || GT_SEL_COLD_LDS
        const bool had_ = cold_admit(cacc, cacc1, cs, thrF[0], thrF[QT - 1], hnqF[0], hnqF[QT - 1], uint32_t(qblock + li),
                                     uint32_t(qblock + 32 + li), tbase, gglob, tr_on, lane, li, h, sy);
#if GT_SEL_COLD_LDS
        had = had_;
#else
        (void)had_;
#endif
#else
        GT_ADMIT2P(cacc, cacc1, cs, 0);
#endif
    }
    (void)thr;
    (void)w;
}

// ---- the cold pass in a LOCAL FRAME -----------------------------------------------------------------------------------
// The single float16 chain rounds every coordinate of x sc to 11 bits: the score of a pair is off by up to ~2^-10 |x||y|, an
// error that grows with the distance of the points from the ORIGIN, while the quantity the pass decides on - is this pair closer
// than either row's radius? - lives at the scale of a cell.  On C3 (|x| ~ 46, neighbours ~ 11 apart) the margin the thresholds
// have to leave for it lets a third of the collected candidates through for nothing (35 % of 117 M: DESIGN 4.4), and the
// re-rank, the largest kernel of the build, evaluates them all.
// Here a unit is scored in the frame of its QUERIES: o = the centre of the 64 query rows (gt_sym_group_centres), both
// operands are float16 roundings of (x - o) sc formed on the fly from the float32 points in sorted order.  Everything is then
// EXACT arithmetic on perturbed points x' = o + u / sc, |x' - x| <= eps |x - o| (eps = 2^-11: the float16 rounding):
//     -|x' - y'|^2 / 2 = u.w - |w|^2 / 2 - |u|^2 / 2           (u, w: the rounded operands; norms of the ROUNDED rows, float32)
// and |x' - y'| differs from |x - y| by at most delta = eps (|u| + |w|) (triangle inequality).  A pair row q needs listed
// (|x - y| <= r_q, rloc) has |x' - y'| <= r_q + delta: the test  s > T_q := -(r_q + delta)^2 / 2 - E  never loses it (E: the
// float32 accumulation, (DP + 16) 2^-24 (|u| + |w|)^2); the same s against T_j decides for the database row.  delta and E
// are formed per unit from the largest operand norms (a few DPP steps).  With |u|, |w| ~ a cell radius the margin in d^2 is
// ~ 2 r delta ~ 0.3 on C3 where the global frame's 2 e was ~ 3.5.
// What is FILED is an upper bound of the true score in the global frame's terms, (|x|^2 - (|x' - y'| - delta)^2) / 2 scaled -
// so that the re-rank's bound on what lies beyond a full table (bound_of_score of the 257th key) holds as it stands.
template <typename SYM>
__device__ __forceinline__ bool cold_admit_local(const f32x16& A0, const f32x16& A1, const float tq0, const float tq1,
                                                 const float hq0, const float hq1, const float fq0, const float fq1,
                                                 const uint32_t qpos0, const uint32_t qpos1, const uint32_t tbase,
                                                 const float* __restrict__ Tl, const float* __restrict__ Bl, const bool tr_on,
                                                 const int lane, const int li, const int h, const SYM& sy) {
    // tq: forward thresholds on the accumulator (T_q - hu_q); hq: -|u_q|^2 / 2; fq: what turns an accumulator into the filed
    // forward score; Tl / Bl (LDS, by row of the sub-tile): the rows' thresholds T_j and their filing terms
    const uint32_t tcap = uint32_t(sy.tcap);
    uint32_t nf0 = 0u, nf1 = 0u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        nf0 += (A0[e] > tq0) ? 1u : 0u;
        nf1 += (A1[e] > tq1) ? 1u : 0u;
    }
    uint32_t k0 = 0u, k1 = 0u;
    if (nf0) k0 = atomicAdd(&sy.tcounts[qpos0], nf0);
    if (nf1) k1 = atomicAdd(&sy.tcounts[qpos1], nf1);
    uint32_t vcnt = 0u;
    if (tr_on) {   // wave-uniform
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) {
            const float4 gv = *reinterpret_cast<const float4*>(Tl + 8 * g_ + 4 * h);
            const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) {
                const int e = 4 * g_ + c_;
                const unsigned long long c0 = __ballot((A0[e] + hq0) > ge[c_]), c1 = __ballot((A1[e] + hq1) > ge[c_]);
                const uint32_t lo = uint32_t(__popcll(c0 & 0xFFFFFFFFull) + __popcll(c1 & 0xFFFFFFFFull));
                const uint32_t hi = uint32_t(__popcll(c0 >> 32) + __popcll(c1 >> 32));
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vcnt) : "s"(lo), "n"(e));
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vcnt) : "s"(hi), "n"(e + 16));
            }
        }
    }
    uint32_t vbase = 0u;
    if (vcnt != 0u) {
        const uint32_t e = uint32_t(lane) & 15u, hh = uint32_t(lane) >> 4;
        vbase = atomicAdd(&sy.tcounts[tbase + 8u * (e >> 2) + 4u * hh + (e & 3u)], vcnt);
    }
    asm volatile("" ::"v"(k0), "v"(k1), "v"(vbase));   // (all three waited for here, in front of the first list store: cold_admit)
    if (nf0) {
        uint64_t* lp = sy.tlists + size_t(qpos0) * size_t(tcap);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (A0[e] > tq0) {
                if (k0 < tcap) list_store(lp + k0, cand_pack(A0[e] + fq0, tbase + uint32_t(8 * (e >> 2) + 4 * h + (e & 3))));
                ++k0;
            }
    }
    if (nf1) {
        uint64_t* lp = sy.tlists + size_t(qpos1) * size_t(tcap);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (A1[e] > tq1) {
                if (k1 < tcap) list_store(lp + k1, cand_pack(A1[e] + fq1, tbase + uint32_t(8 * (e >> 2) + 4 * h + (e & 3))));
                ++k1;
            }
    }
    if (tr_on && __ballot(vcnt != 0u) != 0ull) {
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) {
            const float4 gv = *reinterpret_cast<const float4*>(Tl + 8 * g_ + 4 * h);
            const float4 bv = *reinterpret_cast<const float4*>(Bl + 8 * g_ + 4 * h);
            const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
            const float be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) {
                const int e = 4 * g_ + c_;
                const float s0 = A0[e] + hq0, s1 = A1[e] + hq1;
                const bool p0 = s0 > ge[c_], p1 = s1 > ge[c_];
                const unsigned long long c0 = __ballot(p0), c1 = __ballot(p1);
                if ((c0 | c1) != 0ull) {   // wave-uniform
                    const uint32_t bl = uint32_t(__builtin_amdgcn_readlane(int(vbase), e));
                    const uint32_t bh = uint32_t(__builtin_amdgcn_readlane(int(vbase), e + 16));
                    const uint32_t n0l = uint32_t(__popcll(c0 & 0xFFFFFFFFull)), n0h = uint32_t(__popcll(c0 >> 32));
                    const uint32_t n1l = uint32_t(__popcll(c1 & 0xFFFFFFFFull));
                    const uint32_t pre0 = __builtin_amdgcn_mbcnt_hi(uint32_t(c0 >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(c0), 0u));
                    const uint32_t pre1 = __builtin_amdgcn_mbcnt_hi(uint32_t(c1 >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(c1), 0u));
                    const uint32_t b0 = h ? bh - n0l : bl;
                    const uint32_t b1 = h ? bh + n0h - n1l : bl + n0l;
                    const uint32_t j = tbase + uint32_t(8 * (e >> 2) + (e & 3)) + 4u * uint32_t(h);
                    uint64_t* lj = sy.tlists + size_t(j) * size_t(tcap);
                    const uint32_t t0 = b0 + pre0, t1 = b1 + pre1;
                    if (p0 && t0 < tcap) list_store(lj + t0, cand_pack(s0 + be[c_], qpos0));
                    if (p1 && t1 < tcap) list_store(lj + t1, cand_pack(s1 + be[c_], qpos1));
                }
            }
        }
    }
    return __ballot((nf0 | nf1 | vcnt) != 0u) != 0ull;
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// one row of the sorted float32 points -> the lane's share of the MFMA operand in the frame o (osc = o sc, LDS): features
// [16 s + 8 h, 16 s + 8 h + 8) of every k-step s as float16 (u = fl16(fl32(x sc - o sc))); returns the lane's share of |u|^2
template <int DP>
__device__ __forceinline__ float frag_load_local(Frag<DP, 2>& f, const float* __restrict__ row, const int xs_d,
                                                 const float* __restrict__ osc, const float sc, const int h) {
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < DP / 16; ++s) {
        const int f0 = 16 * s + 8 * h;
        float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
        if (f0 < xs_d) xa = *reinterpret_cast<const float4*>(row + f0);          // (xs_d is a multiple of 4; columns beyond
        if (f0 + 4 < xs_d) xb = *reinterpret_cast<const float4*>(row + f0 + 4);  //  it are zeros in the points AND the centre)
        const float4 oa = *reinterpret_cast<const float4*>(osc + f0), ob = *reinterpret_cast<const float4*>(osc + f0 + 4);
        const float u[8] = {fmaf(xa.x, sc, -oa.x), fmaf(xa.y, sc, -oa.y), fmaf(xa.z, sc, -oa.z), fmaf(xa.w, sc, -oa.w),
                            fmaf(xb.x, sc, -ob.x), fmaf(xb.y, sc, -ob.y), fmaf(xb.z, sc, -ob.z), fmaf(xb.w, sc, -ob.w)};
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = _Float16(u[e]);
        f.hi[s] = v;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            f16x2 p;
            p[0] = v[e];
            p[1] = v[e + 1];
            part = __builtin_amdgcn_fdot2(p, p, part, false);
        }
    }
    return part;
}

#ifndef GT_SEL_COLD_LOCAL_WAVES
#define GT_SEL_COLD_LOCAL_WAVES 4
#endif
template <int DP>
__global__ __launch_bounds__(256, GT_SEL_COLD_LOCAL_WAVES) void sym_cold_local_kernel(const float* __restrict__ hneg, const int32_t nq,
                                                                               const int32_t ntiles, const SymDev sy, const int xcd_chunk) {
    using C = SelCfg<DP, 2>;
    constexpr int QT = 2;
    // (one wave per workgroup: launch_sym_cold)
    __shared__ __attribute__((aligned(16))) float lds_all[1][DP + 96];
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5, wv = 0;
    float* osc = lds_all[wv];   // [DP]  centre of the wave's queries, scaled
    float* hwl = osc + DP;      // [32]  -|w_j|^2 / 2 of the sub-tile's rows
    float* Tl = hwl + 32;       // [32]  their thresholds T_j
    float* Bl = Tl + 32;        // [32]  their filing terms
    // (xcd_chunk >= 0: the waves are dealt to the XCDs in runs of that many - gt_xcd_item: the queue is in the order of the query
    //  groups, the units of a cluster's groups walk the same few hundred sub-tiles, and an L2 should fetch them once)
    const int64_t en0 = (xcd_chunk >= 0 ? gt_xcd_item(blockIdx.x, gridDim.x, xcd_chunk) : int64_t(blockIdx.x)) * GT_SEL_COLD_EPW;
    if (en0 >= int64_t(sy.qn)) return;
    constexpr int BQ = 128 * GT_SEL_TWO_QT, TPB = BQ / C::BN;
    const int T = ntiles, NB = T / TPB, H = (NB - 1) / 2;
    const float sc = sy.sc;
    const int xs_d = sy.xs_d;
#if GT_SEL_COLD_LDS
    // Round 6, as in sym_cold_kernel: the 32 float32 rows of the NEXT unit (xs_d floats each: up to 8 KiB) and their {radius,
    // seed} pairs are copied global -> LDS while this unit is scored and filed; the fragments of this unit were converted out of
    // the same buffer before the copy was issued.
    __shared__ __attribute__((aligned(16))) unsigned char rows_lds[32 * DP * 4 + 256];
    typedef __attribute__((address_space(3))) void lds_void_;
    const uint32_t lds_rows = uint32_t(size_t((lds_void_*)rows_lds)), lds_rh = lds_rows + uint32_t(32 * DP * 4);
    const float* rowsL = reinterpret_cast<const float*>(rows_lds);
    const float* rhL = reinterpret_cast<const float*>(rows_lds + 32 * DP * 4);   // [32] radii | [32] seeds
    const int tile_bytes = 32 * xs_d * 4;
    auto issue = [&](const uint32_t d32_) {
        // (uniform base in scalar registers, the lane's offset in ONE transient register: per-lane 64-bit addresses held across
        //  the loop pushed the kernel over its 128 registers - and a reload from scratch is a `vmcnt(0)` behind the copy)
        const char* base_ = reinterpret_cast<const char*>(sy.xs) + int64_t(d32_) * tile_bytes;
        const int64_t left_ = (int64_t(sy.xs_n) - int64_t(d32_) * 32) * xs_d * 4;   // bytes of the point set from this row on
        const int valid_ = int(left_ < int64_t(tile_bytes) ? left_ : int64_t(tile_bytes));   // (rows beyond it: not copied, masked below)
#pragma unroll
        for (int k_ = 0; k_ < DP / 8; ++k_) {   // 1 KiB pieces of at most 32 x DP x 4 bytes
            if (k_ * 1024 < tile_bytes) {       // (uniform)
                const uint32_t lo_ = uint32_t(lane) * 16u + uint32_t((k_ >> 2) * 4096);
                if (int(lo_) + (k_ & 3) * 1024 + 16 <= valid_)
                    // (the instruction's offset moves BOTH addresses: global base + lane offset + offset -> M0 + 16 lane + offset)
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(lo_), "s"(base_),
                                 "s"(lds_rows + uint32_t((k_ >> 2) * 4096)), "n"((k_ & 3) * 1024) : "memory");
            }
        }
        // lanes 0 .. 31: the rows' radii, lanes 32 .. 63: their seeds - one copy per half, each from a scalar base (a lane's
        // LDS slot is its place in the wave, whatever the execution mask)
        const uint32_t j_ = d32_ * 32u + uint32_t(li);
        const uint32_t jo_ = (j_ < uint32_t(sy.xs_n) ? j_ : uint32_t(sy.xs_n) - 1u) * 4u;
        if (h == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(jo_), "s"(sy.rloc), "s"(lds_rh) : "memory");
        else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(jo_), "s"(hneg), "s"(lds_rh) : "memory");
    };
    bool had = false;
#endif
    const float EPS = 4.90189e-4f;                     // 2^-11 (1 + 2^-8): float16 rounding of an operand, with head-room
    const float C1 = float(DP + 16) * 5.9604645e-8f;   // float32 accumulation of a score, per (|u| + |w|)^2
    Frag<DP, 2> bq[QT], ca;
    float hu[QT], rq[QT], hgq[QT];
    float umax = 0.f;
    uint32_t have_q = 0xFFFFFFFFu;
    // the wave's entries in one load (lane l holds entry en0 + l): the loop reads them from there - one round trip less per unit
    static_assert(GT_SEL_COLD_EPW <= 64, "one queue entry per lane");
    uint2 ent_mine = make_uint2(0u, 0u);
    if (lane < GT_SEL_COLD_EPW && en0 + lane < int64_t(sy.qn)) ent_mine = sy.queue[en0 + lane];
#if GT_SEL_COLD_LDS
    issue(uint32_t(__builtin_amdgcn_readlane(int(ent_mine.y), 0)));
#endif
    for (int ce_ = 0; ce_ < GT_SEL_COLD_EPW; ++ce_) {
        const int64_t en_ = en0 + ce_;
        if (en_ >= int64_t(sy.qn)) break;   // wave-uniform
        const uint2 ent = make_uint2(uint32_t(__builtin_amdgcn_readlane(int(ent_mine.x), ce_)),
                                     uint32_t(__builtin_amdgcn_readlane(int(ent_mine.y), ce_)));
        const int64_t qblock = int64_t(ent.x) * (QT * 32);
        const uint32_t tbase = ent.y * 32u;
        int rel = int(ent.y / uint32_t(C::BN / 32)) - int(qblock / BQ) * TPB;
        if (rel < 0) rel += T;
        const bool tr_on = sy.own_only == 0 && rel >= TPB && rel < TPB * (1 + H);
        if (ent.x != have_q) {   // wave-uniform
            have_q = ent.x;
            __builtin_amdgcn_wave_barrier();
            if (lane < DP) osc[lane] = sy.gcen[size_t(ent.x) * DP + lane];
            if (DP > 64 && lane + 64 < DP) osc[lane + 64] = sy.gcen[size_t(ent.x) * DP + lane + 64];
            __builtin_amdgcn_wave_barrier();
            float um = 0.f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int64_t qg = qblock + qt * 32 + li;
                const bool real = qg < nq;
                const int64_t qc = real ? qg : int64_t(nq) - 1;
                float part = frag_load_local<DP>(bq[qt], sy.xs + qc * int64_t(xs_d), xs_d, osc, sc, h);
                part += __uint_as_float(lane_xor_b32(__float_as_uint(part), 32));
                hu[qt] = real ? -0.5f * part : -INFINITY;
                rq[qt] = real ? sy.rloc[qc] : -INFINITY;
                hgq[qt] = real ? hneg[qc] : 0.f;
                um = fmaxf(um, real ? part : 0.f);
            }
            umax = sqrtf(wave_max_f32(um));
#if GT_SEL_COLD_LDS
            // (nothing of the queries' loads stays in flight behind this block: sym_cold_kernel)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) asm volatile("" ::"v"(rq[qt]), "v"(hgq[qt]), "v"(hu[qt]));
#endif
        }
        // the sub-tile's rows in the same frame
        const int64_t jg = int64_t(tbase) + li;
        const bool jreal = jg < int64_t(sy.xs_n);
        const int64_t jc = jreal ? jg : int64_t(sy.xs_n) - 1;
#if GT_SEL_COLD_LDS
        // the unit's rows have landed: a unit that filed something waited for its atomics, which are younger than the copy
        if (ce_ == 0 || !had) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float nw = frag_load_local<DP>(ca, rowsL + li * xs_d, xs_d, osc, sc, h);
        const float rj_l = rhL[li], hgj_l = rhL[32 + li];
        // rows, radii and seeds are in registers: the buffer takes the next unit's
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ce_ + 1 < GT_SEL_COLD_EPW && en_ + 1 < int64_t(sy.qn)) issue(uint32_t(__builtin_amdgcn_readlane(int(ent_mine.y), ce_ + 1)));
        nw += __uint_as_float(lane_xor_b32(__float_as_uint(nw), 32));
        const float rj = jreal ? rj_l : -INFINITY;
        const float hgj = jreal ? hgj_l : 0.f;
        (void)jc;
#else
        float nw = frag_load_local<DP>(ca, sy.xs + jc * int64_t(xs_d), xs_d, osc, sc, h);
        nw += __uint_as_float(lane_xor_b32(__float_as_uint(nw), 32));
        const float rj = jreal ? sy.rloc[jc] : -INFINITY;
        const float hgj = jreal ? hneg[jc] : 0.f;
#endif
        __builtin_amdgcn_wave_barrier();
        if (h == 0) hwl[li] = jreal ? -0.5f * nw : -INFINITY;
        __builtin_amdgcn_wave_barrier();
        f32x16 cs;
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) {
            const float4 hv_ = *reinterpret_cast<const float4*>(hwl + 8 * g_ + 4 * h);
            cs[4 * g_ + 0] = hv_.x;
            cs[4 * g_ + 1] = hv_.y;
            cs[4 * g_ + 2] = hv_.z;
            cs[4 * g_ + 3] = hv_.w;
        }
        f32x16 cacc = cs, cacc1 = cs;
        mma_chain<DP>(ca, bq[0], cacc);
        mma_chain<DP>(ca, bq[QT - 1], cacc1);
        // margins of this unit
        const float wmax = sqrtf(wave_max_f32(jreal ? nw : 0.f));
        const float span = umax + wmax;
        const float delta = EPS * span + 1e-6f;
        const float Es = 1.0625f * C1 * span * span + 1e-30f;
        auto thr_of = [&](const float r) {   // T = -(r + delta)^2 / 2 - E, rounded down; a row that needs nothing: +inf
            const float rd = r + delta;
            return r >= 0.f ? -0.5000005f * rd * rd - Es : INFINITY;
        };
        auto file_of = [&](const float r, const float hg) {   // what turns -|x' - y'|^2 / 2 into the filed score
            return r >= 0.f ? (r + delta) * delta + Es - hg : 0.f;
        };
        if (h == 0) {
            Tl[li] = thr_of(rj);
            Bl[li] = file_of(rj, hgj);
        }
        __builtin_amdgcn_wave_barrier();
        const float T0 = thr_of(rq[0]), T1 = thr_of(rq[QT - 1]);
        const bool had_ = cold_admit_local(cacc, cacc1, T0 - hu[0], T1 - hu[QT - 1], hu[0], hu[QT - 1], hu[0] + file_of(rq[0], hgq[0]),
                                           hu[QT - 1] + file_of(rq[QT - 1], hgq[QT - 1]), uint32_t(qblock + li),
                                           uint32_t(qblock + 32 + li), tbase, Tl, Bl, tr_on, lane, li, h, sy);
#if GT_SEL_COLD_LDS
        had = had_;
#else
        (void)had_;
#endif
    }
}

// centres of the groups of 64 consecutive sorted rows, scaled: gcen[g][c] = sc x mean of column c over the group's real rows
// (columns beyond the row length: 0).  One wave per group, lane = column.
__global__ __launch_bounds__(256) void sym_group_centres_kernel(const float* __restrict__ xs, const int xs_d, const int64_t n,
                                                                const int64_t g_first, const int64_t g_end, const int dp,
                                                                const float sc, float* __restrict__ gcen) {
    const int lane = threadIdx.x & 63;
    const int64_t g = g_first + int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (g >= g_end) return;
    const int64_t p0 = g * 64, p1 = p0 + 64 < n ? p0 + 64 : n;
    for (int c = lane; c < dp; c += 64) {
        float acc = 0.f;
        if (c < xs_d)
            for (int64_t p = p0; p < p1; ++p) acc += xs[p * xs_d + c];
        gcen[g * dp + c] = p1 > p0 ? (acc / float(p1 - p0)) * sc : 0.f;
    }
}

int launch_queue_compact(gt_ctx* ctx, const SelectArgs& a) {
    // a.lists: the dense queue (capacity a.cap entries); a.counts: [0] total, [1] fullest region, [2] spill overflow
    // flag (all pre-zeroed);
    // a.sym.queue / qcount / qcap: the regions; a.nq: number of wave regions
    hipLaunchKernelGGL(sym_queue_compact_kernel, dim3((unsigned)ceil_div64(a.nq, 16)), dim3(256), 0, ctx->stream, a.sym.queue,
                       a.sym.qcount, int64_t(a.nq), a.sym.qcap, reinterpret_cast<uint2*>(a.lists), uint32_t(a.cap), a.counts,
                       a.counts + 1);
    GT_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(sym_queue_spill_kernel, dim3((unsigned)ceil_div64(a.sym.qspill_cap, 256)), dim3(256), 0, ctx->stream,
                       a.sym.qspill, a.sym.qspill_count, uint32_t(a.sym.qspill_cap), reinterpret_cast<uint2*>(a.lists),
                       uint32_t(a.cap), a.counts, a.counts + 1);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

template <int DP>
int launch_sym_cold(gt_ctx* ctx, const SelectArgs& a) {
    if (a.sym.qn <= 0) return GT_OK;
    if (!a.sym.queue || !a.sym.g || !a.sym.tlists || !a.sym.tcounts || a.sym.tcap <= 0)
        GT_FAIL(ctx, GT_E_ARG, "knn_select: the cold pass needs the queue and the lists of the collect launch");
    // (independent waves: one per workgroup, so that a wave with much to file does not hold three idle slots)
    const int wpb = 1;
    if (a.sym.xs != nullptr) {
        // local frame (sym_cold_local_kernel): the centres of the query groups first
        if (!a.sym.gcen || !a.sym.rloc || a.sym.xs_d <= 0 || (a.sym.xs_d & 3) != 0 || a.sym.xs_d > DP || a.sym.xs_n <= 0 || !(a.sym.sc > 0.f))
            GT_FAIL(ctx, GT_E_ARG, "knn_select: the local-frame cold pass needs the sorted float32 points, the group centres and the rows' radii");
        const int64_t g_first = a.sym.gc_count > 0 ? a.sym.gc_first : 0;
        const int64_t g_end = a.sym.gc_count > 0 ? g_first + a.sym.gc_count : a.n_pad / 64;
        hipLaunchKernelGGL(sym_group_centres_kernel, dim3((unsigned)ceil_div64(g_end - g_first, 4)), dim3(256), 0, ctx->stream, a.sym.xs,
                           a.sym.xs_d, int64_t(a.sym.xs_n), g_first, g_end, DP, a.sym.sc, const_cast<float*>(a.sym.gcen));
        hipLaunchKernelGGL((sym_cold_local_kernel<DP>), dim3((unsigned)ceil_div64(a.sym.qn, wpb * GT_SEL_COLD_EPW)), dim3(64 * wpb), 0,
                           ctx->stream, a.hneg, a.nq, int(a.n_pad / SelCfg<DP, 2>::BN), a.sym,
                           64);   // (runs of 64 waves ~ a cluster's query groups per XCD: 1.86 -> 1.80 ms on C3, nothing on the manifold set)
        GT_HIP(ctx, hipGetLastError());
        return GT_OK;
    }
    hipLaunchKernelGGL((sym_cold_kernel<DP>), dim3((unsigned)ceil_div64(a.sym.qn, wpb * GT_SEL_COLD_EPW)), dim3(64 * wpb), 0, ctx->stream, a.Yp, a.hneg,
                       a.thr_in, a.nq, int(a.n_pad / SelCfg<DP, 2>::BN), a.sym);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

template <int DP, int NT, int MODEX, int PREC>
int launch_one(gt_ctx* ctx, const SelectArgs& a) {
    using C = SelCfg<DP, PREC>;
    constexpr int MODE = MODEX == 3 ? 2 : MODEX;   // 3: MODE 2 with two-stage scoring
    constexpr int BQL = MODEX == 3 ? 128 * GT_SEL_TWO_QT : C::BQ;   // query rows per workgroup
    const int64_t nblocks = ceil_div64(a.nq, BQL);
    const int ntiles = int(a.n_pad / C::BN);
    int nsplit = 1;
    if (MODE == 1) {
        // few query blocks: split the database so that ~4 workgroups per CU are in flight
        const int64_t want = int64_t(ctx->n_cu) * 4;
        nsplit = int(std::min<int64_t>(std::max<int64_t>(1, want / nblocks), std::max(1, ntiles / 8)));
        GT_HIP(ctx, hipMemsetAsync(a.counts, 0, size_t(nblocks) * C::BQ * sizeof(uint32_t), ctx->stream));
    }
    auto kern = knn_select_kernel<DP, NT, MODEX, PREC>;
    GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    int(C::LDS_BYTES)));
    size_t lds_bytes = MODE == 2 ? C::LDS_BYTES_SYM : C::LDS_BYTES;
    if (MODE == 2) {
        GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        int(lds_bytes)));
        if (a.n_pad % BQL != 0 || a.nq > a.n_pad || a.sym.tcap <= 0 || a.sym.nseg < 1)
            GT_FAIL(ctx, GT_E_ARG, "knn_select: symmetric collect needs the padded point set as queries and database");
    }
    if (MODE == 0 && a.sym.sched == 1 && a.n_pad % C::BQ != 0)
        GT_FAIL(ctx, GT_E_ARG, "knn_select: the own-neighbourhood schedule needs whole query blocks");
    if (a.dbg & 128) {   // experiment: one workgroup per CU
        lds_bytes += 24 * 1024;
        GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        int(lds_bytes)));
    }
    const bool own_collect = MODE == 2 && a.sym.own_only != 0;   // a rank's own query blocks against every tile (gt_knn_shard.cpp)
    if (own_collect && (a.sym.nblk <= 0 || a.sym.block0 < 0 || int64_t(a.sym.block0) + a.sym.nblk > a.n_pad / BQL || a.sym.shard_world > 1 ||
                        a.sym.walk_list != nullptr))
        GT_FAIL(ctx, GT_E_ARG, "knn_select: the own-only collect needs a block range and whole walks");
    if (!own_collect && a.sym.nblk > 0 &&
        (MODE != 0 || a.sym.sched != 1 || a.sym.block0 < 0 || int64_t(a.sym.block0) + a.sym.nblk > a.n_pad / C::BQ))
        GT_FAIL(ctx, GT_E_ARG, "knn_select: a block range needs the own-neighbourhood schedule");
    const int64_t grid_x = MODE == 2 ? (own_collect ? int64_t(a.sym.nblk) : nblocks) * a.sym.nseg : (a.sym.nblk > 0 ? a.sym.nblk : nblocks);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid_x, (unsigned)nsplit), dim3(256), lds_bytes, ctx->stream, a.Yp,
                       a.hneg, a.Qp, a.qrows, a.q0, a.nq, ntiles, a.lists, a.counts, a.thr_in, a.thr_out, a.cap, a.dbg, a.prof,
                       a.samp_stride, a.samp_keep, a.samp_end, a.samp2_level, a.samp2_keep,
                       (a.samp_trig > 0 && a.samp_trig <= 32 * NT - C::BN / 2) ? a.samp_trig : a.samp_keep / 2 + 24,
                       (a.final_keep > 0 && a.final_keep <= 64 * NT) ? a.final_keep : 16 * NT, a.sym);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

template <int DP, int PREC>
int launch_dp(gt_ctx* ctx, const SelectArgs& a) {
    if (a.mode == 1) return launch_one<DP, 8, 1, PREC>(ctx, a);
    if (a.mode == 5) return launch_queue_compact(ctx, a);
    if (a.mode == 4) {
        if constexpr (PREC == 2 && DP >= 32 && SelCfg<DP, PREC>::QT == 2) return launch_sym_cold<DP>(ctx, a);
        GT_FAIL(ctx, GT_E_ARG, "knn_select: no cold pass for this kernel shape");
    }
    if (a.mode == 2) {
        if constexpr (PREC == 2) {
            if (a.sym.half_steps > 0) {
                // (the unit loop of the two-stage collect scores the 16-column stage-one copy: the DP = 16 unit's kernel)
                if constexpr (DP == 16 && SelCfg<DP, PREC>::QT == 2) {
                    if (a.sym.half_steps != GT_SEL_TWO_NS1(DP) || !a.sym.hh || !a.sym.thrh || !a.sym.gminh || !a.sym.queue ||
                        !a.sym.qcount || a.sym.qcap <= 0 || !a.sym.qspill || !a.sym.qspill_count || a.sym.qspill_cap <= 0)
                        GT_FAIL(ctx, GT_E_ARG, "knn_select: two-stage scoring needs the half seeds and thresholds");
                    return launch_one<DP, 8, 3, PREC>(ctx, a);
                }
                GT_FAIL(ctx, GT_E_ARG, "knn_select: the two-stage unit loop runs on the 16-column stage-one copy");
            }
            return launch_one<DP, 8, 2, PREC>(ctx, a);
        }
        GT_FAIL(ctx, GT_E_ARG, "knn_select: symmetric collect runs on the single-chain arithmetic only");
    }
    switch (a.nt) {
        case 8: return launch_one<DP, 8, 0, PREC>(ctx, a);
        case 32: return launch_one<DP, 32, 0, PREC>(ctx, a);
    }
    GT_FAIL(ctx, GT_E_ARG, "knn_select: unsupported list size");
}

}  // namespace

// This file is compiled once per (precision, padded feature count): -DGT_SEL_PREC=<0|1> -DGT_SEL_DP=<dp>, so the
// instantiations build in parallel; gt_knn_select_dispatch.cpp routes to the right one.
#if !defined(GT_SEL_DP) || !defined(GT_SEL_PREC)
#error "compile with -DGT_SEL_PREC=<0|1|2> -DGT_SEL_DP=<dp>"
#endif
#define GT_CAT3_(a, b, c, d) a##b##c##d
#define GT_CAT3(a, b, c, d) GT_CAT3_(a, b, c, d)
#if GT_SEL_PREC == 2 && !GT_SEL_QT1
int GT_CAT3(gt_launch_assign_cells_p, GT_SEL_PREC, _dp, GT_SEL_DP)(gt_ctx* ctx, const float* Yc, const float* Yl,
                                                                  const float* hl, int64_t q0, int32_t nq, int32_t L,
                                                                  int32_t need, uint32_t* cell, float* thr0, float* best) {
    // (two query tiles per wave from 2^17 rows: below that the 128-row workgroups fill the machine better)
    if (nq >= (1 << 17) && !(ctx->dbg_select & 16384))
        hipLaunchKernelGGL(assign_cells2_kernel<GT_SEL_DP>, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, ctx->stream, Yc,
                           Yl, hl, q0, nq, L, need, cell, thr0, best);
    else
        hipLaunchKernelGGL(assign_cells_kernel<GT_SEL_DP>, dim3((unsigned)ceil_div64(nq, 128)), dim3(256), 0, ctx->stream, Yc,
                           Yl, hl, q0, nq, L, need, cell, thr0, best);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}
#endif
#if GT_SEL_QT1
// narrow variant: one 32-row query tile per wave (128-row workgroups) - twice the workgroups for launches with few rows
int GT_CAT3(gt_launch_select_narrow_p, GT_SEL_PREC, _dp, GT_SEL_DP)(gt_ctx* ctx, const SelectArgs& a) {
    return launch_dp<GT_SEL_DP, GT_SEL_PREC>(ctx, a);
}
#else
int GT_CAT3(gt_launch_select_p, GT_SEL_PREC, _dp, GT_SEL_DP)(gt_ctx* ctx, const SelectArgs& a) {
    return launch_dp<GT_SEL_DP, GT_SEL_PREC>(ctx, a);
}
#endif
