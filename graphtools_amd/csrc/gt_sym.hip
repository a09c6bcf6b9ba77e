// Symmetric candidate pass for self queries over the whole point set (euclidean, single-chain float16 arithmetic).
//
// The candidate kernel scores query x against database row y as x.y - |y|^2/2; the dot product is the same MFMA chain
// whichever of the two rows is the query, so one 32 x 32 result block can serve both directions.  That only works
// against FIXED per-row thresholds (a row's list is then fed by many workgroups and nobody can compact it), so the pass
// runs in two launches over the points in CELL-SORTED order (gt_order.hip: rows grouped by nearest landmark, position
// p <-> row perm[p]; queries and database share the order, a query block's neighbours sit next to it):
//   A  knn_select_kernel<MODE 0, sched 1>: every row against its own neighbourhood - the rows of the sym_cells cells
//      whose landmarks are nearest to the landmarks of the query block's own cells (contiguous ranges in the sorted
//      order) - plus every sym_stride-th tile of the rest, keeping the need_m best rows seen.  sym_thresholds_kernel
//      evaluates those need_m rows exactly (float64): the largest of their keys, D_K, bounds the need_m-th neighbour's,
//      and every row the caller needs - within radius_key_factor x that - provably scores above the fixed threshold
//      thr[p] it derives.
//   B  knn_select_kernel<MODE 2>: query block I streams blocks I .. I + (NB-1)/2 only; each result is tested against
//      the lane's threshold (forward) and the database row's (transposed).  N^2 d MFMA flop instead of 2 N^2 d, no
//      list compaction, no threshold updates; on clustered data almost every unit leaves through the one compare.
//   C  rerank_sym_kernel (gt_rerank.hip): exact float64 keys of the (up to 256) best candidates per row, completeness
//      bound from thr[p]; rows whose lists overflowed go to the usual repair path.
// The roofline line of bench.py prices this pass at the algorithmic 2 N^2 d and reports the executed
// matrix work next to it.
#include <rocprim/device/device_select.hpp>

#include "gt_common.h"
#include "gt_device.h"
#include "gt_knn.h"
#include "gt_knn_select.h"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace {

// rows of the compact hi-plane copy (2*DP bytes = c16 16-byte chunks) and their seeds in cell-sorted order; pad rows:
// zeros / -inf
__global__ __launch_bounds__(256) void gather_sorted_kernel(const uint4* __restrict__ Yc, const float* __restrict__ hneg,
                                                            const int32_t* __restrict__ perm, const int64_t n,
                                                            const int64_t n_pad, const int c16, uint4* __restrict__ Ys,
                                                            float* __restrict__ hs, float* __restrict__ hs_fin) {
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (f >= n_pad * c16) return;
    const int64_t p = f / c16;
    const int c = int(f % c16);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (p < n) {
        const int64_t r = perm[p];
        v = Yc[r * c16 + c];
        if (c == 0) {
            const float hv = hneg[r];
            hs[p] = hv;
            if (hs_fin) hs_fin[p] = hv;
        }
    } else if (c == 0) {
        hs[p] = -INFINITY;
        if (hs_fin) hs_fin[p] = -3.0e38f;   // (gt_seed.hip: pad rows with a finite seed)
    }
    Ys[f] = v;
}

// The points themselves (and their squared norms) in cell-sorted order.  The exact stages - thresholds, re-rank - gather the
// rows of a point's candidates: neighbours in space, i.e. rows of the same few cells, which sit next to each other in this
// copy and are shared through the L2 by the neighbouring queries.  In the caller's row order every gather is a trip to
// the HBM (PMC at N = 1e6, d = 64: the 16 M seed rows of sym_thresholds_kernel fetched 2.2 GB for 256 MB of points).
template <typename T>
__global__ __launch_bounds__(256) void gather_points_kernel(const T* __restrict__ X, const double* __restrict__ xn,
                                                            const int32_t* __restrict__ perm, const int64_t n, const int d,
                                                            T* __restrict__ Xs, double* __restrict__ xns) {
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (f >= n * int64_t(d)) return;
    const int64_t p = f / d;
    const int c = int(f - p * d);
    const int64_t r = perm[p];
    Xs[f] = X[r * int64_t(d) + c];
    if (c == 0) xns[p] = xn[r];
}
// float32 rows of d values -> rows of dp >= d values, zero padded
__global__ __launch_bounds__(256) void gather_points_pad_kernel(const float* __restrict__ X, const double* __restrict__ xn,
                                                                const int32_t* __restrict__ perm, const int64_t n, const int d,
                                                                const int dp, float* __restrict__ Xs, double* __restrict__ xns) {
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (f >= n * int64_t(dp)) return;
    const int64_t p = f / dp;
    const int c = int(f - p * dp);
    const int64_t r = perm[p];
    Xs[f] = c < d ? X[r * int64_t(d) + c] : 0.f;
    if (c == 0) xns[p] = xn[r];
}
__global__ __launch_bounds__(256) void gather_points16_kernel(const uint4* __restrict__ X, const double* __restrict__ xn,
                                                              const int32_t* __restrict__ perm, const int64_t n, const int c16,
                                                              uint4* __restrict__ Xs, double* __restrict__ xns) {
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (f >= n * int64_t(c16)) return;
    const int64_t p = f / c16;
    const int c = int(f - p * c16);
    const int64_t r = perm[p];
    Xs[f] = X[r * int64_t(c16) + c];
    if (c == 0) xns[p] = xn[r];
}

// Fixed thresholds of launch B from the need_m rows launch A kept for sorted position p (list slots [0, kept)):
//   D_K = the largest exact key (float64 squared distance, scikit-learn's association) among them: need_m distinct rows
//   lie within D_K, so the need_m-th neighbour does.  The caller needs every row with d^2 <= rkf * d^2(need_m-th) <=
//   R2 := rkf * D_K (1 + 1e-5); such a row has true score >= (|x|^2 - R2)/2 and scaled candidate score
//   >= sc^2 ((|x|^2 - R2)/2 - e) =: thr (rounded down).
// g[p] = thr[p] + hneg[p] (rounded down): the form the transposed test of launch B compares against;
// gmin[p/32] = min of g over the 32 rows of a sub-tile.
// 16 lanes per row, 16 rows per 256-thread block: the 16 lanes read each candidate row together (coalesced), the
// partial dot products meet in a shuffle tree.  D_K only has to be an upper bound: the summation order differs from the
// re-rank's, a relative 1e-12 covers it.
template <typename T>
__global__ __launch_bounds__(256) void sym_thresholds_kernel(const int64_t n, const int64_t p_first, const int64_t p_last,
                                                             const int32_t* __restrict__ perm, const T* __restrict__ X,
                                                             const int d, const double* __restrict__ xn,
                                                             const float* __restrict__ hs,
                                                             const uint64_t* __restrict__ lists, const int lstride,
                                                             const uint32_t* __restrict__ counts, const int need_m,
                                                             const double* __restrict__ ymax2p, const ErrModel err,
                                                             const double rkf, float* __restrict__ thr,
                                                             float* __restrict__ g, const uint32_t* __restrict__ cell_sorted,
                                                             const int32_t* __restrict__ nbr, const int M,
                                                             unsigned long long* __restrict__ far_total,
                                                             float* __restrict__ farcnt, const int sorted, const int metric,
                                                             const int32_t* __restrict__ cstart, const int32_t* __restrict__ cend,
                                                             const int bn, const int skip_t0, const int skip_t1, const int stride) {
    // stride (> 0: launch A scored every stride-th tile besides the neighbourhoods): only kept rows of THOSE tiles can be far -
    // what the count is extrapolated from.  (A query block takes the tiles around ALL its cells; with coherent cell numbers the
    // block's other cells are cells of the same cluster, and what a row finds around them is no strided find either.)
    // skip_t0 / skip_t1: kept rows of the tiles [skip_t0, skip_t1) are not counted far - the SAMPLE launches (the first rows of
    // the order, asked before every row is seeded) pass their own range: with coherent cell numbers the sample is a compact region,
    // the strided tile that happens to lie inside it is a neighbour of a sixteenth of the sample's rows rather than of a
    // five-hundredth, and extrapolated by the stride it refuted the pass on the very data it is for
    // cstart / cend / bn: first and one-past-last sorted position of every cell, rows per tile - a kept row counts as NEAR (not
    // "far": found by the strided sample) when its cell is one of the M around the row's own OR its tile is one that those cells'
    // rows reach into.  The second half matters since the cells are numbered coherently (gt_order.hip): the tiles at the ends of
    // a neighbourhood cell's run spill into the next cell in number, which is now a cell of the same cluster - rows launch A
    // rightly found there are neighbourhood finds, not strided ones (counted far they refuted the pass: 0.34 per row on C3).
    // metric 1 (cosine, rows normalised): D_K is the largest key 1 - x.y among the kept rows; a row within rkf x that has
    // x.y >= 1 - R2, i.e. true score x.y - |y|^2 / 2 >= 1 - R2 - ymax^2 / 2 - the rest (error bound, scale, rounding) as below.
    // sorted != 0: X / xn are the copies in cell-sorted order (gather_points_kernel): rows are addressed by position
    const int sub = threadIdx.x & 15, lane64 = threadIdx.x & 63;
    const int64_t p = p_first + int64_t(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (p >= p_last) return;   // whole 16-lane group
    float t = INFINITY, gv = INFINITY;
    if (p < n) {
        const int64_t q = sorted ? p : int64_t(perm[p]);
        // Everything the row needs that can be asked for NOW is (round 5: the far count in front of the distances cost the kernel
        // three dependent round trips before the first candidate row was on its way): the first sixteen list entries serve the
        // distances and the far count, the far count's reads (the cells around the row's own, the kept rows' cells) travel
        // while the distances are formed and are looked at behind them.
        const uint32_t kept = counts[p];
        const uint64_t l0 = uint32_t(sub) < kept ? lists[size_t(p) * lstride + sub] : 0ull;
        const uint32_t cme = far_total ? cell_sorted[p] : 0u;
        const T* xq = X + q * int64_t(d);
        const double qs = xn[q];
        // this lane's slice of the query row, up to 8 values in registers (d <= 128): features sub, sub + 16, ... - or, float32 rows
        // of a multiple of four features (vec4: D_K is an upper bound, any summation order serves), features 4 sub ... 4 sub + 3
        // of every 64: a candidate row is then ONE 16-byte load per lane and 64 features instead of four 4-byte ones
        constexpr bool kF32 = sizeof(T) == 4;
        const bool vec4 = kF32 && (d & 3) == 0;
        double xr[8];
        if (vec4) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) xr[4 * u + k2] = (64 * u + 4 * sub + k2 < d) ? double(xq[64 * u + 4 * sub + k2]) : 0.0;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) xr[u] = (sub + 16 * u < d) ? double(xq[sub + 16 * u]) : 0.0;
        }
        uint32_t nb0 = 0xFFFFFFFFu, nb1 = 0xFFFFFFFFu, ccm0 = 0xFFFFFFFEu;
        int lo0 = 1, hi0 = 0, lo1 = 1, hi1 = 0;   // tiles the lane's neighbourhood cells reach into (empty: lo > hi)
        if (far_total) {
            // the M cells around the row's own sit in two registers per lane of the group (M <= 32); every kept row's cell is
            // passed round the group and compared by all lanes at once
            if (sub < M) nb0 = uint32_t(nbr[size_t(cme) * M + sub]);
            if (sub + 16 < M) nb1 = uint32_t(nbr[size_t(cme) * M + sub + 16]);
            if (uint32_t(sub) < kept) ccm0 = cell_sorted[cand_index(l0)];
            if (cstart) {
                if (nb0 != 0xFFFFFFFFu) {
                    const int s0 = cstart[nb0], e0 = cend[nb0];
                    if (s0 >= 0 && e0 > s0) lo0 = s0 / bn, hi0 = (e0 - 1) / bn;
                }
                if (nb1 != 0xFFFFFFFFu) {
                    const int s1 = cstart[nb1], e1 = cend[nb1];
                    if (s1 >= 0 && e1 > s1) lo1 = s1 / bn, hi1 = (e1 - 1) / bn;
                }
            }
        }
        // candidate ids: lane c of the group fetches entry c (need_m <= 64: four rounds at most)
        double dk = 0.0;
        for (uint32_t c0 = 0; c0 < kept; c0 += 16u) {
            int64_t jmine = 0;
            if (c0 + uint32_t(sub) < kept) {
                const uint32_t pj = cand_index(c0 == 0u ? l0 : lists[size_t(p) * lstride + c0 + sub]);
                jmine = sorted ? int64_t(pj) : int64_t(perm[pj]);
            }
            const uint32_t lim = kept - c0 < 16u ? kept - c0 : 16u;
            // TWO candidate rows at a time: their loads are in flight together (one after the other the loop is a chain of
            // shuffle -> address -> load -> reduce latencies: 1.26 ms of sym_prepare on C3; two: 1.18; four - the choice of rounds 4-5,
            // in the landmarks' own order - 1.40; eight 1.58: the registers of the rows in flight cost occupancy, and with the
            // cells numbered coherently the rows come out of the L2, not from further away; round 6)
#ifndef GT_THR_CB
#define GT_THR_CB 2
#endif
            constexpr int CB = GT_THR_CB;
            for (uint32_t c = 0; c < lim; c += uint32_t(CB)) {
                int64_t j4[CB];
                double acc4[CB];
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    j4[i] = __shfl(jmine, int((c + i) & 15u), 16);
                    acc4[i] = 0.0;
                }
                if (vec4) {
                    if constexpr (kF32) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            if (64 * u + 4 * sub < d) {
                                float4 y4[CB];
#pragma unroll
                                for (int i = 0; i < CB; ++i)
                                    y4[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(X) + j4[i] * int64_t(d) + 64 * u + 4 * sub);
#pragma unroll
                                for (int i = 0; i < CB; ++i) {
                                    acc4[i] = fma(xr[4 * u + 0], double(y4[i].x), acc4[i]);
                                    acc4[i] = fma(xr[4 * u + 1], double(y4[i].y), acc4[i]);
                                    acc4[i] = fma(xr[4 * u + 2], double(y4[i].z), acc4[i]);
                                    acc4[i] = fma(xr[4 * u + 3], double(y4[i].w), acc4[i]);
                                }
                            }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (sub + 16 * u < d) {
                            T y4[CB];
#pragma unroll
                            for (int i = 0; i < CB; ++i) y4[i] = X[j4[i] * int64_t(d) + sub + 16 * u];
#pragma unroll
                            for (int i = 0; i < CB; ++i) acc4[i] = fma(xr[u], double(y4[i]), acc4[i]);
                        }
                }
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    double acc = acc4[i];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) acc += lane_xor_f64(acc, o);   // (row operations: no LDS round trip)
                    if (c + i < lim) dk = fmax(dk, gt_pair_key(qs, acc, xn[j4[i]], metric));
                }
            }
        }
        // how many of the kept rows lie outside the M cells around the row's own: when that is most of them the cells say
        // nothing about this point set (launch A found its neighbours in the strided sample) and the thresholds are loose
        uint32_t far = 0;
        if (far_total) {
            for (uint32_t c0 = 0; c0 < 64u; c0 += 16u) {   // (need_m <= 64; the trip count is the same for the whole wave)
                if (__ballot(c0 < kept) == 0ull) break;
                const bool have = c0 + uint32_t(sub) < kept;
                const uint32_t pjm = have ? cand_index(c0 == 0u ? l0 : lists[size_t(p) * lstride + c0 + sub]) : 0u;
                const uint32_t ccm = c0 == 0u ? ccm0 : (have ? cell_sorted[pjm] : 0xFFFFFFFEu);
                const int tjm = have ? int(pjm / uint32_t(bn)) : -1;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const uint32_t cc = uint32_t(__shfl(int(ccm), c, 16));
                    const int tj = __shfl(tjm, c, 16);
                    const unsigned long long hit = __ballot(nb0 == cc || nb1 == cc || (tj >= lo0 && tj <= hi0) || (tj >= lo1 && tj <= hi1));
                    const bool near = ((hit >> (lane64 & 48)) & 0xFFFFull) != 0ull;
                    far += (sub == 0 && cc != 0xFFFFFFFEu && !near && !(tj >= skip_t0 && tj < skip_t1) &&
                            (stride <= 0 || tj % stride == 0)) ? 1u : 0u;
                }
            }
            if (sub == 0 && far) atomicAdd(far_total, (unsigned long long)far);
        }
        dk *= 1.0 + 1e-12;
        const double y2 = ymax2p[0];
        const double e = gt_err_bound(err, qs, y2);
        t = -3.0e38f;
        if (kept >= uint32_t(need_m)) {
            const double R2 = rkf * dk * (1.0 + 1e-5) + 1e-9 * (qs + y2);
            const double smin = (metric == 1) ? (1.0 - R2 - 0.5 * ymax2p[1]) : 0.5 * (qs - R2);
            const double x = (smin - e - 1e-9 * (qs + y2)) / err.inv_sc2;
            t = float(x);
            if (double(t) >= x) t = nextafterf(t, -INFINITY);
            if (!(t > -3.0e38f)) t = -3.0e38f;
        }
        // (how many of the kept rows were far is handed on: sym_orphan_cut_kernel may declare the row an orphan)
        if (farcnt && sub == 0) farcnt[p] = kept >= uint32_t(need_m) ? float(far) : 0.f;
        gv = nextafterf(t + hs[p], -INFINITY);
    }
    if (sub == 0) {
        thr[p] = t;
        g[p] = gv;
    }
}

__global__ __launch_bounds__(256) void sym_gmin_kernel(const int64_t n_pad, const float* __restrict__ g,
                                                       float* __restrict__ gmin) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    float m = p < n_pad ? g[p] : INFINITY;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 31) == 0 && p < n_pad) gmin[p >> 5] = m;
}

// ---- neighbourhood schedule of launch A -------------------------------------------------------------------------
// M nearest landmarks of every landmark (itself first): one wave per landmark, float32 squared differences of the
// float16 landmark rows (approximate by design: any tile list is correct)
__global__ __launch_bounds__(256) void landmark_neighbours_kernel(const _Float16* __restrict__ Yl, const int L, const int DP,
                                                                  const int M, int32_t* __restrict__ nbr,
                                                                  const uint32_t* __restrict__ cell_lo,
                                                                  const uint32_t* __restrict__ cell_hi) {
    // FOUR waves per landmark for the distances (each a quarter of the rows: the loop is a chain of L2 round trips, a quarter as
    // long this way - round 6: 0.25 -> 0.155 ms at L = 4096), wave 0 alone for the selection
    extern __shared__ float dist[];   // [L]
    const int a = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = int(blockDim.x >> 6);
    // (cell_lo / cell_hi, optional: addresses of the first and last cell whose neighbours anybody will ask for - the cells
    //  of a rank's own rows in a row-sharded build)
    if (cell_lo && (uint32_t(a) < *cell_lo || uint32_t(a) > *cell_hi)) return;
    // rows are DP halves = DP/8 16-byte chunks (DP is a multiple of 16): one vector load per 8 features
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const half8* ra = reinterpret_cast<const half8*>(Yl + size_t(a) * DP);
    const int c8 = DP / 8;
    if (DP == 64) {
        // The usual width.  EIGHT LANES PER ROW: lane (r, c) = (lane / 8, lane % 8) reads chunk c - eight features, 16 bytes - of
        // row b0 + r, a load instruction covers eight whole 128-byte rows, and the eight partial sums meet in three exchanges.
        // (One row per lane - 64 rows, 64 cache lines per load instruction, eight instructions per row - kept the texture
        // addresser busy for 0.31 ms at L = 4096 with the arithmetic idle; round 6.)  Any summation order serves: the lists are
        // approximate by design, and every rank of a sharded build runs this same kernel.
        const int r = lane >> 3, c = lane & 7;
        float fa[8];
        {
            const half8 va = ra[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) fa[e] = float(va[e]);
        }
        for (int b0 = wv * 64; b0 < L; b0 += 64 * nwv) {   // (eight loads in flight: the loop is otherwise one L2 round trip per 8 rows)
            half8 vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 8 * u + r;
                vb[u] = reinterpret_cast<const half8*>(Yl + size_t(b < L ? b : 0) * DP)[c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 8 * u + r;
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float df = fa[e] - float(vb[u][e]);
                    acc = fmaf(df, df, acc);
                }
                acc += __uint_as_float(lane_xor_b32(__float_as_uint(acc), 1));
                acc += __uint_as_float(lane_xor_b32(__float_as_uint(acc), 2));
                acc += __uint_as_float(lane_xor_b32(__float_as_uint(acc), 4));
                if (c == 0 && b < L) dist[b] = acc;
            }
        }
    } else {
        for (int b = wv * 64 + lane; b < L; b += 64 * nwv) {
            const half8* rb = reinterpret_cast<const half8*>(Yl + size_t(b) * DP);
            float acc = 0.f;
            for (int c = 0; c < c8; ++c) {
                const half8 va = ra[c], vb = rb[c];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float df = float(va[e]) - float(vb[e]);
                    acc = fmaf(df, df, acc);
                }
            }
            dist[b] = acc;
        }
    }
    __syncthreads();
    if (wv != 0) return;   // (no workgroup barrier below: the selection is wave 0's)
    // The M nearest, in turn (distance, then index).  Every lane keeps the four best of ITS landmarks (b = lane, lane + 64, ...)
    // in order; a round is the wave's arg-min over the lanes' heads, and the lane that won moves its queue up.  A lane whose
    // queue runs dry while it may still hold candidates - more than four of the M nearest in one residue class: rare - scans its
    // landmarks again (the taken ones are marked in the LDS).  (Until round 5 every round scanned all L distances: 32 x 64
    // reads and compares per lane instead of 64 + 32 x a handful.)
    float cv[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int ci[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    auto refill = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cv[q] = INFINITY;
            ci[q] = 0x7fffffff;
        }
        for (int b = lane; b < L; b += 64) {
            float v = dist[b];
            int vi = b;
            if (v < INFINITY) {   // (insertion into the sorted four; equal values keep the smaller index in front: b ascends)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool lt = v < cv[q];
                    const float tv = lt ? cv[q] : v;
                    const int ti = lt ? ci[q] : vi;
                    cv[q] = lt ? v : cv[q];
                    ci[q] = lt ? vi : ci[q];
                    v = tv;
                    vi = ti;
                }
            }
        }
    };
    refill();
    bool dry = false;   // the queue has been emptied by wins: the lane may hold more (a fifth candidate was never recorded)
    for (int m = 0; m < M; ++m) {
        if (__ballot(dry && cv[0] == INFINITY) != 0ull) {   // (wave-uniform; all lanes rebuild - their queues come out the same)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            refill();
            dry = false;
        }
        float best = cv[0];
        int bi = ci[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __uint_as_float(lane_xor_b32(__float_as_uint(best), o));
            const int oi = int(lane_xor_b32(uint32_t(bi), o));
            if (ov < best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if (bi < L && (bi & 63) == lane) {   // this lane's head was taken
            dist[bi] = INFINITY;
            cv[0] = cv[1]; ci[0] = ci[1];
            cv[1] = cv[2]; ci[1] = ci[2];
            cv[2] = cv[3]; ci[2] = ci[3];
            cv[3] = INFINITY; ci[3] = 0x7fffffff;
            dry = true;
        }
        if (lane == 0) nbr[size_t(a) * M + m] = (bi < L) ? bi : a;
    }
}

// rows of every cell in the sorted order: cell_sorted is non-decreasing, cell c = positions [start[c], start[c] + cnt[c])
__global__ __launch_bounds__(256) void cell_ranges_kernel(const uint32_t* __restrict__ cell_sorted, const int64_t n,
                                                          int32_t* __restrict__ start, int32_t* __restrict__ cnt) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t c = cell_sorted[p];
    if (p == 0 || cell_sorted[p - 1] != c) start[c] = int32_t(p);
    if (p + 1 == n || cell_sorted[p + 1] != c) cnt[c] = int32_t(p + 1);   // end position for now (fixed up by the reader)
}

// Tile list of every query block (one wave each): its own tiles, then the tiles of the M cells nearest to each of the
// (up to 8) cells its rows belong to - nearest first, round robin over the block's cells, until max_nb tiles are
// spoken for - in ascending order, then every stride-th tile that is not among them.  The neighbourhood is collected
// as a bitmap over the T tiles in LDS, so every tile appears at most once.
__global__ __launch_bounds__(64) void sym_schedule_kernel(const uint32_t* __restrict__ cell_sorted, const int64_t n,
                                                          const int32_t* __restrict__ start, const int32_t* __restrict__ endp,
                                                          const int32_t* __restrict__ nbr, const int M, const int NB,
                                                          const int BQ, const int BN, const int T, const int stride,
                                                          const int max_nb, const int tile_stride,
                                                          int32_t* __restrict__ tile_list, int32_t* __restrict__ tile_cnt,
                                                          const int block0, const int outlier_cell) {
    // outlier_cell (-1: none): a block that holds rows of the outlier cell (gt_order.hip: rows that belong to no cluster) takes
    // EIGHT times the strided sample.  Such rows have no neighbourhood; what launch A finds for them decides how far their repair
    // has to look.  In the landmarks' own order the block's other cells were strangers from all over the point set and their
    // neighbourhoods a sample of it; numbered coherently they are one region, the isolated rows' sixteenth-best seed lay four
    // times as far and their repair collected 114 000 rows each instead of 25 000 (15 isolated points in 10^6: 26 ms against 20).
    extern __shared__ uint32_t bm[];   // [ceil(T / 32)]
    const int I = block0 + blockIdx.x, lane = threadIdx.x;
    const int nw = (T + 31) / 32;
    for (int w = lane; w < nw; w += 64) bm[w] = 0u;
    __syncthreads();
    const int tpb = BQ / BN;
    const int own_a = I * tpb, own_b = I * tpb + tpb - 1;
    // distinct cells of the block's rows, in order (the cell ids of the sorted rows are non-decreasing: a cell is a
    // run); the first 16 runs count - a block of 256 rows rarely holds more
    __shared__ uint32_t blk_cells[16];
    const int64_t r0 = int64_t(I) * BQ;
    const int per = (BQ + 63) / 64;   // rows per lane
    int nchg = 0;
    uint32_t chg[4] = {0u, 0u, 0u, 0u};
    {
        const int64_t ra = r0 + int64_t(lane) * per;
        uint32_t prevc = (ra > r0 && ra - 1 < n) ? cell_sorted[ra - 1] : 0xFFFFFFFFu;
        for (int k = 0; k < per && k < 4; ++k) {
            const int64_t r = ra + k;
            if (r < n && r < r0 + BQ) {
                const uint32_t c = cell_sorted[r];
                if (c != prevc) chg[nchg++] = c;
                prevc = c;
            }
        }
    }
    int incl0 = nchg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl0, o);
        if (lane >= o) incl0 += v;
    }
    for (int k = 0; k < nchg; ++k)
        if (incl0 - nchg + k < 16) blk_cells[incl0 - nchg + k] = chg[k];
    const int nruns = __shfl(incl0, 63);
    const int nu = nruns < 16 ? nruns : 16;
    __syncthreads();
    // candidate intervals in priority order idx = m * nu + which
    int carry = tpb;   // tiles spoken for so far (the own ones)
    for (int base = 0; base < nu * M; base += 64) {
        const int idx = base + lane;
        int a = 0, b = -1;
        if (idx < nu * M) {
            const int m = idx / nu, which = idx % nu;
            const uint32_t c = blk_cells[which];
            const int cb = nbr[size_t(c) * M + m];
            const int s = start[cb], e = endp[cb];
            if (s >= 0 && e > s) {
                a = s / BN;
                b = (e - 1) / BN;
            }
        }
        int len = b - a + 1;
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const bool take = len > 0 && carry + incl <= max_nb;
        if (take)
            for (int t = a; t <= b; ++t) atomicOr(&bm[t >> 5], 1u << (t & 31));
        carry += __shfl(incl, 63);
    }
    __syncthreads();
    if (lane == 0)   // the own tiles are emitted first, by hand
        for (int t = own_a; t <= own_b; ++t) bm[t >> 5] &= ~(1u << (t & 31));
    __syncthreads();
    int32_t* out = tile_list + size_t(I) * tile_stride;
    int cnt = 0;
    if (lane < tpb) out[lane] = own_a + lane;
    cnt = tpb;
    for (int w0 = 0; w0 < nw; w0 += 64) {
        const int w = w0 + lane;
        uint32_t word = w < nw ? bm[w] : 0u;
        const int c = __popc(word);
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        int pos = cnt + incl - c;
        while (word) {
            const int bit = __ffs(int(word)) - 1;
            word &= word - 1;
            if (pos < tile_stride) out[pos] = w * 32 + bit;
            ++pos;
        }
        cnt += __shfl(incl, 63);
    }
    if (stride > 0) {
        bool has_outlier = false;
        for (int k = 0; k < nu; ++k) has_outlier |= outlier_cell >= 0 && int(blk_cells[k]) == outlier_cell;
        const int stride_b = has_outlier ? (stride >= 8 ? stride / 8 : 1) : stride;
        for (int k0 = 0; k0 * stride_b < T; k0 += 64) {
            const int t = (k0 + lane) * stride_b;
            const bool want = t < T && !((bm[t >> 5] >> (t & 31)) & 1u) && (t < own_a || t > own_b);
            const unsigned long long wm = __ballot(want);
            const int pos = cnt + __popcll(wm & ((1ull << lane) - 1ull));
            if (want && pos < tile_stride) out[pos] = t;
            cnt += __popcll(wm);
        }
    }
    if (lane == 0) tile_cnt[I] = cnt < tile_stride ? cnt : tile_stride;
}

__global__ __launch_bounds__(256) void sum_i32_kernel(const int32_t* __restrict__ v, const int64_t n,
                                                      unsigned long long* __restrict__ out) {
    long long acc = 0;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) acc += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)acc);
}


// Completeness radius a threshold stands for: the re-rank will claim "every row closer than lb is in the list"
// (rerank_sym_kernel, bound_of_score) - the same expression here
__device__ __forceinline__ double sym_row_lb(float t, double qs, double y2, const ErrModel& err) {
    const double e = gt_err_bound(err, qs, y2);
    return (qs - 2.0 * (double(t) * err.inv_sc2 + e)) - 1e-9 * (qs + y2);
}

// Statistics behind the orphan cut, over the rows [p_first, p_last), added into acc (4 doubles, pre-zeroed):
//   acc[0], acc[1]  sum and number of the radii lb of the rows that have a threshold
//   acc[2], acc[3]  sum and number of squared distances between every 64th row and a pseudo-random partner row - an
//                   estimate of the typical squared distance between two unrelated points of the set (2 x its variance)
template <typename T>
__global__ __launch_bounds__(256) void sym_radius_sum_kernel(const int64_t n, const int64_t p_first, const int64_t p_last,
                                                             const int32_t* __restrict__ perm, const T* __restrict__ X,
                                                             const int d, const double* __restrict__ xn,
                                                             const float* __restrict__ thr,
                                                             const double* __restrict__ ymax2p, const ErrModel err,
                                                             double* __restrict__ acc) {
    double s = 0.0, c = 0.0, ps = 0.0, pc = 0.0;
    const int64_t hi = p_last < n ? p_last : n;
    // (the trip count is the wave's: the whole wave takes part in the partner distances below)
    for (int64_t pb = p_first + int64_t(blockIdx.x) * 256 + (threadIdx.x & ~63u); pb < hi; pb += int64_t(gridDim.x) * 256) {
        const int64_t p = pb + (threadIdx.x & 63);
        const bool in = p < hi;
        const float t = in ? thr[p] : INFINITY;
        const int64_t row = in ? int64_t(perm[p]) : 0;
        if (in && t != INFINITY && t > -3.0e38f) {
            const double lb = sym_row_lb(t, xn[row], ymax2p[0], err);
            s += lb > 0.0 ? lb : 0.0;
            c += 1.0;
        }
        // every 64th row against a pseudo-random partner: the WAVE forms the distance (lane k the features k, k + 64, ...; one
        // lane walking the whole row serially was most of this kernel's 0.19 ms)
        const unsigned long long lead = __ballot(in && (p & 63) == 0);
        if (lead != 0ull) {   // wave-uniform
            const int src = __ffsll((long long)lead) - 1;
            const int64_t prow = __shfl(row, src);
            const int64_t pp = __shfl(p, src);
            const int64_t other = int64_t((uint64_t(pp) * 0x9E3779B97F4A7C15ull >> 20) % uint64_t(n));
            const T* a = X + prow * int64_t(d);
            const T* b = X + other * int64_t(d);
            double dd = 0.0;
            for (int k = int(threadIdx.x & 63); k < d; k += 64) {
                const double df = double(a[k]) - double(b[k]);
                dd = fma(df, df, dd);
            }
            dd = wave_sum_f64(dd);
            if (int(threadIdx.x & 63) == src) {
                ps += dd;
                pc += 1.0;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        c += __shfl_xor(c, o);
        ps += __shfl_xor(ps, o);
        pc += __shfl_xor(pc, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (c > 0.0) {
            atomicAdd(acc, s);
            atomicAdd(acc + 1, c);
        }
        if (pc > 0.0) {
            atomicAdd(acc + 2, ps);
            atomicAdd(acc + 3, pc);
        }
    }
}

// Orphans of the two-stage collect: rows whose threshold would pass a good part of the point set through stage one, queue
// it and score it, only to overflow the list.  Two signs:
//   * orphan_far x (rows launch A kept for it from the strided sample, not from the cells around it) >= need_m: its
//     cell says little about where its neighbours are, D_K is loose;
//   * a radius far beyond the typical one (cut x the mean over all rows) that is no longer small against the typical
//     distance between two unrelated points (pair_frac of it: 1/16 when stage one sees 16 coordinates - a quarter of
//     every distance on isotropic data -, 1/4 in the principal frame): launch A found it SOME need_m rows nearby, none
//     of its real neighbours.
// They collect nothing (+inf), keep the rows launch A found (sym_inject_orphans_kernel) and go to the repair pass, which
// starts from those rows' exact distances.  acc = the statistics of sym_radius_sum_kernel over ALL rows.
__global__ __launch_bounds__(256) void sym_orphan_cut_kernel(const int64_t n, const int32_t* __restrict__ perm,
                                                             const double* __restrict__ xn, float* __restrict__ thr,
                                                             const float* __restrict__ farcnt,
                                                             const double* __restrict__ ymax2p, const ErrModel err,
                                                             const double* __restrict__ acc, const double cut,
                                                             const int orphan_far, const int need_m,
                                                             const double pair_frac) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    const float t = thr[p];
    if (t == INFINITY || !(t > -3.0e38f)) return;
    bool orphan = orphan_far > 0 && farcnt && farcnt[p] * float(orphan_far) >= float(need_m);
    // (the rows of the outlier cell - gt_order.hip - are NOT declared orphans: tried in round 4, on manifold-like data the cell
    //  holds the natural tail of the distribution, two thousand rows whose repairs cost 11 ms)
    if (!orphan && cut > 0.0 && acc[1] > 0.0) {
        const double lb = sym_row_lb(t, xn[perm[p]], ymax2p[0], err);
        const double typical_pair = acc[3] > 0.0 ? acc[2] / acc[3] : 0.0;
        orphan = lb > cut * (acc[0] / acc[1]) && lb > typical_pair * pair_frac;
    }
    if (orphan) thr[p] = INFINITY;
}

// the rows launch A kept for an orphan (thr = +inf on a real row, see sym_thresholds_kernel) become the head of its list
__global__ __launch_bounds__(256) void sym_inject_orphans_kernel(const int64_t n, const int64_t p_first, const int64_t p_last,
                                                                 const float* __restrict__ thr,
                                                                 const uint64_t* __restrict__ lists, const int lstride,
                                                                 const uint32_t* __restrict__ counts,
                                                                 uint64_t* __restrict__ tlists, const int tcap,
                                                                 uint32_t* __restrict__ tcounts) {
    const int sub = threadIdx.x & 15;
    const int64_t p = p_first + int64_t(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (p >= p_last || p >= n || thr[p] != INFINITY) return;
    const uint32_t kept = counts[p] < uint32_t(tcap) ? counts[p] : uint32_t(tcap);
    uint32_t base = 0u;
    if (sub == 0) base = atomicAdd(&tcounts[p], kept);
    base = __shfl(base, 0, 16);
    for (uint32_t c = uint32_t(sub); c < kept; c += 16u)
        if (base + c < uint32_t(tcap)) tlists[size_t(p) * size_t(tcap) + base + c] = lists[size_t(p) * lstride + c];
}

// ---- stage-one subspace of the two-stage collect -------------------------------------------------------------------
// Partial distances work in ANY orthonormal frame: |P (x - y)| <= |x - y| for P with orthonormal rows.  The frame that
// keeps the most of every distance is the one of the leading principal directions, so stage one scores a separate
// 16-column copy Z = P x of the points (cell-sorted like the full copy), P = the 16 leading eigenvectors of the
// covariance of a row sample.  (On data whose features all weigh the same - the benchmark mixture - this is no better
// than any 16 features; on data with a spectrum - PCA-reduced input, manifolds - it is the difference between a filter
// and none.)  The full scores of the cold pass, the thresholds and every exact stage keep the original coordinates.
constexpr int kZ = 16;   // columns of the stage-one copy (one MFMA k-step)

// column sums of the sampled rows (every `step`-th row) -> sums[d] (pre-zeroed)
template <typename T>
__global__ __launch_bounds__(256) void sample_colsum_kernel(const T* __restrict__ X, const int64_t n, const int d,
                                                            const int64_t step, const int64_t ns, double* __restrict__ sums) {
    const int c = threadIdx.x % 64, g = threadIdx.x / 64;
    if (c >= d) return;
    double acc = 0.0;
    for (int64_t s = int64_t(blockIdx.x) * 4 + g; s < ns; s += int64_t(gridDim.x) * 4) acc += double(X[(s * step) * d + c]);
    atomicAdd(sums + c, acc);
}

// covariance of the sampled rows about `mean`: C[a][b] += sum (x_a - m_a)(x_b - m_b)   (C pre-zeroed, d <= 64)
template <typename T>
__global__ __launch_bounds__(256) void sample_cov_kernel(const T* __restrict__ X, const int64_t n, const int d,
                                                         const int64_t step, const int64_t ns,
                                                         const double* __restrict__ sums, double* __restrict__ C) {
    __shared__ float tile[64][65];
    const int64_t s0 = int64_t(blockIdx.x) * 64;
    for (int f = threadIdx.x; f < 64 * 64; f += 256) {
        const int r = f / 64, c = f % 64;
        const int64_t s = s0 + r;
        tile[r][c] = (s < ns && c < d) ? float(double(X[(s * step) * d + c]) - sums[c] / double(ns)) : 0.f;
    }
    __syncthreads();
    // thread t owns the 16 entries (a, b) with a = t / 4, b = 16 (t % 4) ... + 15
    const int a = threadIdx.x / 4, b0 = 16 * (threadIdx.x % 4);
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int r = 0; r < 64; ++r) {
        const float xa = tile[r][a];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = fmaf(xa, tile[r][b0 + e], acc[e]);
    }
    if (a < d)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (b0 + e < d) atomicAdd(C + a * 64 + b0 + e, double(acc[e]));
}

// Z[p] = float16(scz * P x_perm[p]) (kZ columns), hh[p] = -|Z[p]|^2 / 2 in float32; pad rows: zeros / -inf.
// 16 lanes per row: lane k forms column k; the row's features reach every lane through shuffles.
template <typename T>
__global__ __launch_bounds__(256) void sym_project_kernel(const T* __restrict__ X, const int32_t* __restrict__ perm,
                                                          const int64_t n, const int64_t n_pad, const int d,
                                                          const float* __restrict__ P, const float scz,
                                                          _Float16* __restrict__ Z, float* __restrict__ hh) {
    __shared__ float Ps[kZ * 64];
    for (int f = threadIdx.x; f < kZ * 64; f += 256) Ps[f] = (f % 64) < d ? P[(f / 64) * 64 + (f % 64)] : 0.f;
    __syncthreads();
    const int k = threadIdx.x & 15;
    const int64_t p = int64_t(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (p >= n_pad) return;
    float z = 0.f;
    if (p < n) {
        const T* x = X + int64_t(perm[p]) * d;
        float xr[4];   // features k, k + 16, k + 32, k + 48 of the row
#pragma unroll
        for (int u = 0; u < 4; ++u) xr[u] = (k + 16 * u < d) ? float(x[k + 16 * u]) : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            for (int src = 0; src < 16; ++src) acc = fmaf(Ps[k * 64 + 16 * u + src], __shfl(xr[u], src, 16), acc);
        z = acc * scz;
    }
    const _Float16 zh = _Float16(z);
    Z[size_t(p) * kZ + k] = zh;
    float sq = float(zh) * float(zh);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 16);
    if (k == 0) hh[p] = p < n ? -0.5f * sq : -INFINITY;
}

// Radius of row p among the rows of the full compact copy: a pair either row needs listed (true squared distance below
// the bound lb_p the re-rank will claim, see sym_half_thresholds_kernel) has float16 rows at most this far apart.  -inf:
// the row needs nothing (orphan, pad row); +inf: no threshold was seeded, it needs everything.  Rounded up.
__global__ __launch_bounds__(256) void sym_row_radius_kernel(const int64_t n, const int64_t n_pad,
                                                             const int32_t* __restrict__ perm, const double* __restrict__ xn,
                                                             const float* __restrict__ thr, const double* __restrict__ ymax2p,
                                                             const ErrModel err, float* __restrict__ rrow, const double lres) {
    // lres >= 0: largest rounding residual of a row (true units) when err is not the single-chain model, whose `abs` is it
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n_pad) return;
    float rr = -INFINITY;
    if (p < n) {
        const float t = thr[p];
        if (t == INFINITY) {
        } else if (!(t > -3.0e38f)) {
            rr = INFINITY;
        } else {
            const double qs = xn[perm[p]], y2 = ymax2p[0];
            const double e = gt_err_bound(err, qs, y2);
            double lb = (qs - 2.0 * (double(t) * err.inv_sc2 + e)) - 1e-9 * (qs + y2);
            lb = (lb > 0.0 ? lb : 0.0) * (1.0 + 1e-6) + 1e-9 * (qs + y2);
            const double scf = 1.0 / sqrt(err.inv_sc2), Lf = scf * (lres >= 0.0 ? lres : err.abs);
            const double rf = (scf * sqrt(lb) + 2.0 * Lf) * (1.0 + 1e-9);
            rr = float(rf);
            if (double(rr) <= rf) rr = nextafterf(rr, INFINITY);
        }
    }
    rrow[p] = rr;
}

// ---- two-stage scoring of launch B (partial distances) -----------------------------------------------------------
// Thresholds of the partial test from the full thresholds thr[p] (sorted positions).  The re-rank will claim for row p
// "every row closer than lb_p is in the list", lb_p = |x|^2 - 2 (thr_p / sc^2 + e) - 1e-9 (...) (rerank_sym_kernel,
// bound_of_score) - so stage one must let through every pair with |x - y|^2 < lb_p, seen from either side.  Such a pair
// has scaled float16 rows at most sc sqrt(lb) + 2 Ls apart (Ls: largest rounding residual of a row), on any subset of
// the features, so A >= |x_h|^2/2 - rho_p - (rounding of A), rho_p = (sc sqrt(lb_p) + 2 Ls)^2 / 2:
//   forward     A > thrh[p]            thrh[p] = -hh[p] - rho_p - dmax      (rounded down)
//   transposed  A + hh[x] > gh[p]      gh[p]   = -rho_p - dmax              (rounded down; gminh = its sub-tile minima)
// dmax bounds the float32 accumulation of HD + 1 terms and the roundings of hh and of the sum A + hh[x] for any pair.
__global__ __launch_bounds__(256) void sym_half_thresholds_kernel(const int64_t n, const int64_t n_pad,
                                                                  const int32_t* __restrict__ perm,
                                                                  const double* __restrict__ xn,
                                                                  const float* __restrict__ thr,
                                                                  const float* __restrict__ hh,
                                                                  const double* __restrict__ ymax2p, const ErrModel err,
                                                                  const int HD, const double scz, const double Lz,
                                                                  float* __restrict__ thrh, float* __restrict__ gh,
                                                                  float* __restrict__ rrow) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n_pad) return;
    float th = INFINITY, g = INFINITY;
    // rrow: how far apart (rows of the full compact copy) a pair can be that row p needs listed, rounded up; -inf: the
    // row needs nothing (orphan, pad row), +inf: everything
    float rr = -INFINITY;
    if (p < n) {
        const float t = thr[p];
        if (t == INFINITY) {
            // an orphan collects nothing (sym_thresholds_kernel): th = g = +inf
        } else if (!(t > -3.0e38f)) {
            th = -3.0e38f;   // no threshold was seeded for this row: everything passes, as in the full test
            g = -3.0e38f;
            rr = INFINITY;
        } else {
            const double u = 5.9604644775390625e-08;
            const double qs = xn[perm[p]], y2 = ymax2p[0];
            const double e = gt_err_bound(err, qs, y2);
            double lb = (qs - 2.0 * (double(t) * err.inv_sc2 + e)) - 1e-9 * (qs + y2);
            lb = (lb > 0.0 ? lb : 0.0) * (1.0 + 1e-6) + 1e-9 * (qs + y2);
            // scz: scale of the stage-one copy (its rows are scz x an orthonormal projection of the points, up to 1e-6),
            // Lz: largest rounding residual of one of its rows
            const double sc = scz * (1.0 + 1e-6), Ls = Lz;
            const double X2 = (sc * sqrt(y2) + Ls) * (sc * sqrt(y2) + Ls);
            const double dmax = (2.0 * double(HD + 8) * 1.5 + 4.0) * u * X2;
            const double r = sc * sqrt(lb) + 2.0 * Ls;
            {
                // the same radius among the rows of the FULL compact copy (scale 1 / sqrt(inv_sc2), residual norms up to
                // err.abs before scaling): what the bound pass measures its cells in
                const double scf = 1.0 / sqrt(err.inv_sc2), Lf = scf * err.abs;
                const double rf = (scf * sqrt(lb) + 2.0 * Lf) * (1.0 + 1e-9);
                rr = float(rf);
                if (double(rr) <= rf) rr = nextafterf(rr, INFINITY);
            }
            const double rho = 0.5 * r * r + dmax;
            const double a = -double(hh[p]) - rho, b = -rho;
            th = float(a);
            if (double(th) >= a) th = nextafterf(th, -INFINITY);
            g = float(b);
            if (double(g) >= b) g = nextafterf(g, -INFINITY);
            if (!(th > -3.0e38f)) th = -3.0e38f;
            if (!(g > -3.0e38f)) g = -3.0e38f;
        }
    }
    thrh[p] = th;
    gh[p] = g;
    if (rrow) rrow[p] = rr;
}

// How many (64-query group, 32-row sub-tile) pairs would pass stage one?  One wave per pseudo-randomly drawn pair (lane =
// query, the 32 rows in turn), the same partial scores and the same two tests as the collect kernel's unit loop (float32
// FMA instead of the MFMA: a forecast, not a proof).  flagged (pre-zeroed): pairs in which some score passed.
__global__ __launch_bounds__(256) void sym_two_probe_kernel(const _Float16* __restrict__ Ys, const int64_t n, const int DP,
                                                            const int HD, const float* __restrict__ hh,
                                                            const float* __restrict__ thrh, const float* __restrict__ gh,
                                                            const int64_t samples, uint32_t* __restrict__ flagged) {
    const int lane = threadIdx.x & 63;
    const int64_t sidx = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (sidx >= samples) return;
    const uint64_t h1 = (uint64_t(sidx) + 1) * 0x9E3779B97F4A7C15ull, h2 = (uint64_t(sidx) + 1) * 0xC2B2AE3D27D4EB4Full;
    const int64_t q0 = int64_t((h1 >> 17) % uint64_t(n / 64)) * 64, d0 = int64_t((h2 >> 17) % uint64_t(n / 32)) * 32;
    float xq[32];
    const _Float16* xr = Ys + size_t(q0 + lane) * DP;
#pragma unroll
    for (int k = 0; k < 32; ++k) xq[k] = k < HD ? float(xr[k]) : 0.f;
    const float tq = thrh[q0 + lane], hq = hh[q0 + lane];
    float gmin = INFINITY;
    for (int j = 0; j < 32; ++j) gmin = fminf(gmin, gh[d0 + j]);
    bool hit = false;
    for (int j = 0; j < 32; ++j) {
        const _Float16* yr = Ys + size_t(d0 + j) * DP;
        float a = hh[d0 + j];
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < HD) a = fmaf(xq[k], float(yr[k]), a);
        hit = hit || a > tq || (a + hq > gmin);
    }
    if (__ballot(hit) != 0ull && lane == 0) atomicAdd(flagged, 1u);
}

// ---- row-sharded builds (gt_knn_shard.cpp) -----------------------------------------------------------------------
struct ShardSplits {
    int64_t s[GT_SYM_MAX_WORLD + 1];
    int world;
};
__device__ __forceinline__ int shard_owner(const ShardSplits& sp, int64_t row) {
    int o = 0;
    for (int r = 1; r < sp.world; ++r) o += (row >= sp.s[r]) ? 1 : 0;
    return o;
}

__global__ __launch_bounds__(256) void sym_g_kernel(const int64_t n, const int64_t n_pad, const float* __restrict__ thr,
                                                    const float* __restrict__ hs, float* __restrict__ g) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n_pad) return;
    g[p] = (p < n && thr[p] != INFINITY) ? nextafterf(thr[p] + hs[p], -INFINITY) : INFINITY;   // as sym_thresholds_kernel forms it
}

__global__ __launch_bounds__(256) void invperm_kernel(const int32_t* __restrict__ perm, const int64_t n,
                                                      int32_t* __restrict__ inv) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p < n) inv[perm[p]] = int32_t(p);
}

// Candidate records a rank sends to the owners of the rows: {uint32 row (local to the owner), 0, uint64 key}; the
// candidates this rank collected for sorted position p go to the owner of row perm[p].  A list that overflowed here
// (count > tcap) sends its tcap entries and one marker record (key = all ones): the owner hands the row to the repairs.
__global__ __launch_bounds__(256) void shard_count_kernel(const int64_t n, const int32_t* __restrict__ perm,
                                                          const uint32_t* __restrict__ tcounts, const int tcap,
                                                          const ShardSplits sp, unsigned long long* __restrict__ cnt) {
    __shared__ unsigned int hist[GT_SYM_MAX_WORLD];
    if (threadIdx.x < GT_SYM_MAX_WORLD) hist[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p < n) {
        const uint32_t c = tcounts[p];
        const uint32_t rec = c > uint32_t(tcap) ? uint32_t(tcap) + 1u : c;
        if (rec) atomicAdd(&hist[shard_owner(sp, perm[p])], rec);
    }
    __syncthreads();
    if (threadIdx.x < sp.world && hist[threadIdx.x]) atomicAdd(cnt + threadIdx.x, (unsigned long long)hist[threadIdx.x]);
}

// 256 sorted positions per workgroup; cursor[dest] starts at the bucket offset of dest in `out`.  The slots of a
// workgroup's rows are reserved with ONE global atomic per destination (a million same-address atomics would
// serialise); each wave then writes the records of 64 of the rows, lanes across the entries of a row.
__global__ __launch_bounds__(256) void shard_emit_kernel(const int64_t n, const int32_t* __restrict__ perm,
                                                         const uint64_t* __restrict__ tlists,
                                                         const uint32_t* __restrict__ tcounts, const int tcap,
                                                         const ShardSplits sp, unsigned long long* __restrict__ cursor,
                                                         uint4* __restrict__ out) {
    __shared__ unsigned int lcnt[GT_SYM_MAX_WORLD];
    __shared__ unsigned long long lbase[GT_SYM_MAX_WORLD];
    __shared__ unsigned int r_off[256], r_m[256], r_rl[256];
    __shared__ unsigned char r_dest[256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < GT_SYM_MAX_WORLD) lcnt[tid] = 0u;
    __syncthreads();
    const int64_t p = int64_t(blockIdx.x) * 256 + tid;
    uint32_t c = p < n ? tcounts[p] : 0u;
    const bool over = c > uint32_t(tcap);
    const uint32_t m = over ? uint32_t(tcap) : c;
    int dest = 0;
    uint32_t rl = 0u, off = 0u;
    if (c) {
        const int64_t row = perm[p];
        dest = shard_owner(sp, row);
        rl = uint32_t(row - sp.s[dest]);
        off = atomicAdd(&lcnt[dest], m + (over ? 1u : 0u));
    }
    r_off[tid] = off;
    r_m[tid] = m | (over ? 0x80000000u : 0u);
    r_rl[tid] = rl;
    r_dest[tid] = (unsigned char)dest;
    __syncthreads();
    if (tid < sp.world) lbase[tid] = lcnt[tid] ? atomicAdd(cursor + tid, (unsigned long long)lcnt[tid]) : 0ull;
    __syncthreads();
    for (int r = 0; r < 64; ++r) {
        const int t = w * 64 + r;
        const uint32_t mm = r_m[t];
        const uint32_t mr = mm & 0x7FFFFFFFu;
        if (mm == 0u) continue;   // wave-uniform
        const unsigned long long base = lbase[r_dest[t]] + r_off[t];
        const uint32_t rlr = r_rl[t];
        const uint64_t* tp = tlists + size_t(int64_t(blockIdx.x) * 256 + t) * size_t(tcap);
        for (uint32_t i = uint32_t(lane); i < mr; i += 64u) {
            const uint64_t key = tp[i];
            out[base + i] = make_uint4(rlr, 0u, uint32_t(key), uint32_t(key >> 32));
        }
        if ((mm >> 31) && lane == 0) out[base + mr] = make_uint4(rlr, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
}

struct OwnedRow {
    int32_t r0, r1;
    __device__ bool operator()(const int32_t& row) const { return row >= r0 && row < r1; }
};

// the owner's side: records from every rank into the lists of the owned rows (slots from the row counters; a count
// beyond tcap marks the row as overflowed, as in the single-rank pass, and so does bit 31, set by a marker record)
__global__ __launch_bounds__(256) void shard_scatter_kernel(const uint4* __restrict__ recs, const int64_t n_recs,
                                                            const int64_t nloc, const int tcap,
                                                            uint64_t* __restrict__ lists, uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ bad) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n_recs) return;
    const uint4 r = recs[i];
    if (int64_t(r.x) >= nloc) {
        atomicAdd(bad, 1u);
        return;
    }
    if (r.z == 0xFFFFFFFFu && r.w == 0xFFFFFFFFu) {
        atomicOr(&counts[r.x], 0x80000000u);   // the marker takes no slot: every slot below the count holds a key
        return;
    }
    const uint32_t slot = atomicAdd(&counts[r.x], 1u) & 0x7FFFFFFFu;
    if (slot < uint32_t(tcap)) lists[size_t(r.x) * size_t(tcap) + slot] = (uint64_t(r.w) << 32) | uint64_t(r.z);
}


// ---- bound pass of the two-stage collect: whole cells instead of pairs of rows -----------------------------------------
// The collect asks of every pair of rows whether they are closer than either row's radius rrow (what the thresholds
// encode).  The points are sorted by landmark cell, and a cell is a compact set: with the centre c and radius R of a
// cell's rows (in the float16 compact copy the candidate kernels score), no row of cell A is closer to a row of cell B
// than |c_A - c_B| - R_A - R_B.  When that exceeds the largest radius of either cell, NO pair of the two cells can be one
// either row needs - by the triangle inequality, for whatever centres were chosen - and the (64 queries, 32 rows) units
// made of their rows need neither the unit loop nor the cold pass.  On clustered data that is nearly every unit: the
// bound pass lists the units that are left straight into the queue of the cold launch and the collect launch does not
// run.  (All DP columns: the 16 columns of stage one keep a quarter of the squared distance between two cluster centres
// and a quarter of a cell's squared radius, but the rows' radii do not shrink - too many cell pairs stay undecided.
// Centres are float32 means, distances and radii are evaluated in float64 from those stored values; radii are rounded
// up, the comparison carries a relative margin.)
__global__ __launch_bounds__(256) void cell_ball_kernel(const _Float16* __restrict__ Ys, const int DP,
                                                        const float* __restrict__ rrow,
                                                        const int32_t* __restrict__ start, const int32_t* __restrict__ endp,
                                                        float* __restrict__ centre, float* __restrict__ radius,
                                                        float* __restrict__ need) {
    // DP / 8 threads per row, each with one 16-byte load of its eight columns (DP <= 128: at least 16 rows per step)
    __shared__ __attribute__((aligned(16))) float part[32][129];   // per row slot: partial column sums, then partial squared distances
    __shared__ float cs[128];
    __shared__ double red[4];
    __shared__ float redn[4];
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int c = blockIdx.x;
    const int s0 = start[c], s1 = endp[c];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (s0 < 0 || s1 <= s0) {   // empty cell: never looked up
        for (int k = threadIdx.x; k < DP; k += 256) centre[size_t(c) * DP + k] = 0.f;
        if (threadIdx.x == 0) {
            radius[c] = 0.f;
            need[c] = -INFINITY;
        }
        return;
    }
    const int tpr = DP / 8;                 // threads per row
    const int rps = 256 / tpr < 32 ? 256 / tpr : 32;   // rows per step
    const int slot = int(threadIdx.x) / tpr, t = int(threadIdx.x) % tpr;
    const bool on = slot < rps;
    // centre: column sums, slot by slot, then across the slots
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (on)
        for (int p = s0 + slot; p < s1; p += rps) {
            const half8 v = *reinterpret_cast<const half8*>(Ys + size_t(p) * DP + 8 * t);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += float(v[e]);
        }
    if (on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[slot][8 * t + e] = a[e];
    }
    __syncthreads();
    if (int(threadIdx.x) < DP) {
        float sum = 0.f;
        for (int q = 0; q < rps; ++q) sum += part[q][threadIdx.x];
        const float m = sum / float(s1 - s0);
        cs[threadIdx.x] = m;
        centre[size_t(c) * DP + threadIdx.x] = m;
    }
    __syncthreads();
    // radius: float64 squared distance of every row to the stored centre (thread t of the row's group: its eight columns)
    double rmax = 0.0;
    float nmax = -INFINITY;
    double cd[8];
    if (on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cd[e] = double(cs[8 * t + e]);
    }
    for (int p0 = s0; p0 < s1; p0 += rps) {   // (uniform trip count: the group's partial sums meet in the LDS)
        const int p = p0 + slot;
        double d2 = 0.0;
        if (on && p < s1) {
            const half8 v = *reinterpret_cast<const half8*>(Ys + size_t(p) * DP + 8 * t);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double df = double(float(v[e])) - cd[e];
                d2 = fma(df, df, d2);
            }
        }
        // the tpr partial sums of a row sit in consecutive lanes of one wave (tpr divides 64 for DP = 32, 64, 128) or are
        // summed through the LDS otherwise
        if ((64 % tpr) == 0) {
            for (int o = 1; o < tpr; o <<= 1) d2 += __shfl_xor(d2, o);
            if (on && p < s1 && t == 0) {
                rmax = fmax(rmax, sqrt(d2));
                nmax = fmaxf(nmax, rrow[p]);
            }
        } else {
            __syncthreads();
            double* pd = reinterpret_cast<double*>(&part[0][0]);   // [rps][tpr] doubles: 32 x 16 x 8 B <= the array
            if (on) pd[slot * tpr + t] = d2;
            __syncthreads();
            if (on && p < s1 && t == 0) {
                double tot = 0.0;
                for (int q = 0; q < tpr; ++q) tot += pd[slot * tpr + q];
                rmax = fmax(rmax, sqrt(tot));
                nmax = fmaxf(nmax, rrow[p]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rmax = fmax(rmax, __shfl_xor(rmax, o));
        nmax = fmaxf(nmax, __shfl_xor(nmax, o));
    }
    if (lane == 0) {
        red[w] = rmax;
        redn[w] = nmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double r = fmax(fmax(red[0], red[1]), fmax(red[2], red[3])) * (1.0 + 1e-9);
        float rf = float(r);
        if (double(rf) < r) rf = nextafterf(rf, INFINITY);
        radius[c] = rf;
        need[c] = fmaxf(fmaxf(redn[0], redn[1]), fmaxf(redn[2], redn[3]));
    }
}

// bit b of word mask[a][b / 32]: no row of cell a and row of cell b can be a pair either of them needs.
// Workgroup = 64 cells a x 256 cells b: thread t keeps the centre of cell b0 + t in registers, the centres of the a's come
// from the LDS (broadcast reads).  Float32: a difference of two stored floats is rounded once, the squares are summed
// without cancellation - relative error below (DP + 2) 2^-23 - and the comparison carries a margin of 1e-4.
template <int DPC>
__global__ __launch_bounds__(256) void cell_mask_kernel(const int L, const float* __restrict__ centre,
                                                        const float* __restrict__ radius, const float* __restrict__ need,
                                                        uint32_t* __restrict__ mask, const uint32_t* __restrict__ cell_lo,
                                                        const uint32_t* __restrict__ cell_hi) {
    __shared__ float ca[64][DPC];
    const int words = (L + 31) / 32;
    const int a0 = blockIdx.y * 64, b = blockIdx.x * 256 + threadIdx.x;
    // (optional: only the mask rows of the cells [*cell_lo, *cell_hi] will be read - a rank's own queries)
    if (cell_lo && (uint32_t(a0 + 63) < *cell_lo || uint32_t(a0) > *cell_hi)) return;
    const int lane = threadIdx.x & 63;
    for (int f = threadIdx.x; f < 64 * DPC; f += 256) {
        const int a = a0 + f / DPC;
        ca[f / DPC][f % DPC] = a < L ? centre[size_t(a) * DPC + f % DPC] : 0.f;
    }
    float cb[DPC];
    const int bc = b < L ? b : L - 1;
#pragma unroll
    for (int q = 0; q < DPC; ++q) cb[q] = centre[size_t(bc) * DPC + q];
    const float Rb = radius[bc], nb = need[bc];
    __syncthreads();
    for (int i = 0; i < 64; ++i) {
        const int a = a0 + i;
        if (a >= L) break;   // uniform
        float d2 = 0.f;
#pragma unroll
        for (int q = 0; q < DPC; ++q) {
            const float df = ca[i][q] - cb[q];
            d2 = fmaf(df, df, d2);
        }
        const float nd = fmaxf(need[a], nb);
        // (need = -inf on both sides: the cells need nothing at all; +inf: never prunable)
        const bool pr = nd == -INFINITY ||
                        double(sqrtf(d2)) * (1.0 - 1e-4) - (double(radius[a]) + double(Rb)) > double(nd) * (1.0 + 1e-6);
        const unsigned long long m = __ballot(pr && b < L);
        const int wd = (blockIdx.x * 256 + (threadIdx.x & ~63)) / 32;
        if (lane == 0 && wd < words) mask[size_t(a) * words + wd] = uint32_t(m);
        if (lane == 32 && wd + 1 < words) mask[size_t(a) * words + wd + 1] = uint32_t(m >> 32);
    }
}

// first and last cell of the rows of every 32-row sub-tile of the sorted order (0xFFFF, 0: no real row)
__global__ __launch_bounds__(256) void tile_cells_kernel(const uint32_t* __restrict__ cell_sorted, const int64_t n,
                                                         const int64_t ntile32, uint32_t* __restrict__ tcell) {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= ntile32) return;
    const int64_t p0 = t * 32;
    if (p0 >= n) {
        tcell[t] = 0xFFFFu;
        return;
    }
    const int64_t p1 = p0 + 31 < n ? p0 + 31 : n - 1;
    tcell[t] = (cell_sorted[p0] & 0xFFFFu) | (cell_sorted[p1] << 16);
}

// How many units will the undecided cell pairs leave, roughly?  Sum over the undecided pairs (a, b) of (rows of a / 64) x
// (rows of b / 32), halved (a walk visits an unordered pair of blocks once).  One wave per cell; lets the caller skip the
// enumeration below where it would only establish that there are far too many (unclustered data: 97 M units, 3.5 ms).
__global__ __launch_bounds__(64) void bound_estimate_kernel(const int L, const int32_t* __restrict__ start,
                                                            const int32_t* __restrict__ endp, const uint32_t* __restrict__ mask,
                                                            const int words, double* __restrict__ est,
                                                            const uint32_t* __restrict__ cell_lo,
                                                            const uint32_t* __restrict__ cell_hi) {
    // (cell_lo / cell_hi: only the cells [*cell_lo, *cell_hi] ask - a rank's own query groups against every sub-tile, each
    //  unit counted once: no halving.  Boundary cells count whole: the forecast is compared with 4 x the capacity.)
    const int a = blockIdx.x, lane = threadIdx.x;
    if (start[a] < 0) return;
    if (cell_lo && (uint32_t(a) < *cell_lo || uint32_t(a) > *cell_hi)) return;
    const double ra = double(endp[a] - start[a]) / 64.0;
    double acc = 0.0;
    for (int wd = lane; wd < words; wd += 64) {
        uint32_t zero = ~mask[size_t(a) * words + wd];
        while (zero) {
            const int b = wd * 32 + __ffs(int(zero)) - 1;
            zero &= zero - 1u;
            if (b < L && start[b] >= 0) acc += double(endp[b] - start[b]) / 32.0;
        }
    }
    acc = wave_sum_f64(acc);
    if (lane == 0 && acc > 0.0) atomicAdd(est, (cell_lo ? 1.0 : 0.5) * ra * acc);
}

// The units (64 queries q64, 32 rows d32) of the collect launch's walks that the cell bounds cannot rule out -> queue.
// One wave per q64.  For each cell a of its queries the lanes read the words of mask row a; every undecided cell b
// (zero bit) contributes the sub-tiles its rows lie in - those that belong to the walk of q64's 1024-row block (own
// block, H following blocks, the antipodal one when NB is even: knn_select_kernel<MODE 2> with two-stage scoring).  A
// sub-tile (or the queries) can straddle cells: a unit is filed by the FIRST undecided pair (a, b) of its cell ranges.
__global__ __launch_bounds__(256) void bound_queue_kernel(const int nq64, const int T, const int TPB, const int walk, const int L,
                                                         const int world, const int rank, const int group,
                                                         const int own_q0, const int own_full,
                                                         const uint32_t* __restrict__ tcell,
                                                         const int32_t* __restrict__ start, const int32_t* __restrict__ endp,
                                                         const uint32_t* __restrict__ mask, const int words,
                                                         uint2* __restrict__ queue, const uint32_t cap,
                                                         uint32_t* __restrict__ count, const double* __restrict__ forecast,
                                                         const int64_t n_rows, unsigned long long* __restrict__ dbg_cyc) {
    const unsigned long long t_begin = dbg_cyc ? __builtin_readcyclecounter() : 0ull;
    // four groups of 64 queries per workgroup: their slots come from ONE returning atomic (15 600 of them, one per group, were
    // half of this kernel's time)
    __shared__ uint32_t wtot[4], wbase;
    __shared__ uint16_t blist[4][2048];   // per wave: the undecided cells of one stretch (64 words) of a mask row
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // own_full: the groups [own_q0, nq64) against EVERY sub-tile (no walks: a row-sharded build on renumbered points files
    // each pair under the query's side only, the launch covers the rank's own groups)
    const uint32_t q64 = uint32_t(own_q0) + blockIdx.x * 4u + uint32_t(w);
    if (forecast && *forecast > 4.0 * double(cap)) {   // (bound_estimate_kernel: far too many units - report "too many", list none)
        if (blockIdx.x == 0 && threadIdx.x == 0) count[0] = count[1] = 0xFFFFFFFFu;   // (own_full launches start at own_q0)
        return;   // (every wave of the launch alike)
    }
    bool active = int(q64) < nq64;
    const int blk = int(q64) / (TPB * 2);   // (a 1024-row block = TPB 128-row tiles = 2 TPB groups of 64 queries)
    const uint32_t qa = active ? tcell[2 * q64] : 0xFFFFFFFFu, qb = active ? tcell[2 * q64 + 1] : 0xFFFFFFFFu;
    // cells of the 64 queries: [lo, hi] over the two sub-tiles that hold real rows
    uint32_t qlo = 0xFFFFu, qhi = 0u;
    if ((qa & 0xFFFFu) != 0xFFFFu) {
        qlo = qa & 0xFFFFu;
        qhi = qa >> 16;
    }
    if ((qb & 0xFFFFu) != 0xFFFFu) {
        if ((qb & 0xFFFFu) < qlo) qlo = qb & 0xFFFFu;
        if ((qb >> 16) > qhi) qhi = qb >> 16;
    }
    if (qlo == 0xFFFFu) active = false;   // pad queries only (the wave still meets the others at the reservation)
    // row-sharded build: this rank's piece of the block's walk (the partition of knn_select_kernel<MODE 2>: piece
    // (rank + block / group) mod world of `world` equal pieces)
    const int piece = world > 1 ? (rank + blk / group) % world : 0;
    const int rel_lo = int(int64_t(walk) * piece / world), rel_hi = int(int64_t(walk) * (piece + 1) / world);
    auto open_pair = [&](const uint32_t a, const uint32_t b) {
        return ((mask[size_t(a) * words + (b >> 5)] >> (b & 31u)) & 1u) == 0u;
    };
    // two rounds: count this wave's units, take their slots with ONE atomic (a device-scope counter serves ~90 returning
    // atomics per microsecond: one per unit would cost more than the collect launch saves), then write them
    uint32_t mine = 0u, slot = 0u, all_ranks = 0u;   // all_ranks: the units of EVERY rank's pieces (row-sharded builds)
    // A group that holds rows of a cell undecided against EVERY cell (the outlier cell of gt_order.hip: a ball as wide as the
    // point set) needs every tile of its walk, whatever its other cells say: the tiles are counted and filed by arithmetic,
    // 64 lanes side by side - not cell by cell through the mask (one lane per 32 cells, three dependent loads per tile: 1.5 ms
    // for the one group that holds the cell while the whole launch waits for it)
    bool all_open = false;
    if (active)
        for (uint32_t a = qlo; a <= qhi && !all_open; ++a) {
            bool row_open = true;
            for (int w0 = 0; w0 < words; w0 += 64) {
                const int wd = w0 + lane;
                if (wd < words) {
                    uint32_t full = (wd * 32 + 32 <= L) ? 0xFFFFFFFFu : ((1u << (L - wd * 32)) - 1u);
                    if (int(a >> 5) == wd) full &= ~(1u << (a & 31u));   // (a cell against itself is whatever cell_mask_kernel says: not asked)
                    row_open &= (~mask[size_t(a) * words + wd] & full) == full;
                }
            }
            all_open = __all(row_open) != 0;
        }
  for (int round = 0; round < 2; ++round) {
    if (round == 1) {
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = uint32_t(__shfl_up(int(inc), o));
            if (lane >= o) inc += up;
        }
        const uint32_t total = uint32_t(__shfl(int(inc), 63));
        if (world > 1) {
            // the verdict "the cell bounds leave few enough units" has to be the same on every rank: it is taken on the
            // count over ALL pieces, which every rank computes alike (count[1])
            uint32_t ar = all_ranks;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ar += uint32_t(__shfl_xor(int(ar), o));
            if (lane == 0 && ar != 0u) atomicAdd(count + 1, ar);
        }
        if (lane == 0) wtot[w] = total;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t sum = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            wbase = sum != 0u ? atomicAdd(count, sum) : 0u;
        }
        __syncthreads();
        if (total == 0u) {
            if (dbg_cyc && lane == 0) dbg_cyc[q64] = __builtin_readcyclecounter() - t_begin;
            return;
        }
        uint32_t base = wbase;
        for (int v = 0; v < w; ++v) base += wtot[v];
        slot = base + inc - mine;
    }
    if (active && all_open) {
        const uint32_t nt_real = uint32_t((n_rows + 31) / 32);
        for (uint32_t d32 = uint32_t(lane); d32 < nt_real; d32 += 64u) {
            int rel = int(d32 / 4u) - blk * TPB;
            if (rel < 0) rel += T;
            if (!own_full && rel >= walk) continue;
            const bool own_piece = own_full || (rel >= rel_lo && rel < rel_hi);
            if (!own_piece && (world == 1 || round == 1)) continue;
            if (round == 0) {
                ++all_ranks;
                if (own_piece) ++mine;
            } else {
                if (slot < cap) queue[slot] = make_uint2(q64, d32);
                ++slot;
            }
        }
    } else if (active)
    for (uint32_t a = qlo; a <= qhi; ++a) {
        for (int w0 = 0; w0 < words; w0 += 64) {
            const int wd = w0 + lane;
            uint32_t zero = 0u;
            if (wd < words) {
                zero = ~mask[size_t(a) * words + wd];
                if (wd * 32 + 32 > L) zero &= (L - wd * 32 >= 32) ? 0xFFFFFFFFu : ((1u << (L - wd * 32)) - 1u);
            }
            // The undecided cells of this stretch of the mask row are dealt to the lanes ONE BY ONE (compacted through the LDS),
            // not word by word: with coherent cell numbers (gt_order.hip) the dozen cells a cell is undecided against are
            // neighbours in number - one or two words, i.e. one or two lanes did the whole row while 62 watched (C3: 0.68 -> 1.01
            // ms when the numbering became coherent).  The order of the queue does not matter; both rounds deal alike.
            uint32_t nb;
            {
                const uint32_t c = uint32_t(__popc(zero));
                uint32_t inc = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = uint32_t(__shfl_up(int(inc), o));
                    if (lane >= o) inc += up;
                }
                nb = uint32_t(__shfl(int(inc), 63));
                uint32_t at = inc - c;
                uint32_t z = zero;
                while (z != 0u) {
                    blist[w][at++] = uint16_t(uint32_t(wd) * 32u + uint32_t(__ffs(int(z)) - 1));
                    z &= z - 1u;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t it = uint32_t(lane); it < nb; it += 64u) {
                const uint32_t b = blist[w][it];
                const int sb0 = start[b], sb1 = endp[b];
                if (sb0 < 0 || sb1 <= sb0) continue;   // empty cell
                for (uint32_t d32 = uint32_t(sb0) / 32u; d32 <= uint32_t(sb1 - 1) / 32u; ++d32) {
                    int rel = int(d32 / 4u) - blk * TPB;
                    if (rel < 0) rel += T;
                    if (!own_full && rel >= walk) continue;   // not this block's unit (the other block's walk has it)
                    const bool own_piece = own_full || (rel >= rel_lo && rel < rel_hi);
                    if (!own_piece && (world == 1 || round == 1)) continue;   // another rank's piece (counted in round 0)
                    // first undecided pair of (cells of the queries) x (cells of the sub-tile) files the unit
                    const uint32_t dc = tcell[d32];
                    const uint32_t dlo = dc & 0xFFFFu, dhi = dc >> 16;
                    bool first = true;
                    for (uint32_t a2 = qlo; a2 <= a && first; ++a2)
                        for (uint32_t b2 = dlo; b2 <= dhi; ++b2) {
                            if (a2 == a && b2 >= b) break;
                            if (open_pair(a2, b2)) {
                                first = false;
                                break;
                            }
                        }
                    if (!first) continue;
                    if (round == 0) {
                        ++all_ranks;
                        if (own_piece) ++mine;
                    } else {
                        if (slot < cap) queue[slot] = make_uint2(q64, d32);
                        ++slot;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // (the next stretch reuses the list)
        }
    }
  }
    if (dbg_cyc && lane == 0) dbg_cyc[q64] = __builtin_readcyclecounter() - t_begin;
}


// ---- unit skipping of the two-stage collect: balls of the groups of 32 sorted rows in the stage-one space -----------------
// Stage one passes a pair (q, j) when its partial score beats q's threshold or (transposed) j's: in exact arithmetic on the
// stored float16 rows, |z_q - z_j|^2 < 2 rho_p + (a few dmax) for p = q or p = j (sym_half_thresholds_kernel: gh[p] = -rho_p -
// dmax, rounded down; dmax bounds the float32 accumulation and the roundings of the seeds).  So with need_p :=
// sqrt(-2 gh[p] + 4 dmax) (rounded up; -inf for a row that asks for nothing, +inf for one that asks for everything), a pair
// more than max(need_q, need_j) apart in the stage-one space cannot pass.  Per group of 32 consecutive sorted rows (the query
// tiles and the sub-tiles of knn_select_kernel alike): the centre c of its real rows (float32 mean), R = the largest
// distance of a row from the STORED centre (float64, rounded up) and need = the largest need of its rows.  Two groups whose
// centres are farther apart than R_a + R_b + max(need_a, need_b) hold no pair that passes: the collect kernel skips the unit.
// One wave per group: lane = (row l >> 1, eight columns l & 1).
__global__ __launch_bounds__(256) void z_balls_kernel(const _Float16* __restrict__ Z, const float* __restrict__ gh, const int64_t n,
                                                      const int64_t ngroups, const double dmargin, float* __restrict__ zc,
                                                      float* __restrict__ zrn) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    const int64_t g = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (g >= ngroups) return;
    const int r = lane >> 1, hf = lane & 1;
    const int64_t p = g * 32 + r;
    const bool real = p < n;
    float x[8];
    {
        half8 v = {};
        if (real) v = *reinterpret_cast<const half8*>(Z + size_t(p) * kZ + 8 * hf);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = real ? float(v[e]) : 0.f;
    }
    float cnt = real ? 1.f : 0.f;
    float c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = x[e];
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) {   // over the rows (lanes of the same column half)
        cnt += __shfl_xor(cnt, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) c[e] += __shfl_xor(c[e], o);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = cnt > 0.f ? c[e] / cnt : 0.f;
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) zc[size_t(g) * 16 + 8 * hf + e] = c[e];
    }
    double d2 = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const double df = double(x[e]) - double(c[e]);
        d2 = fma(df, df, d2);
    }
    d2 += __shfl_xor(d2, 1);
    double rmax = real ? d2 : 0.0;
    float need = -INFINITY;
    if (real) {
        const float gv = gh[p];
        if (gv == INFINITY) need = -INFINITY;             // an orphan: collects nothing
        else if (!(gv > -1.0e38f)) need = INFINITY;       // no threshold: everything passes
        else {
            const double n2 = -2.0 * double(gv) + dmargin;
            const double nd = sqrt(n2 > 0.0 ? n2 : 0.0) * (1.0 + 1e-9);
            need = float(nd);
            if (double(need) <= nd) need = nextafterf(need, INFINITY);
        }
    }
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) {
        rmax = fmax(rmax, __shfl_xor(rmax, o));
        need = fmaxf(need, __shfl_xor(need, o));
    }
    if (lane == 0) {
        const double rr = sqrt(rmax) * (1.0 + 1e-9);
        float rf = float(rr);
        if (double(rf) <= rr) rf = nextafterf(rf, INFINITY);
        zrn[size_t(g) * 2 + 0] = rf;
        zrn[size_t(g) * 2 + 1] = need;
    }
}

// ---- listed walks of the one-stage collect (knn_select_kernel<MODE 2>, SymDev::walk_list) --------------------------------
// When the cell bounds leave too many (64 x 32) units for the queue of the cold launch, they may still rule out most TILES:
// on points that lie near a low-dimensional sheet the cells around a row are a few per cent of all cells, yet a cell of 244
// rows is compact only down to its own radius - millions of small units, each with a hit or two, none worth a unit of the
// cold launch (one memory round trip per unit) but cheap as 128-row tiles streamed through the LDS.  One workgroup per
// 256-row query block b: `closed` = the AND of the mask rows of the block's cells (bit c: no row of cell c and row of the block
// can be a pair either needs); position rel of the block's walk (tile (b TPB + rel) mod T) is listed when a cell of the
// tile's rows is not closed.  The list is ascending, so the walk keeps its order (own block first).  A list that does not
// fit its `cap` slots is given up (cnt = -1: the block walks everything).  total += tiles listed (the whole walk for a
// block without a list).
__global__ __launch_bounds__(256) void collect_lists_kernel(const int NB, const int T, const int TPB, const int walk, const int L,
                                                            const int words, const uint32_t* __restrict__ tcell,
                                                            const uint32_t* __restrict__ mask, const int cap, const int stride,
                                                            int32_t* __restrict__ tile_list, int32_t* __restrict__ tile_cnt,
                                                            unsigned long long* __restrict__ total) {
    extern __shared__ uint32_t closed[];   // [words]
    __shared__ int wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int spb = TPB * 4;   // 32-row sub-tiles per query block
    uint32_t qlo = 0xFFFFu, qhi = 0u;
    for (int s_ = 0; s_ < spb; ++s_) {
        const uint32_t c = tcell[size_t(b) * spb + s_];
        if ((c & 0xFFFFu) != 0xFFFFu) {
            qlo = (c & 0xFFFFu) < qlo ? (c & 0xFFFFu) : qlo;
            qhi = (c >> 16) > qhi ? (c >> 16) : qhi;
        }
    }
    if (qlo == 0xFFFFu) {   // pad rows only
        if (tid == 0) tile_cnt[b] = 0;
        return;
    }
    for (int wd = tid; wd < words; wd += 256) {
        uint32_t m = 0xFFFFFFFFu;
        for (uint32_t a = qlo; a <= qhi; ++a) m &= mask[size_t(a) * words + wd];
        closed[wd] = m;
    }
    __syncthreads();
    int count = 0;
    int32_t* out = tile_list + size_t(b) * size_t(stride);
    for (int rel0 = 0; rel0 < walk; rel0 += 256) {
        const int rel = rel0 + tid;
        bool need = false;
        if (rel < walk) {
            int t = b * TPB + rel;
            if (t >= T) t -= T;
            uint32_t dlo = 0xFFFFu, dhi = 0u;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const uint32_t c = tcell[size_t(t) * 4 + s_];
                if ((c & 0xFFFFu) != 0xFFFFu) {
                    dlo = (c & 0xFFFFu) < dlo ? (c & 0xFFFFu) : dlo;
                    dhi = (c >> 16) > dhi ? (c >> 16) : dhi;
                }
            }
            if (dlo != 0xFFFFu)
                for (uint32_t c = dlo; c <= dhi && !need; ++c) need = ((closed[c >> 5] >> (c & 31u)) & 1u) == 0u;
        }
        const unsigned long long bm = __ballot(need);
        if (lane == 0) wsum[w] = __popcll(bm);
        __syncthreads();
        int off = count;
        for (int v = 0; v < w; ++v) off += wsum[v];
        off += __popcll(bm & ((1ull << lane) - 1ull));
        if (need && off < cap) out[off] = rel;
        count += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) {
        const bool fits = count <= cap;
        tile_cnt[b] = fits ? count : -1;
        atomicAdd(total, (unsigned long long)(fits ? count : walk));
    }
}

}  // namespace

int gt_sym_gather(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, void* Ys, float* hs, float* hs_fin) {
    const int c16 = ctx->DP / 8;   // 2*DP bytes per row
    const int64_t total = n_pad_s * c16;
    hipLaunchKernelGGL(gather_sorted_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, ctx->stream,
                       ctx->Yc.as<uint4>(), ctx->hneg.as<float>(), perm, ctx->n, n_pad_s, c16, reinterpret_cast<uint4*>(Ys), hs, hs_fin);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_sym_gather_points(gt_ctx* ctx, const int32_t* perm, bool pad4) {
    KnnWork* k = ctx->knn;
    const size_t esz = ctx->dtype == GT_F32 ? 4 : 8;
    const int64_t n = ctx->n;
    const int d = ctx->d;
    k->xs_ready = false;
    k->xs_d = d;
    GT_HIP(ctx, k->xns.reserve(size_t(n) * sizeof(double)));
    if (pad4 && ctx->dtype == GT_F32 && (d & 3) != 0 && d <= 64 && ctx->rerank_lanes4 != 0) {
        const int dp = (d + 3) & ~3;
        GT_HIP(ctx, k->Xs.reserve(size_t(n) * dp * esz));
        hipLaunchKernelGGL(gather_points_pad_kernel, dim3((unsigned)ceil_div64(n * dp, 256)), dim3(256), 0, ctx->stream,
                           (const float*)ctx->X, ctx->xn.as<double>(), perm, n, d, dp, k->Xs.as<float>(), k->xns.as<double>());
        GT_HIP(ctx, hipGetLastError());
        k->xs_d = dp;
        k->xs_ready = true;
        return GT_OK;
    }
    GT_HIP(ctx, k->Xs.reserve(size_t(n) * d * esz));
    const size_t row_bytes = size_t(d) * esz;
    if (row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(ctx->X) & 15) == 0) {
        const int c16 = int(row_bytes / 16);
        hipLaunchKernelGGL(gather_points16_kernel, dim3((unsigned)ceil_div64(n * c16, 256)), dim3(256), 0, ctx->stream,
                           (const uint4*)ctx->X, ctx->xn.as<double>(), perm, n, c16, k->Xs.as<uint4>(), k->xns.as<double>());
    } else if (ctx->dtype == GT_F32) {
        hipLaunchKernelGGL(gather_points_kernel<float>, dim3((unsigned)ceil_div64(n * d, 256)), dim3(256), 0, ctx->stream,
                           (const float*)ctx->X, ctx->xn.as<double>(), perm, n, d, k->Xs.as<float>(), k->xns.as<double>());
    } else {
        hipLaunchKernelGGL(gather_points_kernel<double>, dim3((unsigned)ceil_div64(n * d, 256)), dim3(256), 0, ctx->stream,
                           (const double*)ctx->X, ctx->xn.as<double>(), perm, n, d, k->Xs.as<double>(), k->xns.as<double>());
    }
    GT_HIP(ctx, hipGetLastError());
    k->xs_ready = true;
    return GT_OK;
}

int gt_sym_thresholds(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const float* hs, const uint64_t* lists, int lstride,
                      const uint32_t* counts, int need_m, const ErrModel& err, double rkf, float* thr, float* g, float* gmin,
                      const DevBuf& work, int cells, unsigned long long* far_total, float* farcnt, int64_t p_first,
                      int64_t p_last, bool sample) {
    if (p_last < 0) p_last = n_pad_s;
    const dim3 grid((unsigned)ceil_div64(p_last - p_first, 16));
    // landmark adjacency the schedule of launch A was built from (gt_sym_schedule: nbr [L][M] at the head of `work`)
    const int M = std::min(std::min(cells, 32), ctx->order_L);
    const int32_t* nbr = work.as<int32_t>();
    // (... and the cells' runs behind it: start [L] | end [L])
    const int32_t* cstart = nbr + size_t(ctx->order_L) * M;
    const int32_t* cend = cstart + ctx->order_L;
    const int bn = gt_select_bn(ctx->DP);
    // (the strided sample of launch A, as every caller of gt_sym_schedule derives it)
    const int64_t n_tiles_s = n_pad_s / bn;
    // (... all of which belongs to the COHERENT numbering of the cells, gt_order.hip - it is made only where there is a strided
    //  sample.  In the landmarks' own order "far" is what it always was: a kept row outside the M cells around the row's own.)
    const int stride = (ctx->order_coherent_active != 0 && ctx->sym_stride > 0 && n_tiles_s >= int64_t(8) * ctx->sym_stride) ? ctx->sym_stride : 0;
    const bool skip = sample && stride > 0;
    if (stride <= 0) cstart = cend = nullptr;
    const int skip_t0 = skip ? int(p_first / bn) : 0, skip_t1 = skip ? int((p_last + bn - 1) / bn) : 0;
    const uint32_t* cell_sorted = ctx->order_cell.as<uint32_t>() + ctx->n;
    // the points in sorted order, when the caller has made the copy (gt_sym_gather_points)
    const KnnWork* kw = ctx->knn;
    const bool sorted = kw && kw->xs_ready;
    const void* Xp = sorted ? kw->Xs.p : ctx->X;
    const double* xnp = sorted ? kw->xns.as<double>() : ctx->xn.as<double>();
    const int dx = sorted ? kw->xs_d : ctx->d;   // (row stride = row length: a zero-padded copy adds exact zeros)
    if (ctx->dtype == GT_F32)
        hipLaunchKernelGGL(sym_thresholds_kernel<float>, grid, dim3(256), 0, ctx->stream, ctx->n, p_first, p_last, perm,
                           (const float*)Xp, dx, xnp, hs, lists, lstride, counts, need_m, ctx->ymax.as<double>(), err, rkf,
                           thr, g, cell_sorted, nbr, M, far_total, farcnt, sorted ? 1 : 0, ctx->metric, cstart, cend, bn, skip_t0, skip_t1, stride);
    else
        hipLaunchKernelGGL(sym_thresholds_kernel<double>, grid, dim3(256), 0, ctx->stream, ctx->n, p_first, p_last, perm,
                           (const double*)Xp, ctx->d, xnp, hs, lists, lstride, counts, need_m, ctx->ymax.as<double>(), err, rkf,
                           thr, g, cell_sorted, nbr, M, far_total, farcnt, sorted ? 1 : 0, ctx->metric, cstart, cend, bn, skip_t0, skip_t1, stride);
    GT_HIP(ctx, hipGetLastError());
    if (gmin) {
        hipLaunchKernelGGL(sym_gmin_kernel, dim3((unsigned)ceil_div64(n_pad_s, 256)), dim3(256), 0, ctx->stream, n_pad_s, g, gmin);
        GT_HIP(ctx, hipGetLastError());
    }
    return GT_OK;
}

int gt_sym_radius_sum(gt_ctx* ctx, const int32_t* perm, int64_t p_first, int64_t p_last, const float* thr, const ErrModel& err,
                      double* acc) {
    if (p_last <= p_first) return GT_OK;
    if (ctx->dtype == GT_F32)
        hipLaunchKernelGGL(sym_radius_sum_kernel<float>, dim3(256), dim3(256), 0, ctx->stream, ctx->n, p_first, p_last, perm,
                           (const float*)ctx->X, ctx->d, ctx->xn.as<double>(), thr, ctx->ymax.as<double>(), err, acc);
    else
        hipLaunchKernelGGL(sym_radius_sum_kernel<double>, dim3(256), dim3(256), 0, ctx->stream, ctx->n, p_first, p_last, perm,
                           (const double*)ctx->X, ctx->d, ctx->xn.as<double>(), thr, ctx->ymax.as<double>(), err, acc);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_sym_orphan_cut(gt_ctx* ctx, const int32_t* perm, float* thr, const float* farcnt, const ErrModel& err, const double* acc,
                      int need_m, double pair_frac) {
    hipLaunchKernelGGL(sym_orphan_cut_kernel, dim3((unsigned)ceil_div64(ctx->n, 256)), dim3(256), 0, ctx->stream, ctx->n, perm,
                       ctx->xn.as<double>(), thr, farcnt, ctx->ymax.as<double>(), err, acc, ctx->sym_radius_cut,
                       ctx->sym_orphan_far, need_m, pair_frac);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_sym_inject_orphans(gt_ctx* ctx, int64_t p_first, int64_t p_last, const float* thr, const uint64_t* lists, int lstride,
                          const uint32_t* counts, uint64_t* tlists, int tcap, uint32_t* tcounts) {
    if (p_last <= p_first) return GT_OK;
    hipLaunchKernelGGL(sym_inject_orphans_kernel, dim3((unsigned)ceil_div64(p_last - p_first, 16)), dim3(256), 0, ctx->stream,
                       ctx->n, p_first, p_last, thr, lists, lstride, counts, tlists, tcap, tcounts);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

// thresholds of all rows (gathered from the ranks that seeded them) -> transposed form g and its sub-tile minima
int gt_sym_g_from_thr(gt_ctx* ctx, int64_t n_pad_s, const float* thr, const float* hs, float* g, float* gmin) {
    hipLaunchKernelGGL(sym_g_kernel, dim3((unsigned)ceil_div64(n_pad_s, 256)), dim3(256), 0, ctx->stream, ctx->n, n_pad_s, thr,
                       hs, g);
    GT_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(sym_gmin_kernel, dim3((unsigned)ceil_div64(n_pad_s, 256)), dim3(256), 0, ctx->stream, n_pad_s, g, gmin);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

// Stage-one copy of the points (see "stage-one subspace" above): P_host [16][64] float (rows orthonormal), scale scz
int gt_sym_project(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const float* P_dev, double scz, void* Z, float* hh) {
    const dim3 grid((unsigned)ceil_div64(n_pad_s, 16));
    if (ctx->dtype == GT_F32)
        hipLaunchKernelGGL(sym_project_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)ctx->X, perm, ctx->n, n_pad_s,
                           ctx->d, P_dev, float(scz), reinterpret_cast<_Float16*>(Z), hh);
    else
        hipLaunchKernelGGL(sym_project_kernel<double>, grid, dim3(256), 0, ctx->stream, (const double*)ctx->X, perm, ctx->n,
                           n_pad_s, ctx->d, P_dev, float(scz), reinterpret_cast<_Float16*>(Z), hh);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

// covariance (about the sample mean) of every step-th row: C_dev [64][64] doubles, sums_dev [64] (both pre-zeroed here)
int gt_sym_sample_cov(gt_ctx* ctx, int64_t step, int64_t ns, double* sums_dev, double* C_dev) {
    if (ctx->d > 64) GT_FAIL(ctx, GT_E_ARG, "sample covariance: at most 64 features");
    GT_HIP(ctx, hipMemsetAsync(sums_dev, 0, 64 * sizeof(double), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(C_dev, 0, 64 * 64 * sizeof(double), ctx->stream));
    if (ctx->dtype == GT_F32) {
        hipLaunchKernelGGL(sample_colsum_kernel<float>, dim3(64), dim3(256), 0, ctx->stream, (const float*)ctx->X, ctx->n, ctx->d,
                           step, ns, sums_dev);
        hipLaunchKernelGGL(sample_cov_kernel<float>, dim3((unsigned)ceil_div64(ns, 64)), dim3(256), 0, ctx->stream,
                           (const float*)ctx->X, ctx->n, ctx->d, step, ns, sums_dev, C_dev);
    } else {
        hipLaunchKernelGGL(sample_colsum_kernel<double>, dim3(64), dim3(256), 0, ctx->stream, (const double*)ctx->X, ctx->n,
                           ctx->d, step, ns, sums_dev);
        hipLaunchKernelGGL(sample_cov_kernel<double>, dim3((unsigned)ceil_div64(ns, 64)), dim3(256), 0, ctx->stream,
                           (const double*)ctx->X, ctx->n, ctx->d, step, ns, sums_dev, C_dev);
    }
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_sym_half_thresholds(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const float* thr, const float* hh,
                           const ErrModel& err, int hd, double scz, double Lz, float* thrh, float* gh, float* gminh,
                           float* rrow) {
    hipLaunchKernelGGL(sym_half_thresholds_kernel, dim3((unsigned)ceil_div64(n_pad_s, 256)), dim3(256), 0, ctx->stream, ctx->n,
                       n_pad_s, perm, ctx->xn.as<double>(), thr, hh, ctx->ymax.as<double>(), err, hd, scz, Lz, thrh, gh, rrow);
    GT_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(sym_gmin_kernel, dim3((unsigned)ceil_div64(n_pad_s, 256)), dim3(256), 0, ctx->stream, n_pad_s, gh, gminh);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_sym_two_probe(gt_ctx* ctx, const void* Ys, int hd, const float* hh, const float* thrh, const float* gh, int64_t samples,
                     uint32_t* flagged) {
    if (ctx->n < 64 || hd > kZ) GT_FAIL(ctx, GT_E_ARG, "sym probe: bad shape");   // Ys: the stage-one copy, kZ columns
    GT_HIP(ctx, hipMemsetAsync(flagged, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(sym_two_probe_kernel, dim3((unsigned)ceil_div64(samples, 4)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const _Float16*>(Ys), ctx->n, kZ, hd, hh, thrh, gh, samples, flagged);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

// Bound pass (see cell_ball_kernel): fills `queue` (capacity cap entries) with the units of the two-stage collect's
// walks (1024-row query blocks, 128-row tiles) that the cell bounds leave, *count = how many there are (beyond cap: the
// caller runs the collect launch instead).  Ys: the sorted compact copy [n_pad][DP] float16, rrow: the rows' radii in
// it; work: scratch.
int gt_sym_bound_queue(gt_ctx* ctx, int64_t n_pad_s, const void* Ys, const float* rrow, DevBuf& work, uint2* queue,
                       uint32_t cap, uint32_t* count_dev, int world, int rank, int group, int64_t own_p0, int64_t own_p1) {
    const int L = ctx->order_L;
    if (L <= 0 || L > 65535) GT_FAIL(ctx, GT_E_STATE, "bound pass: no landmark cells");
    if (n_pad_s % 1024 != 0) GT_FAIL(ctx, GT_E_ARG, "bound pass: whole 1024-row query blocks");
    const int words = (L + 31) / 32;
    const int64_t nt32 = n_pad_s / 32;
    const int DP = ctx->DP;
    if (DP > 128) GT_FAIL(ctx, GT_E_ARG, "bound pass: at most 128 padded features");
    // work: start [L] | end [L] | centre [L][DP] | radius [L] | need [L] | mask [L][words] | tcell [n_pad / 32]
    const size_t bytes = (size_t(L) * (2 + DP + 2) + size_t(L) * words + size_t(nt32) + 4) * 4;   // (+ the forecast, a double)
    GT_HIP(ctx, work.reserve(bytes));
    int32_t* start = work.as<int32_t>();
    int32_t* endp = start + L;
    float* centre = reinterpret_cast<float*>(endp + L);
    float* radius = centre + size_t(L) * DP;
    float* need = radius + L;
    uint32_t* mask = reinterpret_cast<uint32_t*>(need + L);
    uint32_t* tcell = mask + size_t(L) * words;
    const uint32_t* cell_sorted = ctx->order_cell.as<uint32_t>() + ctx->n;
    GT_HIP(ctx, hipMemsetAsync(start, 0xFF, 2 * size_t(L) * sizeof(int32_t), ctx->stream));   // -1: empty cell
    GT_HIP(ctx, hipMemsetAsync(count_dev, 0, 2 * sizeof(uint32_t), ctx->stream));   // [0] this rank's units, [1] all ranks'
    const bool own = own_p1 > own_p0;
    const int64_t own_pl = own ? std::min<int64_t>(own_p1, ctx->n) : 0;
    hipLaunchKernelGGL(cell_ranges_kernel, dim3((unsigned)ceil_div64(ctx->n, 256)), dim3(256), 0, ctx->stream, cell_sorted,
                       ctx->n, start, endp);
    hipLaunchKernelGGL(cell_ball_kernel, dim3((unsigned)L), dim3(256), 0, ctx->stream, reinterpret_cast<const _Float16*>(Ys),
                       DP, rrow, start, endp, centre, radius, need);
    {
        const dim3 grid((unsigned)ceil_div64(L, 256), (unsigned)ceil_div64(L, 64));
#define GT_MASK_CASE(DPC_)                                                                                               \
    case DPC_:                                                                                                           \
        hipLaunchKernelGGL(cell_mask_kernel<DPC_>, grid, dim3(256), 0, ctx->stream, L, centre, radius, need, mask,       \
                           own ? cell_sorted + own_p0 : nullptr, own ? cell_sorted + (own_pl - 1) : nullptr);            \
        break;
        switch (DP) {
            GT_MASK_CASE(16) GT_MASK_CASE(32) GT_MASK_CASE(48) GT_MASK_CASE(64) GT_MASK_CASE(80) GT_MASK_CASE(96)
            GT_MASK_CASE(112) GT_MASK_CASE(128)
            default: GT_FAIL(ctx, GT_E_ARG, "bound pass: unexpected padded feature count");
        }
#undef GT_MASK_CASE
    }
    hipLaunchKernelGGL(tile_cells_kernel, dim3((unsigned)ceil_div64(nt32, 256)), dim3(256), 0, ctx->stream, cell_sorted, ctx->n,
                       nt32, tcell);
    GT_HIP(ctx, hipGetLastError());
    if (ctx->dbg_select & 2048) {   // development: what the cells look like
        std::vector<float> rad(L), nd(L);
        std::vector<uint32_t> mk(size_t(L) * words);
        std::vector<int32_t> st(2 * size_t(L));
        GT_HIP(ctx, hipMemcpyAsync(rad.data(), radius, L * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipMemcpyAsync(nd.data(), need, L * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipMemcpyAsync(mk.data(), mask, mk.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipMemcpyAsync(st.data(), start, st.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<float> r2, n2;
        int64_t open_pairs = 0, infn = 0, empty = 0;
        for (int c = 0; c < L; ++c) {
            if (st[c] < 0) {
                ++empty;
                continue;
            }
            r2.push_back(rad[c]);
            if (nd[c] == INFINITY) ++infn;
            else if (nd[c] > -INFINITY) n2.push_back(nd[c]);
            for (int b = 0; b < L; ++b)
                if (st[b] >= 0 && !((mk[size_t(c) * words + (b >> 5)] >> (b & 31)) & 1u)) ++open_pairs;
        }
        std::sort(r2.begin(), r2.end());
        std::sort(n2.begin(), n2.end());
        auto q = [](const std::vector<float>& v, double f) { return v.empty() ? 0.f : v[size_t(f * (v.size() - 1))]; };
        fprintf(stderr, "[gt] bound pass cells: %d (%lld empty), radius min/median/90%%/max %.4g %.4g %.4g %.4g, need median/90%%/99%%/max %.4g %.4g %.4g %.4g (%lld cells need everything), undecided cell pairs %lld\n",
                L, (long long)empty, q(r2, 0), q(r2, 0.5), q(r2, 0.9), q(r2, 1.0), q(n2, 0.5), q(n2, 0.9), q(n2, 0.99), q(n2, 1.0),
                (long long)infn, (long long)open_pairs);
    }
    const double* est_ptr = nullptr;
    {
        // forecast: far more undecided units than the queue holds -> the caller falls through to the collect launch without
        // paying for their enumeration (the same verdict on every rank: the cells are the same everywhere; a rank's own
        // groups against every sub-tile - round 6: the enumeration of 25 M units it would only count cost the manifold set 1.7 ms
        // per rank - ask for their own cells)
        double* est_dev = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(tcell + nt32) + 7) & ~uintptr_t(7));
        GT_HIP(ctx, hipMemsetAsync(est_dev, 0, sizeof(double), ctx->stream));
        hipLaunchKernelGGL(bound_estimate_kernel, dim3((unsigned)L), dim3(64), 0, ctx->stream, L, start, endp, mask, words, est_dev,
                           own ? cell_sorted + own_p0 : nullptr, own ? cell_sorted + (own_pl - 1) : nullptr);
        GT_HIP(ctx, hipGetLastError());
        est_ptr = est_dev;   // (read by the enumeration itself: no trip to the host)
    }
    unsigned long long* dbg_cyc = nullptr;
    DevBuf dbg_buf;
    if (ctx->dbg_select & 4096) {   // development: cycles per query group of the enumeration
        GT_HIP(ctx, dbg_buf.reserve(size_t(n_pad_s / 64 + 8) * sizeof(unsigned long long)));
        GT_HIP(ctx, hipMemsetAsync(dbg_buf.p, 0, size_t(n_pad_s / 64 + 8) * sizeof(unsigned long long), ctx->stream));
        dbg_cyc = dbg_buf.as<unsigned long long>();
    }
    const int T = int(n_pad_s / 128), TPB = 8, NB = T / TPB, H = (NB - 1) / 2;
    const int walk = TPB * (1 + H) + ((NB & 1) ? 0 : (NB > 1 ? TPB : 0));
    if (own_p1 > own_p0) {
        // the rank's own query groups [own_p0 / 64, own_p1 / 64) against every sub-tile (own_p0 a multiple of 64)
        const int q_lo = int(own_p0 / 64), q_hi = int((own_p1 + 63) / 64);
        hipLaunchKernelGGL(bound_queue_kernel, dim3((unsigned)((q_hi - q_lo + 3) / 4)), dim3(256), 0, ctx->stream, q_hi, T, TPB, walk, L,
                           1, 0, 1, q_lo, 1, tcell, start, endp, mask, words, queue, cap, count_dev, est_ptr, ctx->n, dbg_cyc);
        GT_HIP(ctx, hipGetLastError());
        return GT_OK;
    }
    hipLaunchKernelGGL(bound_queue_kernel, dim3((unsigned)((NB * 2 * TPB + 3) / 4)), dim3(256), 0, ctx->stream, NB * 2 * TPB, T, TPB, walk, L,
                       std::max(world, 1), rank, std::max(group, 1), 0, 0, tcell, start,
                       endp, mask, words, queue, cap, count_dev, est_ptr, ctx->n, dbg_cyc);
    GT_HIP(ctx, hipGetLastError());
    if (dbg_cyc) {
        std::vector<unsigned long long> cyc(size_t(n_pad_s / 64));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipMemcpy(cyc.data(), dbg_cyc, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<std::pair<unsigned long long, size_t>> top;
        for (size_t g = 0; g < cyc.size(); ++g) top.emplace_back(cyc[g], g);
        std::sort(top.rbegin(), top.rend());
        const uint32_t* cs = ctx->order_cell.as<uint32_t>() + ctx->n;
        for (int t = 0; t < 6 && t < int(top.size()); ++t) {
            uint32_t c0 = 0, c1 = 0;
            (void)hipMemcpy(&c0, cs + std::min<size_t>(top[t].second * 64, size_t(ctx->n - 1)), 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&c1, cs + std::min<size_t>(top[t].second * 64 + 63, size_t(ctx->n - 1)), 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "[gt] bound queue: group %zu took %llu cycles (cells %u .. %u)\n", top[t].second, top[t].first, c0, c1);
        }
        unsigned long long med = top[top.size() / 2].first;
        fprintf(stderr, "[gt] bound queue: median group %llu cycles\n", med);
        dbg_buf.release();
    }
    return GT_OK;
}


// balls of the groups of 32 sorted rows in the stage-one space (z_balls_kernel): zc [n_pad_s / 32][16], zrn [n_pad_s / 32][2]
int gt_sym_z_balls(gt_ctx* ctx, int64_t n_pad_s, const void* Z, const float* gh, double dmargin, float* zc, float* zrn) {
    const int64_t ng = n_pad_s / 32;
    hipLaunchKernelGGL(z_balls_kernel, dim3((unsigned)ceil_div64(ng, 4)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const _Float16*>(Z), gh, ctx->n, ng, dmargin, zc, zrn);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

// Listed walks for the one-stage collect from the cell masks the bound pass left in `work` (gt_sym_bound_queue ran on this
// point set and order): tile_list [NB][stride], tile_cnt [NB] for the NB = n_pad_s / 256 query blocks, *total_dev (pre-zeroed
// here) = tiles listed in all.  Returns the length of a whole walk in *walk_out.
int gt_sym_collect_lists(gt_ctx* ctx, int64_t n_pad_s, DevBuf& work, int cap, int stride, int32_t* tile_list, int32_t* tile_cnt,
                         unsigned long long* total_dev, int* walk_out) {
    const int L = ctx->order_L;
    if (L <= 0 || L > 65535 || !work.p) GT_FAIL(ctx, GT_E_STATE, "collect lists: no cell masks");
    if (n_pad_s % 256 != 0 || ctx->DP > 64) GT_FAIL(ctx, GT_E_ARG, "collect lists: whole 256-row query blocks of 128-row tiles");
    const int words = (L + 31) / 32;
    const int DP = ctx->DP;
    // (the layout of gt_sym_bound_queue's scratch)
    int32_t* start = work.as<int32_t>();
    int32_t* endp = start + L;
    float* centre = reinterpret_cast<float*>(endp + L);
    float* radius = centre + size_t(L) * DP;
    float* need = radius + L;
    const uint32_t* mask = reinterpret_cast<const uint32_t*>(need + L);
    const uint32_t* tcell = mask + size_t(L) * words;
    const int T = int(n_pad_s / 128), TPB = 2, NB = T / TPB, H = (NB - 1) / 2;
    const int walk = TPB * (1 + H) + ((NB & 1) ? 0 : (NB > 1 ? TPB : 0));
    GT_HIP(ctx, hipMemsetAsync(total_dev, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(collect_lists_kernel, dim3((unsigned)NB), dim3(256), size_t(words) * sizeof(uint32_t), ctx->stream, NB, T, TPB,
                       walk, L, words, tcell, mask, cap, stride, tile_list, tile_cnt, total_dev);
    GT_HIP(ctx, hipGetLastError());
    if (walk_out) *walk_out = walk;
    return GT_OK;
}

int gt_sym_row_radius(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const float* thr, const ErrModel& err, float* rrow,
                      double lres) {
    hipLaunchKernelGGL(sym_row_radius_kernel, dim3((unsigned)ceil_div64(n_pad_s, 256)), dim3(256), 0, ctx->stream, ctx->n, n_pad_s,
                       perm, ctx->xn.as<double>(), thr, ctx->ymax.as<double>(), err, rrow, lres);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_sym_invperm(gt_ctx* ctx, const int32_t* perm, int32_t* inv) {
    hipLaunchKernelGGL(invperm_kernel, dim3((unsigned)ceil_div64(ctx->n, 256)), dim3(256), 0, ctx->stream, perm, ctx->n, inv);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_sym_own_rows(gt_ctx* ctx, const int32_t* perm, int64_t r0, int64_t r1, int32_t* own, DevBuf& tmp) {
    // stream compaction of perm by ownership (order kept): rocPRIM select; the count lands behind the rows
    OwnedRow pred{int32_t(r0), int32_t(r1)};
    size_t bytes = 0;
    unsigned int* count = reinterpret_cast<unsigned int*>(own + (r1 - r0));
    GT_HIP(ctx, rocprim::select(nullptr, bytes, perm, own, count, size_t(ctx->n), pred, ctx->stream));
    GT_HIP(ctx, tmp.reserve(bytes));
    GT_HIP(ctx, rocprim::select(tmp.p, bytes, perm, own, count, size_t(ctx->n), pred, ctx->stream));
    return GT_OK;
}

static ShardSplits make_splits(int world, const int64_t* splits) {
    ShardSplits sp;
    sp.world = world;
    for (int r = 0; r <= world; ++r) sp.s[r] = splits[r];
    return sp;
}

int gt_sym_shard_count(gt_ctx* ctx, const int32_t* perm, const uint32_t* tcounts, int tcap, int world, const int64_t* splits,
                       unsigned long long* cnt) {
    if (world < 1 || world > GT_SYM_MAX_WORLD) GT_FAIL(ctx, GT_E_ARG, "sym shard: bad world size");
    GT_HIP(ctx, hipMemsetAsync(cnt, 0, size_t(world) * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(shard_count_kernel, dim3((unsigned)ceil_div64(ctx->n, 256)), dim3(256), 0, ctx->stream, ctx->n, perm,
                       tcounts, tcap, make_splits(world, splits), cnt);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_sym_shard_emit(gt_ctx* ctx, const int32_t* perm, const uint64_t* tlists, const uint32_t* tcounts, int tcap, int world,
                      const int64_t* splits, unsigned long long* cursor, void* out) {
    hipLaunchKernelGGL(shard_emit_kernel, dim3((unsigned)ceil_div64(ctx->n, 256)), dim3(256), 0, ctx->stream, ctx->n, perm, tlists,
                       tcounts, tcap, make_splits(world, splits), cursor, reinterpret_cast<uint4*>(out));
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_sym_shard_scatter(gt_ctx* ctx, const void* recs, int64_t n_recs, int64_t nloc, int tcap, uint64_t* lists,
                         uint32_t* counts, uint32_t* bad) {
    if (n_recs <= 0) return GT_OK;
    hipLaunchKernelGGL(shard_scatter_kernel, dim3((unsigned)ceil_div64(n_recs, 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const uint4*>(recs), n_recs, nloc, tcap, lists, counts, bad);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

// Tile lists of launch A for the NB query blocks of the sorted order (tile_list [NB][tile_stride], tile_cnt [NB]).
// Uses the landmark rows and the sorted cell ids the query ordering left in the context (gt_order.hip).
int gt_sym_schedule(gt_ctx* ctx, int64_t n_pad_s, int bq, int bn, int cells, int stride, int max_nb, int tile_stride,
                    DevBuf& work, int32_t* tile_list, int32_t* tile_cnt, unsigned long long* tiles_total, int64_t p_first,
                    int64_t p_last) {
    const int L = ctx->order_L;
    if (L <= 0 || !ctx->land_Y.p) GT_FAIL(ctx, GT_E_STATE, "sym schedule: no landmark cells");
    const int M = std::min(std::min(cells, 32), L);
    // work: nbr [L][M] | start [L] | end [L]
    GT_HIP(ctx, work.reserve((size_t(L) * M + 2 * size_t(L)) * sizeof(int32_t)));
    int32_t* nbr = work.as<int32_t>();
    int32_t* start = nbr + size_t(L) * M;
    int32_t* endp = start + L;
    GT_HIP(ctx, hipMemsetAsync(start, 0xFF, 2 * size_t(L) * sizeof(int32_t), ctx->stream));   // -1: empty cell
    const uint32_t* cell_sorted = ctx->order_cell.as<uint32_t>() + ctx->n;
    // p_last > p_first: tile lists (and landmark neighbourhoods) for the query blocks of the sorted positions [p_first, p_last)
    // only - a rank's own rows; the cells of those rows are a run of cell numbers (the order is sorted by cell)
    const bool part = p_last > p_first;
    const int64_t pl = part ? std::min<int64_t>(p_last, ctx->n) : 0;
    hipLaunchKernelGGL(landmark_neighbours_kernel, dim3((unsigned)L), dim3(256), size_t(L) * sizeof(float), ctx->stream,
                       ctx->land_Y.as<_Float16>(), L, ctx->DP, M, nbr, part ? cell_sorted + p_first : nullptr,
                       part ? cell_sorted + (pl - 1) : nullptr);
    GT_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(cell_ranges_kernel, dim3((unsigned)ceil_div64(ctx->n, 256)), dim3(256), 0, ctx->stream, cell_sorted,
                       ctx->n, start, endp);
    GT_HIP(ctx, hipGetLastError());
    const int NB = int(n_pad_s / bq), T = int(n_pad_s / bn);
    const int b0 = part ? int(p_first / bq) : 0, nblk = part ? int(ceil_div64(p_last, bq)) - b0 : NB;
    hipLaunchKernelGGL(sym_schedule_kernel, dim3((unsigned)nblk), dim3(64), size_t((T + 31) / 32) * sizeof(uint32_t), ctx->stream,
                       cell_sorted, ctx->n, start, endp, nbr, M, NB, bq, bn, T, stride, max_nb, tile_stride, tile_list, tile_cnt, b0,
                       ctx->order_coherent_active != 0 ? ctx->order_outlier_cell : -1);
    GT_HIP(ctx, hipGetLastError());
    if (tiles_total && !part) {   // statistics: tiles launch A visits in all (device counter, pre-zeroed by the caller)
        hipLaunchKernelGGL(sum_i32_kernel, dim3(16), dim3(256), 0, ctx->stream, tile_cnt, int64_t(NB), tiles_total);
        GT_HIP(ctx, hipGetLastError());
    }
    return GT_OK;
}
