// Exact fp64 stage of the kNN search.
//
// rerank_kernel: for every query, recompute the squared distance of each candidate exactly the way
//   scikit-learn's EuclideanArgKmin does - d2 = (|x|^2 + (-2 x.y)) + |y|^2 in float64 with float32 inputs
//   upcast (sklearn:_argkmin.pyx.tp:497-505) - sort the candidates by (d2, index), and derive the
//   completeness bound d2_lb: the candidate kernel rejected a database row only when its float32 score
//   s~ = x.y - |y|^2/2 was <= thr, and |s~ - s| <= e (e from the fp32 accumulation bound), hence every
//   rejected row has d2 = |x|^2 - 2 s >= |x|^2 - 2 (thr + e) =: d2_lb.  The first m table entries are
//   the exact m nearest neighbours iff d2[m-1] < d2_lb; rows that fail go to
// fallback_kernel: exhaustive float64 distances to every database row + an exact selection of the m
//   smallest (d2, index) pairs (64-step bitwise search for the m-th key, ordered collection, sort).
//
// One wave per query; candidates are spread over lanes (MP/64 per lane), the query row sits in LDS as
// float64.  Work is O(nq * MP * d) fp64 FMAs + a random gather of MP database rows per query - small
// next to the O(nq * n * d) candidate pass.
#include "gt_common.h"
#include "gt_device.h"
#include "gt_knn.h"

namespace {

constexpr unsigned long long kInfBits = 0x7FF0000000000000ull;

// (the canonical summation order of the exact stages: gt_device.h gt_dot16)
template <typename T>
__device__ __forceinline__ double dot_row(const double* __restrict__ xs, const T* __restrict__ y, int d) {
    return gt_dot16(xs, y, d);
}
template <>
__device__ __forceinline__ double dot_row<float>(const double* __restrict__ xs, const float* __restrict__ y, int d) {
    if ((d & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) & 15) == 0)) return gt_dot16_f4(xs, y, d);
    return gt_dot16(xs, y, d);
}

// F4 (host-decided: float32 rows, d a multiple of 4, 16-byte aligned): the 16-byte-load form alone is instantiated - the
// general form next to it costs the re-rank kernels two thirds of their occupancy in registers
template <typename T, bool F4>
__device__ __forceinline__ double dot_row_sel(const double* __restrict__ xs, const T* __restrict__ y, int d) {
    if constexpr (F4 && sizeof(T) == 4) return gt_dot16_f4(xs, reinterpret_cast<const float*>(y), d);
    else return gt_dot16(xs, y, d);
}

// WT (tables of 256 slots): next to every key the key the OTHER row holds for the same pair (cand_d2t: the same dot product with
// the roles swapped, as in rerank_sym4_kernel), keyt_ok / nokeyt_rows: which rows' tables carry them - so that builds whose
// candidates came from the classic pass (isotropic data) take the pair-resolved tail too (round 6; the verdicts' "transposed keys
// through the classic re-rank").  The keys travel through the sorts as a payload parked in the LDS (wave_sort_asc_pair_fast).
template <typename T, int NT2, bool F4, bool WT = false>
__global__ __launch_bounds__(256) void rerank_kernel(const T* __restrict__ X, const int d, const double* __restrict__ xn,
                                                     const T* __restrict__ Q, const double* __restrict__ qn,
                                                     const double* __restrict__ qn_sel, const int64_t q0, const int64_t nq,
                                                     const uint64_t* __restrict__ lists, const int lstride,
                                                     const uint32_t* __restrict__ counts,
                                                     const float* __restrict__ thr_final,
                                                     const double* __restrict__ ymax2p, const ErrModel err,
                                                     const int metric, const int need_m, double* __restrict__ cand_d2,
                                                     uint32_t* __restrict__ cand_j, uint32_t* __restrict__ cand_n,
                                                     double* __restrict__ d2_lb, uint32_t* __restrict__ fb_count,
                                                     int32_t* __restrict__ fb_rows, uint32_t* __restrict__ gflags,
                                                     const double radius_key_factor, uint32_t* __restrict__ unproven,
                                                     const int32_t* __restrict__ qrows, double* __restrict__ cand_d2t = nullptr,
                                                     uint8_t* __restrict__ keyt_ok = nullptr, int32_t* __restrict__ nokeyt_rows = nullptr,
                                                     uint32_t* __restrict__ nokeyt_count = nullptr) {
    constexpr int MP = NT2 * 64;
    static_assert(!WT || NT2 <= 8, "the payload rides through wave_sort_asc_pair_fast");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    double* xs = reinterpret_cast<double*>(smem_raw) + size_t(w) * d;
    // (WT: behind the waves' query rows, per wave MP parked keys | MP transposed keys | MP row numbers)
    uint64_t* park_hi = nullptr;
    uint64_t* park_x = nullptr;
    uint32_t* park_lo = nullptr;
    if constexpr (WT) {
        const int nw = int(blockDim.x >> 6);
        uint64_t* pb = reinterpret_cast<uint64_t*>(smem_raw) + size_t(nw) * d;
        park_hi = pb + size_t(w) * MP;
        park_x = pb + size_t(nw) * MP + size_t(w) * MP;
        park_lo = reinterpret_cast<uint32_t*>(pb + size_t(2) * nw * MP) + size_t(w) * MP;
    }
    uint64_t hx[NT2];
#pragma unroll
    for (int u = 0; u < NT2; ++u) hx[u] = 0ull;
    // Workgroups are dealt to the 8 XCDs round robin; consecutive lists belong to neighbouring queries (same landmark
    // cell) and re-rank largely the same database rows, so each XCD takes one contiguous eighth of the lists and finds
    // those rows in its own L2.
    const int64_t nb = gridDim.x, xcd = blockIdx.x & 7, base = nb >> 3, rem = nb & 7;
    const int64_t bid = xcd * base + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    const int64_t ql = bid * (blockDim.x >> 6) + w;   // list index: the order the candidate pass dealt the queries in
    if (ql >= nq) return;   // whole wave exits together (ql is wave-uniform); no block-level sync below
    const int64_t q = qrows ? int64_t(qrows[ql]) - q0 : ql;   // row of the tables (rows [q0, q0 + nq) in their own order)

    const T* xrow = Q + (q0 + q) * int64_t(d);
    for (int k = lane; k < d; k += 64) xs[k] = double(xrow[k]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double qnq = qn[q0 + q];
    const uint32_t cnt = counts[ql];
    const uint32_t n = cnt < uint32_t(MP) ? cnt : uint32_t(MP);
    const uint64_t* lp = lists + size_t(ql) * lstride;

    // completeness bound of the candidate pass: every row it rejected or dropped scored <= thr_final
    // qs, y2: the norms of what the candidate pass scored (a coordinate subset for wide data: its squared distance
    // is a lower bound of the full one, so the bounds below hold for the full distance as well)
    const double qs = qn_sel[q0 + q];
    const double y2 = ymax2p[0];    // largest squared norm over the scored columns
    const double y2f = ymax2p[1];   // and over all columns
    const double e = gt_err_bound(err, qs, y2);
    // euclidean: d2 = |x|^2 - 2 s ; cosine: D = 1 - x.y = 1 - s - |y|^2/2 >= 1 - s - ymax^2/2
    auto bound_of_score = [&](float score) {
        const double sv = double(score) * err.inv_sc2;
        // cosine: D = 1 - x.y, x.y = x_S.y_S + x_R.y_R <= (s + |y_S|^2/2) + (|x_R|^2 + |y_R|^2)/2  (S: scored columns)
        const double b = (metric == 1) ? (1.0 - (sv + e) - 0.5 * (qnq - qs) - 0.5 * y2f) : (qs - 2.0 * (sv + e));
        return b - 1e-9 * (qs + y2);   // float64 rounding of the quantities above, with a wide margin
    };
    double lb = INFINITY;
    const float thr_f = thr_final[ql];
    if (thr_f > -INFINITY) lb = bound_of_score(thr_f);   // -inf: the candidate pass rejected nothing for this query

    uint64_t hi[NT2], lo[NT2];
    uint32_t n_tab = n;
    const int pos = need_m - 1;
    bool settled = false;
    const bool may_stop = radius_key_factor > 0.0;
    const double rkf = fabs(radius_key_factor);
    if constexpr (NT2 == 4) {
        // Two batches: the 128 candidates with the best approximate scores are evaluated exactly first.  Every other
        // candidate scored at most s_129, i.e. lies beyond bound_of_score(s_129); if that bound (and the pass bound)
        // clears what the caller needs - the need_m-th key times radius_key_factor - the second batch of row gathers
        // (half of this kernel's HBM traffic) is skipped and the table ends at 128 entries.
        uint64_t ks[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t c = uint32_t(u * 64 + lane);
            ks[u] = (c < n) ? lp[c] : 0ull;   // a valid key is never 0
        }
        wave_bitonic_desc<4>(ks, lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            hi[u] = kInfBits;
            lo[u] = 0xFFFFFFFFull;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ks[u] != 0ull) {
                const uint32_t j = cand_index(ks[u]);
                const double dot = dot_row_sel<T, F4>(xs, X + int64_t(j) * d, d);
                const double xnj = xn[j];
                hi[u] = (uint64_t)__double_as_longlong(gt_pair_key(qnq, dot, xnj, metric));
                lo[u] = j;
                if constexpr (WT) hx[u] = (uint64_t)__double_as_longlong(gt_pair_key(xnj, dot, qnq, metric));
            }
        }
        const uint64_t k129 = __shfl((unsigned long long)ks[2], 0);
        const double lb_rest = (k129 != 0ull) ? bound_of_score(cand_score(k129)) : INFINITY;
        uint64_t h2[2] = {hi[0], hi[1]}, l2[2] = {lo[0], lo[1]};
        uint64_t x2[2] = {hx[0], hx[1]};
        wave_sort_asc_pair_fast<2>(h2, l2, lane, park_hi, park_lo, WT ? x2 : nullptr, park_x);
        const uint64_t sel2 = (pos >> 6) == 0 ? h2[0] : h2[1];
        const double need2 = __longlong_as_double((long long)__shfl((unsigned long long)sel2, pos & 63));
        const double lbm = fmin(lb, lb_rest);
        if (may_stop && pos < 128 && need2 * rkf < lbm) {   // wave-uniform
            hi[0] = h2[0]; hi[1] = h2[1];
            lo[0] = l2[0]; lo[1] = l2[1];
            hx[0] = x2[0]; hx[1] = x2[1];
            lb = lbm;
            n_tab = n < 128u ? n : 128u;
            settled = true;
        } else {
#pragma unroll
            for (int u = 2; u < 4; ++u) {
                if (ks[u] != 0ull) {
                    const uint32_t j = cand_index(ks[u]);
                    const double dot = dot_row_sel<T, F4>(xs, X + int64_t(j) * d, d);
                    const double xnj = xn[j];
                    hi[u] = (uint64_t)__double_as_longlong(gt_pair_key(qnq, dot, xnj, metric));
                    lo[u] = j;
                    if constexpr (WT) hx[u] = (uint64_t)__double_as_longlong(gt_pair_key(xnj, dot, qnq, metric));
                }
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            const uint32_t c = uint32_t(u * 64 + lane);
            hi[u] = kInfBits;
            lo[u] = 0xFFFFFFFFull;
            if (c < n) {
                const uint32_t j = cand_index(lp[c]);
                const double dot = dot_row_sel<T, F4>(xs, X + int64_t(j) * d, d);
                const double xnj = xn[j];
                hi[u] = (uint64_t)__double_as_longlong(gt_pair_key(qnq, dot, xnj, metric));
                lo[u] = j;
                if constexpr (WT) hx[u] = (uint64_t)__double_as_longlong(gt_pair_key(xnj, dot, qnq, metric));
            }
        }
    }
    if (!settled) {
        if constexpr (NT2 <= 8) wave_sort_asc_pair_fast<NT2>(hi, lo, lane, park_hi, park_lo, WT ? hx : nullptr, park_x);
        else wave_bitonic_asc_pair<NT2>(hi, lo, lane);
    }
    // slots [0, max(n_tab, need_m)) are defined (+inf keys past the last candidate); nobody reads further
    const uint32_t n_def = n_tab > uint32_t(need_m) ? n_tab : uint32_t(need_m);
#pragma unroll
    for (int u = 0; u < NT2; ++u) {
        if (uint32_t(u * 64) < n_def) {   // wave-uniform
            cand_d2[size_t(q) * MP + u * 64 + lane] = __longlong_as_double((long long)hi[u]);
            cand_j[size_t(q) * MP + u * 64 + lane] = uint32_t(lo[u]);
            if constexpr (WT) cand_d2t[size_t(q) * MP + u * 64 + lane] = __longlong_as_double((long long)hx[u]);
        }
    }
    // d2 of the need_m-th neighbour (position need_m - 1)
    uint64_t sel = 0;
#pragma unroll
    for (int u = 0; u < NT2; ++u)
        if ((pos >> 6) == u) sel = hi[u];
    const double d2_need = __longlong_as_double((long long)__shfl((unsigned long long)sel, pos & 63));
    const uint64_t second = __shfl((unsigned long long)hi[0], 1);
    if (lane == 0) {
        cand_n[q] = n_tab;
        d2_lb[q] = lb;
        if (!(d2_need < lb)) {
            const uint32_t slot = atomicAdd(fb_count, 1u);
            fb_rows[slot] = int32_t(q);
        }
        if constexpr (WT) {   // (a row handed to the repair pass gets a new table, without the transposed keys: it is listed)
            keyt_ok[q] = (d2_need < lb) ? 1 : 0;
            if (!(d2_need < lb)) nokeyt_rows[atomicAdd(nokeyt_count, 1u)] = int32_t(q0 + q);
        }
        if (unproven && !(d2_need * rkf < lb)) atomicAdd(unproven, 1u);
        if (n > 1 && second == 0ull) atomicOr(gflags, GT_FLAG_DUPLICATES);
    }
}

constexpr uint32_t kNoRow = 0xFFFFFFFFu;

// The caller will need the rows within radius_key_factor x key(need_m-th neighbour) (a build's affinity radius; 1: a plain
// k-nearest table): what lies beyond that - a third of the candidates, the margin of the float16 filter - need not enter the
// table.  Returns the number of leading entries kept (>= need_m; the table is sorted) and lowers `lb`, the table's own
// completeness bound, to the first key dropped.  No cut for a non-positive factor (the extent is not tied to that neighbour).
__device__ __forceinline__ uint32_t sym_table_cut(const uint64_t (&hi)[4], const uint32_t n_tab, const int need_m, const int lane,
                                                  const double radius_key_factor, double& lb) {
    if (!(radius_key_factor > 0.0) || n_tab <= uint32_t(need_m)) return n_tab;
    const int pos = need_m - 1;
    uint64_t sel = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if ((pos >> 6) == u) sel = hi[u];
    const double d2_need = __longlong_as_double((long long)__shfl((unsigned long long)sel, pos & 63));
    const double cut = d2_need * radius_key_factor * (1.0 + 1e-5);
    uint32_t cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool in = uint32_t(u * 64 + lane) < n_tab && __longlong_as_double((long long)hi[u]) <= cut;
        cnt += uint32_t(__popcll(__ballot(in)));
    }
    if (cnt < uint32_t(need_m)) cnt = uint32_t(need_m);
    if (cnt >= n_tab) return n_tab;
    uint64_t fd = 0;   // the first key left out (sorted position cnt)
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if ((cnt >> 6) == uint32_t(u)) fd = hi[u];
    const double first_drop = __longlong_as_double((long long)__shfl((unsigned long long)fd, int(cnt & 63u)));
    lb = fmin(lb, first_drop);
    return cnt;
}

// ---- re-rank of the symmetric candidate pass (knn_select_kernel MODE 2) ---------------------------------------
// List ql belongs to the row at cell-sorted position ql (row perm[ql]); it holds (score, sorted position) keys collected
// against the FIXED threshold thr[ql] by every workgroup that scored a pair with that row: every row that is not in it
// scored <= thr[ql].  The 256 best approximate scores are evaluated exactly in two batches of 128 like above (lists of up
// to 128 / 256 keys take a shorter sorting network); a row whose list overflowed (more than tcap <= 512 keys) is handed
// to the repair path (bound = -inf).
template <typename T, bool F4>
__global__ __launch_bounds__(256, 5) void rerank_sym_kernel(
    const T* __restrict__ X, const int d, const double* __restrict__ xn, const int64_t nq,
    const uint64_t* __restrict__ tlists, const int tcap, const uint32_t* __restrict__ tcounts,
    const float* __restrict__ thr, const double* __restrict__ ymax2p, const ErrModel err, const int need_m,
    const int32_t* __restrict__ perm, double* __restrict__ cand_d2, uint32_t* __restrict__ cand_j,
    uint32_t* __restrict__ cand_n, double* __restrict__ d2_lb, uint32_t* __restrict__ fb_count,
    int32_t* __restrict__ fb_rows, uint32_t* __restrict__ gflags, const double radius_key_factor,
    uint32_t* __restrict__ unproven, unsigned long long* __restrict__ stat, const int want_stats,
    const int32_t* __restrict__ invperm, const int32_t* __restrict__ own_rows, const int64_t own_r0,
    const int xcd_chunk, const T* __restrict__ Xs, const double* __restrict__ xns, const int metric, const int64_t pos0) {
    constexpr int MP = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    double* xs = reinterpret_cast<double*>(smem_raw) + size_t(w) * d;
    const int64_t bid = gt_xcd_item(blockIdx.x, gridDim.x, xcd_chunk * (4 / int(blockDim.x >> 6)));
    const int64_t ql = bid * (blockDim.x >> 6) + w;
    if (ql >= nq) return;
    // single rank: list ql = sorted position ql, tables indexed by the row perm[ql].  Row-sharded (invperm given): the
    // ql-th owned row in the sorted order is own_rows[ql] (neighbouring waves then evaluate overlapping candidate rows),
    // its list and table sit at its local index, its threshold at its sorted position
    // (pos0: the launch covers the sorted positions [pos0, pos0 + nq) - a rank's run of a renumbered point set)
    const int64_t qo = invperm ? int64_t(own_rows[ql]) : int64_t(perm[ql + pos0]);   // row of the bound points
    const int64_t q = qo - own_r0;                                                   // row of the tables (own_r0 = 0 on one rank)
    const int64_t qt = invperm ? int64_t(invperm[qo]) : ql + pos0;                   // index of the threshold
    const int64_t ls = invperm ? q : ql + pos0;                                      // index of the list
    // (Xs / xns: the points and norms in sorted order - the candidates of neighbouring lists are neighbouring rows there)
    const T* xrow = Xs ? Xs + qt * int64_t(d) : X + qo * int64_t(d);
    for (int k = lane; k < d; k += 64) xs[k] = double(xrow[k]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double qnq = xns ? xns[qt] : xn[qo];
    // (row-sharded builds: bit 31 = some rank's partial list of this row overflowed, gt_sym.hip shard_scatter_kernel)
    const uint32_t ct_raw = tcounts[ls];
    const uint32_t ct = ct_raw & 0x7FFFFFFFu;
    const bool overflow = ct > uint32_t(tcap) || (ct_raw >> 31) != 0u;
    const uint32_t n = ct < uint32_t(tcap) ? ct : uint32_t(tcap);   // (a marked row may hold fewer than tcap keys)
    const uint64_t* tp = tlists + size_t(ls) * size_t(tcap);

    const double y2 = ymax2p[0];
    const double e = gt_err_bound(err, qnq, y2);
    // euclidean: d2 = |x|^2 - 2 s; cosine (rows normalised): D = 1 - x.y = 1 - s - |y|^2 / 2 >= 1 - s - ymax^2 / 2 - what the
    // candidate stages guarantee is "every row scoring above thr is listed", whatever the score is turned into
    auto bound_of_score = [&](float score) {
        const double sv = double(score) * err.inv_sc2;
        const double b = (metric == 1) ? (1.0 - (sv + e) - 0.5 * ymax2p[1]) : (qnq - 2.0 * (sv + e));
        return b - 1e-9 * (qnq + y2);
    };
    double lb = overflow ? -INFINITY : bound_of_score(thr[qt]);

    // up to 256 candidates: all of them are evaluated, their order does not matter.  (Sorting the approximate keys first so
    // that the best 128 could settle the row was measured at N = 1e6: 29 % of the rows hold more than 128 candidates and only
    // a third of those stopped after the first batch - the network over 256 keys cost more than the evaluations it saved.)
    const bool both = n > 128u;   // wave-uniform
    uint32_t pc[4];               // sorted positions of the candidates in slots u * 64 + lane
    uint64_t k257 = 0ull;         // best key beyond the table
    if (n <= 256u) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t c = uint32_t(u * 64 + lane);
            pc[u] = (c < n) ? cand_index(tp[c]) : kNoRow;
        }
    } else {
        uint64_t ks[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t c = uint32_t(u * 64 + lane);
            ks[u] = (c < n) ? tp[c] : 0ull;   // a valid key is never 0
        }
        wave_bitonic_desc<8>(ks, lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) pc[u] = ks[u] != 0ull ? cand_index(ks[u]) : kNoRow;
        k257 = __shfl((unsigned long long)ks[4], 0);
    }
    const uint32_t n_eval = n < uint32_t(MP) ? n : uint32_t(MP);
    uint64_t hi[4];
    uint32_t lo[4];
    // (one database row per lane.  Sixteen lanes per row with coalesced 16-byte loads and a rotation sum were tried: 8.1 ms
    //  against 5.8; four lanes per row is rerank_sym4_kernel below)
    // two rows at a time, their loads in flight together (no branch between them)
    auto eval2 = [&](const uint32_t pa, const uint32_t pb, uint64_t& ha, uint32_t& la, uint64_t& hb, uint32_t& lb_) {
        const bool va = pa != kNoRow, vb = pb != kNoRow;
        const uint32_t ja = va ? uint32_t(perm[pa]) : 0u, jb = vb ? uint32_t(perm[pb]) : 0u;   // the rows themselves: what the
        const uint32_t ra = Xs ? (va ? pa : 0u) : ja, rb = Xs ? (vb ? pb : 0u) : jb;             // tables hold, what breaks ties
        const T* base = Xs ? Xs : X;
        const double* nrm = Xs ? xns : xn;
        const double da = va ? dot_row_sel<T, F4>(xs, base + int64_t(ra) * d, d) : 0.0;
        const double db = vb ? dot_row_sel<T, F4>(xs, base + int64_t(rb) * d, d) : 0.0;
        ha = hb = kInfBits;
        la = lb_ = 0xFFFFFFFFu;
        if (va) {
            ha = (uint64_t)__double_as_longlong(gt_pair_key(qnq, da, nrm[ra], metric));
            la = ja;
        }
        if (vb) {
            hb = (uint64_t)__double_as_longlong(gt_pair_key(qnq, db, nrm[rb], metric));
            lb_ = jb;
        }
    };
    eval2(pc[0], pc[1], hi[0], lo[0], hi[1], lo[1]);
    hi[2] = hi[3] = kInfBits;
    lo[2] = lo[3] = 0xFFFFFFFFu;
    const int pos = need_m - 1;
    const double rkf = fabs(radius_key_factor);
    const uint32_t n_tab = n_eval;
    if (!both) {
        uint64_t h2[2] = {hi[0], hi[1]};
        uint32_t l2[2] = {lo[0], lo[1]};
        wave_sort_asc_pair_fast<2>(h2, l2, lane);
        hi[0] = h2[0]; hi[1] = h2[1];
        lo[0] = l2[0]; lo[1] = l2[1];
    } else {
        eval2(pc[2], pc[3], hi[2], lo[2], hi[3], lo[3]);
        if (k257 != 0ull) lb = fmin(lb, bound_of_score(cand_score(k257)));   // candidates beyond the table
        wave_sort_asc_pair_fast<4>(hi, lo, lane);
    }
    const double lb_cand = lb;   // (what the candidate stages proved; the table may be cut shorter below)
    const uint32_t n_all = n_tab;
    const uint32_t n_cut = sym_table_cut(hi, n_tab, need_m, lane, radius_key_factor, lb);
    const uint32_t n_def = n_cut > uint32_t(need_m) ? n_cut : uint32_t(need_m);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (uint32_t(u * 64) < n_def) {   // wave-uniform
            cand_d2[size_t(q) * MP + u * 64 + lane] = __longlong_as_double((long long)hi[u]);
            cand_j[size_t(q) * MP + u * 64 + lane] = uint32_t(lo[u]);
        }
    }
    uint64_t sel = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if ((pos >> 6) == u) sel = hi[u];
    const double d2_need = __longlong_as_double((long long)__shfl((unsigned long long)sel, pos & 63));
    const uint64_t second = __shfl((unsigned long long)hi[0], 1);
    if (lane == 0) {
        cand_n[q] = n_cut;
        d2_lb[q] = lb;
        if (!(d2_need < lb_cand)) {
            const uint32_t slot = atomicAdd(fb_count, 1u);
            fb_rows[slot] = int32_t(q);
        }
        // (an overflowing row is short of room, an orphan - threshold +inf - of seeds, neither of precision: they do not
        //  count against the arithmetic)
        if (unproven && !overflow && k257 == 0ull && thr[qt] != INFINITY && !(d2_need * rkf < lb_cand)) atomicAdd(unproven, 1u);
        // (round-5 advisor: a row with 257 .. tcap candidates whose table cannot prove its radius is neither "unproven" - that
        //  verdict judges the arithmetic - nor an overflow, yet it costs a repair like one: counted on its own, and the ladder
        //  holds overflows + these against its 10 % give-up threshold)
        if (stat && !overflow && k257 != 0ull && thr[qt] != INFINITY && !(d2_need * rkf < lb_cand)) atomicAdd(stat + 6, 1ull);
        if (stat && overflow) atomicAdd(stat + 0, 1ull);
        if (stat && want_stats) {
            // [1] sum of the list lengths [3] longest list [4] rows with more than 256 keys [7] rows with more than 128 keys
            const unsigned long long tot = (unsigned long long)ct;
            atomicAdd(stat + 1, tot);
            atomicMax(stat + 3, tot);
            if (tot > 256ull) atomicAdd(stat + 4, 1ull);
            if (tot > 128ull) atomicAdd(stat + 7, 1ull);
        }
        if (n_all > 1 && second == 0ull) atomicOr(gflags, GT_FLAG_DUPLICATES);
    }
}

// ---- the same re-rank with FOUR LANES PER CANDIDATE ROW (float32 rows, d a multiple of 4, 16-byte aligned) ------------
// rerank_sym_kernel gives every lane a database row of its own: a wave's load instruction touches 64 rows, 16 bytes of a
// 64-byte sector each, and comes back to the sector three more times - 2048 sector requests per row of the graph at the
// L1, which is what the kernel was bound by (5.3 ms at N = 1e6).  Here lane (r, c) = (lane / 4, lane % 4) reads quarter c
// of EVERY 64-byte sector of candidate r's row: the four lanes of a group cover a whole sector in one instruction
// (16 sectors per instruction, 512 requests per row of the graph), and the lane's share of the query - the same elements of
// every candidate - stays in registers (no LDS at all).
// The canonical summation order (gt_device.h gt_dot16: element k goes to partial sum (k >> 2) & 15, tree (l, l+8), (l, l+4),
// (0, 2) + (1, 3)) falls into place: quad g = k >> 2 belongs to lane c = g & 3, the lane's four accumulators are the
// partial sums c, c + 4, c + 8, c + 12, the first two tree levels are in-lane, the last two are one cross-lane add each
// (additions commute: every lane of the group ends with the same bits).
// DB = blocks of 64 features (d <= 64 DB).  Measured: 5.5 -> 5.2 ms at d = 64, 1.07 -> 0.90 ms at d = 36 (N = 3e5); with two
// blocks (d = 100) the registers of the wider query share cost more than the coalescing returns (4.3 -> 6.1 ms): the
// launcher keeps the lane-per-row kernel there.
#ifndef GT_RERANK_EXP
#define GT_RERANK_EXP 0   // development: 1 = tables left unsorted, 2 = one evaluation pass per batch (timing only, wrong results), 3 = tables sorted twice (same results: the sort's share of the time)
#endif
#ifndef GT_RERANK_UNROLL
#define GT_RERANK_UNROLL 2   // evaluation passes whose loads are issued together
#endif
#ifndef GT_RERANK_RPW
#define GT_RERANK_RPW 1   // rows per wave of rerank_sym4_kernel, fixed at compile time (0: eight)
#endif
#ifndef GT_RERANK_WAVES
#define GT_RERANK_WAVES 4   // waves per SIMD the registers are cut for
#endif
#ifndef GT_RERANK_Q32
#define GT_RERANK_Q32 true   // 32-bit composite keys in the table sort (gt_device.h wave_sort_asc_pair_fast)
#endif
template <int DB, bool WT, int WPB>   // WT: the transposed keys travel with the table (cand_d2t, keyt_ok); WPB: waves (rows) per workgroup
__global__ __launch_bounds__(64 * WPB, GT_RERANK_WAVES) void rerank_sym4_kernel(
    const float* __restrict__ X, const int d, const double* __restrict__ xn, const int64_t nq,
    const uint64_t* __restrict__ tlists, const int tcap, const uint32_t* __restrict__ tcounts,
    const float* __restrict__ thr, const double* __restrict__ ymax2p, const ErrModel err, const int need_m,
    const int32_t* __restrict__ perm, double* __restrict__ cand_d2, uint32_t* __restrict__ cand_j,
    uint32_t* __restrict__ cand_n, double* __restrict__ d2_lb, uint32_t* __restrict__ fb_count,
    int32_t* __restrict__ fb_rows, uint32_t* __restrict__ gflags, const double radius_key_factor,
    uint32_t* __restrict__ unproven, unsigned long long* __restrict__ stat, const int want_stats,
    const int32_t* __restrict__ invperm, const int32_t* __restrict__ own_rows, const int64_t own_r0,
    const int xcd_chunk, const float* __restrict__ Xs, const double* __restrict__ xns, double* __restrict__ cand_d2t,
    uint8_t* __restrict__ keyt_ok, int32_t* __restrict__ nokeyt_rows, uint32_t* __restrict__ nokeyt_count, const int metric,
    const int64_t pos0, const int rows_per_wave, const int tab_sorted) {
    // cand_d2t (optional): next to every key of the table, the key the OTHER row holds for the same pair - the same dot
    // product in scikit-learn's association with the roles swapped, (|y|^2 - 2 x.y) + |x|^2 - so that the affinity pass can
    // tell, bit for bit, what the transposed entry is worth (gt_sparse.hip, pair-resolved symmetrisation); keyt_ok[q] = 1
    // when the row's table came from here with them
    constexpr int MP = 256;
    constexpr int NI = 4 * DB;   // 64-byte sectors of a row this kernel can hold
    __shared__ uint64_t park_hi_all[WPB * MP];   // (the sorted tables are picked up by position through the LDS)
    __shared__ uint32_t park_lo_all[WPB * MP];
    __shared__ uint64_t park_x_all[WT ? WPB * MP : 1];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int c = lane & 3, r = lane >> 2;
    uint64_t* park_hi = park_hi_all + w * MP;
    uint32_t* park_lo = park_lo_all + w * MP;
    uint64_t* park_x = park_x_all + w * MP;
    constexpr bool want_t = WT;
    // A wave can take several CONSECUTIVE sorted positions, one after the other, with the next row's counter and first 128 keys
    // loaded (unconditionally: slots beyond the count are masked later) at the top of the current row - the chain counter ->
    // keys (two trips to the HBM: the lists span 4 GB) -> gathers leaves the critical path of all but a wave's first row.
    // Measured in round 5 (C3): 3.67 -> 3.48 ms from one to four rows per wave in a build that could afford the loop's
    // registers, but the loop-carried state costs ~20 VGPRs (168 unconstrained, 108 B of scratch at four waves per SIMD) and four
    // waves per SIMD are worth more (3.16 ms with one row per wave against 3.72 with three waves): GT_RERANK_RPW = 1 compiles
    // the loop and the prefetch away.
    const int64_t bid = gt_xcd_item(blockIdx.x, gridDim.x, xcd_chunk * (4 / WPB));
    const int rpw_ = GT_RERANK_RPW > 0 ? GT_RERANK_RPW : rows_per_wave;
    const int64_t row0 = (bid * WPB + w) * int64_t(rpw_);
    if (row0 >= nq) return;
    uint32_t pf_ct = 0u;
    uint64_t pf_k0 = 0ull, pf_k1 = 0ull;
    auto prefetch = [&](const int64_t qn_) {
        if (!invperm && qn_ < nq) {   // (wave-uniform)
            const int64_t lsn = qn_ + pos0;
            pf_ct = tcounts[lsn];
            const uint64_t* tpn = tlists + size_t(lsn) * size_t(tcap);
            pf_k0 = lane < tcap ? tpn[lane] : 0ull;
            pf_k1 = lane + 64 < tcap ? tpn[64 + lane] : 0ull;
        }
    };
    constexpr bool kPrefetch = GT_RERANK_RPW != 1;   // (one row per wave: the list is read where it is needed, as in rounds 3-4)
    if (kPrefetch) prefetch(row0);
  for (int rr_ = 0; rr_ < rpw_; ++rr_) {
    const int64_t ql = row0 + rr_;
    if (ql >= nq) break;
    // (pos0: the launch covers the sorted positions [pos0, pos0 + nq) - a rank's run of a renumbered point set)
    const int64_t qo = invperm ? int64_t(own_rows[ql]) : int64_t(perm[ql + pos0]);   // row of the bound points
    const int64_t q = qo - own_r0;                                                   // row of the tables (own_r0 = 0 on one rank)
    const int64_t qt = invperm ? int64_t(invperm[qo]) : ql + pos0;                   // index of the threshold
    const int64_t ls = invperm ? q : ql + pos0;                                      // index of the list
    const int64_t tq = tab_sorted ? ql + pos0 : q;                                   // slot of the table (KnnWork::tab_sorted: the sorted position)
    // this lane's share of the query row: elements 16 i + 4 c .. + 3 of every sector i (float64: 32 VGPRs.  Kept in the LDS
    // instead and read back in every pass - tried in round 5 to make room for a fifth wave per SIMD - the kernel took 3.5 ms
    // against 3.2: the reads and the extra spills cost more than they freed)
    double xq[NI][4];
    {
        // (Xs / xns: the points and norms in sorted order, REQUIRED here - candidate rows are read by position, the row
        //  numbers for the tables arrive on the side)
        const float4* xrow4 = reinterpret_cast<const float4*>(Xs + qt * int64_t(d));
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool in = 16 * i + 4 * c < d;
            const float4 v = in ? xrow4[4 * i + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            xq[i][0] = double(v.x);
            xq[i][1] = double(v.y);
            xq[i][2] = double(v.z);
            xq[i][3] = double(v.w);
        }
    }
    const double qnq = xns[qt];
    const bool direct = invperm != nullptr || !kPrefetch;
    const uint32_t ct_raw = direct ? tcounts[ls] : pf_ct;
    const uint32_t ct = ct_raw & 0x7FFFFFFFu;
    const bool overflow = ct > uint32_t(tcap) || (ct_raw >> 31) != 0u;
    const uint32_t n = ct < uint32_t(tcap) ? ct : uint32_t(tcap);
    const uint64_t* tp = tlists + size_t(ls) * size_t(tcap);
    uint64_t key0 = 0ull, key1 = 0ull;
    if constexpr (kPrefetch) {
        key0 = direct ? (uint32_t(lane) < n ? tp[lane] : 0ull) : pf_k0;
        key1 = direct ? (uint32_t(lane) + 64u < n ? tp[64 + lane] : 0ull) : pf_k1;
        if (rr_ + 1 < rpw_) prefetch(ql + 1);
    }   // (the next row's list travels while this one is evaluated)
    const double y2 = ymax2p[0];
    const double e = gt_err_bound(err, qnq, y2);
    // euclidean: d2 = |x|^2 - 2 s; cosine (rows normalised): D = 1 - x.y = 1 - s - |y|^2 / 2 >= 1 - s - ymax^2 / 2 - what the
    // candidate stages guarantee is "every row scoring above thr is listed", whatever the score is turned into
    auto bound_of_score = [&](float score) {
        const double sv = double(score) * err.inv_sc2;
        const double b = (metric == 1) ? (1.0 - (sv + e) - 0.5 * ymax2p[1]) : (qnq - 2.0 * (sv + e));
        return b - 1e-9 * (qnq + y2);
    };
    double lb = overflow ? -INFINITY : bound_of_score(thr[qt]);

    // up to 256 candidates: all of them are evaluated, their order does not matter.  (Sorting the approximate keys first so
    // that the best 128 could settle the row was measured at N = 1e6: 29 % of the rows hold more than 128 candidates and only
    // a third of those stopped after the first batch - the network over 256 keys cost more than the evaluations it saved.)
    const bool both = n > 128u;   // wave-uniform
    uint32_t pc[4];               // sorted positions of the candidates in slots u * 64 + lane
    uint64_t k257 = 0ull;         // best key beyond the table
    if (n <= 256u) {
        if constexpr (kPrefetch) {
            pc[0] = uint32_t(lane) < n ? cand_index(key0) : kNoRow;
            pc[1] = uint32_t(lane) + 64u < n ? cand_index(key1) : kNoRow;
#pragma unroll
            for (int u = 2; u < 4; ++u) {
                const uint32_t cidx = uint32_t(u * 64 + lane);
                pc[u] = (cidx < n) ? cand_index(tp[cidx]) : kNoRow;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t cidx = uint32_t(u * 64 + lane);
                pc[u] = (cidx < n) ? cand_index(tp[cidx]) : kNoRow;
            }
        }
    } else {
        uint64_t ks[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t cidx = uint32_t(u * 64 + lane);
            ks[u] = (cidx < n) ? tp[cidx] : 0ull;   // a valid key is never 0
        }
        wave_bitonic_desc<8>(ks, lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) pc[u] = ks[u] != 0ull ? cand_index(ks[u]) : kNoRow;
        k257 = __shfl((unsigned long long)ks[4], 0);
    }
    const uint32_t n_eval = n < uint32_t(MP) ? n : uint32_t(MP);
    // exact keys of the 128 candidates held as (jA: slot u, jB: slot u + 1), 16 candidates per pass: group r takes
    // candidate 16 p + r of the batch, lane c of the group keeps the result of the passes p with p % 4 == c
    // npass: passes of 16 candidates that hold any (wave-uniform).  The passes are software-pipelined: the gather of pass p + 1
    // (row number, norm, the lane's four quarters of the candidate row) is issued BEFORE the arithmetic of pass p, into a second
    // set of registers - a pass's loads are always in flight behind the previous pass's multiply-adds.  (Round 4 issued two
    // passes' loads together and then waited for both: with ~5 passes per row the wave sat through three full L2 round trips.)
    struct CandLoad {
        float4 v[NI];
        double yn;
        uint32_t j;
    };
    auto issue = [&](const int p, const uint32_t jA, const uint32_t jB, CandLoad& L) {
        const uint32_t pj = uint32_t(__shfl(int(p < 4 ? jA : jB), (p & 3) * 16 + r));   // position of the group's candidate
        L.j = kNoRow;
        L.yn = 0.0;
        if (pj != kNoRow) {
            L.j = uint32_t(perm[pj]);   // the row itself: what the tables hold and what breaks ties
            const float4* y4 = reinterpret_cast<const float4*>(Xs + int64_t(pj) * d);   // (independent of j)
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (16 * i < d) L.v[i] = (16 * i + 4 * c < d) ? y4[4 * i + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            L.yn = xns[pj];
        }
    };
    auto finish = [&](const int p, const CandLoad& L, uint64_t& hA, uint32_t& lA, uint64_t& hB, uint32_t& lB, uint64_t& xA,
                      uint64_t& xB) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // partial sums c, c + 4, c + 8, c + 12
        if (L.j != kNoRow) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (16 * i < d) {   // (uniform; a quad past the end of the row contributes exact zeros, as in gt_dot16
                                    //  where it is simply absent - x + 0 = x)
                    double& acc = (i & 3) == 0 ? a0 : (i & 3) == 1 ? a1 : (i & 3) == 2 ? a2 : a3;
                    if (16 * i + 4 * c < d) {
                        acc = fma(xq[i][0], double(L.v[i].x), acc);
                        acc = fma(xq[i][1], double(L.v[i].y), acc);
                        acc = fma(xq[i][2], double(L.v[i].z), acc);
                        acc = fma(xq[i][3], double(L.v[i].w), acc);
                    }
                }
        }
        const double cc = (a0 + a2) + (a1 + a3);            // b[c] + b[c + 4]
        const double w2 = cc + lane_xor_f64(cc, 2);         // lanes 0, 2: c0 + c2; lanes 1, 3: c1 + c3
        const double dot = w2 + lane_xor_f64(w2, 1);        // (c0 + c2) + (c1 + c3)
        if (L.j != kNoRow && c == (p & 3)) {
            const uint64_t key = (uint64_t)__double_as_longlong(gt_pair_key(qnq, dot, L.yn, metric));
            const uint64_t keyt = want_t ? (uint64_t)__double_as_longlong(gt_pair_key(L.yn, dot, qnq, metric)) : 0ull;
            if (p < 4) {
                hA = key;
                lA = L.j;
                xA = keyt;
            } else {
                hB = key;
                lB = L.j;
                xB = keyt;
            }
        }
    };
    auto eval128 = [&](const uint32_t jA, const uint32_t jB, uint64_t& hA, uint32_t& lA, uint64_t& hB, uint32_t& lB,
                       uint64_t& xA, uint64_t& xB, const int npass_) {
        hA = hB = kInfBits;
        lA = lB = 0xFFFFFFFFu;
        xA = xB = 0ull;
        const int npass = GT_RERANK_EXP == 2 ? 1 : npass_;
        CandLoad L0, L1;
        issue(0, jA, jB, L0);
        for (int p = 0; p < npass; p += 2) {
            if (p + 1 < npass) issue(p + 1, jA, jB, L1);
            finish(p, L0, hA, lA, hB, lB, xA, xB);
            if (p + 1 < npass) {
                if (p + 2 < npass) issue(p + 2, jA, jB, L0);
                finish(p + 1, L1, hA, lA, hB, lB, xA, xB);
            }
        }
    };
    uint64_t hi[4];
    uint32_t lo[4];
    uint64_t hx[4] = {0ull, 0ull, 0ull, 0ull};
    {
        const uint32_t n1 = n_eval < 128u ? n_eval : 128u;
        eval128(pc[0], pc[1], hi[0], lo[0], hi[1], lo[1], hx[0], hx[1], int((n1 + 15u) >> 4));
        hi[2] = hi[3] = kInfBits;
        lo[2] = lo[3] = 0xFFFFFFFFu;
    }
    const int pos = need_m - 1;
    const uint32_t n_tab = n_eval;
    if (n_eval <= 64u) {   // wave-uniform: one key per lane, the network of 64
        uint64_t h1[1] = {hi[0]};
        uint32_t l1[1] = {lo[0]};
        uint64_t x1[1] = {hx[0]};
#if GT_RERANK_EXP != 1
        wave_sort_asc_pair_fast<1, uint32_t, GT_RERANK_Q32>(h1, l1, lane, park_hi, park_lo, want_t ? x1 : nullptr, park_x);
#endif
        hi[0] = h1[0];
        lo[0] = l1[0];
        hx[0] = x1[0];
    } else if (!both) {
        uint64_t h2[2] = {hi[0], hi[1]};
        uint32_t l2[2] = {lo[0], lo[1]};
        uint64_t x2[2] = {hx[0], hx[1]};
#if GT_RERANK_EXP != 1
        wave_sort_asc_pair_fast<2, uint32_t, GT_RERANK_Q32>(h2, l2, lane, park_hi, park_lo, want_t ? x2 : nullptr, park_x);
#endif
#if GT_RERANK_EXP == 3
        wave_sort_asc_pair_fast<2, uint32_t, GT_RERANK_Q32>(h2, l2, lane, park_hi, park_lo, want_t ? x2 : nullptr, park_x);
#endif
        hi[0] = h2[0]; hi[1] = h2[1];
        lo[0] = l2[0]; lo[1] = l2[1];
        hx[0] = x2[0]; hx[1] = x2[1];
    } else {
        eval128(pc[2], pc[3], hi[2], lo[2], hi[3], lo[3], hx[2], hx[3], int((n_eval - 128u + 15u) >> 4));
        if (k257 != 0ull) lb = fmin(lb, bound_of_score(cand_score(k257)));   // candidates beyond the table
#if GT_RERANK_EXP != 1
        wave_sort_asc_pair_fast<4, uint32_t, GT_RERANK_Q32>(hi, lo, lane, park_hi, park_lo, want_t ? hx : nullptr, park_x);
#endif
#if GT_RERANK_EXP == 3
        wave_sort_asc_pair_fast<4, uint32_t, GT_RERANK_Q32>(hi, lo, lane, park_hi, park_lo, want_t ? hx : nullptr, park_x);
#endif
    }
    const double lb_cand = lb;   // (what the candidate stages proved; the table may be cut shorter below)
    const uint32_t n_all = n_tab;
    const uint32_t n_cut = sym_table_cut(hi, n_tab, need_m, lane, radius_key_factor, lb);
    const uint32_t n_def = n_cut > uint32_t(need_m) ? n_cut : uint32_t(need_m);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        // (only the defined slots travel: a row's 80 entries of the 128 its two chunks hold - a gigabyte of the launch's stores)
        if (uint32_t(u * 64 + lane) < n_def) {
            cand_d2[size_t(tq) * MP + u * 64 + lane] = __longlong_as_double((long long)hi[u]);
            cand_j[size_t(tq) * MP + u * 64 + lane] = uint32_t(lo[u]);
            if (want_t) cand_d2t[size_t(tq) * MP + u * 64 + lane] = __longlong_as_double((long long)hx[u]);
        }
    }
    uint64_t sel = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if ((pos >> 6) == u) sel = hi[u];
    const double d2_need = __longlong_as_double((long long)__shfl((unsigned long long)sel, pos & 63));
    const uint64_t second = __shfl((unsigned long long)hi[0], 1);
    if (lane == 0) {
        cand_n[tq] = n_cut;
        d2_lb[tq] = lb;
        if (keyt_ok) {   // (a row handed to the repair pass gets a new table, without them: it is listed for the affinity pass)
            keyt_ok[tq] = (d2_need < lb_cand) ? 1 : 0;
            // (listed as rows of the query matrix - own_r0 + q, like the rows of the radius pass: the affinity launch over the list
            //  subtracts the offset of its rows)
            if (!(d2_need < lb_cand)) nokeyt_rows[atomicAdd(nokeyt_count, 1u)] = int32_t(qo);
        }
        if (!(d2_need < lb_cand)) {
            const uint32_t slot = atomicAdd(fb_count, 1u);
            fb_rows[slot] = int32_t(q);
        }
        // ("unproven" judges the ARITHMETIC of the candidate pass: a row with more candidates than the table holds - k257 - is
        //  short of room, like an overflowing one; with the margin of the local frame such rows are no longer all overflows)
        if (unproven && !overflow && k257 == 0ull && thr[qt] != INFINITY && !(d2_need * fabs(radius_key_factor) < lb_cand)) atomicAdd(unproven, 1u);
        // (round-5 advisor: a row with 257 .. tcap candidates whose table cannot prove its radius is neither "unproven" - that
        //  verdict judges the arithmetic - nor an overflow, yet it costs a repair like one: counted on its own, and the ladder
        //  holds overflows + these against its 10 % give-up threshold)
        if (stat && !overflow && k257 != 0ull && thr[qt] != INFINITY && !(d2_need * fabs(radius_key_factor) < lb_cand)) atomicAdd(stat + 6, 1ull);
        if (stat && overflow) atomicAdd(stat + 0, 1ull);
        if (stat && want_stats) {
            const unsigned long long tot = (unsigned long long)ct;
            atomicAdd(stat + 1, tot);
            atomicMax(stat + 3, tot);
            if (tot > 256ull) atomicAdd(stat + 4, 1ull);
            if (tot > 128ull) atomicAdd(stat + 7, 1ull);
        }
        if (n_all > 1 && second == 0ull) atomicOr(gflags, GT_FLAG_DUPLICATES);
    }
  }   // rows of the wave
}

// ---- exhaustive exact fallback: one workgroup per flagged query ---------------------------------
template <typename T, int NT2>
__global__ __launch_bounds__(256) void fallback_kernel(const T* __restrict__ X, const int64_t n, const int d,
                                                       const double* __restrict__ xn, const T* __restrict__ Q,
                                                       const double* __restrict__ qn, const int64_t q0,
                                                       const int32_t* __restrict__ fb_rows, const int64_t row_off,
                                                       const int metric, const int need_m, double* __restrict__ scratch,
                                                       double* __restrict__ cand_d2, uint32_t* __restrict__ cand_j,
                                                       uint32_t* __restrict__ cand_n, double* __restrict__ d2_lb,
                                                       const int32_t* __restrict__ trow) {
    constexpr int MP = NT2 * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* xs = reinterpret_cast<double*>(smem_raw);                         // [d]
    unsigned long long* out_hi = reinterpret_cast<unsigned long long*>(xs + d);   // [MP]
    uint32_t* out_lo = reinterpret_cast<uint32_t*>(out_hi + MP);              // [MP]
    int* red = reinterpret_cast<int*>(out_lo + MP);                           // [4]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int64_t q = fb_rows[row_off + blockIdx.x];
    double* sc = scratch + size_t(blockIdx.x) * size_t(n);
    const T* xrow = Q + (q0 + q) * int64_t(d);
    for (int k = tid; k < d; k += 256) xs[k] = double(xrow[k]);
    __syncthreads();
    const double qnq = qn[q0 + q];
    for (int64_t j = tid; j < n; j += 256) {
        const double dot = dot_row<T>(xs, X + j * d, d);
        sc[j] = gt_pair_key(qnq, dot, xn[j], metric);
    }
    __syncthreads();
    // bitwise search for the need_m-th smallest key (float64 >= 0: bit pattern is order preserving)
    unsigned long long v = 0ull;
    for (int b = 63; b >= 0; --b) {
        const unsigned long long trial = v | ((1ull << b) - 1ull);
        int c = 0;
        for (int64_t j = tid; j < n; j += 256)
            c += ((unsigned long long)__double_as_longlong(sc[j]) <= trial) ? 1 : 0;
        c = wave_sum_i32(c);
        __syncthreads();
        if (lane == 0) red[w] = c;
        __syncthreads();
        const int total = red[0] + red[1] + red[2] + red[3];
        if (total < need_m) v |= (1ull << b);
    }
    __syncthreads();
    // ordered collection by wave 0: all keys < v, then the lowest-index keys == v
    if (w == 0) {
        for (int p = lane; p < MP; p += 64) {
            out_hi[p] = kInfBits;
            out_lo[p] = 0xFFFFFFFFu;
        }
        int c_less_local = 0;
        for (int64_t j = lane; j < n; j += 64)
            c_less_local += ((unsigned long long)__double_as_longlong(sc[j]) < v) ? 1 : 0;
        const int c_less = wave_sum_i32(c_less_local);
        int base_less = 0, base_eq = 0;
        const int eq_quota = need_m - c_less;
        for (int64_t j0 = 0; j0 < n; j0 += 64) {
            const int64_t j = j0 + lane;
            unsigned long long key = ~0ull;
            if (j < n) key = (unsigned long long)__double_as_longlong(sc[j]);
            const bool is_less = (j < n) && key < v;
            const bool is_eq = (j < n) && key == v;
            int tl, te;
            const int pl = wave_prefix_count(is_less, lane, tl);
            const int pe = wave_prefix_count(is_eq, lane, te);
            if (is_less) {
                out_hi[base_less + pl] = key;
                out_lo[base_less + pl] = uint32_t(j);
            }
            if (is_eq && base_eq + pe < eq_quota) {
                out_hi[c_less + base_eq + pe] = key;
                out_lo[c_less + base_eq + pe] = uint32_t(j);
            }
            base_less += tl;
            base_eq += te;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint64_t hi[NT2], lo[NT2];
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            hi[u] = out_hi[u * 64 + lane];
            lo[u] = out_lo[u * 64 + lane];
        }
        wave_bitonic_asc_pair<NT2>(hi, lo, lane);
        const int64_t tq = trow ? int64_t(trow[q]) : int64_t(q);   // (KnnWork::tab_sorted: the slot the re-rank wrote this row's table to)
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            cand_d2[size_t(tq) * MP + u * 64 + lane] = __longlong_as_double((long long)hi[u]);
            cand_j[size_t(tq) * MP + u * 64 + lane] = uint32_t(lo[u]);
        }
        if (lane == 0) {
            cand_n[tq] = uint32_t(need_m);
            d2_lb[tq] = __longlong_as_double((long long)v);   // everything strictly closer is in the table
        }
    }
}

// ---- collected fallback ---------------------------------------------------------------------------------
// A flagged query (its table could not be proven complete - typically exact distance ties) is repaired without
// touching all n rows: every database row with key <= K_M (the exact need_m-th candidate key, an upper bound of the
// true need_m-th key) scores at least smin(K_M) - e, so one radius-mode candidate pass with that threshold collects
// a superset of the answer at MFMA speed.  fallback_thr_kernel derives the thresholds, collected_select_kernel
// (one workgroup per flagged query) evaluates the exact keys of the collected rows and selects the need_m smallest
// (key, index) pairs with two bitwise searches (key, then index among the ties).
__global__ __launch_bounds__(256) void fallback_thr_kernel(const int32_t* __restrict__ fb_rows, const int64_t n_rows,
                                                           const int64_t row_off, const int64_t q0, const int MP,
                                                           const int need_m, const int metric,
                                                           const double* __restrict__ cand_d2,
                                                           const double* __restrict__ qn,
                                                           const double* __restrict__ qn_full,
                                                           const double* __restrict__ ymax2p, const ErrModel err,
                                                           int32_t* __restrict__ qrows, float* __restrict__ thr,
                                                           const int32_t* __restrict__ trow) {
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (f >= n_rows) return;
    const int64_t q = fb_rows[row_off + f];
    const double key = cand_d2[(trow ? int64_t(trow[q]) : q) * MP + (need_m - 1)] * (1.0 + 1e-12);
    const double qnq = qn[q0 + q];   // norm over the scored columns
    const double y2 = ymax2p[0];
    const double e = gt_err_bound(err, qnq, y2);
    // every row with key' <= key scores at least smin (see bound_of_score in rerank_kernel)
    const double smin = (metric == 1) ? (1.0 - key - 0.5 * (qn_full[q0 + q] - qnq) - 0.5 * ymax2p[1]) : 0.5 * (qnq - key);
    const double x = (smin - e - 1e-9 * (qnq + y2)) / err.inv_sc2;
    float t = float(x);
    if (double(t) >= x) t = nextafterf(t, -INFINITY);
    qrows[f] = int32_t(q0 + q);
    thr[f] = t;
}

template <typename T, int NT2>
__global__ __launch_bounds__(256) void collected_select_kernel(
    const T* __restrict__ X, const int d, const double* __restrict__ xn, const T* __restrict__ Q,
    const double* __restrict__ qn, const int64_t q0, const int32_t* __restrict__ fb_rows, const int64_t row_off,
    const int metric, const int need_m, const uint64_t* __restrict__ clists, const uint32_t* __restrict__ ccounts,
    const int cap, double* __restrict__ scratch, double* __restrict__ cand_d2, uint32_t* __restrict__ cand_j,
    uint32_t* __restrict__ cand_n, double* __restrict__ d2_lb, uint32_t* __restrict__ fail, const int32_t* __restrict__ trow) {
    constexpr int MP = NT2 * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* xs = reinterpret_cast<double*>(smem_raw);                             // [d]
    unsigned long long* out_hi = reinterpret_cast<unsigned long long*>(xs + d);   // [MP]
    uint32_t* out_lo = reinterpret_cast<uint32_t*>(out_hi + MP);                  // [MP]
    int* red = reinterpret_cast<int*>(out_lo + MP);                               // [4]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t f = blockIdx.x;
    const int64_t q = fb_rows[row_off + f];
    const int c = int(ccounts[f] < uint32_t(cap) ? ccounts[f] : uint32_t(cap));
    if (c < need_m) {   // cannot happen if the threshold logic holds; reported, never silently accepted
        if (tid == 0) atomicAdd(fail, 1u);
        return;
    }
    const uint64_t* cl = clists + size_t(f) * cap;
    double* sc = scratch + size_t(f) * cap;
    const T* xrow = Q + (q0 + q) * int64_t(d);
    for (int k = tid; k < d; k += 256) xs[k] = double(xrow[k]);
    __syncthreads();
    const double qnq = qn[q0 + q];
    for (int e = tid; e < c; e += 256) {
        const uint32_t j = cand_index(cl[e]);
        const double dot = dot_row<T>(xs, X + int64_t(j) * d, d);
        sc[e] = gt_pair_key(qnq, dot, xn[j], metric);
    }
    __syncthreads();
    auto block_count = [&](int local) {
        local = wave_sum_i32(local);
        __syncthreads();
        if (lane == 0) red[w] = local;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    // need_m-th smallest key
    unsigned long long v = 0ull;
    for (int b = 63; b >= 0; --b) {
        const unsigned long long trial = v | ((1ull << b) - 1ull);
        int cnt = 0;
        for (int e = tid; e < c; e += 256) cnt += ((unsigned long long)__double_as_longlong(sc[e]) <= trial) ? 1 : 0;
        if (block_count(cnt) < need_m) v |= (1ull << b);
    }
    int cl_local = 0;
    for (int e = tid; e < c; e += 256) cl_local += ((unsigned long long)__double_as_longlong(sc[e]) < v) ? 1 : 0;
    const int c_less = block_count(cl_local);
    const int quota = need_m - c_less;
    // quota-th smallest index among the ties at v
    uint32_t jv = 0u;
    for (int b = 31; b >= 0; --b) {
        const uint32_t trial = jv | ((1u << b) - 1u);
        int cnt = 0;
        for (int e = tid; e < c; e += 256)
            cnt += ((unsigned long long)__double_as_longlong(sc[e]) == v && cand_index(cl[e]) <= trial) ? 1 : 0;
        if (block_count(cnt) < quota) jv |= (1u << b);
    }
    // collect the need_m winners (unordered), then sort them in wave 0
    __shared__ int slot_sh;
    if (tid == 0) slot_sh = 0;
    for (int p = tid; p < MP; p += 256) {
        out_hi[p] = kInfBits;
        out_lo[p] = 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int e = tid; e < c; e += 256) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(sc[e]);
        const uint32_t j = cand_index(cl[e]);
        if (key < v || (key == v && j <= jv)) {
            const int s = atomicAdd(&slot_sh, 1);
            if (s < MP) {
                out_hi[s] = key;
                out_lo[s] = j;
            }
        }
    }
    __syncthreads();
    if (w == 0) {
        uint64_t hi[NT2], lo[NT2];
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            hi[u] = out_hi[u * 64 + lane];
            lo[u] = out_lo[u * 64 + lane];
        }
        wave_bitonic_asc_pair<NT2>(hi, lo, lane);
        const int64_t tq = trow ? int64_t(trow[q]) : int64_t(q);   // (KnnWork::tab_sorted: the slot the re-rank wrote this row's table to)
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            cand_d2[size_t(tq) * MP + u * 64 + lane] = __longlong_as_double((long long)hi[u]);
            cand_j[size_t(tq) * MP + u * 64 + lane] = uint32_t(lo[u]);
        }
        if (lane == 0) {
            cand_n[tq] = uint32_t(need_m);
            d2_lb[tq] = __longlong_as_double((long long)v);   // everything strictly closer is in the table
        }
    }
}

__global__ __launch_bounds__(256) void emit_knn_kernel(const double* __restrict__ cand_d2,
                                                       const uint32_t* __restrict__ cand_j, const int MP,
                                                       const int64_t nq, const int k, const int dtype, const int metric,
                                                       int64_t* __restrict__ out_idx, double* __restrict__ out_dist) {
    const int64_t total = nq * k;
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < total; e += int64_t(gridDim.x) * 256) {
        const int64_t q = e / k;
        const int p = int(e % k);
        const double d2 = cand_d2[q * MP + p];
        const double dist = gt_key_to_dist(d2, dtype, metric);
        out_idx[e] = int64_t(cand_j[q * MP + p]);
        out_dist[e] = dist;
    }
}

template <typename T>
int rerank_t(gt_ctx* ctx, const RerankArgs& a) {
    if (a.wrote_t) *a.wrote_t = false;
    const int wpb = 1;   // rows (waves) per workgroup: see gt_launch_rerank_sym
    const int64_t blocks = ceil_div64(a.nq, wpb);
    const size_t lds = size_t(wpb) * a.d * sizeof(double);
    const bool f4 = sizeof(T) == 4 && (a.d & 3) == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
    if (a.MP == 128) {
        if (f4) { hipLaunchKernelGGL((rerank_kernel<T, 2, true>), dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.qn_sel, a.q0, a.nq, a.lists, a.lstride, a.counts, a.thr_final, a.ymax2,
                           a.err, a.metric, a.need_m, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows,
                           a.gflags, a.radius_key_factor, a.unproven, a.qrows); } else { hipLaunchKernelGGL((rerank_kernel<T, 2, false>), dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.qn_sel, a.q0, a.nq, a.lists, a.lstride, a.counts, a.thr_final, a.ymax2,
                           a.err, a.metric, a.need_m, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows,
                           a.gflags, a.radius_key_factor, a.unproven, a.qrows); }
    } else if (a.MP == 256 && a.cand_d2t && a.keyt_ok && a.nokeyt_rows && a.nokeyt_count) {
        // (with the transposed keys: + MP x 20 bytes of LDS per wave for the sorts' parked payload)
        const size_t lds_t = lds + size_t(wpb) * 256 * 20;
        if (f4) { hipLaunchKernelGGL((rerank_kernel<T, 4, true, true>), dim3((unsigned)blocks), dim3(64 * wpb), lds_t, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.qn_sel, a.q0, a.nq, a.lists, a.lstride, a.counts, a.thr_final, a.ymax2,
                           a.err, a.metric, a.need_m, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows,
                           a.gflags, a.radius_key_factor, a.unproven, a.qrows, a.cand_d2t, a.keyt_ok, a.nokeyt_rows, a.nokeyt_count); } else { hipLaunchKernelGGL((rerank_kernel<T, 4, false, true>), dim3((unsigned)blocks), dim3(64 * wpb), lds_t, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.qn_sel, a.q0, a.nq, a.lists, a.lstride, a.counts, a.thr_final, a.ymax2,
                           a.err, a.metric, a.need_m, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows,
                           a.gflags, a.radius_key_factor, a.unproven, a.qrows, a.cand_d2t, a.keyt_ok, a.nokeyt_rows, a.nokeyt_count); }
        if (a.wrote_t) *a.wrote_t = true;
    } else if (a.MP == 256) {
        if (f4) { hipLaunchKernelGGL((rerank_kernel<T, 4, true>), dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.qn_sel, a.q0, a.nq, a.lists, a.lstride, a.counts, a.thr_final, a.ymax2,
                           a.err, a.metric, a.need_m, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows,
                           a.gflags, a.radius_key_factor, a.unproven, a.qrows); } else { hipLaunchKernelGGL((rerank_kernel<T, 4, false>), dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.qn_sel, a.q0, a.nq, a.lists, a.lstride, a.counts, a.thr_final, a.ymax2,
                           a.err, a.metric, a.need_m, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows,
                           a.gflags, a.radius_key_factor, a.unproven, a.qrows); }
    } else if (a.MP == 512) {
        if (f4) { hipLaunchKernelGGL((rerank_kernel<T, 8, true>), dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.qn_sel, a.q0, a.nq, a.lists, a.lstride, a.counts, a.thr_final, a.ymax2,
                           a.err, a.metric, a.need_m, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows,
                           a.gflags, a.radius_key_factor, a.unproven, a.qrows); } else { hipLaunchKernelGGL((rerank_kernel<T, 8, false>), dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.qn_sel, a.q0, a.nq, a.lists, a.lstride, a.counts, a.thr_final, a.ymax2,
                           a.err, a.metric, a.need_m, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows,
                           a.gflags, a.radius_key_factor, a.unproven, a.qrows); }
    } else {
        GT_FAIL(ctx, GT_E_ARG, "rerank: unsupported table width");
    }
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

template <typename T>
int fallback_t(gt_ctx* ctx, const RerankArgs& a, int64_t n_rows, int64_t row_off, double* scratch) {
    if (a.MP == 128) {
        const size_t lds = size_t(a.d) * 8 + size_t(128) * 12 + 16;
        hipLaunchKernelGGL((fallback_kernel<T, 2>), dim3((unsigned)n_rows), dim3(256), lds, ctx->stream, (const T*)a.X,
                           a.n, a.d, a.xn, (const T*)a.Q, a.qn, a.q0, a.fb_rows, row_off, a.metric, a.need_m, scratch, a.cand_d2,
                           a.cand_j, a.cand_n, a.d2_lb, a.trow);
    } else if (a.MP == 256) {
        const size_t lds = size_t(a.d) * 8 + size_t(256) * 12 + 16;
        hipLaunchKernelGGL((fallback_kernel<T, 4>), dim3((unsigned)n_rows), dim3(256), lds, ctx->stream, (const T*)a.X,
                           a.n, a.d, a.xn, (const T*)a.Q, a.qn, a.q0, a.fb_rows, row_off, a.metric, a.need_m, scratch, a.cand_d2,
                           a.cand_j, a.cand_n, a.d2_lb, a.trow);
    } else if (a.MP == 512) {
        const size_t lds = size_t(a.d) * 8 + size_t(512) * 12 + 16;
        hipLaunchKernelGGL((fallback_kernel<T, 8>), dim3((unsigned)n_rows), dim3(256), lds, ctx->stream, (const T*)a.X,
                           a.n, a.d, a.xn, (const T*)a.Q, a.qn, a.q0, a.fb_rows, row_off, a.metric, a.need_m, scratch, a.cand_d2,
                           a.cand_j, a.cand_n, a.d2_lb, a.trow);
    } else {
        GT_FAIL(ctx, GT_E_ARG, "fallback: unsupported table width");
    }
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

}  // namespace

int gt_launch_rerank(gt_ctx* ctx, const RerankArgs& a) {
    if (a.dtype == GT_F32) return rerank_t<float>(ctx, a);
    return rerank_t<double>(ctx, a);
}

int gt_launch_fallback(gt_ctx* ctx, const RerankArgs& a, int64_t n_rows, int64_t row_off, double* scratch) {
    if (a.dtype == GT_F32) return fallback_t<float>(ctx, a, n_rows, row_off, scratch);
    return fallback_t<double>(ctx, a, n_rows, row_off, scratch);
}

int gt_launch_emit_knn(gt_ctx* ctx, const double* cand_d2, const uint32_t* cand_j, int MP, int64_t nq, int k,
                       int dtype, int metric, int64_t* out_idx, double* out_dist) {
    int64_t blocks = ceil_div64(nq * k, 256);
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(emit_knn_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, cand_d2, cand_j, MP, nq, k,
                       dtype, metric, out_idx, out_dist);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_launch_fallback_thr(gt_ctx* ctx, const RerankArgs& a, int64_t n_rows, int64_t row_off, int32_t* qrows, float* thr) {
    hipLaunchKernelGGL(fallback_thr_kernel, dim3((unsigned)ceil_div64(n_rows, 256)), dim3(256), 0, ctx->stream, a.fb_rows,
                       n_rows, row_off, a.q0, a.MP, a.need_m, a.metric, a.cand_d2, a.qn_sel, a.qn, a.ymax2, a.err, qrows, thr, a.trow);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

template <typename T>
static int collected_t(gt_ctx* ctx, const RerankArgs& a, int64_t n_rows, int64_t row_off, const uint64_t* clists,
                       const uint32_t* ccounts, int cap, double* scratch, uint32_t* fail) {
    const size_t lds = size_t(a.d) * 8 + size_t(a.MP) * 12 + 16;
    if (a.MP == 128)
        hipLaunchKernelGGL((collected_select_kernel<T, 2>), dim3((unsigned)n_rows), dim3(256), lds, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.q0, a.fb_rows, row_off, a.metric, a.need_m, clists, ccounts, cap,
                           scratch, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, fail, a.trow);
    else if (a.MP == 256)
        hipLaunchKernelGGL((collected_select_kernel<T, 4>), dim3((unsigned)n_rows), dim3(256), lds, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.q0, a.fb_rows, row_off, a.metric, a.need_m, clists, ccounts, cap,
                           scratch, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, fail, a.trow);
    else
        hipLaunchKernelGGL((collected_select_kernel<T, 8>), dim3((unsigned)n_rows), dim3(256), lds, ctx->stream, (const T*)a.X,
                           a.d, a.xn, (const T*)a.Q, a.qn, a.q0, a.fb_rows, row_off, a.metric, a.need_m, clists, ccounts, cap,
                           scratch, a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, fail, a.trow);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_launch_collected_select(gt_ctx* ctx, const RerankArgs& a, int64_t n_rows, int64_t row_off, const uint64_t* clists,
                               const uint32_t* ccounts, int cap, double* scratch, uint32_t* fail) {
    if (a.dtype == GT_F32) return collected_t<float>(ctx, a, n_rows, row_off, clists, ccounts, cap, scratch, fail);
    return collected_t<double>(ctx, a, n_rows, row_off, clists, ccounts, cap, scratch, fail);
}

int gt_launch_rerank_sym(gt_ctx* ctx, const RerankArgs& a, const SymRerank& sr) {
    const int wpb = 4;   // (the lane-per-row kernel: measured 3.56 ms with four rows per workgroup, 3.79 with one - N = 3e5, d = 100)
    const int64_t blocks = ceil_div64(a.nq, wpb);
    const size_t lds = size_t(wpb) * a.d * sizeof(double);
    if (a.MP != 256) GT_FAIL(ctx, GT_E_ARG, "rerank_sym: table width 256");
#define GT_RERANK_SYM_LAUNCH(T_, F4_)                                                                                     \
    hipLaunchKernelGGL((rerank_sym_kernel<T_, F4_>), dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream, (const T_*)a.X, a.d,  \
                       a.xn, a.nq, sr.tlists, sr.tcap, sr.tcounts, a.thr_final, a.ymax2, a.err, a.need_m, sr.perm,        \
                       a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows, a.gflags, a.radius_key_factor,      \
                       a.unproven, sr.stat, (ctx->dbg_select & 256) ? 1 : 0, sr.invperm, sr.own_rows, sr.own_r0, 0,      \
                       (const T_*)sr.Xs, sr.xns, a.metric, sr.pos0)
#define GT_RERANK_SYM4_LAUNCH(DB_, WT_, WPB_)                                                                             \
    hipLaunchKernelGGL((rerank_sym4_kernel<DB_, WT_, WPB_>), dim3((unsigned)ceil_div64(a.nq, int64_t(WPB_) * rpw)), dim3(64 * WPB_), 0, ctx->stream, (const float*)a.X, dx, \
                       a.xn, a.nq, sr.tlists, sr.tcap, sr.tcounts, a.thr_final, a.ymax2, a.err, a.need_m, sr.perm,        \
                       a.cand_d2, a.cand_j, a.cand_n, a.d2_lb, a.fb_count, a.fb_rows, a.gflags, a.radius_key_factor,      \
                       a.unproven, sr.stat, (ctx->dbg_select & 256) ? 1 : 0, sr.invperm, sr.own_rows, sr.own_r0, 0,      \
                       (const float*)sr.Xs, sr.xns, sr.cand_d2t, sr.keyt_ok, sr.nokeyt_rows, sr.nokeyt_count, a.metric, sr.pos0, rpw, tabs)
    if (sr.wrote_t) *sr.wrote_t = false;
    // (tables by sorted position: only the kernel that also writes the transposed keys does that - the caller looks at wrote_t)
    const int tabs = (sr.tab_sorted && sr.cand_d2t != nullptr && sr.keyt_ok != nullptr && !sr.invperm) ? 1 : 0;
    // rows per wave of rerank_sym4_kernel (consecutive sorted positions, the next row's list prefetched): enough waves to fill
    // the chip a few times over must remain
    const int rpw = GT_RERANK_RPW > 0 ? GT_RERANK_RPW : 8;
    // (dx: row length = stride of the sorted copy - the points' d, or d zero padded to a multiple of 4)
    const int dx = (sr.Xs && sr.xs_d > 0) ? sr.xs_d : a.d;
    if (dx != a.d && !(a.dtype == GT_F32 && (dx & 3) == 0 && dx <= 64 && ctx->rerank_lanes4 != 0))
        GT_FAIL(ctx, GT_E_STATE, "rerank_sym: a padded sorted copy needs the four-lanes-per-row kernel");
    if (a.dtype == GT_F32) {
        const bool f4 = (a.d & 3) == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
        const bool f4x = sr.Xs != nullptr && (dx & 3) == 0 && (reinterpret_cast<uintptr_t>(sr.Xs) & 15) == 0;
        if (f4x && ctx->rerank_lanes4 != 0 && dx <= 64) {
            const bool wt = sr.cand_d2t != nullptr && sr.keyt_ok != nullptr && sr.nokeyt_rows != nullptr && sr.nokeyt_count != nullptr;
            // one wave per workgroup: rows differ in cost, and a workgroup's slots are only handed on when its last wave is
            // done (four rows per workgroup measured 4.28 against 4.02 ms in round 3; the variant was removed in round 5)
            if (wt) {
                GT_RERANK_SYM4_LAUNCH(1, true, 1);
            } else {
                if (tabs) GT_FAIL(ctx, GT_E_STATE, "rerank_sym: tables by sorted position need the transposed keys");
                GT_RERANK_SYM4_LAUNCH(1, false, 1);
            }
            if (sr.wrote_t) *sr.wrote_t = wt;
        }
        else if (f4) GT_RERANK_SYM_LAUNCH(float, true);
        else GT_RERANK_SYM_LAUNCH(float, false);
    } else {
        GT_RERANK_SYM_LAUNCH(double, false);
    }
#undef GT_RERANK_SYM_LAUNCH
#undef GT_RERANK_SYM4_LAUNCH
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}
