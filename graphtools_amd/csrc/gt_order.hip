// Query ordering for the candidate pass: the rows of a launch grouped by the nearest of L landmark rows.
//
// Measured at N = 1e6 (mix, d = 64): with the 32 queries of a wave drawn from one neighbourhood the candidate kernel
// enters its admission path in 6 % of the units instead of 23 % (a database row that beats one query's threshold
// beats most of them in the same unit), -11 % kernel time.  The database side keeps the caller's row order - the
// entries of the candidate lists stay original row ids - only the order in which query rows are dealt to workgroups
// changes; list i then belongs to row out_rows[i], which the re-rank maps back (RerankArgs::qrows).
// Landmarks: L = n/512 (64 ... 4096) evenly strided rows.  Assignment: one float16 MFMA chain against the landmark
// rows (assign_cells_kernel, gt_knn_select.hip) - approximate by design, any grouping is correct.  Grouping: a
// stable radix sort of (cell, row) pairs (rocPRIM), so the order is deterministic.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "gt_common.h"
#include "gt_device.h"
#include "gt_knn.h"
#include "gt_knn_select.h"

namespace {

__global__ __launch_bounds__(256) void gather_landmarks_kernel(const uint32_t* __restrict__ Yc, const float* __restrict__ hneg,
                                                               const int64_t step, const int L, const int rw,
                                                               uint32_t* __restrict__ Yl, float* __restrict__ hl) {
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (f >= int64_t(L) * rw) return;
    const int64_t l = f / rw;
    const int c = int(f % rw);
    Yl[f] = Yc[l * step * rw + c];
    if (c == 0) hl[l] = hneg[l * step];
}

__global__ __launch_bounds__(256) void iota_rows_kernel(const int64_t q0, const int64_t nq, int32_t* __restrict__ rows) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < nq) rows[i] = int32_t(q0 + i);
}

// row l of out = row l * step of X (rows of row_bytes bytes, a multiple of 4)
__global__ __launch_bounds__(256) void gather_strided_rows_kernel(const uint32_t* __restrict__ X, const int64_t step, const int L,
                                                                  const int rw, uint32_t* __restrict__ out) {
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (f >= int64_t(L) * rw) return;
    out[f] = X[(f / rw) * step * rw + f % rw];
}

// rows lidx[l] * step of the compact copy (and their seeds) as landmark l
__global__ __launch_bounds__(256) void gather_landmarks_idx_kernel(const uint32_t* __restrict__ Yc, const float* __restrict__ hneg,
                                                                   const int32_t* __restrict__ lidx, const int64_t step, const int L,
                                                                   const int rw, uint32_t* __restrict__ Yl, float* __restrict__ hl) {
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (f >= int64_t(L) * rw) return;
    const int64_t l = f / rw;
    const int c = int(f % rw);
    const int64_t row = int64_t(lidx[l]) * step;
    Yl[f] = Yc[row * rw + c];
    if (c == 0) hl[l] = hneg[row];
}
__global__ __launch_bounds__(256) void gather_strided_rows_idx_kernel(const uint32_t* __restrict__ X, const int32_t* __restrict__ lidx,
                                                                      const int64_t step, const int L, const int rw,
                                                                      uint32_t* __restrict__ out) {
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (f >= int64_t(L) * rw) return;
    out[f] = X[int64_t(lidx[f / rw]) * step * rw + f % rw];
}

// ---- the outlier cell ---------------------------------------------------------------------------------------------------
// A cell is what the bound pass reasons about: centre, radius (its farthest member), the largest collection radius of its
// rows.  ONE row far from every landmark - an isolated point, a point half way between two clusters - inflates the ball of the
// cell it lands in, the fat ball stays undecided against thousands of cells, and twenty such rows in a million (real data has
// them) were enough to push the bound pass over its queue and the build from 18 to 84 ms.  Rows whose squared distance to their
// nearest landmark exceeds tau x the mean of all rows are therefore given a cell of their own, the last one: its ball is huge,
// every pair with one of its rows is scored - as it has to be - and the other cells keep the balls of their bulk.  tau: the
// smallest of 4, 8, 16, 32, 64 that leaves the cell at most `cap` rows, most of which are still beyond 4 tau (none fits: no
// outlier cell - the tail is the data's nature, not an exception).  o = -(hneg + best) = |x - landmark|^2 / 2 in the units of the scores.
constexpr int kOutlierTaus = 5;

__global__ __launch_bounds__(256) void outlier_sum_kernel(const int64_t nq, const int64_t q0, const float* __restrict__ hneg,
                                                          const float* __restrict__ best, double* __restrict__ acc) {
    double s = 0.0;
    for (int64_t q = int64_t(blockIdx.x) * 256 + threadIdx.x; q < nq; q += int64_t(gridDim.x) * 256) {
        const float o = -(hneg[q0 + q] + best[q]);
        s += o > 0.f ? double(o) : 0.0;
    }
    // (one atomic per workgroup: a thousand waves adding to one address took 0.1 ms)
    __shared__ double red[4];
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void outlier_count_kernel(const int64_t nq, const int64_t q0, const float* __restrict__ hneg,
                                                            const float* __restrict__ best, const double* __restrict__ acc,
                                                            uint32_t* __restrict__ counts) {
    const float mean = float(acc[0] / double(nq));
    uint32_t c[kOutlierTaus] = {0u, 0u, 0u, 0u, 0u};
    for (int64_t q = int64_t(blockIdx.x) * 256 + threadIdx.x; q < nq; q += int64_t(gridDim.x) * 256) {
        const float o = -(hneg[q0 + q] + best[q]);
#pragma unroll
        for (int t = 0; t < kOutlierTaus; ++t) c[t] += o > float(4 << t) * mean ? 1u : 0u;
    }
    __shared__ int redc[4][kOutlierTaus];
#pragma unroll
    for (int t = 0; t < kOutlierTaus; ++t) {
        const int tot = wave_sum_i32(int(c[t]));
        if ((threadIdx.x & 63) == 0) redc[threadIdx.x >> 6][t] = tot;
    }
    __syncthreads();
    if (threadIdx.x < kOutlierTaus) {
        const int tot = redc[0][threadIdx.x] + redc[1][threadIdx.x] + redc[2][threadIdx.x] + redc[3][threadIdx.x];
        if (tot) atomicAdd(&counts[threadIdx.x], uint32_t(tot));
    }
}

__global__ __launch_bounds__(256) void outlier_relabel_kernel(const int64_t nq, const int64_t q0, const float* __restrict__ hneg,
                                                              const float* __restrict__ best, const double* __restrict__ acc,
                                                              const uint32_t* __restrict__ counts, const uint32_t cap,
                                                              const uint32_t outlier_cell, uint32_t* __restrict__ cell) {
    const int64_t q = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (q >= nq) return;
    // the smallest factor whose tail fits AND is flat - most of its rows are still beyond four times the factor: rows that
    // belong to nothing, not the upper end of a distribution that simply has one (a curled manifold: two thousand rows beyond
    // 4 x the mean, a few dozen beyond 16 x - packing those into a cell of their own costs them their neighbourhood, their
    // seeds, and the build 11 ms of repairs); the two largest factors have nothing to be compared with: a handful of rows
    int pick = -1;
#pragma unroll
    for (int t = kOutlierTaus - 1; t >= 0; --t) {
        const bool flat = (t + 2 < kOutlierTaus) ? counts[t] <= 2u * counts[t + 2] + 16u : counts[t] <= 64u;
        if (counts[t] <= cap && flat) pick = t;
    }
    if (pick < 0) return;
    const float mean = float(acc[0] / double(nq));
    if (-(hneg[q0 + q] + best[q]) > float(4 << pick) * mean) cell[q] = outlier_cell;
}

}  // namespace

// ---- a COHERENT order of the cells -----------------------------------------------------------------------------------------
// The landmarks are evenly strided rows: in the caller's row order, i.e. in no order at all - cell c and cell c + 1 of the sorted
// points have nothing to do with each other, and the eight or so cells that share a cluster (or a patch of a manifold) lie
// anywhere in the order.  Every pass that walks the sorted rows pays for that: the exact stages gather a row's candidates from
// eight places (each new cell of a walk brings a new half megabyte into the L2 instead of finding its cluster's there), and a rank
// of a row-sharded build - a run of whole cells - owns an eighth of EVERY cluster: six in ten of its one-sided entries pointed
// at other ranks' rows (C3, world 8).  So the landmarks themselves are put in an order in which neighbours in space are
// neighbours in number: every S-th landmark is a "super" landmark (S = L / 8: a super cell is about a cluster's worth), every
// landmark goes to its nearest super (the assignment kernel, L rows against S), and the landmarks are sorted by super, stably.
// lidx_out[l']: the strided landmark that becomes landmark l'.  Cheap (L <= 8192 rows), deterministic (every rank of a sharded
// build computes the same order from the same rows), and any order is correct: cells only group.
// key[l] = ((g3[g2[g1[l]]] * S2 + g2[g1[l]]) * S1 + g1[l]: the landmark's groups from the coarsest level down (missing levels: 0)
__global__ __launch_bounds__(256) void compose_group_keys_kernel(const int L, const uint32_t* __restrict__ g1, const int S1,
                                                                 const uint32_t* __restrict__ g2, const int S2,
                                                                 const uint32_t* __restrict__ g3, uint32_t* __restrict__ key) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    const uint32_t a = g1[l];
    const uint32_t b = g2 ? g2[a] : 0u;
    const uint32_t c = g3 ? g3[b] : 0u;
    key[l] = (c * uint32_t(S2 > 0 ? S2 : 1) + b) * uint32_t(S1) + a;
}

static int coherent_landmark_order(gt_ctx* ctx, const int L, const int rw, const uint32_t* Yl, const float* hl, int32_t** lidx_out) {
    *lidx_out = nullptr;
    ctx->order_coherent_active = 0;
    if (ctx->order_coherent == 0 || L < 256) return GT_OK;
    // Only where launch A of the symmetric pass scores a strided sample of the far tiles (gt_knn.cpp stride_a: from 8 x sym_stride
    // tiles, ~8e5 rows).  Below that the one early sign of a point set whose cells say nothing (half the points in one blob: any
    // cell of the blob serves a row as well as the cells around it) is that a query block's OTHER cells - strangers, in the
    // landmarks' own order - contribute as many seeds as the row's own neighbourhood; numbered coherently the block's other
    // cells are neighbours, the sign is gone, and the pass was abandoned only after its lists had overflowed (tests/
    // test_gpu_ladder.py, N = 2e5: 44 ms against 34).  The sets the order is for - a million rows and more - have the sample.
    if (ctx->sym_stride <= 0 || ctx->n / gt_select_bn(ctx->DP) < int64_t(8) * ctx->sym_stride) return GT_OK;
    // Three levels of groups, every level's groups formed like the cells themselves: every k-th item of the level below is a
    // "super" item, every item goes to its nearest super (the assignment kernel).  S1 = L / 4 groups of landmarks, S2 = L / 16
    // groups of those, S3 = L / 128 groups of those; the landmarks are sorted by (group 3, group 2, group 1), stably.  One level
    // (L / 8 groups) left a cluster that holds two supers in two places and put 76 % of the cells' neighbours on the same eighth
    // of the order (C3's geometry, numpy); three levels 89 % - a cluster's pieces are neighbours one level up.
    auto round32 = [](int v) { return std::max(32, v / 32 * 32); };
    int S[3] = {round32(L / 4), round32(L / 16), round32(L / 128)};
    int nlev = 1;
    if (S[1] < S[0]) nlev = 2;
    if (nlev == 2 && S[2] < S[1]) nlev = 3;
    // (the key (g3 S2 + g2) S1 + g1 is below S3 S2 S1)
    int kb = 1;
    while (kb < 32 && (uint64_t(1) << kb) < uint64_t(S[0]) * uint64_t(nlev > 1 ? S[1] : 1) * uint64_t(nlev > 2 ? S[2] : 1)) ++kb;
    size_t sort_bytes = 0;
    GT_HIP(ctx, rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                          size_t(L), 0u, unsigned(kb), ctx->stream));
    // one buffer: the levels' rows and seeds | their group numbers | keys (in, sorted) | thresholds (unused) | iota | order | scratch
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off = (off + bytes + 255) & ~size_t(255);
        return o;
    };
    size_t o_y[3], o_h[3], o_g[3];
    for (int v = 0; v < nlev; ++v) {
        o_y[v] = take(size_t(S[v]) * rw * 4);
        o_h[v] = take(size_t(S[v]) * 4);
        o_g[v] = take(size_t(v == 0 ? L : S[v - 1]) * 4);
    }
    const size_t o_k0 = take(size_t(L) * 4), o_k1 = take(size_t(L) * 4), o_thr = take(size_t(L) * 4), o_io = take(size_t(L) * 4),
                 o_ix = take(size_t(L) * 4), o_tmp = take(sort_bytes + 256);
    GT_HIP(ctx, ctx->land_ord.reserve(off));
    char* base = static_cast<char*>(ctx->land_ord.p);
    const uint32_t* items_Y = Yl;
    const float* items_h = hl;
    int n_items = L;
    uint32_t* grp[3] = {nullptr, nullptr, nullptr};
    for (int v = 0; v < nlev; ++v) {
        uint32_t* sup_Y = reinterpret_cast<uint32_t*>(base + o_y[v]);
        float* sup_h = reinterpret_cast<float*>(base + o_h[v]);
        grp[v] = reinterpret_cast<uint32_t*>(base + o_g[v]);
        hipLaunchKernelGGL(gather_landmarks_kernel, dim3((unsigned)ceil_div64(int64_t(S[v]) * rw, 256)), dim3(256), 0, ctx->stream,
                           items_Y, items_h, int64_t(n_items / S[v]), S[v], rw, sup_Y, sup_h);
        GT_HIP(ctx, hipGetLastError());
        GT_TRY(gt_launch_assign_cells(ctx, ctx->DP, reinterpret_cast<const float*>(items_Y), reinterpret_cast<const float*>(sup_Y),
                                      sup_h, 0, int32_t(n_items), S[v], 1, grp[v], reinterpret_cast<float*>(base + o_thr), nullptr));
        items_Y = sup_Y;
        items_h = sup_h;
        n_items = S[v];
    }
    uint32_t* k0 = reinterpret_cast<uint32_t*>(base + o_k0);
    uint32_t* k1 = reinterpret_cast<uint32_t*>(base + o_k1);
    int32_t* io = reinterpret_cast<int32_t*>(base + o_io);
    int32_t* ix = reinterpret_cast<int32_t*>(base + o_ix);
    hipLaunchKernelGGL(compose_group_keys_kernel, dim3((unsigned)ceil_div64(L, 256)), dim3(256), 0, ctx->stream, L, grp[0], S[0],
                       nlev > 1 ? grp[1] : (const uint32_t*)nullptr, nlev > 1 ? S[1] : 0, nlev > 2 ? grp[2] : (const uint32_t*)nullptr, k0);
    hipLaunchKernelGGL(iota_rows_kernel, dim3((unsigned)ceil_div64(L, 256)), dim3(256), 0, ctx->stream, int64_t(0), int64_t(L), io);
    GT_HIP(ctx, hipGetLastError());
    GT_HIP(ctx, rocprim::radix_sort_pairs(base + o_tmp, sort_bytes, k0, k1, io, ix, size_t(L), 0u, unsigned(kb), ctx->stream));
    *lidx_out = ix;
    ctx->order_coherent_active = 1;
    return GT_OK;
}

// number of landmark cells of a point set of n rows (0: too few rows for a cell order)
static int order_cells_of(const gt_ctx* ctx, int64_t n) {
    return int(std::min<int64_t>(std::min<int64_t>(8192, n / 32 * 32), std::max<int64_t>(64, (n / std::max(ctx->order_cell_rows, 32)) / 32 * 32)));
}

// ---- the cell order of ALL bound rows in two halves, for row-sharded builds (gt_knn_shard.cpp gt_points_cells_*) ----
// Half one, before any working copy of the whole point set exists: the landmark rows (the same evenly strided rows
// gt_query_order takes, prepared from the raw rows - conversions are row-wise, the bits are those of the full copy) and the
// cells of the rows [row0, row1) only: a rank's share of the assignment (1 / world of assign_cells_kernel's work).
// Needs ctx->X, n, d, dtype, DP, prec = 1 with the compact copy, ctx->sc.  cells_out: device uint32 [row1 - row0].
int gt_order_cells_partial(gt_ctx* ctx, int64_t row0, int64_t row1, uint32_t* cells_out, int* active) {
    *active = 0;
    const int64_t kMinRows = ctx->order_min_rows, n = ctx->n;
    if (!ctx->query_order || ctx->prec != 1 || ctx->fast_mode == 0 || ctx->DP == 0 || ctx->wide || n < std::max<int64_t>(kMinRows, 64))
        return GT_OK;
    const int L = order_cells_of(ctx, n);
    const int64_t step = n / L, nloc = row1 - row0;
    const size_t esz = ctx->dtype == GT_F32 ? 4 : 8;
    const int rw_raw = int(size_t(ctx->d) * esz / 4), rw = ctx->DP / 2;
    // landmark rows: raw rows -> working copy (hi plane = land_Y, seeds = land_h)
    GT_HIP(ctx, ctx->land_X.reserve(size_t(L) * ctx->d * esz));
    GT_HIP(ctx, ctx->land_Yp.reserve(size_t(L) * ctx->DP * sizeof(float)));
    GT_HIP(ctx, ctx->land_xn.reserve(size_t(L) * sizeof(double)));
    // (one row more: the stand-in of the outlier cell, as in gt_query_order)
    GT_HIP(ctx, ctx->land_Y.reserve(size_t(L + 1) * rw * sizeof(uint32_t)));
    GT_HIP(ctx, ctx->land_h.reserve(size_t(L + 1) * sizeof(float)));
    GT_HIP(ctx, hipMemsetAsync(ctx->land_Y.as<uint32_t>() + size_t(L) * rw, 0, size_t(rw) * sizeof(uint32_t), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(ctx->land_h.as<float>() + L, 0, sizeof(float), ctx->stream));
    hipLaunchKernelGGL(gather_strided_rows_kernel, dim3((unsigned)ceil_div64(int64_t(L) * rw_raw, 256)), dim3(256), 0, ctx->stream,
                       static_cast<const uint32_t*>(ctx->X), step, L, rw_raw, ctx->land_X.as<uint32_t>());
    GT_HIP(ctx, hipGetLastError());
    GT_TRY(gt_prep_matrix(ctx, ctx->land_X.p, L, ctx->d, ctx->dtype, ctx->DP, L, ctx->land_Yp.as<float>(), ctx->land_xn.as<double>(),
                          ctx->land_h.as<float>(), nullptr, 1, ctx->sc, nullptr, ctx->land_Y.p));
    {
        // the landmarks in a coherent order (above): the same rows, renumbered - gathered and prepared again in that order
        int32_t* lidx = nullptr;
        GT_TRY(coherent_landmark_order(ctx, L, rw, ctx->land_Y.as<uint32_t>(), ctx->land_h.as<float>(), &lidx));
        if (lidx) {
            hipLaunchKernelGGL(gather_strided_rows_idx_kernel, dim3((unsigned)ceil_div64(int64_t(L) * rw_raw, 256)), dim3(256), 0,
                               ctx->stream, static_cast<const uint32_t*>(ctx->X), lidx, step, L, rw_raw, ctx->land_X.as<uint32_t>());
            GT_HIP(ctx, hipGetLastError());
            GT_TRY(gt_prep_matrix(ctx, ctx->land_X.p, L, ctx->d, ctx->dtype, ctx->DP, L, ctx->land_Yp.as<float>(),
                                  ctx->land_xn.as<double>(), ctx->land_h.as<float>(), nullptr, 1, ctx->sc, nullptr, ctx->land_Y.p));
        }
    }
    if (nloc > 0) {
        // working copy of the share, at the head of the context's own buffers (the full copy overwrites it later)
        const int bn = ctx->DP <= 64 ? 128 : 64;
        const int64_t n_pad = ceil_div64(n, bn) * bn;
        GT_HIP(ctx, ctx->Yp.reserve(size_t(n_pad) * ctx->DP * sizeof(float)));
        GT_HIP(ctx, ctx->Yc.reserve(size_t(n_pad) * ctx->DP * sizeof(_Float16)));
        GT_HIP(ctx, ctx->xn.reserve(size_t(n) * sizeof(double)));
        GT_HIP(ctx, ctx->hneg.reserve(size_t(n_pad) * sizeof(float)));
        GT_HIP(ctx, ctx->order_rows.reserve(size_t(nloc) * sizeof(float)));   // (starting thresholds of the assignment: not used)
        const char* Xs = static_cast<const char*>(ctx->X) + size_t(row0) * ctx->d * esz;
        GT_TRY(gt_prep_matrix(ctx, Xs, nloc, ctx->d, ctx->dtype, ctx->DP, nloc, ctx->Yp.as<float>(), ctx->xn.as<double>(),
                              ctx->hneg.as<float>(), nullptr, 1, ctx->sc, nullptr, ctx->Yc.p));
        // the outlier cell (see above): decided for the share from the share's own statistics - the shares are slices of the
        // caller's rows, the cells of all rows are gathered afterwards: one numbering everywhere, whatever each rank decided
        const bool outliers = ctx->order_outliers != 0;
        float* best = nullptr;
        if (outliers) {
            GT_HIP(ctx, ctx->order_tmp.reserve(size_t(nloc) * sizeof(float) + 64));
            best = ctx->order_tmp.as<float>();
        }
        GT_TRY(gt_launch_assign_cells(ctx, ctx->DP, ctx->Yc.as<float>(), ctx->land_Y.as<float>(), ctx->land_h.as<float>(), 0,
                                      int32_t(nloc), L, 1, cells_out, ctx->order_rows.as<float>(), best));
        if (outliers) {
            double* acc = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(best + nloc) + 7) & ~uintptr_t(7));
            uint32_t* counts = reinterpret_cast<uint32_t*>(acc + 1);
            GT_HIP(ctx, hipMemsetAsync(acc, 0, sizeof(double) + kOutlierTaus * sizeof(uint32_t), ctx->stream));
            const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(nloc, 256), int64_t(ctx->n_cu) * 2);
            hipLaunchKernelGGL(outlier_sum_kernel, dim3(grid), dim3(256), 0, ctx->stream, nloc, int64_t(0), ctx->hneg.as<float>(), best, acc);
            hipLaunchKernelGGL(outlier_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, nloc, int64_t(0), ctx->hneg.as<float>(), best, acc,
                               counts);
            const uint32_t cap = uint32_t(std::max<int64_t>(8, nloc / 512));
            hipLaunchKernelGGL(outlier_relabel_kernel, dim3((unsigned)ceil_div64(nloc, 256)), dim3(256), 0, ctx->stream, nloc, int64_t(0),
                               ctx->hneg.as<float>(), best, acc, counts, cap, uint32_t(L), cells_out);
            GT_HIP(ctx, hipGetLastError());
        }
    }
    ctx->cells_L = L + (ctx->order_outliers != 0 ? 1 : 0);
    *active = 1;
    return GT_OK;
}

// Half two: the cells of ALL rows (gathered from the ranks) -> the cell-sorted order (stable radix sort of (cell, row)
// pairs: the same order on every rank).  out_rows: device int32 [n]; the sorted cells end up in order_cell + n.
int gt_order_sort_cells(gt_ctx* ctx, const uint32_t* cells_all, int32_t* out_rows) {
    const int64_t n = ctx->n;
    const int L = ctx->cells_L;
    if (L <= 0) GT_FAIL(ctx, GT_E_STATE, "cell sort: no assignment to finish");
    GT_HIP(ctx, ctx->order_cell.reserve(size_t(n) * sizeof(uint32_t) * 2));
    GT_HIP(ctx, ctx->order_rows.reserve(size_t(n) * sizeof(int32_t)));
    uint32_t* cell = ctx->order_cell.as<uint32_t>();
    GT_HIP(ctx, hipMemcpyAsync(cell, cells_all, size_t(n) * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(iota_rows_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, int64_t(0), n,
                       ctx->order_rows.as<int32_t>());
    GT_HIP(ctx, hipGetLastError());
    int bits = 1;
    while ((1 << bits) < L) ++bits;
    size_t tmp_bytes = 0;
    GT_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, cell, cell + n, ctx->order_rows.as<int32_t>(), out_rows, size_t(n), 0u,
                                          unsigned(bits), ctx->stream));
    GT_HIP(ctx, ctx->order_tmp.reserve(tmp_bytes));
    GT_HIP(ctx, rocprim::radix_sort_pairs(ctx->order_tmp.p, tmp_bytes, cell, cell + n, ctx->order_rows.as<int32_t>(), out_rows,
                                          size_t(n), 0u, unsigned(bits), ctx->stream));
    ctx->order_L = L;
    ctx->order_has_thr0 = 0;
    ctx->order_outlier_cell = ctx->order_outliers != 0 ? L - 1 : -1;
    return GT_OK;
}

int gt_query_order(gt_ctx* ctx, const float* Qc, int64_t q0, int64_t nq, int need, int32_t* out_rows, float* out_thr0,
                   int* active) {
    *active = 0;
    ctx->order_L = 0;
    ctx->order_has_thr0 = 0;
    ctx->order_outlier_cell = -1;
    if (ctx->presorted && Qc == ctx->Yc.as<float>() && ctx->vcell.p && q0 >= 0 && q0 + nq <= ctx->n && nq >= 1) {
        // the bound points were renumbered in cell-sorted order (gt_points_cell_sort): rows [q0, q0 + nq) are grouped already
        GT_HIP(ctx, ctx->order_cell.reserve(size_t(nq) * sizeof(uint32_t) * 2));
        uint32_t* cell = ctx->order_cell.as<uint32_t>();
        GT_HIP(ctx, hipMemcpyAsync(cell, ctx->vcell.as<uint32_t>() + q0, size_t(nq) * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                                   ctx->stream));
        GT_HIP(ctx, hipMemcpyAsync(cell + nq, ctx->vcell.as<uint32_t>() + q0, size_t(nq) * sizeof(uint32_t),
                                   hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(iota_rows_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, ctx->stream, q0, nq, out_rows);
        GT_HIP(ctx, hipGetLastError());
        ctx->order_L = ctx->presorted_L;
        *active = 1;
        return GT_OK;
    }
    ctx->order_coherent_active = 0;   // (an order made here is made anew)
    const int64_t kMinRows = ctx->order_min_rows;   // below this the whole launch is a few workgroup rounds
    // Many query rows against FEW points (a random-landmark assignment: 1e6 rows, 2000 landmarks): nearly every point is a
    // landmark of the order - what the queries gain is the threshold (the need-th best of the scores seen here bounds the need-th
    // best of all from below): the candidate pass then admits a handful of rows per query instead of filling its lists from an
    // open threshold (N = 1e6 against 2000: candidate pass 10.9 -> 1.3 ms, re-rank 8.0 -> 4.6 ms).
    const bool few_points = ctx->n < kMinRows && ctx->n >= 64 && Qc != ctx->Yc.as<float>();
    if (!ctx->query_order || ctx->prec != 1 || ctx->fast_mode == 0 || !ctx->Yc.p || !Qc || nq < kMinRows ||
        (ctx->n < std::max<int64_t>(kMinRows, 64) && !few_points))
        return GT_OK;
    const int rw = ctx->DP / 2;   // dwords per row of the compact copy
    // (up to 8192 cells: beyond a million rows the cells would otherwise grow, and with them the share of clusters that own
    //  no landmark - see the bound pass, gt_sym.hip)
    int L = few_points ? int(std::min<int64_t>(8192, ctx->n / 32 * 32)) : order_cells_of(ctx, ctx->n);
    const int64_t step = ctx->n / L;
    // (one row more than there are landmarks: the outlier cell's stand-in - zeros, the origin - for whoever indexes the
    //  landmark rows by cell; neighbourhoods are approximate by design)
    GT_HIP(ctx, ctx->land_Y.reserve(size_t(L + 1) * rw * sizeof(uint32_t)));
    GT_HIP(ctx, ctx->land_h.reserve(size_t(L + 1) * sizeof(float)));
    GT_HIP(ctx, hipMemsetAsync(ctx->land_Y.as<uint32_t>() + size_t(L) * rw, 0, size_t(rw) * sizeof(uint32_t), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(ctx->land_h.as<float>() + L, 0, sizeof(float), ctx->stream));
    GT_HIP(ctx, ctx->order_cell.reserve(size_t(nq) * sizeof(uint32_t) * 2));
    GT_HIP(ctx, ctx->order_rows.reserve(size_t(nq) * sizeof(int32_t)));
    hipLaunchKernelGGL(gather_landmarks_kernel, dim3((unsigned)ceil_div64(int64_t(L) * rw, 256)), dim3(256), 0, ctx->stream,
                       ctx->Yc.as<uint32_t>(), ctx->hneg.as<float>(), step, L, rw, ctx->land_Y.as<uint32_t>(),
                       ctx->land_h.as<float>());
    GT_HIP(ctx, hipGetLastError());
    if (!few_points) {
        // ... in a coherent order (coherent_landmark_order): the same rows, renumbered
        int32_t* lidx = nullptr;
        GT_TRY(coherent_landmark_order(ctx, L, rw, ctx->land_Y.as<uint32_t>(), ctx->land_h.as<float>(), &lidx));
        if (lidx) {
            hipLaunchKernelGGL(gather_landmarks_idx_kernel, dim3((unsigned)ceil_div64(int64_t(L) * rw, 256)), dim3(256), 0, ctx->stream,
                               ctx->Yc.as<uint32_t>(), ctx->hneg.as<float>(), lidx, step, L, rw, ctx->land_Y.as<uint32_t>(),
                               ctx->land_h.as<float>());
            GT_HIP(ctx, hipGetLastError());
        }
    }
    uint32_t* cell = ctx->order_cell.as<uint32_t>();
    uint32_t* cell_sorted = cell + nq;
    // the outlier cell (above): for the rows of the bound point set (their seeds -|x|^2 / 2 are at hand)
    const bool outliers = ctx->order_outliers != 0 && Qc == ctx->Yc.as<float>() && ctx->hneg.p != nullptr;
    float* best = nullptr;
    if (outliers) {
        GT_HIP(ctx, ctx->order_tmp.reserve(size_t(nq) * sizeof(float) + 64));
        best = ctx->order_tmp.as<float>();
    }
    GT_TRY(gt_launch_assign_cells(ctx, ctx->DP, Qc, ctx->land_Y.as<float>(), ctx->land_h.as<float>(), q0,
                                  int32_t(nq), L, need, cell, out_thr0, best));
    int n_cells = L;
    if (outliers) {
        // acc: sum of the scores (double) | five counts, behind the scores in the same buffer
        double* acc = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(best + nq) + 7) & ~uintptr_t(7));
        uint32_t* counts = reinterpret_cast<uint32_t*>(acc + 1);
        GT_HIP(ctx, hipMemsetAsync(acc, 0, sizeof(double) + kOutlierTaus * sizeof(uint32_t), ctx->stream));
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(nq, 256), int64_t(ctx->n_cu) * 2);
        hipLaunchKernelGGL(outlier_sum_kernel, dim3(grid), dim3(256), 0, ctx->stream, nq, q0, ctx->hneg.as<float>(), best, acc);
        hipLaunchKernelGGL(outlier_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, nq, q0, ctx->hneg.as<float>(), best, acc, counts);
        const uint32_t cap = uint32_t(std::max<int64_t>(64, nq / 512));
        hipLaunchKernelGGL(outlier_relabel_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, ctx->stream, nq, q0,
                           ctx->hneg.as<float>(), best, acc, counts, cap, uint32_t(L), cell);
        GT_HIP(ctx, hipGetLastError());
        n_cells = L + 1;
    }
    hipLaunchKernelGGL(iota_rows_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, ctx->stream, q0, nq,
                       ctx->order_rows.as<int32_t>());
    GT_HIP(ctx, hipGetLastError());
    int bits = 1;
    while ((1 << bits) < n_cells) ++bits;
    size_t tmp_bytes = 0;
    GT_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, cell, cell_sorted, ctx->order_rows.as<int32_t>(), out_rows,
                                          size_t(nq), 0u, unsigned(bits), ctx->stream));
    GT_HIP(ctx, ctx->order_tmp.reserve(tmp_bytes));
    GT_HIP(ctx, rocprim::radix_sort_pairs(ctx->order_tmp.p, tmp_bytes, cell, cell_sorted, ctx->order_rows.as<int32_t>(),
                                          out_rows, size_t(nq), 0u, unsigned(bits), ctx->stream));
    ctx->order_outlier_cell = (outliers && q0 == 0 && nq == ctx->n) ? L : -1;
    ctx->order_L = n_cells;   // cells of the order just built (cell ids of the sorted rows: order_cell + nq)
    ctx->order_has_thr0 = 1;
    *active = 1;
    return GT_OK;
}
