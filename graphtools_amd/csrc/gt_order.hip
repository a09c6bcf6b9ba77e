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

}  // namespace

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
    GT_HIP(ctx, ctx->land_Y.reserve(size_t(L) * rw * sizeof(uint32_t)));
    GT_HIP(ctx, ctx->land_h.reserve(size_t(L) * sizeof(float)));
    hipLaunchKernelGGL(gather_strided_rows_kernel, dim3((unsigned)ceil_div64(int64_t(L) * rw_raw, 256)), dim3(256), 0, ctx->stream,
                       static_cast<const uint32_t*>(ctx->X), step, L, rw_raw, ctx->land_X.as<uint32_t>());
    GT_HIP(ctx, hipGetLastError());
    GT_TRY(gt_prep_matrix(ctx, ctx->land_X.p, L, ctx->d, ctx->dtype, ctx->DP, L, ctx->land_Yp.as<float>(), ctx->land_xn.as<double>(),
                          ctx->land_h.as<float>(), nullptr, 1, ctx->sc, nullptr, ctx->land_Y.p));
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
        GT_TRY(gt_launch_assign_cells(ctx, ctx->DP, ctx->Yc.as<float>(), ctx->land_Y.as<float>(), ctx->land_h.as<float>(), 0,
                                      int32_t(nloc), L, 1, cells_out, ctx->order_rows.as<float>()));
    }
    ctx->cells_L = L;
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
    return GT_OK;
}

int gt_query_order(gt_ctx* ctx, const float* Qc, int64_t q0, int64_t nq, int need, int32_t* out_rows, float* out_thr0,
                   int* active) {
    *active = 0;
    ctx->order_L = 0;
    ctx->order_has_thr0 = 0;
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
    const int64_t kMinRows = ctx->order_min_rows;   // below this the whole launch is a few workgroup rounds
    if (!ctx->query_order || ctx->prec != 1 || ctx->fast_mode == 0 || !ctx->Yc.p || !Qc || nq < kMinRows || ctx->n < std::max<int64_t>(kMinRows, 64))
        return GT_OK;
    const int rw = ctx->DP / 2;   // dwords per row of the compact copy
    // (up to 8192 cells: beyond a million rows the cells would otherwise grow, and with them the share of clusters that own
    //  no landmark - see the bound pass, gt_sym.hip)
    int L = order_cells_of(ctx, ctx->n);
    const int64_t step = ctx->n / L;
    GT_HIP(ctx, ctx->land_Y.reserve(size_t(L) * rw * sizeof(uint32_t)));
    GT_HIP(ctx, ctx->land_h.reserve(size_t(L) * sizeof(float)));
    GT_HIP(ctx, ctx->order_cell.reserve(size_t(nq) * sizeof(uint32_t) * 2));
    GT_HIP(ctx, ctx->order_rows.reserve(size_t(nq) * sizeof(int32_t)));
    hipLaunchKernelGGL(gather_landmarks_kernel, dim3((unsigned)ceil_div64(int64_t(L) * rw, 256)), dim3(256), 0, ctx->stream,
                       ctx->Yc.as<uint32_t>(), ctx->hneg.as<float>(), step, L, rw, ctx->land_Y.as<uint32_t>(),
                       ctx->land_h.as<float>());
    GT_HIP(ctx, hipGetLastError());
    uint32_t* cell = ctx->order_cell.as<uint32_t>();
    uint32_t* cell_sorted = cell + nq;
    GT_TRY(gt_launch_assign_cells(ctx, ctx->DP, Qc, ctx->land_Y.as<float>(), ctx->land_h.as<float>(), q0,
                                  int32_t(nq), L, need, cell, out_thr0));
    hipLaunchKernelGGL(iota_rows_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, ctx->stream, q0, nq,
                       ctx->order_rows.as<int32_t>());
    GT_HIP(ctx, hipGetLastError());
    int bits = 1;
    while ((1 << bits) < L) ++bits;
    size_t tmp_bytes = 0;
    GT_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, cell, cell_sorted, ctx->order_rows.as<int32_t>(), out_rows,
                                          size_t(nq), 0u, unsigned(bits), ctx->stream));
    GT_HIP(ctx, ctx->order_tmp.reserve(tmp_bytes));
    GT_HIP(ctx, rocprim::radix_sort_pairs(ctx->order_tmp.p, tmp_bytes, cell, cell_sorted, ctx->order_rows.as<int32_t>(),
                                          out_rows, size_t(nq), 0u, unsigned(bits), ctx->stream));
    ctx->order_L = L;   // cells of the order just built (cell ids of the sorted rows: order_cell + nq)
    ctx->order_has_thr0 = 1;
    *active = 1;
    return GT_OK;
}
