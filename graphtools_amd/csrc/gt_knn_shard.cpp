// Row-sharded symmetric candidate pass (one process per GPU; the collectives are the host language's, dist.py).
//
// Single rank (gt_knn.cpp, gt_sym.hip): launch A seeds a fixed threshold per row, launch B scores every unordered pair
// of rows once and files the survivors in the lists of BOTH rows, the re-rank turns each list into an exact table.
// Sharded over `world` ranks that all hold the full point set (the all-gather of the d-dim points comes first anyway):
//   plan      every rank builds the same cell-sorted order (deterministic: MFMA assignment + stable radix sort)
//   seed      launch A for the rank's share of the query blocks of that order only (1/world of the work);
//             -> the host all-gathers the thresholds (4 bytes per row)
//   collect   launch B with every block's walk cut into world x nseg pieces: this rank takes the pieces
//             ((rank + block) mod world) x nseg ... - 1/world of the pair scores, the hit-rich pieces next to the
//             diagonal spread round robin; the survivors land in this rank's own partial lists of ALL rows
//   emit      partial lists -> 16-byte records {row local to its owner, key}, bucketed by the owner of the ROW in the
//             caller's row split (the rank that builds that row of the kernel matrix)
//             -> the host moves them with the same all-to-all the triplets use
//   finish    received records -> lists of the owned rows; gt_knn_candidates() for exactly these rows then re-ranks
//             them (rerank_sym_kernel, thresholds found through the inverse permutation) and runs the usual repairs.
// The union of the ranks' partial lists is exactly the list the single-rank pass builds (same thresholds, same scores),
// so the tables - and the graph - are those of the single-rank symmetric build.
#include "gt_knn.h"
#include "gt_knn_select.h"

// cells whose rows seed a row's thresholds in the row-sharded passes: a rank pays the fixed cost of the repair pass (launches,
// read-backs: 0.4 ms) as soon as ONE of its rows needs it, so the neighbourhood stays wider than the single-GPU default
// (8 since round 4's sweep: rank 0 of 8 on C3 repairs nothing with 12 cells, a handful of rows with 8 - 4.89 against 5.20 ms)
static inline int shard_cells(const gt_ctx* ctx) { return std::max(ctx->sym_cells, ctx->sym_cells_shard); }

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <utility>
#include <vector>

namespace {

bool shard_applicable(const gt_ctx* ctx, int need_m) {
    const int bq = gt_select_bq(ctx->DP);
    const bool fast = ctx->prec == 1 && (ctx->fast_mode == 2 || (ctx->fast_mode == 1 && ctx->fast_ok != 0));
    return ctx->sym_mode != 0 && ctx->sym_ok != 0 && fast && (ctx->metric == 0 || (ctx->metric == 1 && ctx->sym_cosine != 0)) && !ctx->wide && ctx->DP != 0 &&
           need_m >= 1 && need_m <= ctx->nt8_max_need && need_m <= 64 && ctx->Yc.p != nullptr &&
           (ctx->sym_mode > 0 || ctx->n >= ctx->sym_min_rows) && ctx->n >= int64_t(8) * bq && ctx->n >= 4096;
}

}  // namespace

// ---- cell-sorted renumbering of the bound points ---------------------------------------------------------------------
// Row-sharded builds (dist.py) give every rank a run of consecutive rows.  In the caller's row order a rank's rows lie
// all over the point set: its candidate pairs are everybody's, and the symmetric pass had to ship every candidate it
// found to the row's owner (14 M records per rank at N = 1e6, world 8).  Renumbered in the cell-sorted order a rank's
// rows are whole landmark cells: the pairs a row needs are scored by the rank that owns the row, nothing is shipped
// before the transposed triplets of the symmetrisation.  The renumbering is a property of the CONTEXT (every stage works
// on the renumbered points and never learns about it); the caller's numbers come back in the last sort of the tail
// (gt_sparse.hip: RowSrc::relabel) and through gt_points_row_ids.
//   applied = 0: the points are too few (or too wide) for a cell order - nothing changed.
// perm (device int32 [n], with the sorted cells in order_cell + n): the bound points become rows perm[0], perm[1], ...
static int renumber_by(gt_ctx* ctx, const int32_t* perm) {
    KnnWork* k = ctx->knn;
    const int64_t n = ctx->n;
    StageSpan span(ctx, "renumber");
    // the points in the new order (the caller's buffer is not referenced any more), their row numbers and cells
    GT_HIP(ctx, ctx->xn.reserve(size_t(n) * sizeof(double)));   // (gathered along; recomputed below)
    GT_TRY(gt_sym_gather_points(ctx, perm));
    std::swap(ctx->X_own, k->Xs);
    ctx->X = ctx->X_own.p;
    k->xs_ready = false;
    GT_HIP(ctx, ctx->vperm.reserve(size_t(n) * sizeof(int32_t)));
    GT_HIP(ctx, ctx->vcell.reserve(size_t(n) * sizeof(uint32_t)));
    GT_HIP(ctx, hipMemcpyAsync(ctx->vperm.p, perm, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    GT_HIP(ctx, hipMemcpyAsync(ctx->vcell.p, ctx->order_cell.as<uint32_t>() + n, size_t(n) * sizeof(uint32_t),
                               hipMemcpyDeviceToDevice, ctx->stream));
    // norms and working copies of the renumbered rows (the same rows: the same norms, the same float16 scale; the landmark
    // rows the order was made with keep describing the cells)
    GT_TRY(gt_prep_points(ctx));
    ctx->presorted = 1;
    ctx->presorted_L = ctx->order_L;
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

extern "C" int gt_points_cell_sort(gt_ctx* ctx, int32_t* applied) {
    if (!ctx || !applied) return GT_E_ARG;
    *applied = 0;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->n <= 0 || !ctx->X) GT_FAIL(ctx, GT_E_STATE, "gt_points_cell_sort: no points bound (call gt_set_points first)");
    if (ctx->presorted) {
        *applied = 1;
        return GT_OK;
    }
    if (ctx->DP == 0 || ctx->wide || ctx->prec != 1 || ctx->fast_mode == 0 || !ctx->Yc.p) return GT_OK;
    if (!ctx->knn) ctx->knn = new KnnWork();
    KnnWork* k = ctx->knn;
    const int64_t n = ctx->n;
    int ordered = 0;
    GT_HIP(ctx, k->qorder.reserve(size_t(n) * sizeof(int32_t)));
    GT_HIP(ctx, k->qthr0.reserve(size_t(n) * sizeof(float)));
    {
        StageSpan span(ctx, "query_order");
        GT_TRY(gt_query_order(ctx, ctx->Yc.as<float>(), 0, n, 1, k->qorder.as<int32_t>(), k->qthr0.as<float>(), &ordered));
    }
    k->ordered = false;
    k->xs_ready = false;
    k->sh_stage = 0;
    if (!ordered || ctx->order_L <= 0) return GT_OK;
    GT_TRY(renumber_by(ctx, k->qorder.as<int32_t>()));
    *applied = 1;
    return GT_OK;
}

// The same renumbering with the cell assignment SPLIT over the ranks of a sharded build (assign_cells_kernel is 0.77 ms
// of replicated work at N = 1e6 when every rank assigns every row, and the full working copy it reads another 0.26 ms):
//   gt_points_cells_begin   binds the gathered points (device memory) WITHOUT making the working copies, prepares the
//                           landmark rows and the rows [row0, row1) only, assigns those rows to their cells
//                           -> cells_out (device uint32 [row1 - row0]); the host all-gathers the ranks' cells (4 B per row)
//   gt_points_cells_finish  cells of all rows (device uint32 [n]) -> stable sort, renumbering, working copies
// applied = 0 from _begin: no cell order for these points (too few / too wide) - they are bound as gt_set_points binds them,
// nothing to gather, _finish must not be called.
extern "C" int gt_points_cells_begin(gt_ctx* ctx, const void* X_dev, int64_t n, int32_t d, int32_t dtype, int64_t row0,
                                     int64_t row1, void* cells_out_dev, int32_t* applied) {
    if (!ctx || !applied) return GT_E_ARG;
    *applied = 0;
    GT_TRY(gt_bind_points(ctx, X_dev, n, d, dtype, 1));
    if (row0 < 0 || row1 > n || row1 < row0 || (row1 > row0 && !cells_out_dev)) GT_FAIL(ctx, GT_E_ARG, "gt_points_cells_begin: bad row range");
    int active = 0;
    if (ctx->DP != 0 && !ctx->wide && ctx->prec == 1 && ctx->fast_mode != 0) {
        ctx->sc = gt_f16_scale(ctx->maxabs);
        if (ctx->metric == 1) {   // (the normalised rows: their own max |x|, as gt_prep_points measures it)
            GT_TRY(gt_max_abs(ctx, ctx->X, ctx->n * int64_t(ctx->d), ctx->dtype, &ctx->maxabs, nullptr));
            ctx->sc = gt_f16_scale(ctx->maxabs);
        }
        StageSpan span(ctx, "query_order");
        GT_TRY(gt_order_cells_partial(ctx, row0, row1, static_cast<uint32_t*>(cells_out_dev), &active));
    }
    if (!active) return gt_prep_bound_points(ctx);   // no cell order: plain gt_set_points
    ctx->cells_pending = 1;
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *applied = 1;
    return GT_OK;
}

extern "C" int gt_points_cells_finish(gt_ctx* ctx, const void* cells_all_dev) {
    if (!ctx || !cells_all_dev) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->cells_pending || ctx->n <= 0 || !ctx->X) GT_FAIL(ctx, GT_E_STATE, "gt_points_cells_finish: call gt_points_cells_begin first");
    ctx->cells_pending = 0;
    if (!ctx->knn) ctx->knn = new KnnWork();
    KnnWork* k = ctx->knn;
    GT_HIP(ctx, k->qorder.reserve(size_t(ctx->n) * sizeof(int32_t)));
    {
        StageSpan span(ctx, "query_order");
        GT_TRY(gt_order_sort_cells(ctx, static_cast<const uint32_t*>(cells_all_dev), k->qorder.as<int32_t>()));
    }
    k->ordered = false;
    k->xs_ready = false;
    k->sh_stage = 0;
    return renumber_by(ctx, k->qorder.as<int32_t>());
}

// the caller's row number of the context's rows [v0, v1) (identity when the points were not renumbered)
extern "C" int gt_points_row_ids(gt_ctx* ctx, int64_t v0, int64_t v1, int32_t* out, int32_t out_on_device) {
    if (!ctx || !out) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (v0 < 0 || v1 > ctx->n || v1 < v0) GT_FAIL(ctx, GT_E_ARG, "gt_points_row_ids: bad row range");
    if (v1 == v0) return GT_OK;
    if (ctx->presorted) {
        GT_HIP(ctx, hipMemcpyAsync(out, ctx->vperm.as<int32_t>() + v0, size_t(v1 - v0) * sizeof(int32_t),
                                   out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return GT_OK;
    }
    std::vector<int32_t> ids(size_t(v1 - v0));
    for (int64_t v = v0; v < v1; ++v) ids[size_t(v - v0)] = int32_t(v);
    GT_HIP(ctx, hipMemcpyAsync(out, ids.data(), ids.size() * sizeof(int32_t),
                               out_on_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

// ---- row-sharded symmetric pass on renumbered points: no exchange at all --------------------------------------------
// The bound points are in cell-sorted order (gt_points_cell_sort) and the rank owns the rows [r0, r1) = whole 1024-row
// blocks of that order (gt_points_shard_splits).  Everything launch A and launch B need of OTHER rows is their
// coordinates, which every rank holds: the rank seeds its own rows, fixes their thresholds, lists the (64 own queries,
// 32 rows) units its own rows' radii cannot rule out - against EVERY sub-tile, other ranks' included - and files each
// survivor under the query only.  A pair of rows of two ranks is scored twice, once by each owner (2 / world of the pair
// scores per rank instead of 1 / world), and in exchange nothing is shipped: no thresholds, no candidate records.  The
// lists equal the single-rank lists of the same rows (same thresholds, same scores), gt_knn_candidates() re-ranks them.
//   applies = 0: not on this point set / these parameters, or the cell bounds leave too many units - the caller's
//   gt_graph_begin then runs the classic pass for the rank's rows (any rank may, on its own: nothing is shared).
int gt_knn_shard_local(gt_ctx* ctx, int64_t r0, int64_t r1, int need_m, double rkf, int32_t* applies) {
    *applies = 0;
    if (ctx->n <= 0 || !ctx->X) GT_FAIL(ctx, GT_E_STATE, "no points bound (call gt_set_points first)");
    if (r0 < 0 || r1 > ctx->n || r1 <= r0) GT_FAIL(ctx, GT_E_ARG, "sym shard: bad row range");
    if (!ctx->knn) ctx->knn = new KnnWork();
    KnnWork* k = ctx->knn;
    k->sh_stage = 0;
    const int bq = gt_select_bq(ctx->DP), bn = gt_select_bn(ctx->DP);
    if (!ctx->presorted || !shard_applicable(ctx, need_m) || ctx->DP < 32 || bq != 256 || ctx->sym_two_stage == 0 ||
        ctx->sym_bounds == 0 || ctx->sym_dense_seed == 0 || need_m > 64)
        return GT_OK;
    if (r0 % 1024 != 0 || (r1 % 1024 != 0 && r1 != ctx->n)) return GT_OK;   // whole blocks (gt_points_shard_splits)
    const int64_t n_pad_s = ceil_div64(ctx->n, 1024) * 1024;
    const int64_t p0 = r0, p1 = ceil_div64(r1, 1024) * 1024;   // (the last rank's run ends with the pad rows)
    const int tcap = ctx->sym_tcap;
    // identity order (the points are sorted), cells from the renumbering
    int ordered = 0;
    GT_HIP(ctx, k->qorder.reserve(size_t(ctx->n) * sizeof(int32_t)));
    GT_HIP(ctx, k->qthr0.reserve(size_t(ctx->n) * sizeof(float)));
    GT_TRY(gt_query_order(ctx, ctx->Yc.as<float>(), 0, ctx->n, need_m, k->qorder.as<int32_t>(), k->qthr0.as<float>(), &ordered));
    k->ordered = false;
    k->xs_ready = false;
    if (!ordered || ctx->order_L <= 0) return GT_OK;
    const int32_t* perm = k->qorder.as<int32_t>();
    GT_HIP(ctx, k->Ycs.reserve(size_t(n_pad_s) * ctx->DP * sizeof(_Float16)));
    GT_HIP(ctx, k->hnegs.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->hnegs_fin.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->tlists.reserve(size_t(n_pad_s) * size_t(tcap) * sizeof(uint64_t)));
    GT_HIP(ctx, k->tcounts.reserve(size_t(n_pad_s) * sizeof(uint32_t)));
    GT_HIP(ctx, k->sym_stat.reserve(8 * sizeof(unsigned long long)));
    GT_HIP(ctx, k->lists.reserve(size_t(p1 - p0) * 64 * sizeof(uint64_t)));   // the dense seeding kernel's keys: own blocks only
    GT_HIP(ctx, k->counts.reserve(size_t(n_pad_s) * sizeof(uint32_t)));
    GT_HIP(ctx, k->thr_final.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_g.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_farcnt.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_racc.reserve(4 * sizeof(double)));
    GT_HIP(ctx, k->sym_rrow.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_qtot.reserve(4 * sizeof(uint32_t)));
    const int64_t bcap = ctx->sym_bound_cap > 0 ? ctx->sym_bound_cap : (int64_t(1) << 22);
    GT_HIP(ctx, k->sym_qdense.reserve(size_t(bcap) * sizeof(uint2)));
    GT_HIP(ctx, hipMemsetAsync(k->sym_farcnt.p, 0, size_t(n_pad_s) * sizeof(float), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(k->sym_stat.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(k->sym_racc.p, 0, 4 * sizeof(double), ctx->stream));
    const int n_tiles_s = int(n_pad_s / bn);
    const int stride_a = ctx->sym_stride > 0 && n_tiles_s >= 8 * ctx->sym_stride ? ctx->sym_stride : 0;
    const int tile_stride = ((stride_a ? n_tiles_s / stride_a + 1 : 0) + ctx->sym_max_nb + bq / bn + 63) / 64 * 64;
    if (tile_stride > 1024) return GT_OK;
    GT_HIP(ctx, k->sym_tiles.reserve(size_t(n_pad_s / bq) * tile_stride * sizeof(int32_t)));
    GT_HIP(ctx, k->sym_tile_cnt.reserve(size_t(n_pad_s / bq) * sizeof(int32_t)));
    ErrModel em = gt_err_model(ctx, 2);
    em.rel += 8.0 * 5.9604644775390625e-08;   // as in the single-rank pass (gt_knn.cpp)
    {
        StageSpan span(ctx, "sym_prepare");
        // the compact copy padded to whole 1024-row blocks (the order is the identity: a copy with pad rows)
        GT_TRY(gt_sym_gather(ctx, perm, n_pad_s, k->Ycs.p, k->hnegs.as<float>(), k->hnegs_fin.as<float>()));
        GT_TRY(gt_sym_schedule(ctx, n_pad_s, bq, bn, shard_cells(ctx), stride_a, ctx->sym_max_nb, tile_stride, k->sym_work,
                               k->sym_tiles.as<int32_t>(), k->sym_tile_cnt.as<int32_t>(),
                               k->sym_stat.as<unsigned long long>() + 5, p0, p1));
        // every row that is not the rank's: no threshold (+inf: asks for nothing, admits nothing, has no radius)
        GT_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(k->thr_final.p), 0x7F800000, size_t(n_pad_s), ctx->stream));
    }
    uint64_t* lists0 = k->lists.as<uint64_t>() - size_t(p0) * 64;   // (addressed by sorted position: only [p0, p1) is touched)
    int64_t seeded_to = p0;
    if (ctx->sym_mode < 0 && ctx->sym_ok < 0 && p1 - p0 >= int64_t(16) * 2048) {
        // first build on these points: a sixteenth of the rank's rows is seeded first and asked the questions the whole
        // launch would be asked (gt_knn.cpp) - a rank whose rows have no cluster structure (isotropic data) gives up after
        // 0.4 ms instead of 1.7 and runs the classic pass
        const int64_t ps = std::max<int64_t>(2048, ((p1 - p0) / 16) / 256 * 256), prows = std::min<int64_t>(p0 + ps, ctx->n) - p0;
        unsigned long long far_s = 0;
        {
            StageSpan span(ctx, "sym_seed");
            GT_TRY(gt_sym_seed_dense(ctx, ctx->DP, k->Ycs.p, k->hnegs_fin.as<float>(), ctx->n, n_pad_s, k->sym_tiles.as<int32_t>(),
                                     k->sym_tile_cnt.as<int32_t>(), tile_stride, bq, p0 / 128, ps / 128, need_m, lists0, 64,
                                     k->counts.as<uint32_t>()));
        }
        {
            StageSpan span(ctx, "sym_prepare");
            GT_TRY(gt_sym_thresholds(ctx, perm, n_pad_s, k->hnegs.as<float>(), lists0, 64, k->counts.as<uint32_t>(), need_m, em,
                                     std::max(1.0, std::fabs(rkf)), k->thr_final.as<float>(), k->sym_g.as<float>(), nullptr,
                                     k->sym_work, shard_cells(ctx), k->sym_stat.as<unsigned long long>() + 2,
                                     k->sym_farcnt.as<float>(), p0, p0 + ps, true));
            GT_HIP(ctx, hipMemcpyAsync(&far_s, k->sym_stat.as<unsigned long long>() + 2, sizeof(far_s), hipMemcpyDeviceToHost, ctx->stream));
            GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        const double est_s = double(need_m) + double(std::max(stride_a, 1)) * double(far_s) / double(std::max<int64_t>(prows, 1));
        if ((stride_a > 0 && est_s > double(tcap) / 8.0) || double(far_s) >= ctx->sym_sample_far * double(prows)) {
            k->sym_far = int64_t(double(far_s) * double(r1 - r0) / double(std::max<int64_t>(prows, 1)));
            ctx->sym_ok = 0;
            return GT_OK;
        }
        seeded_to = p0 + ps;
    }
    if (seeded_to < p1) {
        StageSpan span(ctx, "sym_seed");
        GT_TRY(gt_sym_seed_dense(ctx, ctx->DP, k->Ycs.p, k->hnegs_fin.as<float>(), ctx->n, n_pad_s, k->sym_tiles.as<int32_t>(),
                                 k->sym_tile_cnt.as<int32_t>(), tile_stride, bq, seeded_to / 128, (p1 - seeded_to) / 128, need_m,
                                 lists0, 64, k->counts.as<uint32_t>()));
    }
    k->sym_seed_dense = true;
    k->sh_lstride = 64;
    unsigned long long far = 0;
    {
        StageSpan span(ctx, "sym_prepare");
        // (the exact stages read the points themselves: they ARE in sorted order)
        if (seeded_to < p1)
            GT_TRY(gt_sym_thresholds(ctx, perm, n_pad_s, k->hnegs.as<float>(), lists0, 64, k->counts.as<uint32_t>(), need_m, em,
                                     std::max(1.0, std::fabs(rkf)), k->thr_final.as<float>(), k->sym_g.as<float>(), nullptr,
                                     k->sym_work, shard_cells(ctx), k->sym_stat.as<unsigned long long>() + 2,
                                     k->sym_farcnt.as<float>(), seeded_to, p1));
        GT_TRY(gt_sym_radius_sum(ctx, perm, p0, std::min<int64_t>(p1, ctx->n), k->thr_final.as<float>(), em, k->sym_racc.as<double>()));
        GT_HIP(ctx, hipMemsetAsync(k->tcounts.p, 0, size_t(n_pad_s) * sizeof(uint32_t), ctx->stream));
        GT_HIP(ctx, hipMemcpyAsync(&far, k->sym_stat.as<unsigned long long>() + 2, sizeof(far), hipMemcpyDeviceToHost, ctx->stream));
    }
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    k->sym_far = int64_t(far);
    if (ctx->sym_mode < 0) {
        // the predictor of the single-rank pass (gt_knn.cpp) on this rank's rows
        const double est = double(need_m) + double(std::max(stride_a, 1)) * double(far) / double(r1 - r0);
        if (stride_a > 0 && est > double(tcap) / 8.0) return GT_OK;
    }
    uint32_t left = 0;
    {
        StageSpan span(ctx, "sym_bound");
        // orphans (statistics of the rank's own rows), radii, then the units the cells cannot rule out
        GT_TRY(gt_sym_orphan_cut(ctx, perm, k->thr_final.as<float>(), k->sym_farcnt.as<float>(), em, k->sym_racc.as<double>(),
                                 need_m, 0.25));
        GT_TRY(gt_sym_row_radius(ctx, perm, n_pad_s, k->thr_final.as<float>(), em, k->sym_rrow.as<float>()));
        GT_TRY(gt_sym_bound_queue(ctx, n_pad_s, k->Ycs.p, k->sym_rrow.as<float>(), k->sym_bwork, k->sym_qdense.as<uint2>(),
                                  uint32_t(bcap), k->sym_qtot.as<uint32_t>(), 1, 0, 1, p0, p1));
        GT_HIP(ctx, hipMemcpyAsync(&left, k->sym_qtot.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->dbg_select & 2048)
        fprintf(stderr, "[gt] local shard rows [%lld, %lld): bound pass leaves %u units (capacity %lld), far-kept %llu\n",
                (long long)r0, (long long)r1, left, (long long)bcap, far);
    const bool bound_done = int64_t(left) <= bcap;
    SelectArgs ta;   // the two-stage collect of the rank's own blocks, where the cell bounds leave too many units
    if (!bound_done) {
        // Round 6: points near a low-dimensional sheet (the manifold set: 41 % of the cell pairs undecided) used to send the
        // rank to the classic pass here - 27 ms per rank at world 8 where the single-rank build takes 43.  Now the rank's own
        // 1024-row query blocks stream EVERY tile through stage one of the two-stage collect (16 principal directions,
        // forward test only: 2 / world of the single rank's stage-one work), the units that pass go to the cold launch, filed
        // under the queries only - the same lists.  The frame and its forecast are formed from the same rows on every rank.
        if (n_pad_s % 1024 != 0 || !(ctx->sym_two_stage > 0 || ctx->sym_two_ok != 0)) return GT_OK;
        ta.dp = ctx->DP;
        ta.prec = 2;
        ta.mode = 2;
        ta.nt = 8;
        ta.Yp = ta.Qp = k->Ycs.as<float>();
        ta.hneg = k->hnegs.as<float>();
        ta.n_pad = n_pad_s;
        ta.q0 = 0;
        ta.nq = int32_t(ctx->n);
        ta.counts = k->counts.as<uint32_t>();
        ta.thr_in = k->thr_final.as<float>();
        ta.sym.g = k->sym_g.as<float>();
        ta.sym.gmin = nullptr;
        ta.sym.tlists = k->tlists.as<uint64_t>();
        ta.sym.tcounts = k->tcounts.as<uint32_t>();
        ta.sym.tcap = tcap;
        ta.sym.own_only = 1;
        ta.sym.block0 = int32_t(p0 / 1024);
        ta.sym.nblk = int32_t((p1 - p0) / 1024);
        GT_HIP(ctx, k->sym_gmin.reserve(size_t(n_pad_s / 32) * sizeof(float)));
        GT_TRY(gt_sym_two_stage_prepare(ctx, perm, n_pad_s, em, need_m, ta, true));
        if (ta.sym.half_steps <= 0) return GT_OK;   // stage one is no filter on these points: the classic pass
        const int64_t slots = int64_t(ctx->n_cu) * 3;
        int best = 1;
        double best_cost = 1e30;
        for (int sgm = 1; sgm <= 8; ++sgm) {
            const double cost = double(ceil_div64(int64_t(ta.sym.nblk) * sgm, slots)) / sgm + 0.01 * sgm;
            if (cost < best_cost - 1e-9) best_cost = cost, best = sgm;
        }
        ta.sym.nseg = ctx->sym_nseg > 0 ? std::min(ctx->sym_nseg, 8) : best;
    }
    GT_TRY(gt_sym_inject_orphans(ctx, p0, std::min<int64_t>(p1, ctx->n), k->thr_final.as<float>(), lists0, 64,
                                 k->counts.as<uint32_t>(), k->tlists.as<uint64_t>(), tcap, k->tcounts.as<uint32_t>()));
    if (!bound_done) {
        GT_TRY(gt_sym_queue_prepare(ctx, n_pad_s, ta));
        {
            StageSpan span(ctx, "knn_select");
            GT_TRY(gt_sym_launch_collect(ctx, ta));
        }
        int ok = 0;
        GT_TRY(gt_sym_queue_finish(ctx, ta, &k->sym_cold_entries, &ok));
        if (!ok) {   // the queue overflowed: the lists are incomplete - the classic pass for these rows
            ctx->sym_two_ok = 0;
            return GT_OK;
        }
        if (ctx->sym_two_ok < 0) ctx->sym_two_ok = 1;
        k->sym_cold_local_used = false;
    } else {
        StageSpan span(ctx, "sym_cold");
        SelectArgs dq;
        dq.dp = ctx->DP;
        dq.prec = 2;
        dq.mode = 4;
        dq.nt = 8;
        dq.Yp = dq.Qp = k->Ycs.as<float>();
        dq.hneg = k->hnegs.as<float>();
        dq.n_pad = n_pad_s;
        dq.q0 = 0;
        dq.nq = int32_t(ctx->n);
        dq.thr_in = k->thr_final.as<float>();
        dq.sym.g = k->sym_g.as<float>();   // (not read: own_only)
        dq.sym.tlists = k->tlists.as<uint64_t>();
        dq.sym.tcounts = k->tcounts.as<uint32_t>();
        dq.sym.tcap = tcap;
        dq.sym.queue = k->sym_qdense.as<uint2>();
        dq.sym.qn = int32_t(left);
        dq.sym.own_only = 1;
        k->sym_cold_local_used = false;
        if (ctx->sym_cold_local != 0 && ctx->dtype == GT_F32 && (ctx->d & 3) == 0 && ctx->d <= ctx->DP && ctx->DP <= 64) {
            // the units in the frame of their queries (gt_knn_select.hip sym_cold_local_kernel), as on one GPU: the renumbered
            // points ARE the sorted float32 copy; centres for the rank's own query groups only
            GT_HIP(ctx, k->sym_rloc.reserve(size_t(n_pad_s) * sizeof(float)));
            GT_HIP(ctx, k->sym_gcen.reserve(size_t(n_pad_s / 64) * ctx->DP * sizeof(float)));
            GT_TRY(gt_sym_row_radius(ctx, perm, n_pad_s, k->thr_final.as<float>(), em, k->sym_rloc.as<float>(), 0.0));
            dq.sym.xs = reinterpret_cast<const float*>(ctx->X);
            dq.sym.xs_d = ctx->d;
            dq.sym.xs_n = int32_t(ctx->n);
            dq.sym.gcen = k->sym_gcen.as<float>();
            dq.sym.rloc = k->sym_rloc.as<float>();
            dq.sym.sc = float(ctx->sc);
            dq.sym.gc_first = int32_t(p0 / 64);
            dq.sym.gc_count = int32_t((p1 - p0) / 64);
            k->sym_cold_local_used = true;
        }
        GT_TRY(gt_launch_select(ctx, dq));
        k->sym_cold_entries = int64_t(left);
    }
    k->sym_bound_used = bound_done;
    k->sym_two_used = true;
    k->sym_nseg = bound_done ? 1 : ta.sym.nseg;
    ctx->last_main_prec = 2;
    k->sh_world = 1;
    k->sh_rank = 0;
    k->sh_r0 = r0;
    k->sh_nloc = r1 - r0;
    k->sh_n_pad_s = n_pad_s;
    k->sh_p0 = p0;
    k->sh_p1 = p1;
    k->sh_need = need_m;
    k->sh_rkf = rkf;
    if (ctx->sym_ok < 0) ctx->sym_ok = 1;
    k->sh_stage = 6;   // the lists of the sorted positions [r0, r1) wait in tlists (gt_knn_candidates)
    *applies = 1;
    return GT_OK;
}

// row splits of a sharded build on renumbered points: runs of whole 1024-row blocks, as even as they come
extern "C" int gt_points_shard_splits(gt_ctx* ctx, int32_t world, int64_t* out_splits) {
    if (!ctx || !out_splits || world < 1) return GT_E_ARG;
    if (ctx->n <= 0) GT_FAIL(ctx, GT_E_STATE, "gt_points_shard_splits: no points bound");
    const int64_t nb = ceil_div64(ctx->n, 1024);
    if (nb >= int64_t(2) * world) {
        for (int r = 0; r <= world; ++r) out_splits[r] = std::min<int64_t>(ctx->n, (nb * r / world) * 1024);
    } else {
        // a handful of blocks: even runs of rows (gt_graph_shard_local declines, the classic pass serves such sets)
        for (int r = 0; r <= world; ++r) out_splits[r] = ctx->n * r / world;
    }
    out_splits[world] = ctx->n;
    return GT_OK;
}

// device address of row `row0` of the bound points as the context holds them (its own numbering, the caller's dtype;
// cosine: the normalised rows) - for device-to-device hand-offs (e.g. a rank's rows as the query matrix of another context)
extern "C" int gt_points_device(gt_ctx* ctx, int64_t row0, void** out, int32_t* out_dtype, int32_t* out_d) {
    if (!ctx || !out) return GT_E_ARG;
    if (ctx->n <= 0 || !ctx->X) GT_FAIL(ctx, GT_E_STATE, "gt_points_device: no points bound");
    if (row0 < 0 || row0 >= ctx->n) GT_FAIL(ctx, GT_E_ARG, "gt_points_device: row out of range");
    const size_t esz = ctx->dtype == GT_F32 ? 4 : 8;
    *out = const_cast<char*>(static_cast<const char*>(ctx->X)) + size_t(row0) * size_t(ctx->d) * esz;
    if (out_dtype) *out_dtype = ctx->dtype;
    if (out_d) *out_d = ctx->d;
    return GT_OK;
}

int gt_knn_shard_plan(gt_ctx* ctx, int world, int rank, const int64_t* splits, int need_m, double rkf, int32_t* applies,
                      int64_t* n_pad_sorted, int64_t* sorted_splits) {
    *applies = 0;
    if (ctx->n <= 0 || !ctx->X) GT_FAIL(ctx, GT_E_STATE, "no points bound (call gt_set_points first)");
    if (world < 1 || world > GT_SYM_MAX_WORLD || rank < 0 || rank >= world || !splits)
        GT_FAIL(ctx, GT_E_ARG, "sym shard: bad world/rank/row_splits");
    if (!ctx->knn) ctx->knn = new KnnWork();
    KnnWork* k = ctx->knn;
    k->sh_stage = 0;
    if (!shard_applicable(ctx, need_m) || splits[rank + 1] <= splits[rank]) return GT_OK;
    const int bq = gt_select_bq(ctx->DP);
    // (the two-stage collect kernel works on query blocks of up to 1024 rows; the seeding shares stay 256-row blocks.
    //  The padding depends on the OPTIONS only, never on what earlier builds of this rank found out about the points:
    //  every rank must arrive at the same n_pad_s and the same sorted splits)
    const bool two = ctx->DP >= 32 && bq == 256 && ctx->sym_two_stage != 0;
    const int64_t pad_s = two ? 1024 : bq;
    const int64_t n_pad_s = ceil_div64(ctx->n, pad_s) * pad_s;
    const int64_t NB = n_pad_s / bq;
    if (NB < world) return GT_OK;
    // the cell-sorted order of ALL rows (position -> row in k->qorder)
    int ordered = 0;
    GT_HIP(ctx, k->qorder.reserve(size_t(ctx->n) * sizeof(int32_t)));
    k->xs_ready = false;
    GT_HIP(ctx, k->qthr0.reserve(size_t(ctx->n) * sizeof(float)));
    {
        StageSpan span(ctx, "query_order");
        GT_TRY(gt_query_order(ctx, ctx->Yc.as<float>(), 0, ctx->n, need_m, k->qorder.as<int32_t>(), k->qthr0.as<float>(),
                              &ordered));
    }
    k->ordered = false;
    if (!ordered || ctx->order_L <= 0) return GT_OK;
    k->sh_world = world;
    k->sh_rank = rank;
    for (int r = 0; r <= world; ++r) k->sh_splits[r] = splits[r];
    k->sh_r0 = splits[rank];
    k->sh_nloc = splits[rank + 1] - splits[rank];
    k->sh_n_pad_s = n_pad_s;
    k->sh_need = need_m;
    k->sh_rkf = rkf;
    for (int r = 0; r <= world; ++r) sorted_splits[r] = (NB * r / world) * bq;
    k->sh_p0 = sorted_splits[rank];
    k->sh_p1 = sorted_splits[rank + 1];
    *n_pad_sorted = n_pad_s;
    *applies = 1;
    k->sh_stage = 1;
    return GT_OK;
}

int gt_knn_shard_seed(gt_ctx* ctx, float* thr_local, int64_t* far_local, double* racc_local) {
    KnnWork* k = ctx->knn;
    if (!k || k->sh_stage != 1) GT_FAIL(ctx, GT_E_STATE, "sym shard: seed without a plan");
    k->sh_stage = 0;
    const int bq = gt_select_bq(ctx->DP), bn = gt_select_bn(ctx->DP);
    const int64_t n_pad_s = k->sh_n_pad_s, p0 = k->sh_p0, p1 = k->sh_p1;
    const int need_m = k->sh_need;
    const int tcap = ctx->sym_tcap;
    const size_t lcap = size_t(64) * 8;
    const int32_t* perm = k->qorder.as<int32_t>();
    GT_HIP(ctx, k->Ycs.reserve(size_t(n_pad_s) * ctx->DP * sizeof(_Float16)));
    GT_HIP(ctx, k->hnegs.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_g.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_gmin.reserve(size_t(n_pad_s / 32) * sizeof(float)));
    GT_HIP(ctx, k->tlists.reserve(size_t(n_pad_s) * size_t(tcap) * sizeof(uint64_t)));
    GT_HIP(ctx, k->tcounts.reserve(size_t(n_pad_s) * sizeof(uint32_t)));
    GT_HIP(ctx, k->sym_stat.reserve(8 * sizeof(unsigned long long)));
    GT_HIP(ctx, k->lists.reserve(size_t(std::max<int64_t>(p1 - p0, bq)) * lcap * sizeof(uint64_t)));   // own blocks only
    GT_HIP(ctx, k->counts.reserve(size_t(n_pad_s) * sizeof(uint32_t)));
    GT_HIP(ctx, k->thr_final.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_farcnt.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, hipMemsetAsync(k->sym_farcnt.p, 0, size_t(n_pad_s) * sizeof(float), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(k->sym_stat.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    const int n_tiles_s = int(n_pad_s / bn);
    const int stride_a = ctx->sym_stride > 0 && n_tiles_s >= 8 * ctx->sym_stride ? ctx->sym_stride : 0;
    const int tile_stride = ((stride_a ? n_tiles_s / stride_a + 1 : 0) + ctx->sym_max_nb + bq / bn + 63) / 64 * 64;
    k->sh_stride = stride_a;
    k->sh_tile_stride = tile_stride;
    GT_HIP(ctx, k->sym_tiles.reserve(size_t(n_pad_s / bq) * tile_stride * sizeof(int32_t)));
    GT_HIP(ctx, k->sym_tile_cnt.reserve(size_t(n_pad_s / bq) * sizeof(int32_t)));
    {
        StageSpan span(ctx, "sym_prepare");
        GT_HIP(ctx, k->hnegs_fin.reserve(size_t(n_pad_s) * sizeof(float)));
        GT_TRY(gt_sym_gather(ctx, perm, n_pad_s, k->Ycs.p, k->hnegs.as<float>(), k->hnegs_fin.as<float>()));
        if (ctx->sym_sorted_points != 0) GT_TRY(gt_sym_gather_points(ctx, perm));
        GT_TRY(gt_sym_schedule(ctx, n_pad_s, bq, bn, shard_cells(ctx), stride_a, ctx->sym_max_nb, tile_stride, k->sym_work,
                               k->sym_tiles.as<int32_t>(), k->sym_tile_cnt.as<int32_t>(),
                               k->sym_stat.as<unsigned long long>() + 5));   // (tiles of ALL blocks; this rank walks 1/world)
    }
    // the lists of launch A are addressed by sorted position: a base pointer p0 rows before the buffer keeps the
    // kernels' indexing (only positions [p0, p1) are touched)
    // dense cell blocks with the keys in registers (gt_seed.hip) as in the single-rank pass, or the streaming lists; the dense
    // kernel files exactly need_m keys per row at a stride of 64
    const bool dense_seed = ctx->sym_dense_seed != 0 && need_m <= 64 && tile_stride <= 1024 && n_pad_s % 256 == 0 &&
                            p0 % 128 == 0 && p1 % 128 == 0;
    const size_t lstride = dense_seed ? size_t(64) : lcap;
    k->sh_lstride = int(lstride);
    uint64_t* lists0 = k->lists.as<uint64_t>() - size_t(p0) * lstride;
    ErrModel em = gt_err_model(ctx, 2);
    em.rel += 8.0 * 5.9604644775390625e-08;   // as in the single-rank pass (gt_knn.cpp)
    if (p1 > p0) {
        SelectArgs a;
        a.dp = ctx->DP;
        a.prec = 2;
        a.mode = 0;
        a.nt = 8;
        a.narrow = 0;
        a.Yp = a.Qp = k->Ycs.as<float>();
        a.hneg = k->hnegs.as<float>();
        a.n_pad = n_pad_s;
        a.qrows = nullptr;
        a.q0 = 0;
        a.nq = int32_t(ctx->n);
        a.lists = lists0;
        a.counts = k->counts.as<uint32_t>();
        a.thr_in = nullptr;
        a.thr_out = k->thr_final.as<float>();
        a.dbg = 0;
        a.sym.sched = 1;
        a.sym.tile_list = k->sym_tiles.as<int32_t>();
        a.sym.tile_cnt = k->sym_tile_cnt.as<int32_t>();
        a.sym.tile_stride = tile_stride;
        a.sym.block0 = int32_t(p0 / bq);
        a.sym.nblk = int32_t((p1 - p0) / bq);
        if (ctx->DP <= 64 && bq == 256 && ctx->narrow_mode != 0) {
            // a rank's share is a fraction of a round of 256-row workgroups and every workgroup walks its whole tile
            // list: 128-row workgroups (one query tile per wave) double the waves that hide each other's latencies
            a.narrow = 1;
            a.sym.list_shift = 1;
            a.sym.block0 *= 2;
            a.sym.nblk *= 2;
        }
        int keep = std::max(ctx->samp_keep > 0 ? ctx->samp_keep : 16, need_m);
        keep += keep & 1;
        a.samp_stride = 0;
        a.samp_keep = keep;
        a.samp_trig = ctx->samp_trig > 0 ? ctx->samp_trig : 96;   // (lists of the neighbourhood tiles: compact at 96 keys, 7.5 -> 7.1 ms)
        a.samp_end = 0;
        a.samp2_level = 0;
        a.final_keep = need_m;
        {
            StageSpan span(ctx, "sym_seed");
            if (dense_seed)
                GT_TRY(gt_sym_seed_dense(ctx, ctx->DP, k->Ycs.p, k->hnegs_fin.as<float>(), ctx->n, n_pad_s, k->sym_tiles.as<int32_t>(),
                                         k->sym_tile_cnt.as<int32_t>(), tile_stride, bq, p0 / 128, (p1 - p0) / 128, need_m, lists0,
                                         int(lstride), k->counts.as<uint32_t>()));
            else
                GT_TRY(gt_launch_select(ctx, a));
        }
        k->sym_seed_dense = dense_seed;
        StageSpan span(ctx, "sym_prepare");
        GT_TRY(gt_sym_thresholds(ctx, perm, n_pad_s, k->hnegs.as<float>(), lists0, int(lstride), k->counts.as<uint32_t>(), need_m,
                                 em, std::max(1.0, std::fabs(k->sh_rkf)), k->thr_final.as<float>(), k->sym_g.as<float>(),
                                 nullptr, k->sym_work, shard_cells(ctx), k->sym_stat.as<unsigned long long>() + 2,
                                 k->sym_farcnt.as<float>(), p0, p1));
        // {threshold, far-kept seeds} of every position of the share, interleaved, to the caller
        GT_HIP(ctx, hipMemcpy2DAsync(thr_local, 2 * sizeof(float), k->thr_final.as<float>() + p0, sizeof(float), sizeof(float),
                                     size_t(p1 - p0), hipMemcpyDeviceToDevice, ctx->stream));
        GT_HIP(ctx, hipMemcpy2DAsync(thr_local + 1, 2 * sizeof(float), k->sym_farcnt.as<float>() + p0, sizeof(float), sizeof(float),
                                     size_t(p1 - p0), hipMemcpyDeviceToDevice, ctx->stream));
    }
    // this rank's share of the radius statistics behind the orphan cut (summed over the ranks by the host)
    GT_HIP(ctx, k->sym_racc.reserve(4 * sizeof(double)));
    GT_HIP(ctx, hipMemsetAsync(k->sym_racc.p, 0, 4 * sizeof(double), ctx->stream));
    GT_TRY(gt_sym_radius_sum(ctx, perm, p0, p1, k->thr_final.as<float>(), em, k->sym_racc.as<double>()));
    GT_HIP(ctx, hipMemcpyAsync(racc_local, k->sym_racc.p, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long far = 0;
    GT_HIP(ctx, hipMemcpyAsync(&far, k->sym_stat.as<unsigned long long>() + 2, sizeof(far), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *far_local = int64_t(far);
    k->sh_stage = 2;
    return GT_OK;
}

int gt_knn_shard_collect(gt_ctx* ctx, const float* thr_all, int64_t far_total, const double* racc_total, int32_t* applies,
                         int64_t* send_counts) {
    KnnWork* k = ctx->knn;
    *applies = 0;
    if (!k || k->sh_stage != 2) GT_FAIL(ctx, GT_E_STATE, "sym shard: collect without seeds");
    k->sh_stage = 0;
    const int bq = gt_select_bq(ctx->DP);
    const int64_t n_pad_s = k->sh_n_pad_s;
    const int tcap = ctx->sym_tcap;
    k->sym_far = far_total;
    if (ctx->sym_mode < 0) {
        // the predictor of the single-rank pass on the far-kept count of ALL ranks (every rank sees the same number)
        const double est = double(k->sh_need) + double(std::max(k->sh_stride, 1)) * double(far_total) / double(ctx->n);
        if (k->sh_stride > 0 && est > double(tcap) / 8.0) {
            ctx->sym_ok = 0;
            return GT_OK;
        }
    }
    // thr_all: {threshold, far-kept seeds} of every sorted position, interleaved
    GT_HIP(ctx, hipMemcpy2DAsync(k->thr_final.p, sizeof(float), thr_all, 2 * sizeof(float), sizeof(float), size_t(n_pad_s),
                                 hipMemcpyDeviceToDevice, ctx->stream));
    GT_HIP(ctx, hipMemcpy2DAsync(k->sym_farcnt.p, sizeof(float), thr_all + 1, 2 * sizeof(float), sizeof(float), size_t(n_pad_s),
                                 hipMemcpyDeviceToDevice, ctx->stream));
    // the statistics of ALL rows behind the orphan cut (every rank holds the same sums, applies the same cut)
    GT_HIP(ctx, hipMemcpyAsync(k->sym_racc.p, racc_total, 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    // The orphans (gt_sym.hip sym_orphan_cut_kernel) are declared on every rank alike wherever the two-stage collect is
    // configured at all: which rows collect nothing must not differ between ranks (a row's list is the union of what the
    // ranks collected).  The same holds for the kernel of launch B (below): every verdict behind it is taken on
    // quantities all ranks hold alike.
    const bool cut = n_pad_s % 1024 == 0 && ctx->DP >= 32 && bq == 256 && ctx->sym_two_stage != 0;
    {
        StageSpan span(ctx, "sym_prepare");
        if (cut) {
            ErrModel emc = gt_err_model(ctx, 2);
            emc.rel += 8.0 * 5.9604644775390625e-08;
            GT_TRY(gt_sym_orphan_cut(ctx, k->qorder.as<int32_t>(), k->thr_final.as<float>(), k->sym_farcnt.as<float>(), emc,
                                     k->sym_racc.as<double>(), k->sh_need));
        }
        GT_TRY(gt_sym_g_from_thr(ctx, n_pad_s, k->thr_final.as<float>(), k->hnegs.as<float>(), k->sym_g.as<float>(),
                                 k->sym_gmin.as<float>()));
        GT_HIP(ctx, hipMemsetAsync(k->tcounts.p, 0, size_t(n_pad_s) * sizeof(uint32_t), ctx->stream));
        if (cut) {
            // the orphans among the rows THIS rank seeded start their lists with the rows launch A kept for them; they
            // travel to the owners with the other records
            const size_t lcap = size_t(k->sh_lstride);
            GT_TRY(gt_sym_inject_orphans(ctx, k->sh_p0, k->sh_p1, k->thr_final.as<float>(),
                                         k->lists.as<uint64_t>() - size_t(k->sh_p0) * lcap, int(lcap), k->counts.as<uint32_t>(),
                                         k->tlists.as<uint64_t>(), tcap, k->tcounts.as<uint32_t>()));
        }
    }
    SelectArgs a;
    a.dp = ctx->DP;
    a.prec = 2;
    a.mode = 2;
    a.nt = 8;
    a.Yp = a.Qp = k->Ycs.as<float>();
    a.hneg = k->hnegs.as<float>();
    a.n_pad = n_pad_s;
    a.q0 = 0;
    a.nq = int32_t(ctx->n);
    a.lists = nullptr;
    a.counts = k->counts.as<uint32_t>();
    a.thr_in = k->thr_final.as<float>();
    a.thr_out = nullptr;
    a.dbg = 0;
    a.sym.g = k->sym_g.as<float>();
    a.sym.gmin = k->sym_gmin.as<float>();
    a.sym.tlists = k->tlists.as<uint64_t>();
    a.sym.tcounts = k->tcounts.as<uint32_t>();
    a.sym.tcap = tcap;
    a.sym.shard_world = k->sh_world;
    a.sym.shard_rank = k->sh_rank;
    a.sym.shard_group = std::max(1, ctx->sym_shard_group);
    bool bound_done = false;
    k->sym_bound_used = false;
    if (n_pad_s % 1024 == 0 && ctx->DP >= 32 && bq == 256 &&
        (ctx->sym_two_stage > 0 || (ctx->sym_two_stage < 0 && ctx->sym_two_ok != 0))) {
        ErrModel em = gt_err_model(ctx, 2);
        em.rel += 8.0 * 5.9604644775390625e-08;
        if (ctx->sym_bounds != 0 && ctx->order_L > 0) {
            // bound pass, as in the single-rank pass (gt_knn.cpp): the units of THIS rank's pieces of the walks that the
            // cells of the sorted order cannot rule out go straight to the cold launch (the cell geometry is replicated
            // work, 0.7 ms; every rank decides for its own share - the lists are complete either way)
            const int64_t bcap = ctx->sym_bound_cap > 0 ? ctx->sym_bound_cap : (int64_t(1) << 22);
            GT_HIP(ctx, k->sym_qdense.reserve(size_t(bcap) * sizeof(uint2)));
            GT_HIP(ctx, k->sym_qtot.reserve(4 * sizeof(uint32_t)));
            GT_HIP(ctx, k->sym_rrow.reserve(size_t(n_pad_s) * sizeof(float)));
            uint32_t left = 0, left_all = 0;
            {
                StageSpan span(ctx, "sym_bound");
                GT_TRY(gt_sym_row_radius(ctx, k->qorder.as<int32_t>(), n_pad_s, k->thr_final.as<float>(), em, k->sym_rrow.as<float>()));
                GT_TRY(gt_sym_bound_queue(ctx, n_pad_s, k->Ycs.p, k->sym_rrow.as<float>(), k->sym_bwork, k->sym_qdense.as<uint2>(),
                                          uint32_t(bcap), k->sym_qtot.as<uint32_t>(), k->sh_world, k->sh_rank,
                                          std::max(1, ctx->sym_shard_group)));
                GT_HIP(ctx, hipMemcpyAsync(&left, k->sym_qtot.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                GT_HIP(ctx, hipMemcpyAsync(&left_all, k->sym_qtot.as<uint32_t>() + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            }
            GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            // Which kernel runs launch B must be the SAME on every rank: the bound pass / two-stage collect and the
            // one-stage kernel cut the pair space into different pieces (1024- against 256-row query blocks), a rank on
            // its own would leave pairs unscored.  So the verdict is taken on the units of ALL ranks' pieces, a number
            // every rank computes alike from the gathered thresholds - no collective needed; this rank's own units are
            // at most that many and fit its queue.
            if (k->sh_world == 1) left_all = left;
            if (int64_t(left_all) <= bcap) {
                StageSpan span(ctx, "sym_cold");
                SelectArgs dq = a;
                dq.mode = 4;
                dq.sym.queue = k->sym_qdense.as<uint2>();
                dq.sym.qn = int32_t(left);
                GT_TRY(gt_launch_select(ctx, dq));
                k->sym_cold_entries = int64_t(left);
                bound_done = true;
                k->sym_bound_used = true;
                if (ctx->sym_two_ok < 0) ctx->sym_two_ok = 1;
            }
        }
        // two-stage scoring, as in the single-rank pass (gt_knn.cpp); the forecast runs on the same data on every rank
        if (!bound_done) GT_TRY(gt_sym_two_stage_prepare(ctx, k->qorder.as<int32_t>(), n_pad_s, em, k->sh_need, a, true));
    }
    const bool two_stage = a.sym.half_steps > 0;
    k->sym_two_used = bound_done;
    {
        const int64_t slots = int64_t(ctx->n_cu) * 3, nb = n_pad_s / bq;
        int best = 1;
        double best_cost = 1e30;
        for (int sgm = 1; sgm <= 8; ++sgm) {
            // (a work item is 1/world as long as a single rank's: its fixed costs weigh more)
            const double cost = double(ceil_div64(nb * sgm, slots)) / sgm + 0.1 * sgm;
            if (cost < best_cost - 1e-9) best_cost = cost, best = sgm;
        }
        a.sym.nseg = ctx->sym_nseg > 0 ? std::min(ctx->sym_nseg, 8) : (two_stage ? std::max(1, best / 2) : best);
        k->sym_nseg = a.sym.nseg;
    }
    if (two_stage) GT_TRY(gt_sym_queue_prepare(ctx, n_pad_s, a));
    for (int attempt = 0; attempt < 2 && !bound_done; ++attempt) {
        {
            StageSpan span(ctx, "knn_select");
            GT_TRY(gt_sym_launch_collect(ctx, a));
        }
        if (a.sym.half_steps <= 0) break;
        // the deferred cold pass of the two-stage collect (gt_knn.cpp has the single-rank twin)
        int ok = 0;
        GT_TRY(gt_sym_queue_finish(ctx, a, &k->sym_cold_entries, &ok));
        if (ok) {
            if (ctx->sym_two_ok < 0) ctx->sym_two_ok = 1;
            k->sym_two_used = true;
            break;
        }
        ctx->sym_two_ok = 0;     // stage one is no filter on these points: the one-stage kernel, now and later
        a.sym.half_steps = 0;
        a.sym.nseg = k->sym_nseg = ctx->sym_nseg > 0 ? std::min(ctx->sym_nseg, 8) : 3;
    }
    ctx->last_main_prec = 2;
    GT_HIP(ctx, k->sh_cnt.reserve(size_t(2 * GT_SYM_MAX_WORLD) * sizeof(unsigned long long)));
    unsigned long long host_cnt[GT_SYM_MAX_WORLD];
    {
        StageSpan span(ctx, "sym_exchange");
        GT_TRY(gt_sym_shard_count(ctx, k->qorder.as<int32_t>(), k->tcounts.as<uint32_t>(), tcap, k->sh_world, k->sh_splits,
                                  k->sh_cnt.as<unsigned long long>()));
    }
    GT_HIP(ctx, hipMemcpyAsync(host_cnt, k->sh_cnt.p, size_t(k->sh_world) * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                               ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < k->sh_world; ++r) send_counts[r] = k->sh_send[r] = int64_t(host_cnt[r]);
    *applies = 1;
    k->sh_stage = 3;
    return GT_OK;
}

int gt_knn_shard_emit(gt_ctx* ctx, void* send_buf) {
    KnnWork* k = ctx->knn;
    if (!k || k->sh_stage != 3) GT_FAIL(ctx, GT_E_STATE, "sym shard: emit without collected lists");
    unsigned long long off[GT_SYM_MAX_WORLD];
    unsigned long long acc = 0;
    for (int r = 0; r < k->sh_world; ++r) {
        off[r] = acc;
        acc += (unsigned long long)k->sh_send[r];
    }
    unsigned long long* cursor = k->sh_cnt.as<unsigned long long>() + GT_SYM_MAX_WORLD;
    GT_HIP(ctx, hipMemcpyAsync(cursor, off, size_t(k->sh_world) * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
    if (acc > 0) {
        if (!send_buf) GT_FAIL(ctx, GT_E_ARG, "sym shard: send buffer is NULL");
        StageSpan span(ctx, "sym_exchange");
        GT_TRY(gt_sym_shard_emit(ctx, k->qorder.as<int32_t>(), k->tlists.as<uint64_t>(), k->tcounts.as<uint32_t>(), ctx->sym_tcap,
                                 k->sh_world, k->sh_splits, cursor, send_buf));
    }
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `off` is read by the copy; the caller's stream takes over the buffer
    k->sh_stage = 4;
    return GT_OK;
}

int gt_knn_shard_finish(gt_ctx* ctx, const void* recv, int64_t n_recv) {
    KnnWork* k = ctx->knn;
    if (!k || k->sh_stage != 4) GT_FAIL(ctx, GT_E_STATE, "sym shard: finish without an exchange");
    k->sh_stage = 0;
    if (n_recv < 0 || (n_recv > 0 && !recv)) GT_FAIL(ctx, GT_E_ARG, "sym shard: bad receive buffer");
    const int tcap = ctx->sym_tcap;
    GT_HIP(ctx, k->sh_lists.reserve(size_t(k->sh_nloc) * size_t(tcap) * sizeof(uint64_t)));
    GT_HIP(ctx, k->sh_counts.reserve(size_t(k->sh_nloc) * sizeof(uint32_t)));
    GT_HIP(ctx, k->sh_invperm.reserve(size_t(ctx->n) * sizeof(int32_t)));
    GT_HIP(ctx, k->sh_own.reserve(size_t(k->sh_nloc + 1) * sizeof(int32_t)));
    GT_HIP(ctx, k->unproven.reserve(sizeof(uint32_t)));
    GT_HIP(ctx, hipMemsetAsync(k->sh_counts.p, 0, size_t(k->sh_nloc) * sizeof(uint32_t), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(k->unproven.p, 0, sizeof(uint32_t), ctx->stream));
    uint32_t bad = 0;
    {
        StageSpan span(ctx, "sym_exchange");
        GT_TRY(gt_sym_shard_scatter(ctx, recv, n_recv, k->sh_nloc, tcap, k->sh_lists.as<uint64_t>(), k->sh_counts.as<uint32_t>(),
                                    k->unproven.as<uint32_t>()));
        GT_TRY(gt_sym_invperm(ctx, k->qorder.as<int32_t>(), k->sh_invperm.as<int32_t>()));
        GT_TRY(gt_sym_own_rows(ctx, k->qorder.as<int32_t>(), k->sh_r0, k->sh_r0 + k->sh_nloc, k->sh_own.as<int32_t>(), k->sh_tmp));
    }
    GT_HIP(ctx, hipMemcpyAsync(&bad, k->unproven.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bad) GT_FAIL(ctx, GT_E_ARG, "sym shard: received records for rows this rank does not own");
    k->sh_stage = 5;
    return GT_OK;
}
