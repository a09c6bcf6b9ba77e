// kNN driver state shared between gt_knn.hip (host orchestration), gt_rerank.hip (exact fp64 kernels)
// and gt_sparse.hip (affinity stage reads the candidate tables).
#pragma once
#include "gt_common.h"

// Exact candidate tables for a block of queries, produced by gt_knn_candidates():
//   cand_d2[q][p], cand_j[q][p]  p < MP : float64 squared distance (scikit-learn GEMM form) and database
//                                 index, ascending by (d2, j); unused slots hold +inf / 0xFFFFFFFF
//   cand_n[q]                     number of valid slots
//   d2_lb[q]                      completeness bound: EVERY database row with d2 < d2_lb[q] is in the table
struct KnnWork {
    int MP = 0;          // table width (128 or 512)
    int nt = 0;          // MP / 16
    int64_t nq = 0;      // queries in the tables
    int64_t nq_pad = 0;  // rounded up to the select kernel's query block
    int64_t q0 = 0;      // first query row (self queries) - queries are rows [q0, q0+nq) of the bound points
    bool external = false;
    // external queries (gt_knn_search with Y)
    DevBuf Qraw, Qp, Qc, qn;   // Qc: compact hi-plane copy of the query matrix (single-chain pass)
    DevBuf qn_sel;             // wide data: partial squared norms of the query matrix
    DevBuf lists, counts, thr_final;
    DevBuf cand_d2, cand_j, cand_n, d2_lb;
    DevBuf unproven, qlomax_dev;
    DevBuf qthr0;              //   and the starting thresholds that pass proves (single-chain arithmetic only)
    DevBuf qorder;             // self queries: the rows of the launch grouped by nearest landmark (gt_order.hip)
    bool ordered = false;      //   valid for the current tables (all rows of the bound points, position -> row)
    DevBuf fb_rows, fb_count, fb_scratch, gflags, prof;
    DevBuf fb_qrows, fb_thr, fb_lists, fb_counts, fb_max;   // collected fallback
    // symmetric candidate pass (gt_sym.hip): cell-sorted compact copy + seeds, per-row thresholds in the transposed
    // form and their sub-tile minima, the candidate lists of launch B and their counters
    DevBuf hnegs_fin;                             // seeds of the sorted rows, finite on the pad rows (dense seeding launch)
    DevBuf Ycs, hnegs, sym_g, sym_gmin, tlists, tcounts, sym_stat, sym_work, sym_tiles, sym_tile_cnt;
    DevBuf nokeyt_rows, nokeyt_count;             //   rows without them (handed to a repair pass), their number
    uint32_t nokeyt_n = 0;
    DevBuf cand_d2t, keyt_ok;                     //   keys of the transposed pairs next to cand_d2 (pair-resolved symmetrisation), row flags
    bool keyt_valid = false;                      //   ... written by the last candidate search for every row it proved
    DevBuf Xs, xns;                               //   the points and their squared norms in cell-sorted order (gt_sym_gather_points)
    bool xs_ready = false;                        //   ... valid for the current order (reset whenever the order is rebuilt)
    int xs_d = 0;                                 //   row stride of Xs in elements: d, or d rounded up to a multiple of 4 (float32
                                                  //   rows, zero padded: the four-lanes-per-row re-rank reads 16-byte quarters)
    DevBuf sym_racc, sym_farcnt;                     // radius / spread statistics of the orphan cut (4 doubles), far-kept
                                                  // seeds per sorted position
    DevBuf sym_hh, sym_thrh, sym_gh, sym_gminh;   // two-stage scoring (gt_sym.hip sym_half_*)
    DevBuf sym_z, sym_p, sym_cov;                 //   the stage-one copy Z = P x, its frame P, sample covariance scratch
    DevBuf sym_qspill;                            //   spill area of the queue (+ its counter)
    DevBuf sym_rrow, sym_bwork;                   //   bound pass: radius of every row in the stage-one copy, cell scratch
    DevBuf sym_zc, sym_zrn;                       //   balls of the groups of 32 rows in the stage-one space (unit skipping)
    DevBuf sym_wlist, sym_wcnt;                   //   listed walks of the one-stage collect (gt_sym_collect_lists) + their total
    bool sym_listed = false;                      //   the last symmetric pass ran the one-stage collect over listed walks
    int64_t sym_listed_tiles = 0;                 //   ... (256 query rows x 128 rows) tiles it scored
    // Tables laid out by SORTED POSITION (round 5): a whole single-rank '+' build that will take the pair-resolved tail asks for it
    // (want_tab_sorted, gt_sparse.hip); the symmetric re-rank then writes row perm[t]'s table - cand_d2 / cand_j / cand_d2t and
    // the per-table scalars cand_n / d2_lb / keyt_ok - at slot t, and tab_of_row (the inverse permutation) tells everybody who
    // comes by row where it is.  The affinity pass, the destination bins and the final merge walk the rows in sorted order anyway:
    // their table reads become streams, and the partners' bandwidths they gather are a shared, cache-resident set.
    bool want_tab_sorted = false, tab_sorted = false;
    bool want_keyt_classic = false; // the re-rank of the classic pass writes the transposed keys too (a '+' build that can take the pair-resolved tail)
    bool want_keyt_shard = false;   // the re-rank of a sharded rank's lists (sh_stage 5 / 6) writes the transposed keys too (gt_graph_bandwidth_local)
    DevBuf sym_rloc, sym_gcen;                    //   local-frame cold launch: the radius every row needs listed (scaled), the centres of
                                                  //   the 64-row query groups
    bool sym_cold_local_used = false;
    bool sym_bound_used = false;                  //   the last symmetric pass listed its units by cell bounds (no collect launch)
    DevBuf sym_queue, sym_qcount, sym_qdense, sym_qtot;   //   and the queue of its deferred cold pass (wave regions,
                                                          //   their counts, the compacted queue, {total, overflow})
    int64_t sym_cold_entries = 0;
    int64_t sym_units_scored = 0;                 //   (32 x 32) units stage one of the last two-stage collect scored (not skipped)
    int sym_frame = 0;                            //   frame of stage one: 0 coordinate axes, 1 principal directions
    bool sym_two_used = false;                    //   the last symmetric pass ran the two-stage collect
    bool sym_used = false;
    int sh_lstride = 512;                         //   stride of the seeding lists of the sharded pass (64: dense kernel)
    bool sym_seed_dense = false;                  //   the last seeding launch ran as dense cell blocks (gt_seed.hip)
    int64_t sym_overflow = 0;
    int sym_nseg = 1;
    int64_t sym_far = 0;
    unsigned long long sym_stat_host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t n_fallback_exhaustive = 0;
    int64_t n_fallback = 0;
    // row-sharded symmetric pass (gt_knn_shard.cpp): 0 idle, 1 planned, 2 seeded (own thresholds known), 3 collected
    // (send counts known), 4 emitted, 5 the candidate lists of the owned rows are in sh_lists - the next
    // gt_knn_candidates for exactly these rows re-ranks them instead of running a candidate pass
    int sh_stage = 0;
    int sh_world = 1, sh_rank = 0;
    int64_t sh_r0 = 0, sh_nloc = 0, sh_n_pad_s = 0, sh_p0 = 0, sh_p1 = 0;
    int sh_need = 0;
    double sh_rkf = 1.0;
    int sh_stride = 0, sh_tile_stride = 0;
    int64_t sh_splits[65] = {0};
    int64_t sh_send[64] = {0};
    DevBuf sh_invperm, sh_lists, sh_counts, sh_cnt, sh_own, sh_tmp;
};

int gt_select_bn_for(int dp);
int gt_select_columns(gt_ctx* ctx, int want);   // gt_prep.hip: wide data, columns of largest variance
int gt_prep_matrix(gt_ctx* ctx, const void* Xdev, int64_t n, int d, int dtype, int DP, int64_t n_pad, float* Yp,
                   double* xn, float* hneg, double* ymax2, int prec, double sc, double* lomax2 = nullptr, void* Yc = nullptr,
                   const int32_t* sel = nullptr, int dsel = 0, double* xn_sel = nullptr);
// max |x|; nonfinite (optional): bit 0 = NaN present, bit 1 = infinity present
int gt_max_abs(gt_ctx* ctx, const void* Xdev, int64_t total, int dtype, double* out_host, uint32_t* nonfinite);
// GT_E_NONFINITE with the message of sklearn's check_array when flags != 0
int gt_fail_nonfinite(gt_ctx* ctx, uint32_t flags, int dtype);
double gt_f16_scale(double maxabs);
// row-wise l2 normalisation in the input dtype (sklearn normalize: zero rows untouched); in place allowed
int gt_normalize_rows(gt_ctx* ctx, const void* X, void* out, int64_t n, int d, int dtype);

// Error model of the candidate scores (scaled units s~ = sc^2 (x.y - |y|^2/2) + error):
//   |s~/sc^2 - s| <= rel * (|y|^2/2 + |x||y|) + abs * (|x| + |y|)
// rel covers the accumulation (any summation order of the MFMA chain, with head-room) and, for the split-float16
// back end, the 2^-22 representation residual and the dropped lo.lo term; abs covers float16 underflow of lo parts.
// |score/sc^2 - (x.y - |y|^2/2)| <= rel (|y|^2/2 + |x||y|) + rel_dot |x||y| + abs (|x| + |y|) + cst
struct ErrModel {
    double rel;
    double rel_dot;
    double abs;
    double cst;
    double inv_sc2;
};
__host__ __device__ inline double gt_err_bound(const ErrModel& m, double x2, double y2) {
    const double xy = sqrt(x2 * y2);
    return m.rel * (0.5 * y2 + xy) + m.rel_dot * xy + m.abs * (sqrt(x2) + sqrt(y2)) + m.cst;
}
ErrModel gt_err_model(const gt_ctx* ctx, int prec);   // prec: arithmetic of the pass (0 f32, 1 split f16, 2 single f16)

// Build exact candidate tables with the first `need_m` entries of every row guaranteed to be the true
// need_m nearest neighbours.  Queries: rows [q0, q0+nq) of the bound points, or (external) the matrix in
// ctx->knn->Qraw prepared by the caller.
// `radius_key_factor` (optional): the caller will need every row within factor x key(need_m-th neighbour).  Used to
// judge whether the single-chain float16 main pass is adequate for this point set and to let the re-rank stop after its
// first batch of candidates (speed, never correctness).  Negative: the extent is not tied to the need_m-th neighbour
// (caller-given bandwidth) - the re-rank evaluates every candidate, |factor| serves the judgement.
int gt_knn_candidates(gt_ctx* ctx, int64_t q0, int64_t nq, bool external, int need_m, double radius_key_factor = 1.0);
// Upload / convert an external query matrix (same dtype and width as the bound points) into ctx->knn->Qraw
// (original dtype, normalised for the cosine metric), Qp (working copy) and qn (float64 squared norms).
int gt_prepare_queries(gt_ctx* ctx, const void* Y, int64_t m, int32_t y_on_device);

// ---- metric helpers shared by the float64 stages ---------------------------------------------------------
// The sortable non-negative "key" of a (query, database row) pair is the squared euclidean distance in
// scikit-learn's association (|x|^2 + (-2 x.y)) + |y|^2 clamped at 0 (sklearn:_argkmin.pyx.tp:497-505), or, for
// the cosine metric, clip(1 - xhat.yhat, 0, 2) (sklearn:metrics/pairwise.py:1169-1184) on the normalised points.
#ifdef __HIPCC__
__device__ __forceinline__ double gt_pair_key(double qn, double dot, double yn, int metric) {
    if (metric == 1) {
        double t = 1.0 - dot;
        t = t > 0.0 ? t : 0.0;
        return t < 2.0 ? t : 2.0;
    }
    double t = qn + (-2.0 * dot);
    t = t + yn;
    return t > 0.0 ? t : 0.0;
}
// distance as the reference sees it: _rdist_to_dist in the input dtype (euclidean), the input dtype's rounding
// of the float64 value (cosine)
__device__ __forceinline__ double gt_key_to_dist(double key, int dtype, int metric) {
    if (metric == 1) return (dtype == GT_F32) ? double(float(key)) : key;
    return (dtype == GT_F32) ? double(sqrtf(float(key))) : sqrt(key);
}
#endif

// gt_rerank.hip
struct RerankArgs {
    const void* X;          // database, original dtype, n x d
    int dtype;
    int64_t n;
    int d;
    const double* xn;       // database squared norms
    const void* Q;          // query matrix (original dtype); rows addressed as q0 + q
    const double* qn;       // squared norms of the query matrix rows (indexed like Q)
    const double* qn_sel;   // the norms the candidate pass saw (== qn unless wide data: partial norms), for the bounds
    int64_t q0;
    int64_t nq;
    const uint64_t* lists;
    int lstride;
    const uint32_t* counts;
    const float* thr_final;   // last admission threshold of the candidate pass (scaled score units)
    const double* ymax2;
    ErrModel err;
    int metric;
    int need_m;
    int MP;
    double* cand_d2;
    uint32_t* cand_j;
    uint32_t* cand_n;
    double* d2_lb;
    uint32_t* fb_count;
    int32_t* fb_rows;
    uint32_t* gflags;
    double radius_key_factor = 1.0;   // see gt_knn_candidates
    uint32_t* unproven = nullptr;     // optional counter: rows with key(need_m-th) * radius_key_factor >= bound
    const int32_t* qrows = nullptr;   // list i of the candidate pass belongs to row qrows[i] (else q0 + i)
    const int32_t* trow = nullptr;    // row -> slot of its table (KnnWork::tab_sorted: the repairs write where the re-rank did), else the row
    // optional, together (tables of 256 slots): the transposed keys next to the tables, as the symmetric re-rank writes them
    double* cand_d2t = nullptr;
    uint8_t* keyt_ok = nullptr;
    int32_t* nokeyt_rows = nullptr;
    uint32_t* nokeyt_count = nullptr;
    bool* wrote_t = nullptr;          // (out) the launch wrote them
};
int gt_launch_rerank(gt_ctx* ctx, const RerankArgs& a);
// symmetric candidate pass (gt_sym.hip): segments of list ql = cell-sorted position ql, row perm[ql]
struct SymRerank {
    const uint64_t* tlists;    // [n_pad][tcap]
    const uint32_t* tcounts;
    int tcap;
    const int32_t* perm;       // sorted position -> row
    unsigned long long* stat;  // optional counters [8] (rerank_sym_kernel)
    // row-sharded builds: the lists belong to the owned rows own_r0 + ql (tables indexed by ql), thresholds are found
    // through the inverse permutation (row -> sorted position)
    const int32_t* invperm = nullptr;
    const int32_t* own_rows = nullptr;   // the owned rows in the order of their sorted positions
    int64_t own_r0 = 0;
    // lists of a RANGE of sorted positions (no invperm): list / threshold ql + pos0, table row perm[ql + pos0] - own_r0
    int64_t pos0 = 0;
    // the points / norms in sorted order (optional): candidate rows are then read by position
    const void* Xs = nullptr;
    const double* xns = nullptr;
    int xs_d = 0;              // row stride of Xs in elements (0: the points' own d)
    // transposed keys next to the table's own (optional; rerank_sym4_kernel only): [nq][256] and one flag per row
    double* cand_d2t = nullptr;
    uint8_t* keyt_ok = nullptr;
    int32_t* nokeyt_rows = nullptr;    // rows whose table will come from a repair pass (no transposed keys), and their number
    uint32_t* nokeyt_count = nullptr;
    bool* wrote_t = nullptr;   // out: the launch that ran fills them
    bool tab_sorted = false;   // rerank_sym4_kernel with the transposed keys only: the table of list ql goes to slot ql (not to its row's)
};
int gt_launch_rerank_sym(gt_ctx* ctx, const RerankArgs& a, const SymRerank& sr);
// gt_sym.hip
// hs_fin (optional): the same seeds with -3e38 instead of -inf on the pad rows (gt_seed.hip)
int gt_sym_gather(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, void* Ys, float* hs, float* hs_fin = nullptr);
// the caller's points (all ctx->n rows, their dtype) and norms in sorted order -> KnnWork::Xs / xns, xs_ready
// pad4: float32 rows whose length is no multiple of 4 are zero padded to one (KnnWork::xs_d) - the zeros add nothing to any
// dot product, and the copy serves the 16-byte loads of rerank_sym4_kernel (d = 50: C2, C5)
int gt_sym_gather_points(gt_ctx* ctx, const int32_t* perm, bool pad4 = false);
// rows [p_first, p_last) of the sorted order only (p_last < 0: all); gmin = nullptr: sub-tile minima not formed
int gt_sym_thresholds(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const float* hs, const uint64_t* lists, int lstride,
                      const uint32_t* counts, int need_m, const ErrModel& err, double rkf, float* thr, float* g, float* gmin,
                      const DevBuf& work, int cells, unsigned long long* far_total, float* farcnt = nullptr,
                      int64_t p_first = 0, int64_t p_last = -1, bool sample = false);
// two-stage scoring of launch B: half seeds of the sorted rows, partial-distance thresholds from the full ones
// stage-one copy of the two-stage collect: Z [n_pad][16] float16 = scz x P x (P_dev [16][64], orthonormal rows) in sorted
// order, hh = its half seeds; sample covariance for the principal frame
int gt_sym_project(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const float* P_dev, double scz, void* Z, float* hh);
int gt_sym_sample_cov(gt_ctx* ctx, int64_t step, int64_t ns, double* sums_dev, double* C_dev);
// forecast of stage one: of `samples` pseudo-random (64 queries, 32 rows) pairs, how many pass (flagged: device counter)
int gt_sym_two_probe(gt_ctx* ctx, const void* Ys, int hd, const float* hh, const float* thrh, const float* gh, int64_t samples,
                     uint32_t* flagged);
struct SelectArgs;
// launch B: the one-stage kernel on the full copy, or (sym.half_steps > 0) the two-stage unit loop on the stage-one copy
int gt_sym_launch_collect(gt_ctx* ctx, const SelectArgs& a);
// queue of the two-stage collect launch `a` (mode 2 with sym.half_steps): sizes and binds the wave regions
int gt_sym_queue_prepare(gt_ctx* ctx, int64_t n_pad_s, SelectArgs& a);
// half seeds + partial-distance thresholds + forecast: binds them to `a` (sym.half_steps > 0) when stage one is expected to
// pass few enough pairs for the queue, leaves `a` a one-stage launch otherwise
// (declares the orphans first - thr, g, gmin are updated - when it goes ahead)
// cut_done: the orphans were already declared (row-sharded builds do it on every rank alike, whatever each rank decides here)
int gt_sym_two_stage_prepare(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const ErrModel& em, int need_m, SelectArgs& a,
                             bool cut_done = false);
// after that launch: compacts the queue and runs the cold pass; *entries = pairs scored, *ok = 0 when the queue
// overflowed (nothing was filed: the caller runs the one-stage kernel instead)
int gt_sym_queue_finish(gt_ctx* ctx, const SelectArgs& a, int64_t* entries, int* ok);
int gt_sym_half_thresholds(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const float* thr, const float* hh,
                           const ErrModel& err, int hd, double scz, double Lz, float* thrh, float* gh, float* gminh,
                           float* rrow = nullptr);
// radius of every sorted row in the full compact copy (what a pair it needs listed can be apart at most)
int gt_sym_row_radius(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const float* thr, const ErrModel& err, float* rrow,
                      double lres = -1.0);
// bound pass of the two-stage collect (gt_sym.hip cell_ball_kernel): the units the cell bounds cannot rule out -> queue
// (world / rank / group: a row-sharded build lists the units of its own pieces of the walks)
// own_p1 > own_p0: the query groups of the sorted positions [own_p0, own_p1) against EVERY sub-tile (no walks; own_p0 a
// multiple of 64) - the queue of a cold launch that files under the queries only (SymDev::own_only)
int gt_sym_bound_queue(gt_ctx* ctx, int64_t n_pad_s, const void* Ys, const float* rrow, DevBuf& work, uint2* queue,
                       uint32_t cap, uint32_t* count_dev, int world = 1, int rank = 0, int group = 1, int64_t own_p0 = 0,
                       int64_t own_p1 = 0);
// balls of the groups of 32 sorted rows in the stage-one space: the unit skipping of the two-stage collect (gt_sym.hip)
int gt_sym_z_balls(gt_ctx* ctx, int64_t n_pad_s, const void* Z, const float* gh, double dmargin, float* zc, float* zrn);
// listed walks of the one-stage collect (gt_sym.hip collect_lists_kernel) from the cell masks gt_sym_bound_queue left in `work`
int gt_sym_collect_lists(gt_ctx* ctx, int64_t n_pad_s, DevBuf& work, int cap, int stride, int32_t* tile_list, int32_t* tile_cnt,
                         unsigned long long* total_dev, int* walk_out);
// row-sharded symmetric pass (gt_knn_shard.cpp)
#define GT_SYM_MAX_WORLD 64
int gt_sym_g_from_thr(gt_ctx* ctx, int64_t n_pad_s, const float* thr, const float* hs, float* g, float* gmin);
// statistics of the thresholds' radii and of the point set's spread over the rows [p_first, p_last), added into acc
// (device, 4 doubles: gt_sym.hip sym_radius_sum_kernel); the orphans of the two-stage collect - by their far-kept seeds
// (farcnt, from gt_sym_thresholds) or by a radius that is an outlier against those statistics - lose their threshold
// (+inf: repaired directly)
int gt_sym_radius_sum(gt_ctx* ctx, const int32_t* perm, int64_t p_first, int64_t p_last, const float* thr, const ErrModel& err,
                      double* acc);
int gt_sym_orphan_cut(gt_ctx* ctx, const int32_t* perm, float* thr, const float* farcnt, const ErrModel& err, const double* acc,
                      int need_m, double pair_frac = 0.0625);
// orphans of the sorted positions [p_first, p_last) (thr = +inf on a real row): the rows launch A kept -> head of tlists
int gt_sym_inject_orphans(gt_ctx* ctx, int64_t p_first, int64_t p_last, const float* thr, const uint64_t* lists, int lstride,
                          const uint32_t* counts, uint64_t* tlists, int tcap, uint32_t* tcounts);
int gt_sym_invperm(gt_ctx* ctx, const int32_t* perm, int32_t* inv);
// the rows [r0, r1) in the order they have in perm -> own (device int32 [r1 - r0]); tmp: scratch
int gt_sym_own_rows(gt_ctx* ctx, const int32_t* perm, int64_t r0, int64_t r1, int32_t* own, DevBuf& tmp);
int gt_sym_shard_count(gt_ctx* ctx, const int32_t* perm, const uint32_t* tcounts, int tcap, int world, const int64_t* splits,
                       unsigned long long* cnt);
int gt_sym_shard_emit(gt_ctx* ctx, const int32_t* perm, const uint64_t* tlists, const uint32_t* tcounts, int tcap, int world,
                      const int64_t* splits, unsigned long long* cursor, void* out);
int gt_sym_shard_scatter(gt_ctx* ctx, const void* recs, int64_t n_recs, int64_t nloc, int tcap, uint64_t* lists,
                         uint32_t* counts, uint32_t* bad);
// p_last > p_first: the tile lists of the query blocks of the sorted positions [p_first, p_last) only (a rank's own rows)
int gt_sym_schedule(gt_ctx* ctx, int64_t n_pad_s, int bq, int bn, int cells, int stride, int max_nb, int tile_stride,
                    DevBuf& work, int32_t* tile_list, int32_t* tile_cnt, unsigned long long* tiles_total = nullptr,
                    int64_t p_first = 0, int64_t p_last = 0);
// gt_seed.hip: the threshold-seeding launch as dense cell blocks (`need` distinct near rows per sorted position)
int gt_sym_seed_dense(gt_ctx* ctx, int dp, const void* Ys, const float* hs, int64_t n, int64_t n_pad, const int32_t* tile_list,
                      const int32_t* tile_cnt, int tile_stride, int list_rows, int64_t block0, int64_t nblk, int need,
                      uint64_t* lists, int lstride, uint32_t* counts);
int gt_launch_fallback(gt_ctx* ctx, const RerankArgs& a, int64_t n_rows, int64_t row_off, double* scratch);
int gt_launch_fallback_thr(gt_ctx* ctx, const RerankArgs& a, int64_t n_rows, int64_t row_off, int32_t* qrows, float* thr);
int gt_launch_collected_select(gt_ctx* ctx, const RerankArgs& a, int64_t n_rows, int64_t row_off, const uint64_t* clists,
                               const uint32_t* ccounts, int cap, double* scratch, uint32_t* fail);
int gt_launch_max_u32(gt_ctx* ctx, const uint32_t* v, int64_t n, uint32_t* out);   // gt_sparse.hip (out pre-zeroed)
int gt_launch_emit_knn(gt_ctx* ctx, const double* cand_d2, const uint32_t* cand_j, int MP, int64_t nq, int k,
                       int dtype, int metric, int64_t* out_idx, double* out_dist);
