// Host orchestration of the kNN search: candidate pass (MFMA) -> exact re-rank (fp64) -> exhaustive
// fallback for the rows whose candidate table could not be proven complete.
#include "gt_knn.h"
#include "gt_graph_state.h"
#include "gt_hostcopy.h"
#include "gt_knn_select.h"

#include <algorithm>
#include <vector>
#include <cstdio>

int gt_select_bn_for(int dp) { return gt_select_bn(dp); }

void gt_free_knn_work(gt_ctx* ctx) {
    if (!ctx->knn) return;
    KnnWork* k = ctx->knn;
    for (DevBuf* b : {&k->Qraw, &k->Qp, &k->Qc, &k->qn, &k->qn_sel, &k->lists, &k->counts, &k->thr_final, &k->cand_d2, &k->cand_j, &k->cand_n,
                      &k->d2_lb, &k->fb_rows, &k->fb_count, &k->fb_scratch, &k->gflags, &k->prof, &k->fb_qrows, &k->fb_thr, &k->fb_lists,
                      &k->fb_counts, &k->fb_max, &k->unproven, &k->qorder, &k->qthr0, &k->qlomax_dev, &k->Ycs, &k->hnegs,
                      &k->sym_g, &k->sym_gmin, &k->tlists, &k->tcounts, &k->sym_stat, &k->sym_work, &k->sym_tiles,
                      &k->sym_tile_cnt, &k->sh_invperm, &k->sh_lists, &k->sh_counts, &k->sh_cnt, &k->sh_own, &k->sh_tmp, &k->sym_hh, &k->sym_thrh, &k->sym_gh, &k->sym_gminh, &k->sym_queue, &k->sym_qcount, &k->sym_qdense, &k->sym_qtot, &k->sym_racc, &k->sym_farcnt, &k->sym_z, &k->sym_p, &k->sym_cov, &k->sym_qspill, &k->sym_rrow, &k->sym_bwork, &k->sym_rloc, &k->sym_gcen, &k->hnegs_fin, &k->Xs, &k->xns, &k->cand_d2t, &k->keyt_ok, &k->nokeyt_rows, &k->nokeyt_count})
        b->release();
    delete k;
    ctx->knn = nullptr;
}

ErrModel gt_err_model(const gt_ctx* ctx, int prec) {
    const double u = 5.9604644775390625e-08;  // 2^-24
    ErrModel m;
    m.rel_dot = 0.0;
    m.cst = 0.0;
    if (prec == 1) {
        // 3*DP + 1 summands in fp32 (factor 2 head-room, also covers truncating alignment inside the MFMA),
        // + 3 * 2^-22 for the two residuals and the dropped lo.lo products (x1.01 for second-order terms)
        m.rel = 2.0 * double(3 * ctx->DP + 4) * u + 1.01 * 3.0 * 4.0 * u;
        // float16 underflow of a lo part: absolute 2^-25 per element in scaled units
        m.abs = std::sqrt(double(ctx->DP)) * 2.0 * u / ctx->sc;
    } else if (prec == 2) {
        // hi planes only: with x sc = xh + xl (xl the exact float16 rounding residual), the chain computes xh.yh and
        // drops xh.yl + xl.yh + xl.yl, bounded by |x| Ly + Lx |y| + Lx Ly with L = max residual row norm (measured on
        // the bound points and the query matrix, gt_prep.hip) - far tighter than the worst case 2^-10 |x||y|;
        // on top of it the fp32 accumulation of DP + 1 summands
        const double L = std::max(ctx->lomax, ctx->qlomax);
        m.rel = 2.0 * double(ctx->DP + 4) * u;
        m.abs = 1.001 * L;
        m.cst = L * L;
    } else {
        // DP fused multiply-adds on top of the rounded seed; float64 inputs add 2u from the float32 conversion
        m.rel = 2.0 * double(ctx->DP + 6) * u;
        m.abs = 0.0;
    }
    m.inv_sc2 = 1.0 / (ctx->sc * ctx->sc);
    return m;
}

// Fraction of rows the single-chain pass may leave unproven before the split chains take over for this point set.
// A repaired row costs one radius-mode stream of the database (~0.4 us per row and 1e6 points), the split chains
// cost ~2.2x the single chain on every row: the break-even is near 45 % - stay well below it.
static const double kFastFailFrac = 0.20;

int gt_knn_candidates(gt_ctx* ctx, int64_t q0, int64_t nq, bool external, int need_m, double radius_key_factor) {
    if (ctx->n <= 0 || !ctx->X) GT_FAIL(ctx, GT_E_STATE, "no points bound (call gt_set_points first)");
    if (ctx->DP == 0)
        GT_FAIL(ctx, GT_E_LIMIT, "kNN on the HIP path: more than 2048 features are not supported (reduce with n_pca)");
    if (need_m < 1 || int64_t(need_m) > ctx->n) GT_FAIL(ctx, GT_E_ARG, "k must be in [1, n_samples]");
    if (nq <= 0) GT_FAIL(ctx, GT_E_ARG, "no query rows");
    if (!ctx->knn) ctx->knn = new KnnWork();
    KnnWork* k = ctx->knn;
    int nt;
    if (need_m <= ctx->nt8_max_need)
        nt = 8;
    else if (need_m <= 448)
        nt = 32;
    else
        GT_FAIL(ctx, GT_E_LIMIT, "k > 448 neighbours is not supported by the HIP path yet");
    // arithmetic of the main pass (see gt_common.h fast_mode): single-chain float16 when allowed and not yet refuted
    int main_prec = ctx->prec;
    const bool fast_auto = ctx->prec == 1 && ctx->fast_mode == 1;
    if (ctx->prec == 1 && (ctx->fast_mode == 2 || (fast_auto && ctx->fast_ok != 0 && nq >= 4096))) main_prec = 2;
    const int mkeep = 16 * nt;   // list budget of the streaming selection
    // exact table width M' = entries kept at the end of the stream.  The lists have room for 64*nt, and after the
    // threshold-seeding phase they typically end with ~16x the seed budget anyway: keeping 256 instead of 128 moves the
    // completeness bound out (it pays for the wider error bound of the single chain, and rows with more than 128
    // neighbours inside their radius stay out of the radius pass) for one more re-rank batch per row.
    const int MP = (nt == 8) ? 256 : mkeep;
    const int bq = gt_select_bq(ctx->DP);
    k->MP = MP;
    k->nt = nt;
    k->nq = nq;
    k->nq_pad = ceil_div64(nq, bq) * bq;
    k->q0 = q0;
    k->external = external;
    if (!external) ctx->qlomax = 0.0;   // the residual bound of a previous external query matrix does not apply
    k->n_fallback = 0;
    k->n_fallback_exhaustive = 0;
    k->keyt_valid = false;
    const size_t lcap = size_t(64) * nt;
    GT_HIP(ctx, k->lists.reserve(size_t(k->nq_pad) * lcap * sizeof(uint64_t)));
    GT_HIP(ctx, k->counts.reserve(size_t(k->nq_pad) * sizeof(uint32_t)));
    GT_HIP(ctx, k->thr_final.reserve(size_t(k->nq_pad) * sizeof(float)));
    GT_HIP(ctx, k->cand_d2.reserve(size_t(nq) * MP * sizeof(double)));
    GT_HIP(ctx, k->cand_j.reserve(size_t(nq) * MP * sizeof(uint32_t)));
    GT_HIP(ctx, k->cand_n.reserve(size_t(nq) * sizeof(uint32_t)));
    GT_HIP(ctx, k->d2_lb.reserve(size_t(nq) * sizeof(double)));
    GT_HIP(ctx, k->fb_rows.reserve(size_t(nq) * sizeof(int32_t)));
    GT_HIP(ctx, k->fb_count.reserve(sizeof(uint32_t)));
    GT_HIP(ctx, k->gflags.reserve(sizeof(uint32_t)));
    GT_HIP(ctx, hipMemsetAsync(k->fb_count.p, 0, sizeof(uint32_t), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(k->gflags.p, 0, sizeof(uint32_t), ctx->stream));

    SelectArgs sa;
    sa.dp = ctx->DP;
    sa.prec = main_prec;
    sa.final_keep = MP;
    sa.mode = 0;
    sa.nt = nt;
    sa.Yp = ctx->Yp.as<float>();
    sa.hneg = ctx->hneg.as<float>();
    sa.n_pad = ctx->n_pad;
    sa.Qp = external ? k->Qp.as<float>() : ctx->Yp.as<float>();
    sa.qrows = nullptr;
    sa.q0 = external ? 0 : q0;
    sa.nq = int32_t(nq);
    sa.lists = k->lists.as<uint64_t>();
    sa.counts = k->counts.as<uint32_t>();
    sa.thr_out = k->thr_final.as<float>();
    sa.dbg = ctx->dbg_select;
    {
        // threshold-seeding phase: worthwhile when the stream is long and the wanted count is well below M'
        int keep = std::max(ctx->samp_keep > 0 ? ctx->samp_keep : 16, need_m);
        keep += keep & 1;
        const int64_t ntiles = ctx->n_pad / (ctx->DP <= 64 ? 128 : 64);   // tiles of the candidate kernels
        int stride = 1, levels = 0;   // largest power of two <= the option
        while (stride * 2 <= ctx->samp_stride) stride *= 2, ++levels;
        if (stride > 1 && keep <= mkeep / 2 && ntiles >= int64_t(8) * stride) {
            sa.samp_stride = stride;
            sa.samp_keep = keep;
            sa.samp_trig = ctx->samp_trig;
            // second cut once 2^samp2_level / stride of the tiles are seen
            int keep2 = std::max(ctx->samp2_keep, 3 * keep);
            keep2 += keep2 & 1;
            if (ctx->samp2_level > 0 && ctx->samp2_level < levels && keep2 <= mkeep / 2) {
                sa.samp2_level = ctx->samp2_level;
                sa.samp2_keep = keep2;
            }
            int end = ctx->samp_end < 0 ? keep : std::max(ctx->samp_end > 0 ? ctx->samp_end : 0, ctx->samp_end > 0 ? need_m : 0);
            end += end & 1;
            sa.samp_end = end <= mkeep / 2 ? end : 0;
        }
    }
    if (ctx->dbg_select & 64) {
        // one record per wave; the symmetric pass launches up to 8 work items per query block
        GT_HIP(ctx, k->prof.reserve(size_t(k->nq_pad / bq) * 8 * 4 * 8 * sizeof(unsigned long long)));
        GT_HIP(ctx, hipMemsetAsync(k->prof.p, 0, size_t(k->nq_pad / bq) * 8 * 4 * 8 * sizeof(unsigned long long), ctx->stream));
        sa.prof = k->prof.as<unsigned long long>();
    }
    GT_HIP(ctx, k->unproven.reserve(sizeof(uint32_t)));
    k->tab_sorted = false;
    RerankArgs ra;
    ra.X = ctx->X;
    ra.dtype = ctx->dtype;
    ra.n = ctx->n;
    ra.d = ctx->d;
    ra.xn = ctx->xn.as<double>();
    ra.Q = external ? k->Qraw.p : ctx->X;
    ra.qn = external ? k->qn.as<double>() : ctx->xn.as<double>();
    ra.qn_sel = !ctx->wide ? ra.qn : (external ? k->qn_sel.as<double>() : ctx->xn_sel.as<double>());
    ra.q0 = external ? 0 : q0;
    ra.nq = nq;
    ra.lists = k->lists.as<uint64_t>();
    ra.lstride = int(lcap);
    ra.counts = k->counts.as<uint32_t>();
    ra.thr_final = k->thr_final.as<float>();
    ra.ymax2 = ctx->ymax.as<double>();
    ra.metric = ctx->metric;
    ra.need_m = need_m;
    ra.MP = MP;
    ra.cand_d2 = k->cand_d2.as<double>();
    ra.cand_j = k->cand_j.as<uint32_t>();
    ra.cand_n = k->cand_n.as<uint32_t>();
    ra.d2_lb = k->d2_lb.as<double>();
    ra.fb_count = k->fb_count.as<uint32_t>();
    ra.fb_rows = k->fb_rows.as<int32_t>();
    ra.gflags = k->gflags.as<uint32_t>();
    ra.radius_key_factor = radius_key_factor;
    ra.unproven = k->unproven.as<uint32_t>();
    bool have_thr0 = false;
    // row-sharded symmetric pass (gt_knn_shard.cpp): the candidate lists of exactly these rows are waiting
    // (stage 6: collected by the rank itself on renumbered points, gt_knn_shard_local - lists by sorted position = row)
    const bool sh_local = k->sh_stage == 6;
    bool sh_ready = (k->sh_stage == 5 || k->sh_stage == 6) && !external && q0 == k->sh_r0 && nq == k->sh_nloc &&
                    need_m == k->sh_need && radius_key_factor == k->sh_rkf && MP == 256;
    k->sh_stage = 0;
    if (sh_ready) {
        k->ordered = false;   // k->qorder holds the sorted order of ALL rows (candidate ids are positions in it)
    } else {
        // deal the query rows to workgroups grouped by nearest landmark (list i <-> row qorder[i]; gt_order.hip)
        int ordered = 0;
        const float* Qc = external ? k->Qc.as<float>() : ctx->Yc.as<float>();
        GT_HIP(ctx, k->qorder.reserve(size_t(nq) * sizeof(int32_t)));
        k->xs_ready = false;   // (the sorted copy of the points follows the order)
        GT_HIP(ctx, k->qthr0.reserve(size_t(nq) * sizeof(float)));
        {
            StageSpan span(ctx, "query_order");
            GT_TRY(gt_query_order(ctx, Qc, external ? 0 : q0, nq, need_m, k->qorder.as<int32_t>(), k->qthr0.as<float>(),
                                  &ordered));
        }
        if (ordered) sa.qrows = ra.qrows = k->qorder.as<int32_t>();
        k->ordered = ordered != 0 && !external && q0 == 0 && nq == ctx->n;
        have_thr0 = ordered != 0 && need_m <= 32 && ctx->order_has_thr0 != 0;
    }
    // Symmetric pass (gt_sym.hip): self queries over the whole point set, euclidean, single-chain arithmetic, grouped
    // query order available (its permutation is the cell-sorted order both sides of the pass share)
    const int bq_sym = gt_select_bq(ctx->DP);
    const bool use_sym = ctx->sym_mode != 0 && !external && q0 == 0 && nq == ctx->n && sa.qrows != nullptr &&
                         (ctx->metric == 0 || (ctx->metric == 1 && ctx->sym_cosine != 0)) && !ctx->wide && nt == 8 && need_m <= 64 &&
                         ctx->Yc.p != nullptr &&
                         (ctx->sym_mode > 0 || nq >= ctx->sym_min_rows) &&
                         ctx->order_L > 0 && nq >= int64_t(8) * bq_sym;
    k->sym_used = false;
    bool sym_now = use_sym && ctx->sym_ok != 0;
    bool two_failed = false;   // the two-stage collect of this call overflowed its queue
    uint32_t n_fb = 0;
    for (;;) {
        if (sh_ready) {
            sh_ready = false;
            ErrModel em = gt_err_model(ctx, 2);
            em.rel += 8.0 * 5.9604644775390625e-08;
            ra.err = em;
            // (the counters of the seeding stage - far-kept rows [2], tiles [5] - stay; only the re-rank's own start at 0)
            GT_HIP(ctx, hipMemsetAsync(k->sym_stat.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
            GT_HIP(ctx, hipMemsetAsync(k->sym_stat.as<unsigned long long>() + 6, 0, sizeof(unsigned long long), ctx->stream));
            GT_HIP(ctx, hipMemsetAsync(k->unproven.p, 0, sizeof(uint32_t), ctx->stream));
            SymRerank sr;
            sr.tcap = ctx->sym_tcap;
            sr.perm = k->qorder.as<int32_t>();
            sr.stat = k->sym_stat.as<unsigned long long>();
            sr.own_r0 = q0;
            if (sh_local) {
                // lists, thresholds and rows by sorted position, which IS the row of the renumbered points
                sr.tlists = k->tlists.as<uint64_t>();
                sr.tcounts = k->tcounts.as<uint32_t>();
                sr.pos0 = q0;
                sr.Xs = ctx->X;
                sr.xns = ctx->xn.as<double>();
            } else {
                sr.tlists = k->sh_lists.as<uint64_t>();
                sr.tcounts = k->sh_counts.as<uint32_t>();
                sr.invperm = k->sh_invperm.as<int32_t>();
                sr.own_rows = k->sh_own.as<int32_t>();
                if (k->xs_ready) {
                    sr.Xs = k->Xs.p;
                    sr.xns = k->xns.as<double>();
                    sr.xs_d = k->xs_d;
                }
            }
            bool wrote_t = false;
            if (k->want_keyt_shard && ctx->symm_pairs != 0 && MP == 256) {
                // the pair-resolved tail on a rank of a sharded build (gt_graph_bandwidth_local): the keys of the transposed pairs
                // next to the tables, as in the single-rank pass below (the tables stay by row)
                GT_HIP(ctx, k->cand_d2t.reserve(size_t(nq) * MP * sizeof(double)));
                GT_HIP(ctx, k->keyt_ok.reserve(size_t(nq)));
                GT_HIP(ctx, k->nokeyt_rows.reserve(size_t(nq) * sizeof(int32_t)));
                GT_HIP(ctx, k->nokeyt_count.reserve(sizeof(uint32_t)));
                GT_HIP(ctx, hipMemsetAsync(k->nokeyt_count.p, 0, sizeof(uint32_t), ctx->stream));
                sr.cand_d2t = k->cand_d2t.as<double>();
                sr.keyt_ok = k->keyt_ok.as<uint8_t>();
                sr.nokeyt_rows = k->nokeyt_rows.as<int32_t>();
                sr.nokeyt_count = k->nokeyt_count.as<uint32_t>();
                sr.wrote_t = &wrote_t;
            }
            {
                StageSpan span(ctx, "rerank");
                GT_TRY(gt_launch_rerank_sym(ctx, ra, sr));
            }
            k->keyt_valid = wrote_t;
            k->nokeyt_n = 0;
            uint32_t n_unproven = 0;
            {
                // (one group through the pinned mailbox: four copies into pageable memory were four staged round trips, ~18 us each)
                ReadBack rb(ctx);
                if (wrote_t) GT_HIP(ctx, rb.add(&k->nokeyt_n, k->nokeyt_count.p, sizeof(uint32_t)));
                GT_HIP(ctx, rb.add(&n_fb, k->fb_count.p, sizeof(uint32_t)));
                GT_HIP(ctx, rb.add(&n_unproven, k->unproven.p, sizeof(uint32_t)));
                GT_HIP(ctx, rb.add(k->sym_stat_host, k->sym_stat.p, 8 * sizeof(unsigned long long)));
                GT_HIP(ctx, rb.sync());
            }
            k->sym_overflow = int64_t(k->sym_stat_host[0]);
            k->sym_used = true;
            ctx->last_main_prec = 2;
            // (the verdicts of the single-rank pass stay local: a rank whose share went badly redoes ITS rows with the
            //  classic pass - its own rows against all points need nobody else)
            if (ctx->dbg_select & 2048)
                fprintf(stderr, "[gt] shard rerank: rows %lld unproven %u repairs %u overflow %lld\n", (long long)nq, n_unproven,
                        n_fb, (long long)k->sym_overflow);
            // (rows whose table is full and still cannot prove its radius count like overflows: sym_stat[6], gt_rerank.hip)
            const bool too_many = ctx->sym_mode < 0 && double(k->sym_overflow + int64_t(k->sym_stat_host[6])) > 0.10 * double(nq);
            const bool unproved = fast_auto && double(n_unproven) > kFastFailFrac * double(nq);
            if (too_many || unproved) {
                if (too_many) ctx->sym_ok = 0;
                if (unproved) ctx->fast_ok = 0, main_prec = 1;
                k->sym_used = false;
                k->keyt_valid = false;
                k->nokeyt_n = 0;
                n_fb = 0;
                GT_HIP(ctx, hipMemsetAsync(k->fb_count.p, 0, sizeof(uint32_t), ctx->stream));
                GT_HIP(ctx, hipMemsetAsync(k->gflags.p, 0, sizeof(uint32_t), ctx->stream));
                continue;
            }
            break;
        }
        if (sym_now && main_prec == 2) {
            // two-stage scoring: half the features first, against partial-distance thresholds (gt_sym.hip sym_half_*);
            // its collect kernel works on query blocks of up to 1024 rows
            const bool two_stage = !two_failed && ctx->DP >= 32 && bq_sym == 256 &&
                                   (ctx->sym_two_stage > 0 || (ctx->sym_two_stage < 0 && ctx->sym_two_ok != 0));
            const int64_t pad_s = two_stage ? 1024 : bq_sym;
            const int64_t n_pad_s = ceil_div64(nq, pad_s) * pad_s;
            const int tcap = ctx->sym_tcap;
            const int32_t* perm = k->qorder.as<int32_t>();
            GT_HIP(ctx, k->Ycs.reserve(size_t(n_pad_s) * ctx->DP * sizeof(_Float16)));
            GT_HIP(ctx, k->hnegs.reserve(size_t(n_pad_s) * sizeof(float)));
            GT_HIP(ctx, k->sym_g.reserve(size_t(n_pad_s) * sizeof(float)));
            GT_HIP(ctx, k->sym_gmin.reserve(size_t(n_pad_s / 32) * sizeof(float)));
            GT_HIP(ctx, k->tlists.reserve(size_t(n_pad_s) * size_t(tcap) * sizeof(uint64_t)));
            GT_HIP(ctx, k->tcounts.reserve(size_t(n_pad_s) * sizeof(uint32_t)));
            GT_HIP(ctx, k->sym_stat.reserve(8 * sizeof(unsigned long long)));
            ErrModel em = gt_err_model(ctx, 2);
            em.rel += 8.0 * 5.9604644775390625e-08;   // the transposed test adds two float32 roundings to a score
            ra.err = em;
            const double rkf = std::max(1.0, std::fabs(radius_key_factor));
            const int bn_sym = gt_select_bn(ctx->DP);
            const int n_tiles_s = int(n_pad_s / bn_sym);
            const int stride_a = ctx->sym_stride > 0 && n_tiles_s >= 8 * ctx->sym_stride ? ctx->sym_stride : 0;
            const int tile_stride = ((stride_a ? n_tiles_s / stride_a + 1 : 0) + ctx->sym_max_nb + bq_sym / bn_sym + 63) / 64 * 64;
            GT_HIP(ctx, k->sym_tiles.reserve(size_t(n_pad_s / bq_sym) * tile_stride * sizeof(int32_t)));
            GT_HIP(ctx, k->sym_tile_cnt.reserve(size_t(n_pad_s / bq_sym) * sizeof(int32_t)));
            GT_HIP(ctx, hipMemsetAsync(k->sym_stat.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
            {
                StageSpan span(ctx, "sym_prepare");
                GT_HIP(ctx, k->hnegs_fin.reserve(size_t(n_pad_s) * sizeof(float)));
                GT_TRY(gt_sym_gather(ctx, perm, n_pad_s, k->Ycs.p, k->hnegs.as<float>(), k->hnegs_fin.as<float>()));
                if (ctx->sym_sorted_points != 0) GT_TRY(gt_sym_gather_points(ctx, perm, true));
                GT_TRY(gt_sym_schedule(ctx, n_pad_s, bq_sym, bn_sym, ctx->sym_cells, stride_a, ctx->sym_max_nb, tile_stride,
                                       k->sym_work, k->sym_tiles.as<int32_t>(), k->sym_tile_cnt.as<int32_t>(),
                                       k->sym_stat.as<unsigned long long>() + 5));
            }
            SelectArgs a = sa;
            a.prec = 2;
            a.narrow = 0;
            a.Yp = a.Qp = k->Ycs.as<float>();
            a.hneg = k->hnegs.as<float>();
            a.n_pad = n_pad_s;
            a.qrows = nullptr;
            a.q0 = 0;
            a.thr_in = nullptr;
            a.mode = 0;
            a.sym.sched = 1;
            a.sym.tile_list = k->sym_tiles.as<int32_t>();
            a.sym.tile_cnt = k->sym_tile_cnt.as<int32_t>();
            a.sym.tile_stride = tile_stride;
            if (ctx->narrow_mode != 0 && ctx->DP <= 64 && bq_sym == 256) {
                // 128-row workgroups (one query tile per wave) on the tile lists made for 256-row blocks, as the sharded
                // seeding share runs them: every workgroup walks its whole tile list with dependent list maintenance in
                // between - twice the waves hide more of each other's latencies (7.8 -> 7.4 ms at N = 1e6)
                a.narrow = 1;
                a.sym.list_shift = 1;
            }
            a.samp_stride = 0;
            int keep = std::max(ctx->samp_keep > 0 ? ctx->samp_keep : 16, need_m);
            keep += keep & 1;
            a.samp_keep = keep;
            a.samp_trig = ctx->samp_trig > 0 ? ctx->samp_trig : 96;   // (lists of the neighbourhood tiles: compact at 96 keys, 7.5 -> 7.1 ms)
            a.samp_end = 0;
            a.samp2_level = 0;
            a.final_keep = need_m;
            // dense cell blocks with the keys in registers (gt_seed.hip), or the streaming lists of the candidate kernel
            const bool dense_seed = ctx->sym_dense_seed != 0 && need_m <= 64 && tile_stride <= 1024 && n_pad_s % 256 == 0;
            const int seed_lstride = dense_seed ? 64 : int(lcap);
            int64_t seeded_to = 0;   // sorted positions [0, seeded_to) were seeded (and got their thresholds) by the sample below
            if (ctx->sym_mode < 0 && ctx->sym_ok < 0 && dense_seed && n_pad_s >= int64_t(16) * 8192) {
                // First build on this point set, verdict "do the seeds come from the cells around the row?" still open: ask a
                // SAMPLE first - the first sixteenth of the sorted positions (cells are numbered by landmark, landmarks are
                // evenly strided rows: a prefix of the cells is a spatially random sample) - before every row is seeded.  On
                // points without cluster structure (isotropic Gaussian, N = 2e5, d = 24: 18 ms against 11 ms for the classic
                // pass alone, tests/test_gpu_ladder.py) the lost attempt shrinks from ~6 ms to ~1 ms.
                const int64_t ps = std::max<int64_t>(4096, (n_pad_s / 16) / 256 * 256), prows = std::min<int64_t>(ps, nq);
                unsigned long long far_s = 0;
                {
                    StageSpan span(ctx, "sym_seed");
                    GT_TRY(gt_sym_seed_dense(ctx, ctx->DP, k->Ycs.p, k->hnegs_fin.as<float>(), nq, n_pad_s, k->sym_tiles.as<int32_t>(),
                                             k->sym_tile_cnt.as<int32_t>(), tile_stride, bq_sym, 0, ps / 128, need_m,
                                             k->lists.as<uint64_t>(), seed_lstride, k->counts.as<uint32_t>()));
                }
                {
                    StageSpan span(ctx, "sym_prepare");
                    GT_HIP(ctx, k->sym_farcnt.reserve(size_t(n_pad_s) * sizeof(float)));
                    GT_TRY(gt_sym_thresholds(ctx, perm, n_pad_s, k->hnegs.as<float>(), k->lists.as<uint64_t>(), seed_lstride,
                                             k->counts.as<uint32_t>(), need_m, em, rkf, k->thr_final.as<float>(),
                                             k->sym_g.as<float>(), nullptr, k->sym_work, ctx->sym_cells,
                                             k->sym_stat.as<unsigned long long>() + 2, k->sym_farcnt.as<float>(), 0, ps, true));
                    GT_HIP(ctx, hipMemcpyAsync(&far_s, k->sym_stat.as<unsigned long long>() + 2, sizeof(far_s), hipMemcpyDeviceToHost,
                                               ctx->stream));
                    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                }
                const double est_s = double(need_m) + double(std::max(stride_a, 1)) * double(far_s) / double(prows);
                if (ctx->dbg_select & 2048)
                    fprintf(stderr, "[gt] seeding sample: %lld rows, far-kept %llu (%.2f per row), estimate %.1f (limit %.1f)\n",
                            (long long)prows, far_s, double(far_s) / double(prows), est_s, double(tcap) / 8.0);
                // (the two verdicts of the full launch, on the sample: the far-kept predictor where a strided sample of the other
                //  tiles was scored, and "every second row kept a seed from outside the cells around it" - GT_SAMPLE_FAR rows)
                if ((stride_a > 0 && est_s > double(tcap) / 8.0) || double(far_s) >= ctx->sym_sample_far * double(prows)) {
                    k->sym_far = int64_t(double(far_s) * double(nq) / double(prows));
                    ctx->sym_ok = 0;
                    sym_now = false;
                    continue;
                }
                seeded_to = ps;   // (accepted: the sample's rows are done, the launches below take the rest)
            }
            {
                StageSpan span(ctx, "sym_seed");
                if (dense_seed)
                    GT_TRY(gt_sym_seed_dense(ctx, ctx->DP, k->Ycs.p, k->hnegs_fin.as<float>(), nq, n_pad_s, k->sym_tiles.as<int32_t>(),
                                             k->sym_tile_cnt.as<int32_t>(), tile_stride, bq_sym, seeded_to / 128,
                                             seeded_to > 0 ? (n_pad_s - seeded_to) / 128 : 0, need_m,
                                             k->lists.as<uint64_t>(), seed_lstride, k->counts.as<uint32_t>()));
                else
                    GT_TRY(gt_launch_select(ctx, a));
            }
            k->sym_seed_dense = dense_seed;
            if (ctx->dbg_select & 1024) return GT_OK;   // experiment: stop behind the seeding launch (tables are NOT valid)
            {
                StageSpan span(ctx, "sym_prepare");
                GT_HIP(ctx, k->sym_farcnt.reserve(size_t(n_pad_s) * sizeof(float)));
                GT_HIP(ctx, k->sym_racc.reserve(4 * sizeof(double)));
                GT_HIP(ctx, hipMemsetAsync(k->sym_racc.p, 0, 4 * sizeof(double), ctx->stream));
                GT_TRY(gt_sym_thresholds(ctx, perm, n_pad_s, k->hnegs.as<float>(), k->lists.as<uint64_t>(), seed_lstride,
                                         k->counts.as<uint32_t>(), need_m, em, rkf, k->thr_final.as<float>(),
                                         k->sym_g.as<float>(), k->sym_gmin.as<float>(), k->sym_work, ctx->sym_cells,
                                         k->sym_stat.as<unsigned long long>() + 2, k->sym_farcnt.as<float>(), seeded_to, -1));
                // statistics for the orphan cut of the two-stage collect (gt_sym_two_stage_prepare)
                GT_TRY(gt_sym_radius_sum(ctx, perm, 0, n_pad_s, k->thr_final.as<float>(), em, k->sym_racc.as<double>()));
                GT_HIP(ctx, hipMemsetAsync(k->tcounts.p, 0, size_t(n_pad_s) * sizeof(uint32_t), ctx->stream));
            }
            if (ctx->sym_mode < 0) {
                // launch A kept need_m rows per point: when most of them came from the strided sample instead of the cells
                // around the point, the cells say nothing about this point set (no cluster structure at their scale), the
                // thresholds are loose and launch B would drown in candidates - the classic pass is the right tool
                unsigned long long far_total = 0;
                GT_HIP(ctx, hipMemcpyAsync(&far_total, k->sym_stat.as<unsigned long long>() + 2, sizeof(far_total),
                                           hipMemcpyDeviceToHost, ctx->stream));
                GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                k->sym_far = int64_t(far_total);
                // the strided sample shows 1 / stride of the rows outside the neighbourhood: every kept row from it stands
                // for `stride` rows at least as close.  An estimated need_m + stride * far rows inside D_K per point, times
                // the growth out to the collection radius (x 5 ... 10), has to stay well inside the lists.
                const double est = double(need_m) + double(std::max(stride_a, 1)) * double(far_total) / double(nq);
                if (stride_a > 0 && est > double(tcap) / 8.0) {
                    ctx->sym_ok = 0;
                    sym_now = false;
                    continue;
                }
            }
            a.mode = 2;
            a.sym.sched = 0;
            a.sym.g = k->sym_g.as<float>();
            a.sym.gmin = k->sym_gmin.as<float>();
            a.sym.tlists = k->tlists.as<uint64_t>();
            a.sym.tcounts = k->tcounts.as<uint32_t>();
            a.sym.tcap = tcap;
            {
                // work items per query block: the walk of a block is cut into nseg pieces so that the last round of
                // workgroups is as full as the others (3 workgroups of this kernel per CU)
                const int64_t slots = int64_t(ctx->n_cu) * 3, nb = n_pad_s / bq_sym;
                int best = 1;
                double best_cost = 1e30;
                for (int sgm = 1; sgm <= 8; ++sgm) {
                    const double cost = double(ceil_div64(nb * sgm, slots)) / sgm + 0.03 * sgm;
                    if (cost < best_cost - 1e-9) best_cost = cost, best = sgm;
                }
                a.sym.nseg = ctx->sym_nseg > 0 ? std::min(ctx->sym_nseg, 8) : best;
                k->sym_nseg = a.sym.nseg;
            }
            a.thr_in = k->thr_final.as<float>();
            // Bound pass first: on clustered points the cells of the sorted order rule out nearly every (64 queries, 32
            // rows) unit without looking at a row; what they leave goes straight to the cold launch, and neither the
            // stage-one copy nor the collect launch is needed.  (More than 4 M units left: the unit loop is the better
            // filter - its preparation follows, with the orphans already cut.)
            bool bound_done = false, bound_tried = false;
            uint32_t bound_left = 0;
            k->sym_bound_used = false;
            if (two_stage && ctx->sym_bounds != 0 && n_pad_s % 1024 == 0 && ctx->order_L > 0 && ctx->sym_two_stage != 0 &&
                (ctx->sym_two_stage > 0 || ctx->sym_two_ok != 0)) {
                const int64_t bcap = ctx->sym_bound_cap > 0 ? ctx->sym_bound_cap : (int64_t(1) << 22);
                GT_HIP(ctx, k->sym_qdense.reserve(size_t(bcap) * sizeof(uint2)));
                GT_HIP(ctx, k->sym_qtot.reserve(4 * sizeof(uint32_t)));
                GT_HIP(ctx, k->sym_rrow.reserve(size_t(n_pad_s) * sizeof(float)));
                {
                    StageSpan span(ctx, "sym_bound");
                    // the orphans lose their thresholds (a handful of rows: their radii would keep whole cells undecided),
                    // the forms derived from the thresholds are made again
                    GT_TRY(gt_sym_orphan_cut(ctx, perm, k->thr_final.as<float>(), k->sym_farcnt.as<float>(), em,
                                             k->sym_racc.as<double>(), need_m, 0.25));
                    GT_TRY(gt_sym_g_from_thr(ctx, n_pad_s, k->thr_final.as<float>(), k->hnegs.as<float>(), k->sym_g.as<float>(),
                                             k->sym_gmin.as<float>()));
                    GT_TRY(gt_sym_row_radius(ctx, perm, n_pad_s, k->thr_final.as<float>(), em, k->sym_rrow.as<float>()));
                    GT_TRY(gt_sym_bound_queue(ctx, n_pad_s, k->Ycs.p, k->sym_rrow.as<float>(), k->sym_bwork,
                                              k->sym_qdense.as<uint2>(), uint32_t(bcap), k->sym_qtot.as<uint32_t>()));
                    GT_HIP(ctx, hipMemcpyAsync(&bound_left, k->sym_qtot.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                }
                GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                bound_tried = true;
                bound_done = int64_t(bound_left) <= bcap;
                if (ctx->dbg_select & 2048) {
                    fprintf(stderr, "[gt] bound pass: %u units left (capacity %lld)\n", bound_left, (long long)bcap);
                    if (int64_t(bound_left) <= bcap && bound_left > 0) {   // development: how the units spread over the query groups
                        std::vector<uint2> q(bound_left);
                        (void)hipMemcpy(q.data(), k->sym_qdense.p, size_t(bound_left) * sizeof(uint2), hipMemcpyDeviceToHost);
                        std::vector<uint32_t> per(size_t(n_pad_s / 64), 0u);
                        for (auto& u : q) per[u.x] += 1u;
                        uint32_t mx = 0, arg = 0;
                        uint64_t over1k = 0;
                        for (size_t g = 0; g < per.size(); ++g) {
                            if (per[g] > mx) mx = per[g], arg = uint32_t(g);
                            if (per[g] > 1000u) ++over1k;
                        }
                        fprintf(stderr, "[gt] bound pass: most units in one query group %u (group %u of %zu), groups with more than 1000 units %llu\n",
                                mx, arg, per.size(), (unsigned long long)over1k);
                    }
                }
            }
            // Too many units for the queue - but the cell masks are there, and they may still rule out most TILES (points near a
            // low-dimensional sheet: the cells around a row are a few per cent of all cells, while a 244-row cell is compact
            // only down to its own radius - millions of small units).  Then the one-stage collect streams only the listed
            // tiles of every query block's walk (gt_sym.hip collect_lists_kernel), scoring and filing in one launch: neither
            // the stage-one copy (a PCA on the host, two projections) nor the 16-column stream over ALL pairs nor a cold
            // launch of one round trip per unit.  Taken when the lists hold at most a quarter of the walks (a listed tile
            // costs ~5 x an unlisted tile of the 16-column stream, plus what the cold launch would have cost).
            k->sym_listed = false;
            k->sym_listed_tiles = 0;
            if (two_stage && bound_tried && !bound_done && ctx->sym_listed != 0 && bq_sym == 256 && bn_sym == 128) {
                const int64_t nb = n_pad_s / bq_sym;
                const int wcap = 2048;   // entries per block (a block whose list is longer walks everything)
                GT_HIP(ctx, k->sym_wlist.reserve(size_t(nb) * wcap * sizeof(int32_t)));
                GT_HIP(ctx, k->sym_wcnt.reserve(size_t(nb) * sizeof(int32_t) + 16));
                unsigned long long* tot_dev = reinterpret_cast<unsigned long long*>(
                    (reinterpret_cast<uintptr_t>(k->sym_wcnt.as<int32_t>() + nb) + 7) & ~uintptr_t(7));
                int walk = 0;
                unsigned long long listed = 0;
                {
                    StageSpan span(ctx, "sym_bound");
                    GT_TRY(gt_sym_collect_lists(ctx, n_pad_s, k->sym_bwork, wcap, wcap, k->sym_wlist.as<int32_t>(),
                                                k->sym_wcnt.as<int32_t>(), tot_dev, &walk));
                    GT_HIP(ctx, hipMemcpyAsync(&listed, tot_dev, sizeof(listed), hipMemcpyDeviceToHost, ctx->stream));
                    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                }
                const double frac = double(listed) / std::max(1.0, double(nb) * double(walk));
                if (ctx->dbg_select & 2048)
                    fprintf(stderr, "[gt] listed walks: %llu tiles of %lld (%.2f %%), walk %d\n", listed, (long long)(nb * walk),
                            100.0 * frac, walk);
                if (ctx->sym_listed > 0 || frac <= 0.25) {
                    a.sym.walk_list = k->sym_wlist.as<int32_t>();
                    a.sym.walk_cnt = k->sym_wcnt.as<int32_t>();
                    a.sym.walk_stride = wcap;
                    k->sym_listed = true;
                    k->sym_listed_tiles = int64_t(listed);
                }
            }
            if (two_stage && !bound_done && !k->sym_listed)
                GT_TRY(gt_sym_two_stage_prepare(ctx, perm, n_pad_s, em, need_m, a, bound_tried));
            const bool two_now = bound_done || a.sym.half_steps > 0;
            k->sym_cold_local_used = false;
            // (behind the bound pass only: the units stage one of the two-stage collect lets through - 7.6 M on the manifold
            //  set - are mostly far pairs that file nothing; there the launch is all operand traffic and the frame costs
            //  14.0 -> 20.9 ms for 0.4 ms of re-rank)
            if (bound_done && ctx->sym_cold_local != 0 && k->xs_ready && ctx->dtype == GT_F32 && (k->xs_d & 3) == 0 &&
                k->xs_d <= ctx->DP && ctx->DP >= 32 && ctx->DP <= 64) {
                // the cold launch (behind the bound pass, or behind stage one) in the frame of its queries: what every row needs
                // listed, as a radius (the one the re-rank will claim: lb of thr - the thresholds are final here), and room for
                // the query groups' centres
                GT_HIP(ctx, k->sym_rloc.reserve(size_t(n_pad_s) * sizeof(float)));
                GT_HIP(ctx, k->sym_gcen.reserve(size_t(n_pad_s / 64) * ctx->DP * sizeof(float)));
                GT_TRY(gt_sym_row_radius(ctx, perm, n_pad_s, k->thr_final.as<float>(), em, k->sym_rloc.as<float>(), 0.0));
                a.sym.xs = k->Xs.as<float>();
                a.sym.xs_d = k->xs_d;
                a.sym.xs_n = int32_t(ctx->n);
                a.sym.gcen = k->sym_gcen.as<float>();
                a.sym.rloc = k->sym_rloc.as<float>();
                a.sym.sc = float(ctx->sc);
                k->sym_cold_local_used = true;
            }
            if (!two_now && !k->sym_listed && ctx->sym_mode < 0 && double(k->sym_far) >= 0.5 * double(nq)) {
                // neither the cell bounds nor the partial distances prune this point set, and every second row found a seed
                // outside the cells around it: no cluster structure at the cells' scale.  The one-stage collect would score
                // every pair once with admissions all along - measured slower than the classic pass on such data (isotropic
                // Gaussian, d = 24, N = 3e5: 33 ms against 16) - so the classic pass runs, now and from here on
                ctx->sym_ok = 0;
                sym_now = false;
                continue;
            }
            // (the orphans that were declared start their lists with the rows launch A kept for them)
            if (two_now || bound_tried)
                GT_TRY(gt_sym_inject_orphans(ctx, 0, n_pad_s, k->thr_final.as<float>(), k->lists.as<uint64_t>(), seed_lstride,
                                             k->counts.as<uint32_t>(), k->tlists.as<uint64_t>(), tcap, k->tcounts.as<uint32_t>()));
            if (two_now) {
                if (ctx->sym_nseg <= 0) {
                    // (the collect workgroups of this kernel take 1024 rows: NB x nseg items must fill the rounds)
                    const int64_t slots = int64_t(ctx->n_cu) * 3, nb = ceil_div64(n_pad_s, 1024);
                    int best = 1;
                    double best_cost = 1e30;
                    for (int sgm = 1; sgm <= 8; ++sgm) {
                        const double cost = double(ceil_div64(nb * sgm, slots)) / sgm + 0.01 * sgm;
                        if (cost < best_cost - 1e-9) best_cost = cost, best = sgm;
                    }
                    a.sym.nseg = k->sym_nseg = best;
                }
                if (!bound_done) GT_TRY(gt_sym_queue_prepare(ctx, n_pad_s, a));
            }
            if (bound_done) {
                StageSpan span(ctx, "sym_cold");
                SelectArgs dq = a;
                dq.mode = 4;
                dq.sym.queue = k->sym_qdense.as<uint2>();
                dq.sym.qn = int32_t(bound_left);
                GT_TRY(gt_launch_select(ctx, dq));
                k->sym_cold_entries = int64_t(bound_left);
                k->sym_bound_used = true;
                if (ctx->sym_two_ok < 0) ctx->sym_two_ok = 1;
            }
            if (!bound_done) {
                StageSpan span(ctx, "knn_select");
                GT_TRY(gt_sym_launch_collect(ctx, a));
            }
            if (two_now && !bound_done) {
                int ok = 0;
                GT_TRY(gt_sym_queue_finish(ctx, a, &k->sym_cold_entries, &ok));
                if (!ok) {
                    // stage one let (nearly) everything through: partial distances say nothing on this point set.  Start
                    // launch B over with the one-stage kernel, now and for the later builds on these points.
                    ctx->sym_two_ok = 0;
                    two_failed = true;   // (also when the option forces the two-stage collect: not a second time)
                    continue;
                }
                if (ctx->sym_two_ok < 0) ctx->sym_two_ok = 1;
            }
            k->sym_two_used = two_now;
            ctx->last_main_prec = 2;
            k->sym_used = true;
            if (ctx->dbg_select & 4) return GT_OK;   // experiment: candidate pass only (tables are NOT valid)
            GT_HIP(ctx, hipMemsetAsync(k->unproven.p, 0, sizeof(uint32_t), ctx->stream));
            SymRerank sr;
            sr.tlists = k->tlists.as<uint64_t>();
            sr.tcounts = k->tcounts.as<uint32_t>();
            sr.tcap = tcap;
            sr.perm = perm;
            sr.stat = k->sym_stat.as<unsigned long long>();
            if (k->xs_ready) {
                sr.Xs = k->Xs.p;
                sr.xns = k->xns.as<double>();
                sr.xs_d = k->xs_d;
            }
            bool wrote_t = false;
            if (ctx->symm_pairs != 0 && MP == 256) {
                // the keys of the transposed pairs next to the tables (pair-resolved symmetrisation, gt_sparse.hip)
                GT_HIP(ctx, k->cand_d2t.reserve(size_t(nq) * MP * sizeof(double)));
                GT_HIP(ctx, k->keyt_ok.reserve(size_t(nq)));
                GT_HIP(ctx, k->nokeyt_rows.reserve(size_t(nq) * sizeof(int32_t)));
                GT_HIP(ctx, k->nokeyt_count.reserve(sizeof(uint32_t)));
                GT_HIP(ctx, hipMemsetAsync(k->nokeyt_count.p, 0, sizeof(uint32_t), ctx->stream));
                sr.cand_d2t = k->cand_d2t.as<double>();
                sr.keyt_ok = k->keyt_ok.as<uint8_t>();
                sr.nokeyt_rows = k->nokeyt_rows.as<int32_t>();
                sr.nokeyt_count = k->nokeyt_count.as<uint32_t>();
                sr.wrote_t = &wrote_t;
                sr.tab_sorted = k->want_tab_sorted && q0 == 0 && nq == ctx->n;
            }
            {
                StageSpan span(ctx, "rerank");
                GT_TRY(gt_launch_rerank_sym(ctx, ra, sr));
            }
            k->keyt_valid = wrote_t;
            k->tab_sorted = wrote_t && sr.tab_sorted;
            if (k->tab_sorted) {
                // tables by sorted position: whoever comes by row (the repairs below, the listed rows of the affinity pass) asks
                // the inverse permutation for the slot
                GT_HIP(ctx, k->sh_invperm.reserve(size_t(nq) * sizeof(int32_t)));
                GT_TRY(gt_sym_invperm(ctx, perm, k->sh_invperm.as<int32_t>()));
                ra.trow = k->sh_invperm.as<int32_t>();
            }
            k->nokeyt_n = 0;
            uint32_t n_unproven = 0;
            {
                ReadBack rb(ctx);
                if (wrote_t) GT_HIP(ctx, rb.add(&k->nokeyt_n, k->nokeyt_count.p, sizeof(uint32_t)));
                GT_HIP(ctx, rb.add(&n_fb, k->fb_count.p, sizeof(uint32_t)));
                GT_HIP(ctx, rb.add(&n_unproven, k->unproven.p, sizeof(uint32_t)));
                GT_HIP(ctx, rb.add(k->sym_stat_host, k->sym_stat.p, 8 * sizeof(unsigned long long)));
                GT_HIP(ctx, rb.sync());
            }
            k->sym_overflow = int64_t(k->sym_stat_host[0]);
            if (ctx->sym_mode < 0 && double(k->sym_overflow + int64_t(k->sym_stat_host[6])) > 0.10 * double(nq)) {
                // neighbourhoods too large for the fixed lists (every overflowing row costs a repair): this point set
                // goes through the classic pass, now and from here on
                ctx->sym_ok = 0;
                sym_now = false;
                k->sym_used = false;
                k->keyt_valid = false;
                k->tab_sorted = false;
                ra.trow = nullptr;
                GT_HIP(ctx, hipMemsetAsync(k->fb_count.p, 0, sizeof(uint32_t), ctx->stream));
                GT_HIP(ctx, hipMemsetAsync(k->gflags.p, 0, sizeof(uint32_t), ctx->stream));
                continue;
            }
            if (fast_auto) {
                const bool ok = double(n_unproven) <= kFastFailFrac * double(nq);
                ctx->fast_ok = ok ? 1 : 0;
                if (!ok) {
                    main_prec = 1;
                    k->sym_used = false;
                    k->keyt_valid = false;
                    k->tab_sorted = false;
                    ra.trow = nullptr;
                    GT_HIP(ctx, hipMemsetAsync(k->fb_count.p, 0, sizeof(uint32_t), ctx->stream));
                    GT_HIP(ctx, hipMemsetAsync(k->gflags.p, 0, sizeof(uint32_t), ctx->stream));
                    continue;
                }
            }
            if (ctx->sym_ok < 0) ctx->sym_ok = 1;   // (the verdicts are in for this point set: later builds skip the seeding sample)
            break;
        }
        sa.prec = main_prec;
        // Few query rows (a shard of a multi-GPU build): 128-row workgroups double the number of workgroups, which fills
        // the machine better than it costs in shared-operand reuse - as long as the 256-row workgroups would leave CUs
        // without one.  Round 4, isotropic N = 1e6, d = 64 sharded (tools/gpu_shard_local_probe.py): 31 k rows 13.1 against
        // 19.2 ms, 62 k rows 15.0 against 19.4; from 125 k rows (488 workgroups of 256 rows on 256 CUs) the wide kernel wins:
        // 23.4 against 30.0 ms, 250 k rows 45.4 against 49.7.
        sa.narrow = (main_prec == 2 && ctx->DP <= 64 &&
                     (ctx->narrow_mode < 0 ? nq <= int64_t(ctx->n_cu) * 256 * 3 / 2 : ctx->narrow_mode == 1)) ? 1 : 0;
        // the single chain streams the compact hi-plane copies (rows of 2*DP bytes), the others the full working copy
        sa.Yp = main_prec == 2 ? ctx->Yc.as<float>() : ctx->Yp.as<float>();
        sa.Qp = main_prec == 2 ? (external ? k->Qc.as<float>() : ctx->Yc.as<float>())
                               : (external ? k->Qp.as<float>() : ctx->Yp.as<float>());
        ra.err = gt_err_model(ctx, main_prec);
        // the assignment pass scored in the single-chain arithmetic: its bound holds for that arithmetic only
        sa.thr_in = (have_thr0 && main_prec == 2) ? k->qthr0.as<float>() : nullptr;
        {
            StageSpan span(ctx, "knn_select");
            GT_TRY(gt_launch_select(ctx, sa));
        }
        ctx->last_main_prec = main_prec;
        if (ctx->dbg_select & 4) return GT_OK;   // experiment: candidate pass only (tables are NOT valid)
        GT_HIP(ctx, hipMemsetAsync(k->unproven.p, 0, sizeof(uint32_t), ctx->stream));
        bool wrote_t = false;
        if ((k->want_keyt_classic || k->want_keyt_shard) && ctx->symm_pairs != 0 && MP == 256 && !external) {
            // a '+' build that can take the pair-resolved tail (gt_sparse.hip): the keys of the transposed pairs next to the tables,
            // as the symmetric re-rank writes them (rerank_kernel<., ., ., true>)
            GT_HIP(ctx, k->cand_d2t.reserve(size_t(nq) * MP * sizeof(double)));
            GT_HIP(ctx, k->keyt_ok.reserve(size_t(nq)));
            GT_HIP(ctx, k->nokeyt_rows.reserve(size_t(nq) * sizeof(int32_t)));
            GT_HIP(ctx, k->nokeyt_count.reserve(sizeof(uint32_t)));
            GT_HIP(ctx, hipMemsetAsync(k->nokeyt_count.p, 0, sizeof(uint32_t), ctx->stream));
            ra.cand_d2t = k->cand_d2t.as<double>();
            ra.keyt_ok = k->keyt_ok.as<uint8_t>();
            ra.nokeyt_rows = k->nokeyt_rows.as<int32_t>();
            ra.nokeyt_count = k->nokeyt_count.as<uint32_t>();
            ra.wrote_t = &wrote_t;
        }
        {
            StageSpan span(ctx, "rerank");
            GT_TRY(gt_launch_rerank(ctx, ra));
        }
        k->keyt_valid = wrote_t;
        k->nokeyt_n = 0;
        uint32_t n_unproven = 0;
        {
            ReadBack rb(ctx);
            if (wrote_t) GT_HIP(ctx, rb.add(&k->nokeyt_n, k->nokeyt_count.p, sizeof(uint32_t)));
            GT_HIP(ctx, rb.add(&n_fb, k->fb_count.p, sizeof(uint32_t)));
            GT_HIP(ctx, rb.add(&n_unproven, k->unproven.p, sizeof(uint32_t)));
            GT_HIP(ctx, rb.sync());
        }
        if (main_prec == 2 && fast_auto) {
            // verdict for this point set: the wide error bound of the single chain must leave (almost) every row
            // provably complete, otherwise the repairs would cost more than the split chains
            const bool ok = double(n_unproven) <= kFastFailFrac * double(nq);
            if (nq >= 4096) ctx->fast_ok = ok ? 1 : 0;
            if (!ok) {
                main_prec = 1;
                GT_HIP(ctx, hipMemsetAsync(k->fb_count.p, 0, sizeof(uint32_t), ctx->stream));
                GT_HIP(ctx, hipMemsetAsync(k->gflags.p, 0, sizeof(uint32_t), ctx->stream));
                continue;
            }
        }
        break;
    }
    k->n_fallback = n_fb;
    // repairs run on the accurate arithmetic of the working copy
    ra.err = gt_err_model(ctx, ctx->prec);
    ra.unproven = nullptr;
    if (n_fb > 0) {
        // Repair at MFMA speed: radius-mode candidate pass around each flagged query's need_m-th candidate key, exact
        // keys + (key, index) selection on what it collected (gt_rerank.hip).  The exhaustive kernel is the last
        // resort (a collected list that would need every database row).
        GT_HIP(ctx, k->fb_max.reserve(2 * sizeof(uint32_t)));
        const float* Qp_sel = external ? k->Qp.as<float>() : ctx->Yp.as<float>();
        int64_t cap = 1024;
        int64_t off = 0;
        while (off < int64_t(n_fb)) {
            // batch rows so that lists (8 B) + float64 scratch (8 B) stay within ~4 GiB
            int64_t rows = std::min<int64_t>(int64_t(n_fb) - off, std::max<int64_t>(1, (int64_t(4) << 30) / (cap * 16)));
            const int64_t rows_pad = ceil_div64(rows, bq) * bq;
            GT_HIP(ctx, k->fb_qrows.reserve(size_t(rows) * sizeof(int32_t)));
            GT_HIP(ctx, k->fb_thr.reserve(size_t(rows) * sizeof(float)));
            GT_HIP(ctx, k->fb_counts.reserve(size_t(rows_pad) * sizeof(uint32_t)));
            GT_HIP(ctx, k->fb_lists.reserve(size_t(rows_pad) * size_t(cap) * sizeof(uint64_t)));
            uint32_t host_max[2] = {0, 0};
            {
                StageSpan span(ctx, "fallback", 3);
                GT_TRY(gt_launch_fallback_thr(ctx, ra, rows, off, k->fb_qrows.as<int32_t>(), k->fb_thr.as<float>()));
                SelectArgs fa;
                fa.dp = ctx->DP;
                fa.prec = ctx->prec;   // accurate arithmetic (thresholds from ra.err above)
                fa.mode = 1;
                fa.Yp = ctx->Yp.as<float>();
                fa.hneg = ctx->hneg.as<float>();
                fa.n_pad = ctx->n_pad;
                fa.Qp = Qp_sel;
                fa.qrows = k->fb_qrows.as<int32_t>();
                fa.q0 = 0;
                fa.nq = int32_t(rows);
                fa.lists = k->fb_lists.as<uint64_t>();
                fa.counts = k->fb_counts.as<uint32_t>();
                fa.thr_in = k->fb_thr.as<float>();
                fa.cap = int32_t(cap);
                GT_TRY(gt_launch_select(ctx, fa));
                GT_HIP(ctx, hipMemsetAsync(k->fb_max.p, 0, 2 * sizeof(uint32_t), ctx->stream));
                GT_TRY(gt_launch_max_u32(ctx, k->fb_counts.as<uint32_t>(), rows, k->fb_max.as<uint32_t>()));
                GT_HIP(ctx, hipMemcpyAsync(host_max, k->fb_max.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
            if (ctx->dbg_select & 2048)
                fprintf(stderr, "[gt] repair batch: %lld rows, capacity %lld, longest collected list %u\n", (long long)rows,
                        (long long)cap, host_max[0]);
            if (int64_t(host_max[0]) > cap) {
                if (cap >= ctx->n_pad) GT_FAIL(ctx, GT_E_STATE, "fallback: inconsistent collected count");
                cap = std::min<int64_t>(ctx->n_pad, std::max<int64_t>(cap * 8, int64_t(host_max[0]) + 64));
                continue;   // retry this batch with the larger capacity
            }
            if (cap >= ctx->n_pad / 2) {
                // degenerate (nearly every row ties): the exhaustive kernel is as good as anything
                GT_HIP(ctx, k->fb_scratch.reserve(size_t(rows) * size_t(ctx->n) * sizeof(double)));
                StageSpan span(ctx, "fallback");
                GT_TRY(gt_launch_fallback(ctx, ra, rows, off, k->fb_scratch.as<double>()));
                k->n_fallback_exhaustive += rows;
            } else {
                GT_HIP(ctx, k->fb_scratch.reserve(size_t(rows) * size_t(cap) * sizeof(double)));
                StageSpan span(ctx, "fallback");
                GT_TRY(gt_launch_collected_select(ctx, ra, rows, off, k->fb_lists.as<uint64_t>(), k->fb_counts.as<uint32_t>(),
                                                  int(cap), k->fb_scratch.as<double>(), k->fb_max.as<uint32_t>() + 1));
                GT_HIP(ctx, hipMemcpyAsync(host_max, k->fb_max.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (host_max[1] > 0) {
                    // never expected: fall back to the exhaustive kernel for the whole batch rather than trust it
                    GT_HIP(ctx, k->fb_scratch.reserve(size_t(rows) * size_t(ctx->n) * sizeof(double)));
                    GT_TRY(gt_launch_fallback(ctx, ra, rows, off, k->fb_scratch.as<double>()));
                    k->n_fallback_exhaustive += rows;
                }
            }
            off += rows;
        }
    }
    return GT_OK;
}

// An orthonormal basis of (approximately) the k leading principal directions of the symmetric m x m matrix C (row-major):
// orthogonal iteration Q <- orth(C Q), a few steps from the k coordinate axes of largest variance.  Stage one only needs an
// ORTHONORMAL frame that keeps most of the variance - any such frame gives valid partial distances, the results never depend
// on it - so the directions need not be converged eigenvectors: 8 steps of 64 x 64 x 16 cost ~0.1 ms on the host where the
// cyclic Jacobi sweeps over the full 64 x 64 problem took 2 ms (round 5's timeline of the manifold set: a 2.0 ms hole in the
// stream behind sample_cov_kernel).  Q: [m][k] row-major, columns orthonormal (modified Gram-Schmidt, twice).
static void leading_subspace(const std::vector<double>& C, int m, int k, std::vector<double>& Q) {
    std::vector<int> order(m);
    for (int i = 0; i < m; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return C[size_t(x) * m + x] > C[size_t(y) * m + y]; });
    Q.assign(size_t(m) * k, 0.0);
    for (int c = 0; c < k; ++c) Q[size_t(order[c]) * k + c] = 1.0;
    std::vector<double> Z(size_t(m) * k);
    auto orthonormalise = [&](std::vector<double>& A) {
        for (int pass = 0; pass < 2; ++pass)
            for (int c = 0; c < k; ++c) {
                for (int p = 0; p < c; ++p) {
                    double dot = 0.0;
                    for (int r = 0; r < m; ++r) dot += A[size_t(r) * k + p] * A[size_t(r) * k + c];
                    for (int r = 0; r < m; ++r) A[size_t(r) * k + c] -= dot * A[size_t(r) * k + p];
                }
                double nn = 0.0;
                for (int r = 0; r < m; ++r) nn += A[size_t(r) * k + c] * A[size_t(r) * k + c];
                if (!(nn > 1e-300)) {   // (a direction without variance: any unit vector orthogonal to the others will do)
                    for (int r = 0; r < m; ++r) A[size_t(r) * k + c] = 0.0;
                    A[size_t(order[c]) * k + c] = 1.0;
                    nn = 1.0;
                    if (pass == 0) continue;   // (the second pass orthogonalises it)
                }
                const double inv = 1.0 / std::sqrt(nn);
                for (int r = 0; r < m; ++r) A[size_t(r) * k + c] *= inv;
            }
    };
    for (int it = 0; it < 8; ++it) {
        for (int r = 0; r < m; ++r)
            for (int c = 0; c < k; ++c) {
                double acc = 0.0;
                for (int j = 0; j < m; ++j) acc += C[size_t(r) * m + j] * Q[size_t(j) * k + c];
                Z[size_t(r) * k + c] = acc;
            }
        orthonormalise(Z);
        Q.swap(Z);
    }
}

// The frame of stage one (gt_sym.hip "stage-one subspace"): the 16 leading principal directions of a row sample, or - the
// option off, or no more features than columns - the first 16 coordinate axes.  P_host: [16][64] floats, rows orthonormal.
static int stage_one_frame(gt_ctx* ctx, float* P_host, bool principal) {
    KnnWork* k = ctx->knn;
    const int d = ctx->d;
    std::fill(P_host, P_host + 16 * 64, 0.f);
    if (!principal || d <= 16 || d > 64) {
        for (int c = 0; c < 16 && c < d; ++c) P_host[c * 64 + c] = 1.f;
        return GT_OK;
    }
    const int64_t ns = std::min<int64_t>(ctx->n, 32768), step = std::max<int64_t>(1, ctx->n / ns);
    GT_HIP(ctx, k->sym_cov.reserve((64 * 64 + 64) * sizeof(double)));
    double* C_dev = k->sym_cov.as<double>();
    GT_TRY(gt_sym_sample_cov(ctx, step, ns, C_dev + 64 * 64, C_dev));
    std::vector<double> C64(64 * 64);
    GT_HIP(ctx, hipMemcpyAsync(C64.data(), C_dev, 64 * 64 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<double> A(size_t(d) * d), V;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) A[size_t(i) * d + j] = 0.5 * (C64[i * 64 + j] + C64[j * 64 + i]);
    leading_subspace(A, d, 16, V);
    // (what the partial distances rely on is the frame's orthonormality, nothing else: checked, the coordinate axes otherwise)
    double worst = 0.0;
    for (int a = 0; a < 16; ++a)
        for (int b = a; b < 16; ++b) {
            double dot = 0.0;
            for (int m = 0; m < d; ++m) dot += V[size_t(m) * 16 + a] * V[size_t(m) * 16 + b];
            worst = std::max(worst, std::fabs(dot - (a == b ? 1.0 : 0.0)));
        }
    if (!(worst <= 1e-9)) {
        for (int c = 0; c < 16 && c < d; ++c) P_host[c * 64 + c] = 1.f;
        return GT_OK;
    }
    for (int c = 0; c < 16; ++c)
        for (int m = 0; m < d; ++m) P_host[c * 64 + m] = float(V[size_t(m) * 16 + c]);
    return GT_OK;
}

int gt_sym_two_stage_prepare(gt_ctx* ctx, const int32_t* perm, int64_t n_pad_s, const ErrModel& em, int need_m, SelectArgs& a,
                             bool cut_done) {
    KnnWork* k = ctx->knn;
    const int hd = 16;
    GT_HIP(ctx, k->sym_hh.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_thrh.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_gh.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_gminh.reserve(size_t(n_pad_s / 32) * sizeof(float)));
    GT_HIP(ctx, k->sym_rrow.reserve(size_t(n_pad_s) * sizeof(float)));
    GT_HIP(ctx, k->sym_qtot.reserve(4 * sizeof(uint32_t)));
    GT_HIP(ctx, k->sym_z.reserve(size_t(n_pad_s) * 16 * sizeof(_Float16)));
    GT_HIP(ctx, k->sym_p.reserve(16 * 64 * sizeof(float)));
    // the stage-one copy Z = scz * P x: |P x| <= |x| <= sqrt(ymax2), the scale keeps that inside float16's normal range;
    // its rounding residual per row: float16 (2^-11 relative per column) + the float32 projection
    double y2 = 0.0;
    GT_HIP(ctx, hipMemcpyAsync(&y2, ctx->ymax.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double ymax = std::sqrt(std::max(y2, 1e-300));
    const double scz = gt_f16_scale(ymax);
    const double Lz = 1.01 * (4.8828125e-4 + double(ctx->d + 8) * 1.1920928955078125e-7) * scz * ymax;
    const int64_t samples = 8192;
    // Frames in turn: the first 16 coordinate axes (free), then the 16 leading principal directions of a row sample
    // (a covariance pass and a 64 x 64 eigenproblem on the host: only when the axes do not separate).  The forecast
    // decides; with the two-stage collect forced on, the first frame offered is taken (select_sym_pca = 2: the
    // principal one).
    // (round 6: the principal frame FIRST wherever it exists - it keeps at least the variance the coordinate frame keeps, so a
    //  coordinate frame that would pass is a principal frame that passes, and the projection + forecast of a coordinate frame
    //  that fails - 0.8 ms on the manifold set - is not paid in front of it; the frame itself is a 0.3 ms matter now)
    const bool can_pca = ctx->sym_pca != 0 && ctx->d > 16 && ctx->d <= 64;
    bool accepted = false;
    for (int frame = can_pca ? 1 : 0; frame < (can_pca ? 2 : 1) && !accepted; ++frame) {
        float P_host[16 * 64];
        {
            StageSpan span(ctx, "sym_prepare");
            GT_TRY(stage_one_frame(ctx, P_host, frame == 1));   // (synchronises when it runs the PCA)
        }
        uint32_t flagged = 0;
        {
            StageSpan span(ctx, "sym_prepare");
            GT_HIP(ctx, hipMemcpyAsync(k->sym_p.p, P_host, sizeof(P_host), hipMemcpyHostToDevice, ctx->stream));
            GT_TRY(gt_sym_project(ctx, perm, n_pad_s, k->sym_p.as<float>(), scz, k->sym_z.p, k->sym_hh.as<float>()));
            GT_TRY(gt_sym_half_thresholds(ctx, perm, n_pad_s, k->thr_final.as<float>(), k->sym_hh.as<float>(), em, hd, scz, Lz,
                                          k->sym_thrh.as<float>(), k->sym_gh.as<float>(), k->sym_gminh.as<float>(),
                                          k->sym_rrow.as<float>()));
            if (ctx->sym_two_stage < 0)
                GT_TRY(gt_sym_two_probe(ctx, k->sym_z.p, hd, k->sym_hh.as<float>(), k->sym_thrh.as<float>(),
                                        k->sym_gh.as<float>(), samples, k->sym_qtot.as<uint32_t>()));
            GT_HIP(ctx, hipMemcpyAsync(&flagged, k->sym_qtot.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        }
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (P_host is on the stack: its upload is done too)
        if (ctx->sym_two_stage > 0) {
            accepted = true;
        } else {
            if (ctx->dbg_select & 2048)
                fprintf(stderr, "[gt] two-stage forecast (%s frame): %u of %lld sampled pairs pass stage one\n",
                        frame ? "principal" : "coordinate", flagged, (long long)samples);
            // the queue holds one pair in 8; forecast beyond one in 24: this frame is a poor filter on this point set
            accepted = int64_t(flagged) * 24 <= samples;
        }
        k->sym_frame = frame;
    }
    if (!accepted) {
        ctx->sym_two_ok = 0;
        return GT_OK;
    }
    if (!cut_done) {
        // going ahead: the orphans lose their thresholds (a handful of rows; the forecast above was made with them), the
        // forms derived from the thresholds are made again
        StageSpan span(ctx, "sym_prepare");
        // (a radius is "no longer small" from 1/16 of the typical distance between unrelated rows when stage one sees 16
        //  coordinates - a quarter of every distance on isotropic data -, from 1/4 in the principal frame)
        GT_TRY(gt_sym_orphan_cut(ctx, perm, k->thr_final.as<float>(), k->sym_farcnt.as<float>(), em, k->sym_racc.as<double>(), need_m,
                                 k->sym_frame == 1 ? 0.25 : 0.0625));
        GT_TRY(gt_sym_g_from_thr(ctx, n_pad_s, k->thr_final.as<float>(), k->hnegs.as<float>(), k->sym_g.as<float>(),
                                 k->sym_gmin.as<float>()));
        GT_TRY(gt_sym_half_thresholds(ctx, perm, n_pad_s, k->thr_final.as<float>(), k->sym_hh.as<float>(), em, hd, scz, Lz,
                                      k->sym_thrh.as<float>(), k->sym_gh.as<float>(), k->sym_gminh.as<float>(),
                                      k->sym_rrow.as<float>()));
    }
    a.sym.half_steps = 1;
    a.sym.hh = k->sym_hh.as<float>();
    a.sym.thrh = k->sym_thrh.as<float>();
    a.sym.gminh = k->sym_gminh.as<float>();
    a.sym.zrows = k->sym_z.as<float>();
    if (ctx->sym_two_skip != 0) {
        // unit skipping (round 6): balls of the groups of 32 sorted rows in the stage-one space, from the FINAL thresholds
        StageSpan span(ctx, "sym_prepare");
        const double u = 5.9604644775390625e-08;
        const double sc = scz * (1.0 + 1e-6);
        const double X2 = (sc * ymax + Lz) * (sc * ymax + Lz);
        const double dmax = (2.0 * double(hd + 8) * 1.5 + 4.0) * u * X2;   // (sym_half_thresholds_kernel's)
        GT_HIP(ctx, k->sym_zc.reserve(size_t(n_pad_s / 32) * 16 * sizeof(float)));
        GT_HIP(ctx, k->sym_zrn.reserve(size_t(n_pad_s / 32) * 2 * sizeof(float)));
        GT_TRY(gt_sym_z_balls(ctx, n_pad_s, k->sym_z.p, k->sym_gh.as<float>(), 4.0 * dmax, k->sym_zc.as<float>(), k->sym_zrn.as<float>()));
        a.sym.zc = k->sym_zc.as<float>();
        a.sym.zrn = k->sym_zrn.as<float>();
    }
    return GT_OK;
}

int gt_sym_launch_collect(gt_ctx* ctx, const SelectArgs& a) {
    if (a.sym.half_steps <= 0) return gt_launch_select(ctx, a);
    SelectArgs h = a;
    h.dp = 16;
    h.Yp = h.Qp = a.sym.zrows;
    return gt_launch_select(ctx, h);
}

int gt_sym_queue_prepare(gt_ctx* ctx, int64_t n_pad_s, SelectArgs& a) {
    KnnWork* k = ctx->knn;
    // one region per wave of the collect launch, sized so that all of them together hold about one pair in 8 - beyond
    // that stage one is not doing its job; a wave whose region is full spills into a shared area (4 M pairs)
    // (own-only collect of a rank: nblk query blocks against every tile)
    const bool own = a.sym.own_only != 0 && a.sym.nblk > 0;
    const int64_t nwaves = (own ? int64_t(a.sym.nblk) : ceil_div64(n_pad_s, 128 * GT_SEL_TWO_QT)) * a.sym.nseg * 4;
    const int64_t units = own ? int64_t(a.sym.nblk) * (128 * GT_SEL_TWO_QT / 64) * (n_pad_s / 32)
                              : (n_pad_s / 64) * (n_pad_s / 32) / 2 / std::max(1, a.sym.shard_world);
    int64_t rcap = 1024;   // (a wave next to the diagonal of a clustered set notes several hundred pairs)
    while (rcap < 8192 && rcap * nwaves < units / 8) rcap *= 2;
    if (ctx->sym_queue_cap > 0) rcap = ctx->sym_queue_cap;
    GT_HIP(ctx, k->sym_queue.reserve(size_t(nwaves) * size_t(rcap) * sizeof(uint2)));
    GT_HIP(ctx, k->sym_qcount.reserve(size_t(nwaves) * sizeof(uint32_t)));
    GT_HIP(ctx, hipMemsetAsync(k->sym_qcount.p, 0, size_t(nwaves) * sizeof(uint32_t), ctx->stream));
    // (row-sharded builds: the spill area holds EVERY unit of the rank's pieces - an overflow on one rank alone would send
    //  that rank to the one-stage kernel, whose pieces of the pair space are not the other ranks')
    const int64_t spill_cap = a.sym.shard_world > 1 ? std::min<int64_t>(units + 1024, (int64_t(1) << 31) - 1024)
                                                    : (ctx->sym_spill_cap > 0 ? ctx->sym_spill_cap : int64_t(1) << 22);
    GT_HIP(ctx, k->sym_qspill.reserve(size_t(spill_cap) * sizeof(uint2) + 16));
    a.sym.queue = k->sym_queue.as<uint2>();
    a.sym.qcount = k->sym_qcount.as<uint32_t>();
    a.sym.qcap = int32_t(rcap);
    // (the spill counter sits behind the spill entries)
    a.sym.qspill = k->sym_qspill.as<uint2>();
    a.sym.qspill_count = reinterpret_cast<uint32_t*>(k->sym_qspill.as<uint2>() + spill_cap);
    a.sym.qspill_cap = int32_t(spill_cap);
    GT_HIP(ctx, hipMemsetAsync(a.sym.qspill_count, 0, 2 * sizeof(uint32_t), ctx->stream));   // (+ the count of scored units)
    return GT_OK;
}

int gt_sym_queue_finish(gt_ctx* ctx, const SelectArgs& a, int64_t* entries, int* ok) {
    KnnWork* k = ctx->knn;
    *ok = 0;
    const int64_t nwaves = ((a.sym.own_only != 0 && a.sym.nblk > 0) ? int64_t(a.sym.nblk) : ceil_div64(a.n_pad, 128 * GT_SEL_TWO_QT)) *
                           a.sym.nseg * 4;
    const int64_t dense_cap = a.sym.shard_world > 1 ? nwaves * int64_t(a.sym.qcap) + a.sym.qspill_cap
                                                    : std::min<int64_t>(nwaves * int64_t(a.sym.qcap) + a.sym.qspill_cap, int64_t(1) << 25);
    GT_HIP(ctx, k->sym_qdense.reserve(size_t(dense_cap) * sizeof(uint2)));
    GT_HIP(ctx, k->sym_qtot.reserve(4 * sizeof(uint32_t)));
    GT_HIP(ctx, hipMemsetAsync(k->sym_qtot.p, 0, 4 * sizeof(uint32_t), ctx->stream));
    StageSpan span(ctx, "sym_cold");
    SelectArgs c = a;
    c.mode = 5;
    c.nq = int32_t(nwaves);
    c.lists = k->sym_qdense.as<uint64_t>();
    c.cap = int32_t(dense_cap);
    c.counts = k->sym_qtot.as<uint32_t>();
    GT_TRY(gt_launch_select(ctx, c));
    uint32_t tot[3] = {0, 0, 0};   // entries, fullest region, spill overflow
    uint32_t scored = 0;
    GT_HIP(ctx, hipMemcpyAsync(tot, k->sym_qtot.p, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipMemcpyAsync(&scored, a.sym.qspill_count + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    k->sym_units_scored = int64_t(scored);
    *entries = int64_t(tot[0]);
    if (ctx->dbg_select & 2048)
        fprintf(stderr, "[gt] two-stage queue: %u entries, fullest region %u, %lld regions of %d, dense capacity %lld\n", tot[0], tot[1],
                (long long)nwaves, a.sym.qcap, (long long)dense_cap);
    if (tot[2] != 0 || int64_t(tot[0]) > dense_cap) return GT_OK;   // entries were dropped: nothing was filed
    SelectArgs d = a;
    d.mode = 4;
    d.sym.queue = k->sym_qdense.as<uint2>();
    d.sym.qn = int32_t(tot[0]);
    GT_TRY(gt_launch_select(ctx, d));
    *ok = 1;
    return GT_OK;
}

int gt_prepare_queries(gt_ctx* ctx, const void* Y, int64_t m, int32_t y_on_device) {
    if (!Y || m <= 0) GT_FAIL(ctx, GT_E_ARG, "query matrix is empty");
    if (ctx->n <= 0 || !ctx->X) GT_FAIL(ctx, GT_E_STATE, "no points bound (call gt_set_points first)");
    if (ctx->DP == 0) GT_FAIL(ctx, GT_E_LIMIT, "kNN on the HIP path: n_features > 128 needs the euclidean metric and <= 2048 features");
    if (!ctx->knn) ctx->knn = new KnnWork();
    KnnWork* kw = ctx->knn;
    const size_t esz = ctx->dtype == GT_F32 ? 4 : 8;
    const int bq = gt_select_bq(ctx->DP);
    const int64_t mpad = ceil_div64(m, bq) * bq;
    GT_HIP(ctx, kw->Qraw.reserve(size_t(m) * ctx->d * esz));
    GT_HIP(ctx, hipMemcpyAsync(kw->Qraw.p, Y, size_t(m) * ctx->d * esz,
                               y_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (ctx->metric == 1) GT_TRY(gt_normalize_rows(ctx, kw->Qraw.p, kw->Qraw.p, m, ctx->d, ctx->dtype));
    GT_HIP(ctx, kw->Qp.reserve(size_t(mpad) * ctx->DP * sizeof(float)));
    GT_HIP(ctx, kw->qn.reserve(size_t(m) * sizeof(double)));
    if (ctx->prec == 1) {
        // the query matrix shares the database's power-of-two scale; re-scale both if it would overflow float16
        double qmax = 0.0;
        uint32_t nonfinite = 0;
        GT_TRY(gt_max_abs(ctx, kw->Qraw.p, m * int64_t(ctx->d), ctx->dtype, &qmax, &nonfinite));
        if (nonfinite) return gt_fail_nonfinite(ctx, nonfinite, ctx->dtype);
        if (qmax * ctx->sc >= 32768.0) {
            const double keep = ctx->maxabs;
            ctx->sc = gt_f16_scale(std::max(qmax, keep));
            GT_TRY(gt_prep_matrix(ctx, ctx->X, ctx->n, ctx->d, ctx->dtype, ctx->DP, ctx->n_pad, ctx->Yp.as<float>(),
                                  ctx->xn.as<double>(), ctx->hneg.as<float>(), ctx->ymax.as<double>(), 1, ctx->sc,
                                  ctx->lomax_dev.as<double>(), ctx->fast_mode != 0 ? ctx->Yc.p : nullptr,
                                  ctx->wide ? ctx->sel_idx.as<int32_t>() : nullptr, ctx->dsel,
                                  ctx->wide ? ctx->xn_sel.as<double>() : nullptr));
            double lo2 = 0.0;
            GT_HIP(ctx, hipMemcpyAsync(&lo2, ctx->lomax_dev.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->lomax = std::sqrt(lo2) / ctx->sc;
            ctx->fast_ok = -1;
        }
    }
    GT_HIP(ctx, kw->qlomax_dev.reserve(sizeof(double)));
    const bool want_hi = ctx->prec == 1 && ctx->fast_mode != 0;
    if (want_hi) GT_HIP(ctx, kw->Qc.reserve(size_t(mpad) * ctx->DP * sizeof(_Float16)));
    if (ctx->wide) GT_HIP(ctx, kw->qn_sel.reserve(size_t(m) * sizeof(double)));
    GT_TRY(gt_prep_matrix(ctx, kw->Qraw.p, m, ctx->d, ctx->dtype, ctx->DP, mpad, kw->Qp.as<float>(), kw->qn.as<double>(),
                          nullptr, nullptr, ctx->prec, ctx->sc, ctx->prec == 1 ? kw->qlomax_dev.as<double>() : nullptr,
                          want_hi ? kw->Qc.p : nullptr, ctx->wide ? ctx->sel_idx.as<int32_t>() : nullptr, ctx->dsel,
                          ctx->wide ? kw->qn_sel.as<double>() : nullptr));
    ctx->qlomax = 0.0;
    if (ctx->prec == 1) {
        double lo2 = 0.0;
        GT_HIP(ctx, hipMemcpyAsync(&lo2, kw->qlomax_dev.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->qlomax = std::sqrt(lo2) / ctx->sc;
    }
    return GT_OK;
}

extern "C" int gt_knn_stats(const gt_ctx* ctx, int64_t* out12) {
    if (!ctx || !out12) return GT_E_ARG;
    const KnnWork* k = ctx->knn;
    out12[0] = k && k->sym_used ? 1 : 0;
    out12[1] = k ? k->sym_overflow : 0;
    out12[2] = k ? k->n_fallback : 0;
    out12[3] = k ? k->n_fallback_exhaustive : 0;
    for (int i = 0; i < 8; ++i) out12[4 + i] = (k && k->sym_used) ? int64_t(k->sym_stat_host[i]) : 0;
    if (k && k->sym_used) out12[6] = k->sym_nseg;   // work items per query block of launch B
    // ((32 x 32) units stage one of the two-stage collect scored - the others were skipped by their balls - unless the list
    //  statistics of dbg_select bit 256 are on, whose slot this is)
    if (k && k->sym_used && k->sym_two_used && !k->sym_bound_used && !(ctx->dbg_select & 256)) out12[8] = k->sym_units_scored;
    if (k && k->sym_used) out12[5] = k->sym_two_used ? k->sym_cold_entries : 0;   // pairs the cold launch scored in full
    // (one-stage collect over listed walks: the (64 x 32) units its tiles hold - 16 per (256 x 128) tile)
    if (k && k->sym_used && k->sym_listed && !k->sym_two_used) out12[5] = k->sym_listed_tiles * 16;
    if (k && k->sym_used) out12[7] = (k->sym_two_used ? 1 : 0) | (k->sym_seed_dense ? 2 : 0) | (k->sym_cold_local_used ? 4 : 0) | (k->tab_sorted ? 8 : 0) |
                                      ((k->tab_sorted && ctx->graph && ctx->graph->pairs_fused) ? 16 : 0) | ((k->sym_listed && !k->sym_two_used) ? 32 : 0);   // bit 0: two-stage collect ran, bit 1: dense seeding kernel, bit 2: cold launch in the local frame, bit 3: tables by sorted position, bit 4: the affinity pass looked the destinations up
    if (k && k->sym_used) out12[4] = (k->sym_two_used && k->sym_bound_used) ? 1 : 0;   // units listed by cell bounds, no collect launch
    out12[10] = k ? k->sym_far : 0;                 // kept rows of launch A outside the neighbourhood cells (all points)
    return GT_OK;
}

extern "C" int gt_knn_search(gt_ctx* ctx, int64_t row0, int64_t row1, const void* Y, int64_t m, int32_t y_on_device,
                             int32_t k, int64_t* out_idx, double* out_dist, int32_t out_on_device, uint32_t* flags) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->reset_stages();
    if (!out_idx || !out_dist) GT_FAIL(ctx, GT_E_ARG, "gt_knn_search: output pointers are required");
    int64_t nq;
    bool external = (Y != nullptr);
    if (!ctx->knn) ctx->knn = new KnnWork();
    KnnWork* kw = ctx->knn;
    if (external) {
        GT_TRY(gt_prepare_queries(ctx, Y, m, y_on_device));
        kw = ctx->knn;
        nq = m;
        row0 = 0;
    } else {
        if (row0 < 0 || row1 > ctx->n || row1 <= row0) GT_FAIL(ctx, GT_E_ARG, "gt_knn_search: bad row range");
        nq = row1 - row0;
    }
    GT_TRY(gt_knn_candidates(ctx, row0, nq, external, k));
    int64_t* d_idx = out_idx;
    double* d_dist = out_dist;
    DevBuf tmp_i, tmp_d;
    if (!out_on_device) {
        GT_HIP(ctx, tmp_i.reserve(size_t(nq) * k * sizeof(int64_t)));
        GT_HIP(ctx, tmp_d.reserve(size_t(nq) * k * sizeof(double)));
        d_idx = tmp_i.as<int64_t>();
        d_dist = tmp_d.as<double>();
    }
    int rc = gt_launch_emit_knn(ctx, kw->cand_d2.as<double>(), kw->cand_j.as<uint32_t>(), kw->MP, nq, k, ctx->dtype,
                                ctx->metric, d_idx, d_dist);
    if (rc == GT_OK && !out_on_device) {
        rc = gt_copy_to_host(ctx, out_idx, d_idx, size_t(nq) * k * sizeof(int64_t));
        if (rc == GT_OK) rc = gt_copy_to_host(ctx, out_dist, d_dist, size_t(nq) * k * sizeof(double));
    }
    uint32_t fl = 0;
    if (rc == GT_OK) {
        hipError_t e = hipMemcpyAsync(&fl, kw->gflags.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) rc = GT_E_HIP;
    }
    hipError_t es = hipStreamSynchronize(ctx->stream);
    tmp_i.release();
    tmp_d.release();
    if (rc != GT_OK) return rc;
    if (es != hipSuccess) {
        ctx->set_error(std::string("stream sync: ") + hipGetErrorString(es));
        return GT_E_HIP;
    }
    if (kw->n_fallback > 0) fl |= GT_FLAG_FALLBACK_ROWS;
    if (flags) *flags = fl;
    return GT_OK;
}
