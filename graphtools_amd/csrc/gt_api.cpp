// Context management, point binding and the small device-memory helpers of the C ABI.
#include "gt_common.h"
#include "gt_hostcopy.h"
#include "gt_knn.h"
#include "gt_knn_select.h"

#include <cstdlib>

static thread_local std::string g_create_error;

void gt_free_knn_work(gt_ctx* ctx);     // gt_knn.hip
void gt_free_graph_state(gt_ctx* ctx);  // gt_sparse.hip
void gt_free_landmark_state(gt_ctx* ctx);  // gt_landmark.hip
void gt_free_pca_state(gt_ctx* ctx);       // gt_pca.hip

extern "C" {

int gt_abi_version(void) { return GT_ABI_VERSION; }

int gt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gt_ctx_create(int device, gt_ctx** out) {
    if (!out) return GT_E_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        return GT_E_HIP;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "device ordinal out of range";
        return GT_E_ARG;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return GT_E_HIP;
    }
    gt_ctx* ctx = new gt_ctx();
    ctx->device = device;
    ctx->stream = gt_handle_take_stream(device, false);
    e = ctx->stream ? hipSuccess : hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete ctx;
        return GT_E_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (const char* pr = std::getenv("GT_KNN_PRECISION")) {
        if (std::string(pr) == "f32") ctx->prec = 0;
        if (std::string(pr) == "f16") { ctx->prec = 1; ctx->fast_mode = 0; }
        if (std::string(pr) == "f16x1") { ctx->prec = 1; ctx->fast_mode = 2; }
        if (std::string(pr) == "auto") { ctx->prec = 1; ctx->fast_mode = 1; }
    }
    // test hook: group the query rows of launches of at least this many rows (default 32768)
    if (const char* mr = std::getenv("GT_QUERY_ORDER_MIN_ROWS")) ctx->order_min_rows = std::max(1, std::atoi(mr));
    // test hooks of the symmetric candidate pass (gt_sym.hip): force it on (1) / off (0), smallest launch, sample stride
    if (const char* v = std::getenv("GT_DBG_SELECT")) ctx->dbg_select = std::atoi(v);   // development switches
    if (const char* v = std::getenv("GT_SYMMETRIC")) ctx->sym_mode = std::atoi(v);
    if (const char* v = std::getenv("GT_SYM_TWO_STAGE")) ctx->sym_two_stage = std::atoi(v);
    if (const char* v = std::getenv("GT_SYM_MIN_ROWS")) ctx->sym_min_rows = std::max(1, std::atoi(v));
    if (const char* v = std::getenv("GT_SYM_DENSE_SEED")) ctx->sym_dense_seed = std::atoi(v) != 0 ? 1 : 0;
    if (const char* v = std::getenv("GT_SYM_STRIDE")) ctx->sym_stride = std::max(0, std::atoi(v));
    if (const char* v = std::getenv("GT_SYM_CELLS")) ctx->sym_cells = std::max(1, std::atoi(v));
    if (const char* v = std::getenv("GT_ORDER_CELL_ROWS")) ctx->order_cell_rows = std::max(32, std::atoi(v));
    *out = ctx;
    return GT_OK;
}

void gt_ctx_destroy(gt_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    gt_pool_quiesced_begin();   // (one device synchronisation for the ~100 buffers let go below)
    gt_free_knn_work(ctx);
    gt_free_graph_state(ctx);
    gt_free_landmark_state(ctx);
    gt_free_pca_state(ctx);
    ctx->reset_stages();
    gt_handle_park_events(ctx->device, ctx->event_pool);
    ctx->X_own.release();
    ctx->Yp.release();
    ctx->Yc.release();
    ctx->small_tmp.release();
    ctx->sel_idx.release();
    ctx->xn_sel.release();
    ctx->colstat.release();
    ctx->lomax_dev.release();
    ctx->xn.release();
    ctx->hneg.release();
    ctx->ymax.release();
    ctx->dense_degree.release();
    ctx->dense_bw.release();
    ctx->X_norm.release();
    for (DevBuf* b : {&ctx->land_Y, &ctx->land_h, &ctx->order_cell, &ctx->order_rows, &ctx->order_tmp, &ctx->vperm, &ctx->vcell, &ctx->land_X, &ctx->land_Yp, &ctx->land_xn, &ctx->land_ord}) b->release();
    if (ctx->mail) (void)hipHostFree(ctx->mail);
    if (ctx->side_event) (void)hipEventDestroy(ctx->side_event);
    if (ctx->side_stream) {
        (void)hipStreamSynchronize(ctx->side_stream);
        gt_handle_park_stream(ctx->device, true, ctx->side_stream);
    }
    gt_handle_park_stream(ctx->device, false, ctx->stream);
    gt_pool_quiesced_end();
    delete ctx;
}

const char* gt_last_error(const gt_ctx* ctx) {
    if (!ctx) return g_create_error.c_str();
    return ctx->err.c_str();
}

double gt_stage_ms(const gt_ctx* ctx, const char* stage) {
    if (!ctx || !stage) return -1.0;
    auto it = ctx->stages.find(stage);
    if (it == ctx->stages.end()) return -1.0;
    (void)hipStreamSynchronize(ctx->stream);
    double total = 0.0;
    for (auto& s : it->second.spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.first, s.second) == hipSuccess) total += ms;
    }
    return total;
}

int gt_stage_launches(const gt_ctx* ctx, const char* stage) {
    if (!ctx || !stage) return -1;
    auto it = ctx->stages.find(stage);
    if (it == ctx->stages.end()) return -1;
    return it->second.launches;
}

}  // extern "C"

// Everything gt_set_points does before the working copies are made: the caller's matrix bound (uploaded when it is host
// memory), finiteness check + max |x|, cosine normalisation, the padded feature count / wide-data column choice.
int gt_bind_points(gt_ctx* ctx, const void* X, int64_t n, int32_t d, int32_t dtype, int32_t on_device) {
    if (!ctx) return GT_E_ARG;
    if (!X || n <= 0 || d <= 0) GT_FAIL(ctx, GT_E_ARG, "gt_set_points: empty input");
    if (dtype != GT_F32 && dtype != GT_F64) GT_FAIL(ctx, GT_E_ARG, "gt_set_points: dtype must be GT_F32 or GT_F64");
    if (n >= (int64_t(1) << 31) - 1) GT_FAIL(ctx, GT_E_LIMIT, "gt_set_points: n must be < 2^31 - 1");
    GT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->reset_stages();
    ctx->presorted = 0;   // (a renumbering belongs to the points it was made for)
    ctx->order_coherent_active = 0;
    ctx->cells_pending = 0;
    const size_t esz = dtype == GT_F32 ? 4 : 8;
    if (on_device) {
        ctx->X = X;
    } else {
        GT_HIP(ctx, ctx->X_own.reserve(size_t(n) * d * esz));
        GT_TRY(gt_copy_from_host(ctx, ctx->X_own.p, X, size_t(n) * d * esz));
        ctx->X = ctx->X_own.p;
    }
    {
        // finiteness (the reference's NearestNeighbors.fit runs sklearn's check_array) + max |x| for the float16 scale
        uint32_t nonfinite = 0;
        GT_TRY(gt_max_abs(ctx, ctx->X, n * int64_t(d), dtype, &ctx->maxabs, &nonfinite));
        if (nonfinite) {
            ctx->n = 0;
            ctx->X = nullptr;
            return gt_fail_nonfinite(ctx, nonfinite, dtype);
        }
    }
    if (ctx->metric == 1) {
        // cosine: every later stage works on the row-normalised points (distance = 1 - xhat.yhat)
        GT_HIP(ctx, ctx->X_norm.reserve(size_t(n) * d * esz));
        GT_TRY(gt_normalize_rows(ctx, ctx->X, ctx->X_norm.p, n, d, dtype));
        ctx->X = ctx->X_norm.p;
    }
    ctx->n = n;
    ctx->fast_ok = -1;
    ctx->sym_ok = -1;
    ctx->symm_pair_ok = 1;
    ctx->sym_two_ok = -1;
    ctx->d = d;
    ctx->dtype = dtype;
    ctx->DP = gt_choose_dp_prec(d, ctx->prec);
    ctx->wide = false;
    ctx->dsel = 0;
    if (ctx->DP == 0 && d <= 2048) {   // the float64 stages keep one row per wave in LDS (64 KB)
        // more features than the candidate kernels hold: filter on the 128 columns of largest variance (gt_common.h)
        ctx->wide = true;
        ctx->DP = 128;
        GT_TRY(gt_select_columns(ctx, 128));
    }
    return GT_OK;
}

// ... and the working copies (norms, float16 planes, seeds) of the bound points
int gt_prep_bound_points(gt_ctx* ctx) {
    if (ctx->DP == 0) {
        // more than 2048 features: the exact dense path and landmark assignment still work on the raw points
        ctx->n_pad = 0;
        GT_HIP(ctx, ctx->xn.reserve(size_t(ctx->n) * sizeof(double)));
        GT_HIP(ctx, ctx->ymax.reserve(sizeof(double)));
        GT_TRY(gt_prep_matrix(ctx, ctx->X, ctx->n, ctx->d, ctx->dtype, 0, 0, nullptr, ctx->xn.as<double>(), nullptr,
                              ctx->ymax.as<double>(), 0, 1.0));
    } else {
        GT_TRY(gt_prep_points(ctx));
    }
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

extern "C" {

int gt_set_points(gt_ctx* ctx, const void* X, int64_t n, int32_t d, int32_t dtype, int32_t on_device) {
    GT_TRY(gt_bind_points(ctx, X, n, d, dtype, on_device));
    return gt_prep_bound_points(ctx);
}

int gt_last_knn_precision(const gt_ctx* ctx) { return ctx ? ctx->last_main_prec : GT_E_ARG; }

int gt_set_option(gt_ctx* ctx, const char* name, const char* value) {
    if (!ctx || !name || !value) return GT_E_ARG;
    const std::string k(name), v(value);
    if (k == "knn_precision") {
        if (v == "f32") {
            ctx->prec = 0;
        } else if (v == "f16") {          // split float16 (3 chains) only
            ctx->prec = 1;
            ctx->fast_mode = 0;
        } else if (v == "f16x1") {        // single float16 chain for the main pass, always
            ctx->prec = 1;
            ctx->fast_mode = 2;
        } else if (v == "auto") {         // single chain when the data tolerate it (default)
            ctx->prec = 1;
            ctx->fast_mode = 1;
        } else {
            GT_FAIL(ctx, GT_E_ARG, "knn_precision must be 'auto', 'f16x1', 'f16' or 'f32'");
        }
        ctx->n = 0;   // the working copy depends on the precision: points must be bound again
        return GT_OK;
    }
    if (k == "metric") {
        if (v == "euclidean")
            ctx->metric = 0;
        else if (v == "cosine")
            ctx->metric = 1;
        else
            GT_FAIL(ctx, GT_E_ARG, "metric must be 'euclidean' or 'cosine'");
        ctx->n = 0;   // points must be bound again
        return GT_OK;
    }
    if (k == "select_samp_stride") {
        ctx->samp_stride = std::atoi(value);
        return GT_OK;
    }
    if (k == "select_samp_end") {
        ctx->samp_end = std::atoi(value);
        return GT_OK;
    }
    if (k == "query_order") {
        if (v == "auto" || v == "on" || v == "1")
            ctx->query_order = 1;
        else if (v == "off" || v == "0")
            ctx->query_order = 0;
        else
            GT_FAIL(ctx, GT_E_ARG, "query_order must be 'auto' or 'off'");
        return GT_OK;
    }
    if (k == "select_nt8_max_need") {
        ctx->nt8_max_need = std::min(112, std::max(1, std::atoi(value)));
        return GT_OK;
    }
    if (k == "select_narrow") {
        ctx->narrow_mode = v == "auto" ? -1 : std::atoi(value);
        return GT_OK;
    }
    if (k == "symmetrize_pairs_shard") {
        ctx->symm_pairs_shard = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "symmetrize_pairs_huge") {
        ctx->symm_pair_huge = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "query_order_coherent") {
        ctx->order_coherent = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "query_order_outliers") {
        ctx->order_outliers = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "query_order_cell_rows") {
        ctx->order_cell_rows = std::atoi(value);
        return GT_OK;
    }
    if (k == "select_samp_keep") {
        ctx->samp_keep = std::atoi(value);
        return GT_OK;
    }
    if (k == "query_order_min_rows") {
        ctx->order_min_rows = std::max(1, std::atoi(value));
        return GT_OK;
    }
    if (k == "select_symmetric") {
        ctx->sym_mode = v == "auto" ? -1 : std::atoi(value);
        return GT_OK;
    }
    if (k == "distance_dtype") {
        const std::string v = value;
        if (v != "data" && v != "float64") GT_FAIL(ctx, GT_E_ARG, "distance_dtype must be 'data' or 'float64'");
        ctx->dist_f64 = v == "float64" ? 1 : 0;
        return GT_OK;
    }
    if (k == "dense_rows") {
        ctx->dense_rows = v == "auto" ? -1 : std::atoi(value);
        return GT_OK;
    }
    if (k == "dense_rows_fused") {
        ctx->dense_rows_fused = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "dense_rows_cap") {
        ctx->dense_rows_cap = std::atoll(value);
        return GT_OK;
    }
    if (k == "dense_fused_rowsum") {
        ctx->dense_fused_rowsum = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "symmetrize_pairs") {
        ctx->symm_pairs = std::min(2, std::max(0, std::atoi(value)));   // 2 (default): with the tables by sorted position
        return GT_OK;
    }
    if (k == "select_sym_cosine") {
        ctx->sym_cosine = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "select_sym_sorted_points") {
        ctx->sym_sorted_points = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "rerank_lanes4") {
        ctx->rerank_lanes4 = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "select_sym_dense_seed") {
        ctx->sym_dense_seed = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "select_sym_stride") {
        ctx->sym_stride = std::max(0, std::atoi(value));
        return GT_OK;
    }
    if (k == "select_sym_orphan_far") {
        ctx->sym_orphan_far = std::max(0, std::atoi(value));
        return GT_OK;
    }
    if (k == "select_sym_two_stage") {
        ctx->sym_two_stage = v == "auto" ? -1 : std::atoi(value);
        return GT_OK;
    }
    if (k == "symmetrize_bins") {
        ctx->symm_bins = v == "auto" ? -1 : std::atoi(value);
        return GT_OK;
    }
    if (k == "symmetrize_key32") {
        ctx->symm_key32 = std::atoi(value);
        return GT_OK;
    }
    if (k == "symmetrize_bin_shift") {
        const int sh = std::atoi(value);
        if (sh != 0 && (sh < 8 || sh > 12)) return GT_E_ARG;
        ctx->symm_bin_shift = sh;
        return GT_OK;
    }
    if (k == "select_sym_cold_local") {
        ctx->sym_cold_local = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "select_sym_two_skip") {
        ctx->sym_two_skip = std::atoi(value) != 0 ? 1 : 0;
        return GT_OK;
    }
    if (k == "select_sym_listed") {
        ctx->sym_listed = v == "auto" ? -1 : std::atoi(value);
        return GT_OK;
    }
    if (k == "select_sym_bounds") {
        ctx->sym_bounds = v == "auto" ? -1 : std::atoi(value);
        return GT_OK;
    }
    if (k == "select_sym_bound_cap") {
        ctx->sym_bound_cap = std::max<long long>(0, std::atoll(value));
        return GT_OK;
    }
    if (k == "select_sym_pca") {
        ctx->sym_pca = std::atoi(value);
        return GT_OK;
    }
    if (k == "select_sym_spill_cap") {
        ctx->sym_spill_cap = std::max(0, std::atoi(value));
        return GT_OK;
    }
    if (k == "select_sym_queue_cap") {
        ctx->sym_queue_cap = std::max(0, std::atoi(value));
        return GT_OK;
    }
    if (k == "select_sym_shard_group") {
        ctx->sym_shard_group = std::max(1, std::atoi(value));
        return GT_OK;
    }
    if (k == "select_sym_nseg") {
        ctx->sym_nseg = std::min(8, std::max(0, std::atoi(value)));
        return GT_OK;
    }
    if (k == "select_sym_cells") {
        ctx->sym_cells = std::min(32, std::max(1, std::atoi(value)));
        return GT_OK;
    }
    if (k == "select_sym_min_rows") {
        ctx->sym_min_rows = std::atoll(value);
        return GT_OK;
    }
    if (k == "select_sym_tcap") {
        ctx->sym_tcap = std::min(512, std::max(64, std::atoi(value)));
        return GT_OK;
    }
    if (k == "dbg_select") {
        ctx->dbg_select = std::atoi(value);
        return GT_OK;
    }
    GT_FAIL(ctx, GT_E_ARG, "unknown option");
}

int gt_dev_alloc(gt_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) {
        ctx->set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
        return GT_E_ALLOC;
    }
    return GT_OK;
}

int gt_dev_free(gt_ctx* ctx, void* p) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GT_HIP(ctx, hipFree(p));
    return GT_OK;
}

int gt_dev_upload(gt_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    return gt_copy_from_host(ctx, dst_dev, src_host, bytes);
}

int gt_dev_download(gt_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    return gt_copy_to_host(ctx, dst_host, src_dev, bytes);
}

int gt_dev_sync(gt_ctx* ctx) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

// Order the context's stream against a stream of the caller (a collective library's, torch's current stream) without
// blocking the host: direction 0 = the context's later work waits for what the caller's stream holds now, 1 = the caller's
// stream waits for what the context has queued so far.
int gt_stream_order(gt_ctx* ctx, void* other_stream, int32_t direction) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t other = static_cast<hipStream_t>(other_stream);   // (nullptr: the legacy default stream)
    hipEvent_t ev;
    GT_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, direction == 0 ? other : ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(direction == 0 ? ctx->stream : other, ev, 0);
    (void)hipEventDestroy(ev);   // (released once the wait has been satisfied)
    GT_HIP(ctx, e);
    return GT_OK;
}

}  // extern "C"
