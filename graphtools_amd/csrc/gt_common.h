// Host-side plumbing shared by the translation units of libgraphtools_amd.so:
// context object, device buffers, error reporting, per-stage hipEvent timing.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <chrono>
#include <cstdlib>
#include <initializer_list>
#include <string>
#include <utility>
#include <vector>

#include "../../include/graphtools_amd.h"

// ---- error helpers ---------------------------------------------------------------------------
// (an allocation the device cannot satisfy is a LIMIT of the build, not a failure of the runtime: GT_E_LIMIT, with what was
//  being reserved - callers that have another route, e.g. TraditionalGraph's all-pairs path, take it)
#define GT_HIP(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            const bool _oom = (_e == hipErrorOutOfMemory);                                      \
            (ctx)->set_error(std::string(_oom ? "the working set of this build does not fit the GPU's memory - " : "") + \
                             std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" +   \
                             std::to_string(__LINE__) + ")");                                   \
            return _oom ? GT_E_LIMIT : GT_E_HIP;                                                \
        }                                                                                       \
    } while (0)

#define GT_TRY(expr)                 \
    do {                             \
        int _rc = (expr);            \
        if (_rc != GT_OK) return _rc; \
    } while (0)

#define GT_FAIL(ctx, code, msg)   \
    do {                          \
        (ctx)->set_error(msg);    \
        return (code);            \
    } while (0)

// Device memory of the library goes through a small process-wide cache (gt_devpool.cpp): a graph build holds ~15 GB
// of workspace at N = 1e6, and handing that back to the driver with hipFree only to hipMalloc it again for the next
// graph of the process costs hundreds of milliseconds.  Blocks of every size are parked per device when a
// context lets go of them (up to GT_POOL_MAX_GB, default 64) and reused for requests they fit within a factor of two;
// gt_release_cached_memory() / an allocation failure empties the cache.
hipError_t gt_pool_alloc(void** p, size_t bytes, size_t* got);
void gt_pool_free(void* p, size_t bytes);
// a batch of releases behind ONE device synchronisation (a context closing): gt_pool_free skips its own between the two calls
void gt_pool_quiesced_begin();
void gt_pool_quiesced_end();
// streams and events of closed contexts, parked per device and taken over by the next context (gt_devpool.cpp)
hipStream_t gt_handle_take_stream(int device, bool side);        // nullptr: none parked
void gt_handle_park_stream(int device, bool side, hipStream_t s);
hipEvent_t gt_handle_take_event(int device);                      // parked, or a new one
void gt_handle_park_events(int device, std::vector<hipEvent_t>& ev);

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        release();
        if (bytes == 0) bytes = 16;
        size_t want = (bytes + 255) & ~size_t(255);
        size_t got = 0;
        hipError_t e = gt_pool_alloc(&p, want, &got);
        if (e != hipSuccess) {
            p = nullptr;
            cap = 0;
            return e;
        }
        cap = got;
        return hipSuccess;
    }
    void release() {
        if (p) gt_pool_free(p, cap);
        p = nullptr;
        cap = 0;
    }
};

// Scratch buffers of one call: released on every way out of the scope (DevBuf itself has no destructor - most of them are
// long-lived members of the context - so a local one needs this, or an early GT_HIP return leaks it).
struct DevBufScope {
    DevBuf* bufs[6];
    int n = 0;
    DevBufScope(std::initializer_list<DevBuf*> l) {
        for (DevBuf* b : l)
            if (n < 6) bufs[n++] = b;
    }
    ~DevBufScope() {
        for (int i = 0; i < n; ++i) bufs[i]->release();
    }
    DevBufScope(const DevBufScope&) = delete;
    DevBufScope& operator=(const DevBufScope&) = delete;
};

struct StageAcc {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    int launches = 0;
};

struct GraphState;  // gt_sparse.hip
struct KnnWork;     // gt_knn.hip

struct gt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // created on first use: a launch that may overlap the main stream's (see side_launch_begin / end)
    hipEvent_t side_event = nullptr;
    // Small read-backs (counters, flags, totals) land in a pinned mailbox: copies into pageable memory are staged one host round
    // trip each (~17 us of idle stream per copy on the C3 timeline, three or four in a row at the read-back points of a build);
    // into pinned memory they queue behind each other and the one synchronisation that follows takes them all.
    unsigned char* mail = nullptr;   // 1 KB, hipHostMalloc on first use
    size_t mail_used = 0;
    bool mail_busy = false;          // a ReadBack group holds the mailbox
    void* mail_slot(size_t bytes) {
        if (!mail && hipHostMalloc(reinterpret_cast<void**>(&mail), 1024, hipHostMallocDefault) != hipSuccess) mail = nullptr;
        bytes = (bytes + 7) & ~size_t(7);
        if (!mail || mail_used + bytes > 1024) return nullptr;
        void* r = mail + mail_used;
        mail_used += bytes;
        return r;
    }
    std::string err;
    std::map<std::string, StageAcc> stages;
    std::vector<hipEvent_t> event_pool;
    int n_cu = 256;

    // ---- bound points (gt_set_points) ----
    const void* X = nullptr;  // device pointer, original dtype, n x d row-major
    DevBuf X_own;             // owned copy when the caller passed host memory
    int64_t n = 0;
    int32_t d = 0;
    int32_t dtype = GT_F32;
    int32_t DP = 0;      // padded feature count of the working copy
    int32_t prec = 1;    // working copy / candidate arithmetic: 0 = float32 MFMA, 1 = split-float16 MFMA (default)
    // prec 1 only - single-chain float16 main pass (hi planes, score error 2^-10 |x||y|) instead of the three
    // split chains: 0 never, 1 auto (try it, fall back to the split chains for this point set when more than
    // kFastFailFrac of the rows cannot be proven complete), 2 always.  Repairs always run on the split chains.
    int32_t fast_mode = 1;
    int32_t fast_ok = -1;        // auto: verdict for the bound points (-1 unknown, 0 no, 1 yes); reset by gt_set_points
    int32_t last_main_prec = 1;  // arithmetic of the last main candidate pass (0 f32, 1 split f16, 2 single f16)
    double sc = 1.0;     // power-of-two scale applied to the working copy (prec 1: max|x|*sc in [2^13, 2^14))
    double maxabs = 0.0; // max |x_ij| of the bound points
    double lomax = 0.0;  // prec 1: max over rows of |x - hi(x)|_2 (true units), the float16 rounding residual norm
    double qlomax = 0.0; //         same for the external query matrix of the current call
    DevBuf lomax_dev;
    // Wide data (more features than the candidate kernels hold in registers, euclidean metric): the candidate pass
    // runs on the `dsel` coordinates of largest variance.  The squared distance over a coordinate subset is a LOWER
    // bound of the full one, so every bound of the completeness proof holds with the partial norms (xn_sel) while the
    // float64 stages use the full vectors; how much survives the filter only decides how many rows need repairs.
    bool wide = false;
    int32_t dsel = 0;
    DevBuf sel_idx;      // int32 [dsel] selected columns
    DevBuf xn_sel;       // float64 [n] squared norms over the selected columns (== xn when !wide)
    DevBuf colstat;      // float64 [2 d] column sums / sums of squares
    DevBuf small_tmp;    // a few persistent bytes for scalar reductions (no hipMalloc / hipFree on the per-call paths:
                         // both can stall for seconds in a process that also runs RCCL)
    int32_t metric = 0;  // 0 euclidean, 1 cosine (points are row-normalised copies; distance = 1 - x.y)
    DevBuf X_norm;       // cosine: normalised points in the input dtype
    int32_t samp_stride = 32; // candidate pass: threshold-seeding phase over every samp_stride-th tile (<= 1: off)
    int32_t samp_keep = 0;    //   list budget of that phase (0: the number of neighbours wanted, at least 16)
    int32_t samp_end = -1;    //   list budget at the end of that phase (0: none, -1: same as samp_keep)
    int32_t query_order = 1;  // candidate pass: deal the query rows to workgroups grouped by nearest landmark (gt_order.hip)
    int32_t order_min_rows = 32768;  //   launches with fewer query rows (or fewer points) are not grouped
    int32_t order_outliers = 1;      //   rows far from every landmark get a cell of their own (gt_order.hip; 0: off)
    int32_t order_cell_rows = 244;   //   rows per landmark cell (L = n / order_cell_rows landmarks, 64 ... 8192; small cells: a cluster
                                     //   without a landmark of its own swells the cells it lands in - see the bound pass, gt_sym.hip)
    DevBuf land_Y, land_h, order_cell, order_rows, order_tmp;
    int32_t samp_trig = 0;    //   level 0: entries per half-list that trigger a cut (0: samp_keep / 2 + 24)
    int32_t samp2_level = 3;  //   second cut of the lists once 2^samp2_level / samp_stride of the tiles are seen (0: none)
    int32_t samp2_keep = 64;  //   to this many entries (at least 3 * samp_keep)
    int32_t nt8_max_need = 88;  // tables of up to this many neighbours use the 128-entry list budget (else 512)
    int32_t dense_rows = -1;  // exact graph from float32 distances, '+' rule: row-streaming form with the transposed half as a list (-1: from 16384 rows, 0 never, 1 always)
    int32_t dense_rows_fused = 1; //   its list of kept affinities comes out of the bandwidth pass (0: from a pass of its own)
    int64_t dense_rows_cap = 0;   //   entries its list of kept affinities may hold (0: 1024 per row, at least 2^24; beyond: the tile-pair form)
    int32_t narrow_mode = -1; // 128-row-workgroup candidate kernels: -1 auto (few query rows), 0 never, 1 whenever available
    int32_t dbg_select = 0;   // experiment switches forwarded to the candidate kernel (results invalid when set)
    // symmetric candidate pass for self queries over the whole point set (gt_sym.hip): -1 auto (large launches), 0 off, 1 on
    int32_t sym_mode = -1;
    int64_t sym_min_rows = 65536;
    int32_t sym_stride = 768;   //   threshold-seeding launch: every sym_stride-th tile besides the row's own neighbourhood (0: none; 384 until round 4's sweep, tools/seed_sweep.py)
    int32_t sym_cosine = 1;          //   the symmetric pass also serves the cosine metric (rows are normalised: the candidate stages are the euclidean ones)
    int32_t sym_sorted_points = 1;   //   symmetric pass: the exact stages read the points from a copy in cell-sorted order
    int32_t rerank_lanes4 = 1;  //   re-rank of the symmetric pass: four lanes per candidate row (float32 rows, d % 4 == 0, d <= 128)
    int32_t sym_dense_seed = 1; //   threshold-seeding launch: 1 = dense cell blocks, keys in registers (gt_seed.hip), 0 = streaming lists
    int32_t sym_cells_shard = 12;   //   ... at least this many in the row-sharded passes (gt_knn_shard.cpp)
    int32_t sym_cells = 8;     //   ... which is the rows of this many nearest cells (landmarks) of the block's own cells (12 until round 4's sweep),
    int32_t sym_max_nb = 384;   //   at most this many tiles
    int32_t sym_tcap = 512;     //   capacity of a row's candidate list in launch B (<= 512)
    double sym_sample_far = 1.0;    //   seeding sample: far-kept seeds per row from which the symmetric pass is given up at once
                                    //   (isotropic Gaussians 2.0 / 3.6, half the points in one blob 1.4; manifold 0.13, mixtures ~0:
                                    //   tools/ladder_probe.py) - what the full launch decides later at 0.5 once the bound pass and the
                                    //   two-stage forecast have failed too
    int32_t sym_ok = -1;        //   auto: 0 once the bound point set has overflowed the lists of launch B (reset by gt_set_points)
    int32_t sym_nseg = 0;       //   work items per query block in launch B (0: chosen to fill the last round of workgroups)
    int32_t sym_orphan_far = 0;  //  a row is an orphan when this many times its far-kept seeds reach the seeds wanted (0: off - the default since the end of round 4: the rule sent ~150 rows of C3 to the repair pass that the lists hold without it; 4 until then)
    double sym_radius_cut = 4.0; //  rows whose completeness radius (squared) exceeds this many times the mean are repaired directly
    int32_t sym_two_stage = -1; //   launch B scores half the features first (partial distances): -1 auto, 0 off, 1 on
    int32_t symm_bins = -1;     // single-rank symmetrisation through destination bins (gt_sparse.hip): -1 auto, 0 off, 1 on
    int32_t symm_key32 = 1;     //   per-row sorts of the symmetrisation on 32-bit keys where columns and positions fit (0: 64-bit keys)
    // row-sharded builds with knn_max: the counts of the reference's search-expansion loop travel through the host
    int32_t stage_counts_only = 0, stage_totals_valid = 0, stage_n = 0;
    int64_t stage_local[4] = {0, 0, 0, 0}, stage_totals[4] = {0, 0, 0, 0};
    int32_t dist_f64 = 0;       //   distances from the float64 keys in float64 whatever the points' dtype (option "distance_dtype")
    int32_t in_graph_build = 0; //   (set by gt_graph_build around its gt_graph_begin: every row is here, the tail is its own)
    int32_t symm_pairs = 2;     //   pair-resolved symmetrisation (gt_sparse.hip): every row settles its mutual pairs itself, only one-sided entries travel
                                //   (2: and the re-rank lays the tables out by sorted position for it - KnnWork::tab_sorted; 1: tables by row)
    int32_t symm_pairs_shard = 1;  //   ... also on the ranks of a row-sharded build, given the bandwidths of all rows (gt_graph_bandwidth_local / gt_graph_set_bandwidths)
    int32_t symm_pair_huge = 1; //   ... union rows beyond the register sorts are finished by a segmented sort (0: they refute the path: the general tail)
    int32_t keep_stages = 0;    //   (gt_graph_build's second attempt after a refutation: the stage timers are not reset)
    int32_t symm_pair_ok = 1;   //   ... not refuted for the bound point set (a union row beyond the register sorts)
    int32_t symm_bin_shift = 0; //   log2 of the rows per bin (0: 9, more from 2 M rows; development / tests: 8 ... 12)
    int32_t sym_cold_local = 1; //   the cold launch scores its units in the frame of their queries (gt_knn_select.hip sym_cold_local_kernel:
                                //   float16 roundings of (x - o) sc from the sorted float32 points) - the margin of the float16 chain shrinks
                                //   from |x||y| 2^-10 to cell size; 0: the compact copy in the global frame (rounds 2-4)
    int32_t sym_two_skip = 1;   //   two-stage collect: units whose balls in the stage-one space are too far apart are not scored (0: off)
    int32_t sym_listed = -1;    //   one-stage collect over listed walks when the bound pass leaves too many units but few tiles (gt_sym.hip
                                //   collect_lists_kernel): -1 auto (lists <= a quarter of the walks), 0 off, 1 whenever the lists exist
    int32_t sym_bounds = -1;    //   bound pass in front of the two-stage collect (cell balls): -1 auto / 1 on, 0 off
    int64_t sym_bound_cap = 0;  //   units the bound pass may leave before the collect launch runs instead (0: 4 M; tests)
    int32_t sym_pca = 1;        //   stage one scores the 16 leading principal directions (0: the first 16 features)
    int32_t sym_queue_cap = 0;  //   entries per wave region of the two-stage queue (0: sized from the problem; development / tests)
    int32_t sym_spill_cap = 0;  //   entries of the shared spill area behind the regions (0: 4 M; development / tests)
    int32_t sym_two_ok = -1;    //   verdict of the last launch for the bound point set (cold-path share), -1 unknown
    int32_t sym_shard_group = 32;   //   row-sharded launch B: query blocks per rotation step of the walk pieces
    int32_t order_outlier_cell = -1;   // id of the outlier cell of the last query order (its rows are orphans of the symmetric pass), -1: none
    int32_t order_L = 0;        // landmark cells of the last query order (gt_order.hip)
    // Cell-sorted renumbering (gt_points_cell_sort, gt_knn_shard.cpp): the bound points ARE the caller's points in the
    // cell-sorted order - row v of the context is the caller's row vperm[v], a landmark cell is a run of consecutive rows,
    // and gt_query_order answers with the identity.  Row-sharded builds own contiguous runs of cells that way (a rank's
    // neighbourhoods are its own rows); the tail writes the caller's column numbers (relabelled at the final sort).
    int32_t presorted = 0;
    int32_t presorted_L = 0;    //   cells of the renumbering
    int32_t order_has_thr0 = 0; //   the last query order left starting thresholds in its out_thr0 (not in the presorted case)
    DevBuf vperm, vcell;        //   int32 [n] row of the caller for every row of the context; uint32 [n] its cell (non-decreasing)
    int32_t cells_pending = 0;  //   gt_points_cells_begin has bound the points and assigned a share of them: gt_points_cells_finish is due
    int32_t cells_L = 0;        //     landmark cells of that assignment
    int32_t order_coherent = 1;      //   the landmarks are numbered so that neighbours in space are neighbours in number (gt_order.hip coherent_landmark_order)
    int32_t order_coherent_active = 0;   //   ... and the cell order in force was made that way
    DevBuf land_ord;                 //   its scratch
    DevBuf land_X, land_Yp, land_xn;   //   landmark rows of the sharded assignment: raw rows, their working copy, norms
    int64_t n_pad = 0;   // rows of the working copy (multiple of the db tile)
    DevBuf Yp;           // working copy: [n_pad] rows of 4*DP bytes (float32, or float16 hi plane | lo plane)
    DevBuf Yc;           // prec 1 with fast_mode: compact copy of the hi plane, [n_pad] rows of 2*DP bytes
    DevBuf xn;           // double [n]    squared row norms
    DevBuf hneg;         // float [n_pad] -sc^2 |y|^2/2 (-inf on pad rows)
    DevBuf ymax;         // float [1]     max row norm (as float bits, atomicMax on uint)
    float ymax_host = 0.f;

    int32_t dense_fused_rowsum = 1;   //   float32 matrices: row sums accumulated by the tile kernel (0: a pass of their own)
    DevBuf dense_degree, dense_bw;
    int64_t dense_n = 0;

    KnnWork* knn = nullptr;
    GraphState* graph = nullptr;
    void* landmark = nullptr;   // LandmarkState (gt_landmark.hip)
    void* pca = nullptr;        // PcaState (gt_pca.hip)

    // (every failure is reported through here: HIP's per-thread last-error slot is emptied with it, or the error would
    //  resurface in the launch check - hipGetLastError() - of the next, unrelated call on this thread)
    void set_error(const std::string& m) {
        err = m;
        (void)hipGetLastError();
    }

    hipEvent_t get_event() {
        if (!event_pool.empty()) {
            hipEvent_t e = event_pool.back();
            event_pool.pop_back();
            return e;
        }
        return gt_handle_take_event(device);
    }
    void reset_stages() {
        for (auto& kv : stages) {
            for (auto& s : kv.second.spans) {
                event_pool.push_back(s.first);
                event_pool.push_back(s.second);
            }
        }
        stages.clear();
    }
};

// RAII span: records a hipEvent pair on the ctx stream around the launches of one stage
struct StageSpan {
    gt_ctx* ctx;
    StageAcc* acc;
    hipEvent_t e0, e1;
    const char* nm;
    static bool trace_on() {   // GT_STAGE_TRACE=1: every stage is announced and waited for (development: which stage faults)
        static const bool on = std::getenv("GT_STAGE_TRACE") != nullptr;
        return on;
    }
    StageSpan(gt_ctx* c, const char* name, int launches = 1) : ctx(c), nm(name) {
        if (trace_on()) {
            (void)hipStreamSynchronize(c->stream);
            std::fprintf(stderr, "[gt_stage] > %s\n", name);
        }
        acc = &c->stages[name];
        e0 = c->get_event();
        e1 = c->get_event();
        acc->launches += launches;
        (void)hipEventRecord(e0, c->stream);
    }
    ~StageSpan() {
        if (trace_on()) {
            const hipError_t e = hipStreamSynchronize(ctx->stream);
            std::fprintf(stderr, "[gt_stage] < %s (%s)\n", nm, hipGetErrorString(e));
        }
        (void)hipEventRecord(e1, ctx->stream);
        acc->spans.emplace_back(e0, e1);
    }
};

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A group of small device -> host read-backs with ONE synchronisation: add() queues a copy into the context's pinned mailbox
// (straight into `dst` when the mailbox is full or missing), sync() waits for the stream and hands the values out.  The group
// lives on the stack of the function whose locals receive the values.
struct ReadBack {
    gt_ctx* ctx;
    struct Item {
        void* dst;
        const void* slot;
        size_t bytes;
    };
    Item items[8];
    int n = 0;
    bool owner = false;     // this group holds the mailbox (one at a time: a second group alive at once copies straight into `dst`)
    bool pending = false;   // copies were queued and not yet waited for
    explicit ReadBack(gt_ctx* c) : ctx(c) {
        // (round-5 advisor: the constructor used to reset the shared mailbox unconditionally - a helper opening its own group
        //  between another group's add() and sync() would have handed out the same slots)
        if (!c->mail_busy) {
            c->mail_busy = true;
            c->mail_used = 0;
            owner = true;
        }
    }
    ~ReadBack() {
        // an early return between add() and sync() (GT_HIP on a failed launch) must not leave copies in flight into the caller's
        // stack frame (the direct path) or into slots the next group will hand out again
        if (pending) (void)hipStreamSynchronize(ctx->stream);
        if (owner) ctx->mail_busy = false;
    }
    ReadBack(const ReadBack&) = delete;
    ReadBack& operator=(const ReadBack&) = delete;
    hipError_t add(void* dst, const void* dev, size_t bytes) {
        pending = true;
        void* slot = (owner && n < 8) ? ctx->mail_slot(bytes) : nullptr;
        if (!slot) return hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
        items[n++] = Item{dst, slot, bytes};
        return hipMemcpyAsync(slot, dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t sync() {
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        pending = false;
        if (e == hipSuccess)
            for (int i = 0; i < n; ++i) std::memcpy(items[i].dst, items[i].slot, items[i].bytes);
        n = 0;
        if (owner) ctx->mail_used = 0;
        return e;
    }
};

// GT_TRACE=1: host wall-clock trace of the coarse steps of a call (development aid; synchronises the stream)
struct HostTrace {
    gt_ctx* ctx;
    const char* what;
    bool on;
    std::chrono::steady_clock::time_point t0;
    HostTrace(gt_ctx* c, const char* w) : ctx(c), what(w) {
        static const bool enabled = std::getenv("GT_TRACE") != nullptr;
        on = enabled;
        if (on) {
            (void)hipStreamSynchronize(c->stream);
            t0 = std::chrono::steady_clock::now();
        }
    }
    ~HostTrace() {
        if (on) {
            (void)hipStreamSynchronize(ctx->stream);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::fprintf(stderr, "[gt_trace] %-28s %10.3f ms\n", what, ms);
        }
    }
};

// gt_prep.hip
int gt_prep_points(gt_ctx* ctx);
// gt_api.cpp: the two halves of gt_set_points
int gt_bind_points(gt_ctx* ctx, const void* X, int64_t n, int32_t d, int32_t dtype, int32_t on_device);
int gt_prep_bound_points(gt_ctx* ctx);
// choose the padded feature count for d (0 if unsupported)
int gt_choose_dp(int d);
