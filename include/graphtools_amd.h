/*
 * graphtools_amd.h - C ABI of the MI355X (gfx950) implementation of the graphtools
 * kNN -> affinity kernel -> diffusion operator hot path.
 *
 * The reference (KrishnaswamyLab/graphtools v2.1.0) is pure Python; its "FFI" for
 * this path is the set of numpy-level calls it makes into scikit-learn / scipy.
 * Every entry point below names the reference interface (file:line) it replaces.
 * The library is plain `extern "C"`: pointers + sizes, no C++/torch types.  It is
 * loaded with ctypes by graphtools_amd/_hip.py; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (GT_E_*); the message is
 *     available from gt_last_error(ctx) (or gt_last_error(NULL) for ctx creation).
 *   - one gt_ctx per thread of control; a ctx is bound to one HIP device and owns
 *     one HIP stream plus all device workspace.  Calls are blocking.
 *   - "dev" pointers are HIP device pointers on the ctx's device; "host" pointers
 *     are ordinary process memory.  Functions that accept either take an explicit
 *     `on_device` flag.
 *   - matrices are row-major and C-contiguous.  dtype: GT_F32 / GT_F64.
 *   - the C side never owns caller memory and never frees caller pointers.
 */
#ifndef GRAPHTOOLS_AMD_H
#define GRAPHTOOLS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GT_ABI_VERSION 1

/* status codes */
#define GT_OK 0
#define GT_E_ARG (-1)     /* invalid argument / unsupported configuration */
#define GT_E_HIP (-2)     /* HIP runtime error */
#define GT_E_ALLOC (-3)   /* device allocation failed */
#define GT_E_STATE (-4)   /* call sequence error */
#define GT_E_LIMIT (-5)   /* size outside what the HIP path supports */
#define GT_E_NONFINITE (-6) /* gt_set_points: the data contain NaN or infinity; gt_last_error carries the message of
                             * sklearn's check_array (raised by NearestNeighbors.fit in the reference, graphs.py:763-768) */

/* dtypes */
#define GT_F32 0
#define GT_F64 1

/* kernel symmetrisation (graphtools/base.py:557-577) */
#define GT_SYMM_NONE 0
#define GT_SYMM_ADD 1   /* (K + K^T) / 2 */
#define GT_SYMM_MUL 2   /* K o K^T */
#define GT_SYMM_MNN 3   /* theta*min(K,K^T) + (1-theta)*max(K,K^T) */

/* result flags (bit set in *flags outputs) */
#define GT_FLAG_DUPLICATES 1u      /* some distances[:,1] == 0  (graphs.py:787-817 -> RuntimeWarning on host) */
#define GT_FLAG_ZERO_DIAGONAL 2u   /* K has a zero on the diagonal (base.py:553-554) */
#define GT_FLAG_FALLBACK_ROWS 4u   /* informational: some rows took the exact fp64 fallback */
#define GT_FLAG_RADIUS_ROWS 8u     /* informational: some rows needed the radius pass */

/* which-selectors for gt_graph_fetch_* */
#define GT_CSR_K 0
#define GT_CSR_P 1
#define GT_VEC_BANDWIDTH 0
#define GT_VEC_DEGREE 1

typedef struct gt_ctx gt_ctx;

/* Parameters of kNNGraph(...)  (graphtools/graphs.py:608-672; api.Graph graphtools/api.py:14-42). */
typedef struct gt_knn_params {
    int32_t knn;               /* user knn (self excluded); internal k' = knn + 1 (graphs.py:783-784) */
    int32_t kernel_symm;       /* GT_SYMM_* */
    double decay;              /* alpha-decay exponent; NaN = None -> binary connectivity kernel (graphs.py:872-877) */
    double thresh;             /* affinity threshold (clamped to >= DBL_EPSILON when decay is set, graphs.py:628-629) */
    double bandwidth_scale;    /* graphs.py:892 */
    double theta;              /* mnn symmetrisation constant */
    double anisotropy;         /* base.py:579-592 */
    const double* bandwidth;   /* host pointer: NULL, 1 value (fixed) or n values (per row)   (graphs.py:886-897) */
    int64_t bandwidth_len;     /* 0, 1 or n */
    int64_t knn_max;           /* <= 0: None.  >0: at most knn_max+1 nearest columns per row (graphs.py:783, 869) */
} gt_knn_params;

/* ---- context ----------------------------------------------------------------------------- */
int gt_abi_version(void);
int gt_ctx_create(int device, gt_ctx** out);
void gt_ctx_destroy(gt_ctx* ctx);
const char* gt_last_error(const gt_ctx* ctx);
int gt_device_count(void);
/* Device memory released by contexts is parked in a process-wide cache (blocks >= 1 MiB, up to GT_POOL_MAX_GB = 64 GB
 * per device) so that the next context of the process does not pay hipMalloc for its workspace again; this hands
 * the parked blocks back to the driver. */
int gt_release_cached_memory(void);
/* per-stage GPU time (ms, hipEvent on the ctx stream) of the most recent call that ran `stage`;
 * stages: "prep" "query_order" "sym_prepare" "sym_seed" "sym_bound" "sym_cold" "knn_select" "rerank" "fallback" "radius"
 *         "affinity" "symmetrize" (with its parts "symm_bins" "symm_merge" "symm_huge" "symm_compact") "normalize" "renumber"
 *         "sym_exchange" "dense_bandwidth" "dense_kernel" "dense_normalize" "dense_rows_listed" "dense_rows_placed"
 *         "landmark" "landmark_assign" "pca_gram" "pca_matmul" "pca_tmatmul" "spmm".  Returns <0 if never run. */
double gt_stage_ms(const gt_ctx* ctx, const char* stage);
/* number of launches accumulated for `stage` in the most recent call (for roofline: ms / launches) */
int gt_stage_launches(const gt_ctx* ctx, const char* stage);

/* Options (call before gt_set_points).  Every name gt_set_option accepts is listed between the OPTIONS markers below, one
 * per line (tests/test_abi.py checks the list against the parser in csrc/gt_api.cpp in both directions, tests/test_gpu_dropin.py
 * sets each of them on the device); anything else returns GT_E_ARG "unknown option".  Results never depend on an option unless
 * its line says so: exact ordering and values always come from the float64 stages, rows whose candidate table cannot be proven
 * complete are repaired - the tuning switches only move time.  Values are decimal integers unless stated; "auto" where listed.
 * OPTIONS-BEGIN
 *   knn_precision            "auto" | "f16x1" | "f16" | "f32": arithmetic of the candidate pass.  auto (default): one float16 MFMA
 *                            chain on the high plane of every value when the bound data tolerate its wider score error bound
 *                            (judged after the first pass: at most 20 % of the rows left to the repair passes), else "f16" = three
 *                            chains on two float16 planes (2^-22); "f16x1" forces the single chain; "f32" float32 MFMA.
 *                            Environment variable GT_KNN_PRECISION sets the default.  Points must be bound again afterwards.
 *   metric                   "euclidean" | "cosine" (graphs.py:763-768 hands the metric to scikit-learn).  CHANGES RESULTS, by
 *                            definition.  Points must be bound again afterwards.
 *   distance_dtype           "data" | "float64".  CHANGES RESULTS, by design: "float64" takes the distances of a float32 point set
 *                            from the float64 keys unrounded (scipy pdist semantics, the exact graph built through this path)
 *                            instead of rounding them to float32 as scikit-learn's kneighbors does (graphs.py:883).
 *   query_order              "auto" | "off": deal the query rows of large launches to workgroups grouped by nearest landmark.
 *   query_order_min_rows     launches with fewer query rows are not grouped (32768).
 *   query_order_cell_rows    rows per landmark cell (244).
 *   query_order_outliers     0 | 1: rows far from every landmark get a cell of their own (1).
 *   query_order_coherent     0 | 1: the landmark cells are numbered so that neighbours in space are neighbours in number - the
 *                            sorted rows of a cluster lie together, a rank of a row-sharded build owns whole clusters (1).
 *   select_samp_stride       classic pass: threshold-seeding phase over every n-th tile (32; <= 1: off).
 *   select_samp_keep         list budget of that phase (0: the neighbours wanted, at least 16).
 *   select_samp_end          list budget at the end of that phase (-1: same as select_samp_keep, 0: none).
 *   select_nt8_max_need      tables of up to this many neighbours use the 512-slot lists (88), larger ones the 2048-slot lists.
 *   select_narrow            "auto" | 0 | 1: 128-row-workgroup candidate kernels (auto: launches with few query rows).
 *   select_symmetric         "auto" | 0 | 1: self queries over the whole point set score every unordered pair of rows once and
 *                            test the result for both rows (gt_sym.hip); 0 forces the classic pass (every query against every row).
 *   select_sym_min_rows      auto runs the symmetric pass from this many rows (65536).
 *   select_sym_stride        threshold-seeding launch: every n-th tile besides the row's own neighbourhood (768; 0: none).
 *   select_sym_cells         ... whose size is the rows of this many nearest cells (8; 1 ... 32).
 *   select_sym_dense_seed    0 | 1: dense cell-block seeding kernel with the keys in registers (gt_seed.hip) (1).
 *   select_sym_tcap          capacity of a row's candidate list (512; 64 ... 512).
 *   select_sym_nseg          work items per query block of the collect launch (0: chosen to fill the last round; <= 8).
 *   select_sym_shard_group   row-sharded collect: query blocks per rotation step of the walk pieces (32).
 *   select_sym_two_stage     "auto" | 0 | 1: the collect scores 16 leading directions first (partial distances).
 *   select_sym_pca           0 | 1: ... the 16 leading principal directions (1) or the first 16 features (0).
 *   select_sym_bounds        "auto" | 0 | 1: bound pass (cell balls) in front of the collect.
 *   select_sym_two_skip      0 | 1: two-stage collect: a (32 queries x 32 rows) unit whose two groups' balls in the stage-one space
 *                            are farther apart than either group asks for is not scored (1).
 *   select_sym_listed        "auto" | 0 | 1: when the bound pass leaves more units than its queue holds but its cell masks rule
 *                            out most tiles (auto: the listed tiles are at most a quarter of the walks), the one-stage collect
 *                            streams the listed tiles only - no stage-one copy, no cold launch.
 *   select_sym_bound_cap     units the bound pass may leave before the two-stage collect runs instead (0: 4 M).
 *   select_sym_queue_cap     entries per wave region of the two-stage queue (0: sized from the problem).
 *   select_sym_spill_cap     entries of the shared spill area behind the regions (0: 4 M).
 *   select_sym_cold_local    0 | 1: the cold launch scores its units in the frame of their queries (1).
 *   select_sym_orphan_far    a row whose far-kept seeds reach n times the seeds wanted is repaired (0: off).
 *   select_sym_cosine        0 | 1: the symmetric pass also serves the cosine metric (1).
 *   select_sym_sorted_points 0 | 1: the exact stages read a cell-sorted copy of the points (1).
 *   rerank_lanes4            0 | 1: re-rank of the symmetric pass with four lanes per candidate row (1).
 *   symmetrize_bins          "auto" | 0 | 1: single-rank symmetrisation through destination bins (gt_sparse.hip).
 *   symmetrize_bin_shift     log2 of the rows per bin (0: 9, more from 2 M rows; else 8 ... 12).
 *   symmetrize_key32         0 | 1: per-row sorts of the symmetrisation on 32-bit keys where columns and positions fit (1).
 *   symmetrize_pairs         0 | 1 | 2: pair-resolved symmetrisation of '+' builds - every row settles its mutual pairs itself,
 *                            only one-sided entries are transposed; 2 (default): with the tables by sorted position.
 *   symmetrize_pairs_shard   0 | 1: ... also on the ranks of a row-sharded build whose caller gathers the bandwidths
 *                            (gt_graph_bandwidth_local / gt_graph_set_bandwidths) (1; 0: gt_graph_bandwidth_local answers "no").
 *   symmetrize_pairs_huge    0 | 1: union rows beyond the register sorts are finished by a segmented sort (1) or refute the path.
 *   dense_rows               "auto" | 0 | 1: exact graph from float32 distances, '+' rule, in the row-streaming form (auto: from
 *                            16384 rows).
 *   dense_rows_fused         0 | 1: its list of kept affinities comes out of the bandwidth pass (1).
 *   dense_rows_cap           entries that list may hold (0: 1024 per row, at least 2^24; beyond it: the tile-pair form).
 *   dense_fused_rowsum       0 | 1: float32 matrices: row sums accumulated by the tile kernel (1).
 *   dbg_select               development switches (bit mask); INVALIDATES RESULTS for some bits.
 * OPTIONS-END */
int gt_set_option(gt_ctx* ctx, const char* name, const char* value);
/* arithmetic the most recent main candidate pass ran on: 0 float32, 1 split float16 (3 chains), 2 single float16 chain */
int gt_last_knn_precision(const gt_ctx* ctx);
/* statistics of the most recent kNN candidate search: out[0] = 1 if the symmetric pass ran, out[1] = rows whose
 * symmetric lists overflowed (repaired), out[2] = rows repaired in all, out[3] = of those by the exhaustive kernel,
 * out[4..11] = counters of the symmetric pass: [6] work items per query block, [5] (64 x 32) units scored in full (cold launch / listed one-stage collect), [7] bit 0 two-stage collect, bit 1 dense seeding kernel (gt_seed.hip), bit 2 cold launch in the local frame, bit 3 tables by sorted position, bit 4 destinations looked up by the affinity pass, bit 5 one-stage collect over listed walks, [9] tiles visited by the seeding launch,
 * the others list lengths (rerank_sym_kernel, only with dbg_select bit 256) */
int gt_knn_stats(const gt_ctx* ctx, int64_t* out12);

/* ---- points ------------------------------------------------------------------------------ */
/* Bind the data matrix (n x d).  Replaces NearestNeighbors(...).fit(data_nu) (graphs.py:763-768):
 * builds the padded working copies of the candidate pass (float16 planes or float32, see "knn_precision"),
 * float64 row norms and the -|y|^2/2 accumulator seeds on the device.
 * X may be a host or a device pointer; it must stay valid until the next gt_set_points / destroy
 * when it is a device pointer of dtype F32/F64 (the exact re-rank reads it). */
int gt_set_points(gt_ctx* ctx, const void* X, int64_t n, int32_t d, int32_t dtype, int32_t on_device);

/* ---- kNN search -------------------------------------------------------------------------- */
/* Replaces knn_tree.kneighbors(Y, n_neighbors=k) (graphs.py:883 / sklearn EuclideanArgKmin{32,64}):
 * queries are rows [row0,row1) of the bound points when Y == NULL, else the m x d matrix Y
 * (same dtype as the points, host or device per y_on_device).  Outputs (host or device per
 * out_on_device): out_idx int64 [m*k], out_dist float64 [m*k], ascending by (distance, index);
 * distances carry scikit-learn's rounding: float32 points -> (double)sqrtf((float)d2), float64 -> sqrt(d2).
 * Indices are bit-exact w.r.t. the fp64 GEMM-form ordering (ties broken by index). */
int gt_knn_search(gt_ctx* ctx, int64_t row0, int64_t row1, const void* Y, int64_t m, int32_t y_on_device,
                  int32_t k, int64_t* out_idx, double* out_dist, int32_t out_on_device, uint32_t* flags);

/* ---- kNN graph: kernel + diffusion operator ------------------------------------------------ */
/* Row-sharded builds with knn_max (graphs.py:916-976): the reference's search-expansion loop escalates while more than a tenth
 * of ALL rows still have their whole table inside their radius, so the ranks must agree on those counts.  Before
 * gt_graph_begin: every rank calls gt_graph_stage_counts (its counts for the loop's up to 4 steps; n_counts = 0: nothing to
 * exchange), the host sums them (all-reduce) and hands the sums to gt_graph_set_stage_totals.  Without the exchange a sharded
 * build with knn_max caps every row at knn_max. */
int gt_graph_stage_counts(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank, const int64_t* row_splits,
                          int64_t* counts4, int32_t* n_counts);
int gt_graph_set_stage_totals(gt_ctx* ctx, const int64_t* totals4, int32_t n_counts);
/* The three calls replace, for rows [row0,row1) of the graph,
 *   kNNGraph.build_kernel            graphs.py:771-785, 819-982 (+ _build_csr_from_neighbors :450-559)
 *   BaseGraph.symmetrize_kernel      base.py:557-577
 *   BaseGraph.apply_anisotropy       base.py:579-592
 *   BaseGraph.P (normalize l1)       base.py:629-646
 *   BaseGraph.kernel_degree          base.py:648-666
 * Row sharding: `row_splits` (world+1 ascending int64, host) gives the owner of every row; this
 * process is `rank` and builds rows [row_splits[rank], row_splits[rank+1]).  world == 1 is the
 * single-GPU case.
 *
 *   gt_graph_begin : kNN search, bandwidths, radius pass, affinities for the owned rows; counts the
 *                    transposed entries each peer will receive.  send_counts[world] (host, out) is in
 *                    units of 16-byte triplets {uint32 row, uint32 col, double value}.
 *   gt_graph_emit  : writes the transposed triplets, bucketed by destination rank in rank order, into
 *                    the caller's device buffer (sum(send_counts) * 16 bytes).  The caller moves them
 *                    (RCCL all-to-all over xGMI through torch.distributed; a no-op when world == 1).
 *   gt_graph_finish: merges the owned rows with the received triplets (n_recv of them, device pointer),
 *                    symmetrises, applies anisotropy, row-normalises.  Results stay on the device. */
int gt_graph_begin(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank,
                   const int64_t* row_splits, int64_t* send_counts);
int gt_graph_emit(gt_ctx* ctx, void* send_buf_dev);
int gt_graph_finish(gt_ctx* ctx, const void* recv_buf_dev, int64_t n_recv, int64_t* out_nnz, uint32_t* flags);
/* Anisotropy needs the degree (row sum) of EVERY column.  With world == 1 gt_graph_finish applies it
 * itself.  With world > 1 and anisotropy != 0 gt_graph_finish stops after the symmetrised kernel and its
 * degrees; the caller all-gathers the degree vectors (GT_VEC_DEGREE) into one n-vector on the device and
 * calls gt_graph_anisotropy, which rescales K (base.py:579-592) and then forms P. */
int gt_graph_anisotropy(gt_ctx* ctx, const double* degree_all_dev);
/* Row-sharded builds with the '+' rule, optional: the PAIR-RESOLVED TAIL on every rank.  BaseGraph.symmetrize_kernel
 * (base.py:557-577) forms (K + K.T) / 2; an entry K[i, j] whose transposed partner K[j, i] is kept too is a MUTUAL pair, and
 * whether it is follows from row j's bandwidth alone (the key row j holds for the pair is the same dot product in scikit-learn's
 * association with the roles swapped, written next to the table by the re-rank).  A rank that knows the bandwidths of ALL rows
 * therefore settles its mutual pairs itself, whichever rank the partner lives on; only the one-sided entries travel (44 M of
 * 116 M at BASELINE config 3), no pair ever meets its partner in a union row, and gt_graph_finish sorts each row once and
 * writes K, P and the degrees straight into the CSR.  Between gt_graph_shard_local and gt_graph_begin every rank calls
 *   gt_graph_bandwidth_local  the first half of gt_graph_begin (candidate tables, re-rank, bandwidths) for the rank's rows;
 *                             bw_local_dev (device, float64 [rows of the rank], out, on the library's stream).  applies (out)
 *                             = 0: the parameters are not this tail's ('*' / mnn rule, knn_max, anisotropy, a binary kernel,
 *                             symmetrize_pairs / symmetrize_pairs_shard = 0) - the same answer on every rank, nothing was done,
 *                             gt_graph_begin runs the whole build the general way
 *   -> the host all-gathers the slices into float64 [n] in the order of the context's rows (collective: 8 B per row)
 *   gt_graph_set_bandwidths   bw_all_dev (device; kept alive by the caller until gt_graph_begin has returned)
 * and then gt_graph_begin (same params / world / rank / row_splits: the second half) / gt_graph_emit / all-to-all /
 * gt_graph_finish as above.  gt_graph_begin without gt_graph_set_bandwidths finishes the build the general way.  A rank
 * whose tables carry no transposed keys (classic candidate pass, repaired rows) forms them from the dot products: every
 * rank can always follow, which is what lets the ranks decide without a vote.  K, P and the degrees equal the general
 * tail's bit for bit (x / 2 + y / 2 == (x + y) / 2 in binary floating point). */
int gt_graph_bandwidth_local(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank, const int64_t* row_splits,
                             double* bw_local_dev, int32_t* applies);
int gt_graph_set_bandwidths(gt_ctx* ctx, const double* bw_all_dev);
/* Row-sharded builds, optional first phase: the symmetric candidate pass (every unordered pair of rows scored once,
 * DESIGN.md 3.6) split over the ranks - each rank scores 1/world of the pairs instead of its rows against everything.
 * Every rank holds all points and calls, in order (a rank may stop after any call whose `applies` comes back 0, as long
 * as ALL ranks stop - the host all-reduces the flag - and gt_graph_begin then runs the classic candidate pass):
 *   gt_graph_sym_plan    applies (out): this context can run it for these parameters; n_pad_sorted, sorted_splits
 *                        [world + 1] (out): positions of the shared cell-sorted row order whose thresholds each rank seeds
 *   gt_graph_sym_seed    thr_local (device, float32 [sorted_splits[rank+1] - sorted_splits[rank]][2], out: the threshold
 *                        and the far-kept seed count of every position of the share), far_local (out),
 *                        radius_local [4] (out: sums and counts behind the orphan cut - the rows' completeness radii
 *                        and a sample of squared distances between unrelated rows)
 *                        -> the host all-gathers thr_local into float32 [n_pad_sorted][2] and sums far_local and
 *                           radius_local over the ranks
 *   gt_graph_sym_collect thr_all (device), far_total, radius_total [4]; applies (out), send_counts [world] (out, 16-byte records
 *                        {uint32 row local to its owner, uint32 0, uint64 candidate key})
 *   gt_graph_sym_emit    records bucketed by destination rank into the caller's device buffer
 *                        -> the host moves them with the all-to-all it uses for the triplets
 *   gt_graph_sym_finish  received records (device, n_recv of them) -> candidate lists of the owned rows
 * and then gt_graph_begin with the same params / world / rank / row_splits, which consumes the lists.  No reference
 * counterpart: the reference is single-process (its search is sklearn's kneighbors, graphs.py:883). */
int gt_graph_sym_plan(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank, const int64_t* row_splits,
                      int32_t* applies, int64_t* n_pad_sorted, int64_t* sorted_splits);
int gt_graph_sym_seed(gt_ctx* ctx, void* thr_local_dev, int64_t* far_local, double* radius_local);
int gt_graph_sym_collect(gt_ctx* ctx, const void* thr_all_dev, int64_t far_total, const double* radius_total, int32_t* applies,
                         int64_t* send_counts);
int gt_graph_sym_emit(gt_ctx* ctx, void* send_buf_dev);
int gt_graph_sym_finish(gt_ctx* ctx, const void* recv_buf_dev, int64_t n_recv);
/* Row-sharded builds, the default since round 4: CELL-SORTED RENUMBERING.  The reference's n_jobs = -1 (api.py:35,
 * graphs.py:763-768) spends every core on one kneighbors call; the multi-GPU counterpart splits the rows of K over the
 * ranks.  In the caller's row order a rank's rows lie all over the point set and every candidate pair is everybody's;
 * renumbered by landmark cell (the order the single-GPU pass sorts by anyway) a rank owns whole cells, its rows' candidate
 * pairs are found by the rank itself, and NOTHING is exchanged before the transposed triplets of the symmetrisation:
 *   gt_set_points          every rank binds ALL points (RCCL all-gather of the row slices first)
 *   gt_points_cell_sort    renumbers the bound points in cell-sorted order (deterministic: every rank arrives at the
 *                          same numbering); applied = 0: too few / too wide points - nothing changed, the caller's
 *                          numbering stays (the flow below still works, gt_graph_shard_local then declines)
 *   gt_points_shard_splits row_splits [world + 1] in the NEW numbering: runs of whole 1024-row blocks
 *   gt_graph_shard_local   candidate lists of the rank's own rows, no communication; applies = 0: gt_graph_begin runs
 *                          the classic candidate pass for the rank's rows instead (each rank decides for itself)
 *   gt_graph_begin / gt_graph_emit / all-to-all of the triplets / gt_graph_finish as above, in the new numbering
 *   gt_points_row_ids      the caller's row number of every row of the context (the rank's rows: [row0, row1) of
 *                          gt_graph_rows); the CSR's COLUMN indices are the caller's numbers already (relabelled in the
 *                          final per-row sort, so K, P and the degrees equal the single-GPU build's bit for bit)
 * Degree vectors handed to gt_graph_anisotropy / gt_graph_diff_aff are indexed by the caller's row numbers. */
int gt_points_cell_sort(gt_ctx* ctx, int32_t* applied);
/* ... with the cell assignment itself split over the ranks (it is 1 ms of replicated work at N = 1e6 otherwise):
 *   gt_points_cells_begin   instead of gt_set_points: binds the gathered points (device memory), assigns the rows
 *                           [row0, row1) - the rank's share, any split of [0, n) - to their landmark cells
 *                           -> cells_out_dev (uint32 [row1 - row0]); applied = 0: no cell order for these points, they are
 *                           bound as gt_set_points binds them and gt_points_cells_finish must not follow
 *                           -> the host all-gathers the ranks' cells (4 bytes per row)
 *   gt_points_cells_finish  cells of ALL rows (device uint32 [n]) -> the renumbering of gt_points_cell_sort */
int gt_points_cells_begin(gt_ctx* ctx, const void* X_dev, int64_t n, int32_t d, int32_t dtype, int64_t row0, int64_t row1,
                          void* cells_out_dev, int32_t* applied);
int gt_points_cells_finish(gt_ctx* ctx, const void* cells_all_dev);
int gt_points_shard_splits(gt_ctx* ctx, int32_t world, int64_t* out_splits);
int gt_points_row_ids(gt_ctx* ctx, int64_t row0, int64_t row1, int32_t* out, int32_t out_on_device);
int gt_graph_shard_local(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank, const int64_t* row_splits,
                         int32_t* applies);
/* device address of row `row0` of the bound points as the context holds them (its numbering; out_dtype / out_d optional) -
 * a rank's own rows as the device-resident query matrix of gt_knn_search on another context (random landmarking over
 * the ranks, graphs.py:1200-1213: dist.ShardedKnnGraph.random_landmark_clusters) */
int gt_points_device(gt_ctx* ctx, int64_t row0, void** out, int32_t* out_dtype, int32_t* out_d);
/* single-GPU convenience: begin + emit + finish with an internal buffer */
int gt_graph_build(gt_ctx* ctx, const gt_knn_params* params, int64_t* out_nnz, uint32_t* flags);

/* Tail of BaseGraph._build_kernel + BaseGraph.P (base.py:534-555, 557-592, 629-646) for a kernel the caller assembled
 * from device-built blocks - MNNGraph.build_kernel (graphs.py:1857-1936) composes per-batch kNN kernels and
 * cross-batch build_kernel_to_data blocks: host CSR (int64 indptr, int32 indices, float64 data, n x n, unique columns
 * per row, values >= 0) -> symmetrisation (kernel_symm, theta), anisotropy, row normalisation on the device.  Results
 * are fetched like those of gt_graph_build. */
/* Host helper of that composition (no device work): copies the rows of one CSR block given in batch-local ids into
 * their rows of the assembled kernel - entry e of local row r lands at cursor[rows_global[r]] + (e - indptr[r]) with
 * column cols_global[indices[e]] and value data[e] * scale[r] (scale may be NULL); cursor[] advances by the row
 * lengths.  Replaces the scipy COO concatenation + sort of the reference's assembly (graphs.py:1905-1936). */
int gt_host_place_block(int64_t nrows, const int64_t* indptr, const int32_t* indices, const double* data,
                        const int64_t* rows_global, const int64_t* cols_global, const double* scale, int64_t* cursor,
                        int32_t* out_indices, double* out_data);
int gt_csr_graph_build(gt_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* indices, const double* data,
                       int32_t kernel_symm, double theta, double anisotropy, int64_t* out_nnz, uint32_t* flags);

/* Out-of-sample kernel, replaces kNNGraph.build_kernel_to_data(Y) + the normalize() of DataGraph.extend_to_data
 * (graphs.py:819-982, base.py:1166-1193): rows = the m query points Y (same dtype / width as the bound points,
 * host or device), columns = the n bound points; `knn` neighbours (no +1), no symmetrisation.  K_yx and its
 * row-normalised form (the transition matrix) are fetched with gt_graph_fetch_csr(GT_CSR_K / GT_CSR_P). */
int gt_graph_extend(gt_ctx* ctx, const void* Y, int64_t m, int32_t y_on_device, const gt_knn_params* params,
                    int64_t* out_nnz, uint32_t* flags);

/* Results of the most recent gt_graph_finish, for the owned rows.
 * CSR is canonical (sorted columns, no duplicates): data float64 [nnz], indices int32 [nnz] (global
 * column ids), indptr int64 [rows+1].  Any of the three pointers may be NULL.  on_device selects
 * host or device destination memory. */
int gt_graph_rows(const gt_ctx* ctx, int64_t* row0, int64_t* row1, int64_t* nnz);
int gt_graph_fetch_csr(gt_ctx* ctx, int32_t which, double* data, int32_t* indices, int64_t* indptr,
                       int32_t on_device);
/* K and P of the owned rows to the host in one pass over the link (the host-complete build, SURVEY 8d: scipy CSR K and P
 * out): structure and K values are copied; the P values are derived from them on the host by the copy threads while later
 * chunks are on the link - P[e] = K[e] / degree[row], the division the device made for its own P (base.py:645 normalize l1;
 * bit-identical) - so the P values never cross PCIe.  All outputs are host arrays of the caller. */
int gt_graph_fetch_kp(gt_ctx* ctx, double* K_data, int32_t* indices, int64_t* indptr, double* P_data);

/* Dense copy of the owned rows of K or P: out[nloc][n_total], zeros where the sparse form has no entry.  out_dtype GT_F32 or
 * GT_F64.  Serves the exact graph built FROM POINTS (TraditionalGraph, graphtools/graphs.py:1546-1609: pdist -> bandwidth ->
 * exp(-(d/bw)^decay) -> entries below thresh zeroed): everything outside the thresh radius is exactly 0 there, so the build is
 * the kNN path's (candidate distances on the matrix cores, float64 refinement inside the radius only) and this call writes
 * the dense form the reference returns (linear zero fill + scatter: 6.7 TB/s at N = 2e5).  Option "distance_dtype" = "float64" makes a float32 point set's distances float64
 * (scipy's pdist converts to double) instead of scikit-learn's float32. */
int gt_graph_to_dense(gt_ctx* ctx, int32_t which, void* out, int32_t out_dtype, int32_t out_on_device);
int gt_graph_fetch_vec(gt_ctx* ctx, int32_t which, double* out, int32_t on_device);
/* out[owned rows][ncols] = (K or P)[owned rows, :] @ X[n][ncols]  (float64, row-major) with the operator left on the
 * device: the building block of the diffusion steps P^t X that consume `diff_op` downstream of the reference
 * (`graph.diff_op.dot(data)` in PHATE / MAGIC).  Entries of a row are accumulated in column
 * order with separate multiply and add, like scipy's csr @ dense.  on_device: X and out are device pointers. */
int gt_graph_spmm(gt_ctx* ctx, int32_t which, const double* X, int64_t ncols, double* out, int32_t on_device);
/* statistics of the last build: out[0]=rows that took the exact fallback, out[1]=rows that took the
 * radius pass, out[2]=nnz of the unsymmetrised kernel, out[3]=radius-pass capacity retries */
/* Replaces BaseGraph.diff_aff (base.py:668-698): D^-1/2 K D^-1/2 with D = the row sums of K, as values on the CSR
 * structure of K (gt_graph_fetch_csr).  degree_all_dev: device vector of the degrees of ALL rows (sharded builds, after
 * the all-gather) or NULL on a single rank.  out: float64 [nnz of the owned rows], host or device. */
int gt_graph_diff_aff(gt_ctx* ctx, const double* degree_all_dev, double* out, int32_t on_device);

int gt_graph_stats(const gt_ctx* ctx, int64_t* out4);

/* ---- exact dense graph (TraditionalGraph) -------------------------------------------------- */
/* Replaces TraditionalGraph.build_kernel (graphs.py:1514-1610) + symmetrise + P for a dense n x n
 * problem on one GPU.  `precomputed`:
 *   GT_PRECOMPUTED_NONE      X_or_D = n x d points (dtype F32/F64); distances are float64 difference form like
 *                            scipy pdist; K, P are float64.
 *   GT_PRECOMPUTED_DISTANCE  X_or_D = n x n distances; dtype is preserved (float32 D -> float32 K, P).
 *   GT_PRECOMPUTED_AFFINITY  X_or_D = n x n affinities: the unsymmetrised kernel IS the caller's matrix
 *                            (graphs.py:1532-1536); entries below `thresh` are zeroed (:1596-1609), then the common tail
 *                            (symmetrise, anisotropy, P); knn / decay / bandwidth are not used.
 *   GT_PRECOMPUTED_ADJACENCY the same with the diagonal set to 1 first (graphs.py:1537-1545).
 * bandwidth: NULL -> (knn+1)-th smallest entry of each row (graphs.py:1583-1587), else 1 or n values.
 * Outputs (device or host per out_on_device, any may be NULL): K [n*n], P [n*n] in the result dtype.
 * When `inplace` != 0 and the input is a device-resident n x n matrix, K overwrites it (for
 * problems where D alone fills most of HBM) and out_K is ignored. */
#define GT_PRECOMPUTED_NONE 0
#define GT_PRECOMPUTED_DISTANCE 1
#define GT_PRECOMPUTED_AFFINITY 2
#define GT_PRECOMPUTED_ADJACENCY 3
int gt_dense_graph_build(gt_ctx* ctx, const void* X_or_D, int64_t n, int32_t d, int32_t dtype, int32_t on_device,
                         int32_t precomputed, int32_t knn, double decay, double thresh,
                         const double* bandwidth, int64_t bandwidth_len, double bandwidth_scale,
                         int32_t kernel_symm, double theta, double anisotropy, int32_t inplace,
                         void* out_K, void* out_P, int32_t out_on_device, uint32_t* flags);

/* Replaces TraditionalGraph.build_kernel_to_data (graphs.py:1612-1678): the dense kernel from m new points Y (same
 * dtype and width as the points the last from-data gt_dense_graph_build bound) to those points.
 *   pdx = cdist(Y, X) in float64 difference form; bandwidth NULL -> the knn-th smallest entry of each row of pdx
 *   (:1654-1656), else 1 or m values; * bandwidth_scale; K = exp(-(pdx / bandwidth)^decay), NaN -> 1, < thresh -> 0.
 * out_K: float64 [m * n], host or device. */
int gt_dense_extend(gt_ctx* ctx, const void* Y, int64_t m, int32_t y_on_device, int32_t knn, double decay, double thresh,
                    const double* bandwidth, int64_t bandwidth_len, double bandwidth_scale, double* out_K,
                    int32_t out_on_device);

/* degree (row sums of K, GT_VEC_DEGREE) or bandwidth (GT_VEC_BANDWIDTH) of the last dense build, n float64 to host */
int gt_dense_fetch_vec(gt_ctx* ctx, int32_t which, double* out_host);

/* ---- landmark operator (LandmarkGraph) ------------------------------------------------------ */
/* Replaces LandmarkGraph._landmarks_to_data + the products of build_landmark_op
 * (graphs.py:1169-1182, 1232-1246) on the kernel of the most recent gt_graph_finish (owned rows).
 *   clusters : int32 [n] (host) dense labels in [0, n_landmark) for ALL n points (the caller applies
 *              np.unique(..., return_inverse=True), graphs.py:1170).
 * gt_landmark_build forms, for the owned rows n,
 *   pnm[n, c]  = sum_{j in cluster c} K[n, j]          (= pmn[c, n]; K is symmetric)
 *   transitions[n, :] = pnm[n, :] / sum_c pnm[n, c]     (CSR over owned rows x n_landmark, fetched separately)
 *   M[c, c']   = sum_n pnm[n, c] * transitions[n, c']   R[c] = sum_n pnm[n, c]      (float64 [L*L], [L])
 * M and R are partial sums over the owned rows; with world > 1 the caller sums them over ranks (RCCL
 * all-reduce) before gt_landmark_scale, which finishes landmark_op[c, :] = M[c, :] / R[c] in place. */
int gt_landmark_build(gt_ctx* ctx, const int32_t* clusters, int32_t n_landmark, double* out_M, double* out_R,
                      int32_t out_on_device, int64_t* out_transitions_nnz);
int gt_landmark_scale(gt_ctx* ctx, double* M_inout, const double* R, int32_t n_landmark, int32_t on_device);
/* transitions of the last gt_landmark_build: data float64 [nnz], indices int32 [nnz], indptr int64 [rows+1] */
int gt_landmark_fetch_transitions(gt_ctx* ctx, double* data, int32_t* indices, int64_t* indptr, int32_t on_device);
/* ---- tall thin float64 matrices on the device (spectral landmark front end) ------------------ */
/* The dense steps of sklearn.utils.extmath.randomized_svd(diff_aff, n_svd) (graphs.py:1215-1219) next to gt_graph_spmm,
 * on row-major n x k device matrices of at most 128 columns (allocate with gt_dev_alloc):
 *   gt_thin_scale_rows  A[r][:] *= v[r]^power            (v: device vector, e.g. the degrees with power -1/2)
 *   gt_thin_gram        out (k x k, host) = A^T A         (Cholesky-QR of a power-iteration block)
 *   gt_thin_rmul        B = A R, R (k x m) on the host, B a different device matrix */
int gt_thin_scale_rows(gt_ctx* ctx, double* A_dev, int64_t n, int32_t k, const double* v_dev, double power);
int gt_thin_gram(gt_ctx* ctx, const double* A_dev, int64_t n, int32_t k, double* out_host);
int gt_thin_rmul(gt_ctx* ctx, const double* A_dev, int64_t n, int32_t k, const double* R_host, int32_t m, double* B_dev);

/* ---- PCA pre-reduction (Data._reduce_data) ------------------------------------------------ */
/* Replaces the dense products of sklearn PCA(n_pca, svd_solver="randomized").fit(data) / .transform(data)
 * (base.py:227-294 -> sklearn.utils.extmath.randomized_svd: M @ Q, M.T @ Q on the centred n x d matrix).  The matrix
 * stays on the device; the thin factors (at most 128 columns) travel to the host, where the caller runs the reference's
 * own small LAPACK steps (graphtools_amd/_pca.py).  float32 data only (float64 data keeps sklearn's float64 path).
 *   gt_pca_begin    bind X (n x d float32; host memory is uploaded), column means and centred sums of squares (float64)
 *   gt_pca_matmul   thin[dst] = S W - 1 sub^T;  src 0: S = X (W is d x k), src 1/2: S = thin[src] (W is wrows x k);
 *                   W float64 row-major on the host, sub (k values, may be NULL) the centring term mean^T W
 *   gt_pca_tmatmul  out (d x k float64, host) = X^T thin[ybuf][:, :k];  colsum (k, may be NULL) = column sums of thin
 *   gt_pca_gram     out (k x k float64, host) = thin^T thin, accumulated in float64
 *   gt_pca_fetch    thin[ybuf][:, :k] as float32 [n][k] (host or device memory)
 *   gt_pca_end      release the workspace */
int gt_pca_begin(gt_ctx* ctx, const float* X, int64_t n, int32_t d, int32_t x_on_device, double* mean_out,
                 double* sumsq_centered_out);
int gt_pca_matmul(gt_ctx* ctx, int32_t src, const double* W, int32_t wrows, int32_t k, const double* sub, int32_t dst);
int gt_pca_tmatmul(gt_ctx* ctx, int32_t ybuf, int32_t k, double* out, double* colsum);
int gt_pca_gram(gt_ctx* ctx, int32_t ybuf, int32_t k, double* out);
int gt_pca_fetch(gt_ctx* ctx, int32_t ybuf, int32_t k, float* out, int32_t on_device);
int gt_pca_end(gt_ctx* ctx);

/* Random-landmark assignment (graphs.py:1200-1213): clusters[i] = argmin_j |x_i - x_{landmarks[j]}| over the
 * bound points, rows [row0,row1); ties -> lowest j.  mode 0: scipy cdist arithmetic (float64 difference
 * form), mode 1: sklearn euclidean_distances arithmetic (float64 GEMM form rounded to the input dtype).
 * landmarks: int64 [n_landmark] (host), out_clusters: int32 [row1-row0] (host). */
int gt_nearest_landmark(gt_ctx* ctx, int64_t row0, int64_t row1, const int64_t* landmarks, int32_t n_landmark,
                        int32_t mode, int32_t* out_clusters);

/* The same assignment when the landmarks are the BOUND points and the rows to assign are the queries (graphs.py:1200-1213 at
 * scale: sklearn euclidean_distances arithmetic, argmin's first-index rule): gt_knn_search of Y [m rows, host or device] for
 * the k nearest bound points, then out_labels[i] = the lowest index among those at exactly the nearest's distance - formed on
 * the device, so m labels cross PCIe instead of 2 m k table entries.  out_labels: int64 [m] (host). */
int gt_knn_first_nearest(gt_ctx* ctx, const void* Y, int64_t m, int32_t y_on_device, int32_t k, int64_t* out_labels,
                         uint32_t* flags);

/* ---- device memory helpers (so a host language without a HIP binding can keep data resident) -- */
int gt_dev_alloc(gt_ctx* ctx, size_t bytes, void** out);
int gt_dev_free(gt_ctx* ctx, void* p);
int gt_dev_upload(gt_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int gt_dev_download(gt_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int gt_dev_sync(gt_ctx* ctx);
/* order the context's stream against a stream of the caller without blocking the host (multi-GPU host code: the
 * collectives of graphtools_amd/dist.py run on torch's streams): direction 0 = the context's later work waits for
 * everything queued on `other_stream` so far, 1 = `other_stream` waits for everything the context has queued so far */
int gt_stream_order(gt_ctx* ctx, void* other_stream, int32_t direction);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHTOOLS_AMD_H */
