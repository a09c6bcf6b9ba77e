// Device-resident state of a kNN graph build (gt_sparse.hip), shared with the landmark kernels.
#pragma once
#include <vector>

#include "gt_common.h"

struct Triplet {
    uint32_t row;   // destination (global) row
    uint32_t col;   // global column
    double val;
};
static_assert(sizeof(Triplet) == 16, "triplet layout");

// entry of a union row (own entries tagged 0, received ones tagged 1): one 16-byte record, so that the scattered
// writes of the received part touch one cache line per entry
struct UEntry {
    uint32_t key;   // (column << 1) | tag
    uint32_t pad;
    double val;
};
static_assert(sizeof(UEntry) == 16, "union entry layout");

struct GraphState {
    gt_knn_params p{};
    std::vector<double> bw_host;
    int world = 1, rank = 0;
    std::vector<int64_t> splits;
    int64_t r0 = 0, r1 = 0, nloc = 0;
    int64_t n_total = 0;   // number of columns of K / P
    bool begun = false, finished = false;
    bool aniso_applied = false;   // finish_normalize has rescaled K by the degrees (not idempotent)
    // query side: rows of the graph are rows [qoff, qoff + nloc) of Qmat (= the bound points unless `external`,
    // i.e. build_kernel_to_data(Y), graphs.py:819-982)
    bool external = false;
    const void* Qmat = nullptr;
    const double* qnorm = nullptr;
    int64_t qoff = 0;
    int need_m = 0;
    int limit = 0;          // eligible table entries per row
    double radius_factor = 0.0;
    // per-row
    DevBuf bw, bw_user, rowsrc, lenN, lenT, cursor, off, outlen, indptr, degree;
    DevBuf spmm_in, spmm_out;   // staging of gt_graph_spmm for host operands
    DevBuf tablen;   // int32 [nloc]: entries of the candidate-table row the affinity pass looked at
    DevBuf rec_s;    // SlotRec [nloc]: row, table length, bandwidth by sorted position (tables by sorted position only, KnnWork::tab_sorted)
    DevBuf bwpos;    // BwPos [nloc]: bandwidth and sorted position by row (same)
    DevBuf midrows;  // int32 [nloc]: rows of 129 ... kBigRow union entries (fused-destination builds: merge_final_kernel over a list)
    DevBuf sC;       // int64 [nloc + 1]: scan of the tables' lengths by sorted position (fused destination count: posj lies by it)
    // radius pass
    DevBuf over_rows, over_count, rthr, rlists, rcounts, rK, rmax;
    int64_t n_over = 0;
    int32_t rcap = 0;
    int64_t radius_retries = 0;
    // exchange
    DevBuf ownercnt, ownerpos, cnt_sorted, pos_sorted, scan_tmp, selfbuf, splits_dev, edges;
    std::vector<int64_t> send_counts_host;
    // merge
    DevBuf Ukey, Uval, Vkey, Vval, bigrows, hugerows, bigcount, bigscratch_k, bigscratch_v, bigsoff, aniso_tmp, scan_own;
    DevBuf indices, Kdata, Pdata, flags;
    DevBuf deg_caller;   // renumbered points, one rank: the degrees in the caller's numbering (anisotropy, diff_aff)
    DevBuf bincnt, binoff;   // destination bins of the single-rank transpose: triplets per bin (+ the emit cursors), scan
    DevBuf ucol, uval;       // fused tail: received entries in fixed slot rows [row][capT] (columns, values)
    bool bins_used = false;
    bool pairs = false;        // this build's affinities settled the mutual pairs (negative = final values; graph_finish_pairs)
    bool pairs_fused = false;  // ... and counted the destinations of the one-sided entries on the way (tables by sorted position, no
                               //     row of the radius pass): posj (cursor) lies by sC, bincnt is filled - bin_count_kernel is skipped
    int32_t bin_shift = 9, bin_count = 0;   // rows per destination bin (log2), bins
    int64_t sc_total = 0;                   // entries of posj when it lies by sC
    // row-sharded pair-resolved tail (gt_graph_bandwidth_local -> gt_graph_set_bandwidths -> gt_graph_begin -> ... -> gt_graph_finish)
    bool half_begun = false;     // gt_graph_bandwidth_local ran the first half of gt_graph_begin for half_params / world / rank / splits
    gt_knn_params half_params{};
    DevBuf bw_all;               // float64 [n_total]: the bandwidths of all rows, gathered by the caller
    bool bw_all_valid = false;   // ... handed over for the build that is half begun
    bool pairs_shard = false;    // this rank settles its mutual pairs with bw_all; only one-sided entries travel (graph_finish_pairs_shard)
    bool pairs_shard_bins = false;   // ... and its local one-sided entries go through its own destination bins (every row a table row):
                                     //     posj (cursor) by sC, bincnt filled; only entries for OTHER ranks' rows are emitted
    DevBuf colid;                // int32 [nloc]: the caller's number of every local row (the column it is in its partners' rows)
    DevBuf ident;                // int32 [nloc]: 0 ... nloc - 1 (a shard's rows are their own positions: FusedSrc::pos)
    bool relabelled = false;   // the CSR's columns are the caller's row numbers of a renumbered point set (rows: gt_points_row_ids)
    int64_t nnz0 = 0, nnz = 0;
};

