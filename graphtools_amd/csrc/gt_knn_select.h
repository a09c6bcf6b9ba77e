// Launch interface of the MFMA candidate kernels (gt_knn_select.hip).
#pragma once
#include "gt_common.h"

// query tiles per wave of the two-stage collect kernel: its workgroups take 128 x this many rows (host and kernels agree)
#ifndef GT_SEL_TWO_QT
#define GT_SEL_TWO_QT 8
#endif
// Device-side extras of the symmetric self-query path (MODE 2) and of the MODE 0 launch that seeds its thresholds
struct SymDev {
    const float* g = nullptr;       // [n_pad] thr_j + hneg_j: row j admits query q iff (score_qj + hneg_q) > g_j; +inf on pad rows
    const float* gmin = nullptr;    // [n_pad / 32] minimum of g over each 32-row sub-tile
    uint64_t* tlists = nullptr;     // [n_pad][tcap] candidates of every row, forward and transposed (slots from tcounts)
    uint32_t* tcounts = nullptr;    // [n_pad] (pre-zeroed; the true count, may exceed tcap)
    int32_t tcap = 0;
    int32_t nseg = 1;               // MODE 2: work items per query block (grid = blocks x nseg)
    int32_t sched = 0;              // MODE 0: 1 = the queries are the database rows; query block I visits the tiles
                                    //         tile_list[I * tile_stride + 0 .. tile_cnt[I]) with the level-0 list budget
    const int32_t* tile_list = nullptr;
    const int32_t* tile_cnt = nullptr;
    int32_t tile_stride = 0;
    // row-sharded builds (gt_knn_shard.cpp): the sched 1 launch covers the query blocks [block0, block0 + nblk) only
    // (nblk = 0: all of them); a MODE 2 launch walks, for every query block I, the nseg pieces
    // ((shard_rank + I / shard_group) % shard_world) * nseg + 0 .. nseg-1 of the block's walk cut into shard_world * nseg
    // pieces - the ranks together cover every piece once, the hit-rich first pieces go round, and the blocks of a
    // group (neighbours on an XCD) stream the same window of the database
    int32_t block0 = 0;
    int32_t nblk = 0;
    int32_t shard_world = 1;
    int32_t shard_rank = 0;
    int32_t shard_group = 1;
    // MODE 2, two-stage scoring (partial distances): the unit loop runs only the first `half_steps` k-steps of the chain,
    // seeded with hh = -|y_h|^2/2 over those features: the result is |x_h|^2/2 - |x_h - y_h|^2/2, and a pair whose PARTIAL
    // distance already exceeds the row's radius can be dropped; thrh / gminh are the thresholds of that test in the
    // forward / transposed form (gt_sym.hip sym_half_thresholds_kernel).  Units that pass are recomputed in full on the
    // cold path and tested as before.  half_steps = 0: off (the unit loop scores all features).
    int32_t half_steps = 0;
    const float* zrows = nullptr;   // [n_pad][16] float16: the stage-one copy of the points (rows of the unit loop)
    const float* hh = nullptr;      // [n_pad]
    const float* thrh = nullptr;    // [n_pad]
    const float* gminh = nullptr;   // [n_pad / 32]
    // two-stage scoring defers the cold path: the unit loop only notes the (64-query group, 32-row sub-tile) pairs that
    // passed stage one - plain stores into a region of the queue that belongs to the wave alone, nothing to wait for -
    // and a second launch (mode 4, one wave per entry of the compacted queue) scores them in full and files the
    // survivors.  The streaming workgroups never stall on list traffic.
    uint2* queue = nullptr;         // collect launch: [waves][qcap] regions; mode 4: the compacted entries
    uint32_t* qcount = nullptr;     // collect launch: [waves] entries noted by each wave (beyond qcap: dropped, the
                                    // caller falls back)
    int32_t qcap = 0;               // entries per wave region
    uint2* qspill = nullptr;        // shared spill area for the waves whose region is full (slots from an atomic: rare)
    uint32_t* qspill_count = nullptr;
    int32_t qspill_cap = 0;
    int32_t qn = 0;                 // mode 4: entries to process
    int32_t own_only = 0;      // mode 2 (two-stage collect): the query blocks [block0, block0 + nblk) of 1024 rows against EVERY
                               // tile, forward test only (a rank's own rows, gt_knn_shard_local);
                               // mode 4: every unit files under its QUERIES only (row-sharded builds on renumbered points: the
                               // queue pairs the rank's own query groups with every sub-tile the cell bounds leave - the
                               // rows' side belongs to whoever owns the rows)
    int32_t list_shift = 0;    // sched 1: the tile list of query block b is list b >> list_shift (narrow 128-row
                               // workgroups on lists made for 256-row blocks)
    // mode 4 in a LOCAL FRAME (sym_cold_local_kernel): the units are scored on float16 roundings of (x - o) sc, o = the centre
    // of the unit's 64 queries - the rounding error scales with the size of a cell, not with the distance from the origin.
    // xs: the float32 points in sorted order (row stride xs_d floats, a multiple of 4, xs_n rows), gcen [n_pad / 64][dp]:
    // the groups' centres (scaled by sc), rloc [n_pad]: the radius every row needs listed, scaled (-inf: nothing, +inf:
    // everything).  nullptr: the launch scores the compact copy as before.
    const float* xs = nullptr;
    int32_t xs_d = 0;
    int32_t xs_n = 0;
    const float* gcen = nullptr;
    const float* rloc = nullptr;
    float sc = 0.f;
    // two-stage collect, unit skipping by balls in the stage-one space (gt_sym.hip z_balls_kernel): for every group of 32
    // consecutive sorted rows the centre zc [n_pad / 32][16] of its stage-one rows, and zrn [n_pad / 32][2] = {radius of the rows
    // around it, largest stage-one radius a row of the group asks for}.  A (32 queries x 32 rows) unit whose balls are farther
    // apart than either group's need cannot hold a pair that passes stage one: its MFMA and its tests are skipped.  nullptr: off.
    const float* zc = nullptr;
    const float* zrn = nullptr;
    // MODE 2 (one-stage collect) over LISTED walks (gt_sym.hip collect_lists_kernel): query block b visits only the positions
    // walk_list[b * walk_stride + 0 .. walk_cnt[b]) of its walk (ascending; a position rel is the tile (b TPB + rel) mod T) -
    // the tiles whose cells the cell bounds of the bound pass could not rule out against the block's own cells; the nseg work
    // items of a block share its list.  walk_cnt[b] < 0: the list did not fit, the block walks everything.  nullptr: off.
    const int32_t* walk_list = nullptr;
    const int32_t* walk_cnt = nullptr;
    int32_t walk_stride = 0;
    int32_t gc_first = 0;      // the centres of the groups [gc_first, gc_first + gc_count) are formed (gc_count = 0: all)
    int32_t gc_count = 0;
};

struct SelectArgs {
    int dp = 0;            // padded feature count (gt_choose_dp)
    int prec = 0;          // 0: float32 operands, 1: split float16 planes (hi + lo), 2: hi planes of the split copy only
    int mode = 0;          // 0: top-M' selection, 1: radius collect, 2: symmetric collect (self queries, prec 2),
                           // 4: the deferred cold pass of a two-stage symmetric collect (sym.queue, sym.qn),
                           // 5: compaction of the wave regions of that collect's queue (lists = dense queue of `cap`
                           //    entries, counts = {total, overflow flag}, nq = wave regions)
    SymDev sym;            // mode 2 / mode 0 with sym.sched = 1
    int nt = 8;            // selection: keys per lane in the compaction sort; list capacity 64*nt, M' = 16*nt
    const float* Yp = nullptr;    // database working copy, [n_pad] rows of 4*dp bytes
    const float* hneg = nullptr;  // [n_pad]
    int64_t n_pad = 0;
    const float* Qp = nullptr;    // query matrix, [*][dp]
    const int32_t* qrows = nullptr;  // optional row ids into Qp (else q0 + i)
    int64_t q0 = 0;
    int32_t nq = 0;
    uint64_t* lists = nullptr;    // [nq_pad][64*nt] (selection) or [nq_pad][cap] (radius)
    uint32_t* counts = nullptr;   // [nq_pad]
    const float* thr_in = nullptr;  // radius mode: per-query score threshold [nq]; selection mode: optional starting
                                    // threshold per ROW (indexed by row - q0), must be <= the wanted scores
    float* thr_out = nullptr;       // selection mode: final admission threshold per query [nq_pad]
    int32_t cap = 0;                // radius mode: list capacity per query
    unsigned long long* prof = nullptr;   // optional per-wave cycle counters [nblocks*4][8] (development)
    int32_t samp_stride = 0;        // selection: > 1 (power of two) = visit every samp_stride-th tile first with a keep-samp_keep budget
    int32_t samp_keep = 0;          //   (must be >= the number of neighbours wanted, even, <= 8*nt)
    int32_t samp_end = 0;           //   > 0: after the last sampled tile every list is cut to its samp_end best
    int32_t samp2_level = 0;        //   > 0: second cut, to the samp2_keep best, once 2^samp2_level / samp_stride of the tiles are seen
    int32_t samp2_keep = 0;
    int32_t samp_trig = 0;          //   level 0: a half-list is cut when it holds more than this (0: samp_keep / 2 + 24)
    int32_t narrow = 0;             // selection, prec 2, dp <= 64: 128-row workgroups (launches with few query rows)
    int32_t final_keep = 0;         // selection: entries kept per query at the end (0: M' = 16*nt; at most 64*nt)
    int32_t dbg = 0;                // experiment switches (bit 0: no survivors, bit 1: no compaction sort)
};

int gt_launch_select(gt_ctx* ctx, const SelectArgs& a);
// nearest of the L landmark rows Yl (compact hi-plane rows, seeds hl) for the rows [q0, q0 + nq) of Yc (approximate)
int gt_launch_assign_cells(gt_ctx* ctx, int dp, const float* Yc, const float* Yl, const float* hl, int64_t q0, int32_t nq,
                           int32_t L, int32_t need, uint32_t* cell, float* thr0, float* best = nullptr);
// gt_order.hip: the rows [q0, q0 + nq) grouped by nearest landmark -> out_rows (device, int32 [nq]); *active = 0 when
// the launch is too small to bother or the compact copy is not available (out_rows untouched)
// out_thr0 (device, float [nq], indexed by row - q0): a score at least `need` (<= 32) database rows reach in the
// single-chain arithmetic - a valid starting threshold of a MODE 0 launch in that arithmetic (SelectArgs::thr_in)
// Qc: compact hi-plane copy of the query matrix (the bound points themselves, or an external matrix prepared with the
// same scale); rows [q0, q0 + nq) of it are ordered
int gt_query_order(gt_ctx* ctx, const float* Qc, int64_t q0, int64_t nq, int need, int32_t* out_rows, float* out_thr0,
                   int* active);
int gt_choose_dp_prec(int d, int prec);  // padded feature count for a precision (0 if unsupported)
int gt_select_bq(int dp);  // query rows per workgroup
int gt_select_bn(int dp);  // database rows per tile
// the cell order of all bound rows in two halves (row-sharded builds: a rank assigns 1 / world of the rows, the cells are
// all-gathered, every rank sorts) - gt_order.hip
int gt_order_cells_partial(gt_ctx* ctx, int64_t row0, int64_t row1, uint32_t* cells_out, int* active);
int gt_order_sort_cells(gt_ctx* ctx, const uint32_t* cells_all, int32_t* out_rows);
