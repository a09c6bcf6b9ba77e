// Launch interface of the MFMA candidate kernels (gt_knn_select.hip).
#pragma once
#include "gt_common.h"

struct SelectArgs {
    int dp = 0;            // padded feature count (gt_choose_dp)
    int mode = 0;          // 0: top-M' selection, 1: radius collect
    int nt = 8;            // selection: keys per lane in the compaction sort; list capacity 64*nt, M' = 16*nt
    const float* Yp = nullptr;    // database, [n_pad][dp]
    const float* hneg = nullptr;  // [n_pad]
    int64_t n_pad = 0;
    const float* Qp = nullptr;    // query matrix, [*][dp]
    const int32_t* qrows = nullptr;  // optional row ids into Qp (else q0 + i)
    int64_t q0 = 0;
    int32_t nq = 0;
    uint64_t* lists = nullptr;    // [nq_pad][64*nt] (selection) or [nq_pad][cap] (radius)
    uint32_t* counts = nullptr;   // [nq_pad]
    const float* thr_in = nullptr;  // radius mode: per-query score threshold [nq]
    int32_t cap = 0;                // radius mode: list capacity per query
};

int gt_launch_select(gt_ctx* ctx, const SelectArgs& a);
int gt_select_bq(int dp);  // query rows per workgroup
int gt_select_bn(int dp);  // database rows per tile
