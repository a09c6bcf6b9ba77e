// Routes a candidate-kernel launch to the translation unit compiled for its precision / padded feature count.
#include "gt_knn_select.h"

// keep in sync with SELECT_UNITS in graphtools_amd/_build.py
#define GT_SEL_P0_LIST(X) X(0, 16) X(0, 32) X(0, 56) X(0, 64) X(0, 104) X(0, 128)
#define GT_SEL_P1_LIST(X) X(1, 16) X(1, 32) X(1, 48) X(1, 64) X(1, 80) X(1, 96) X(1, 112) X(1, 128)
#define GT_SEL_P2_LIST(X) X(2, 16) X(2, 32) X(2, 48) X(2, 64) X(2, 80) X(2, 96) X(2, 112) X(2, 128)
#define GT_SEL_NARROW_LIST(X) X(2, 16) X(2, 32) X(2, 48) X(2, 64)
#define GT_DECL(PR_, DP_) int gt_launch_select_p##PR_##_dp##DP_(gt_ctx*, const SelectArgs&);
#define GT_DECL_NARROW(PR_, DP_) int gt_launch_select_narrow_p##PR_##_dp##DP_(gt_ctx*, const SelectArgs&);
#define GT_DECL_ASSIGN(PR_, DP_) \
    int gt_launch_assign_cells_p##PR_##_dp##DP_(gt_ctx*, const float*, const float*, const float*, int64_t, int32_t, int32_t, int32_t, uint32_t*, float*, float*);
GT_SEL_P2_LIST(GT_DECL_ASSIGN)
GT_SEL_NARROW_LIST(GT_DECL_NARROW)
GT_SEL_P0_LIST(GT_DECL)
GT_SEL_P1_LIST(GT_DECL)
GT_SEL_P2_LIST(GT_DECL)

int gt_choose_dp_prec(int d, int prec) {
#define GT_PICK(PR_, DP_) if (d <= DP_) return DP_;
    if (prec == 0) {
        GT_SEL_P0_LIST(GT_PICK)
    } else {
        GT_SEL_P1_LIST(GT_PICK)
    }
    return 0;
}

int gt_choose_dp(int d) { return gt_choose_dp_prec(d, 0); }
int gt_select_bq(int dp) { return dp <= 64 ? 256 : 128; }   // also a multiple of the narrow variant's 128 rows
// row padding granule of the working copies: a multiple of every kernel's tile (128 / 64 rows)
int gt_select_bn(int dp) { return dp <= 64 ? 128 : 64; }

int gt_launch_select(gt_ctx* ctx, const SelectArgs& a) {
    if (a.narrow && a.mode == 0) {
#define GT_CASE_NARROW(PR_, DP_) if (a.prec == PR_ && a.dp == DP_) return gt_launch_select_narrow_p##PR_##_dp##DP_(ctx, a);
        GT_SEL_NARROW_LIST(GT_CASE_NARROW)
    }
#define GT_CASE(PR_, DP_) if (a.prec == PR_ && a.dp == DP_) return gt_launch_select_p##PR_##_dp##DP_(ctx, a);
    GT_SEL_P0_LIST(GT_CASE)
    GT_SEL_P1_LIST(GT_CASE)
    GT_SEL_P2_LIST(GT_CASE)
    GT_FAIL(ctx, GT_E_LIMIT, "knn_select: feature dimension > 128 is not supported by the HIP path yet");
}

int gt_launch_assign_cells(gt_ctx* ctx, int dp, const float* Yc, const float* Yl, const float* hl, int64_t q0, int32_t nq,
                           int32_t L, int32_t need, uint32_t* cell, float* thr0, float* best) {
#define GT_CASE_ASSIGN(PR_, DP_) if (dp == DP_) return gt_launch_assign_cells_p##PR_##_dp##DP_(ctx, Yc, Yl, hl, q0, nq, L, need, cell, thr0, best);
    GT_SEL_P2_LIST(GT_CASE_ASSIGN)
    GT_FAIL(ctx, GT_E_LIMIT, "assign_cells: unsupported feature count");
}
