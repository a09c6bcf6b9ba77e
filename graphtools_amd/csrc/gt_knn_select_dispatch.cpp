// Routes a candidate-kernel launch to the translation unit compiled for its padded feature count.
#include "gt_knn_select.h"

#define GT_SEL_DP_LIST(X) X(16) X(32) X(56) X(64) X(104) X(128)
#define GT_DECL(dp) int gt_launch_select_dp##dp(gt_ctx*, const SelectArgs&);
GT_SEL_DP_LIST(GT_DECL)

int gt_choose_dp(int d) {
#define GT_PICK(dp) if (d <= dp) return dp;
    GT_SEL_DP_LIST(GT_PICK)
    return 0;
}

int gt_select_bq(int dp) { return dp <= 64 ? 256 : 128; }
int gt_select_bn(int dp) { return dp <= 64 ? 128 : 64; }

int gt_launch_select(gt_ctx* ctx, const SelectArgs& a) {
    switch (a.dp) {
#define GT_CASE(dp) case dp: return gt_launch_select_dp##dp(ctx, a);
        GT_SEL_DP_LIST(GT_CASE)
    }
    GT_FAIL(ctx, GT_E_LIMIT, "knn_select: feature dimension > 128 is not supported by the HIP path yet");
}
