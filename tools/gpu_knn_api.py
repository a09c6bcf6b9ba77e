#!/usr/bin/env python
"""Development probe: the kNN table of all rows through the C ABI at benchmark size (gt_knn_search, host arrays out).
GT_KS="16,96" neighbour counts; GT_NT8_MAX overrides the list-budget crossover (select_nt8_max_need)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

if __name__ == "__main__":
    X = make_mix(1000000, 64, 1)
    ctx = _hip.Context(0)
    if os.environ.get("GT_NT8_MAX"):
        ctx.set_option("select_nt8_max_need", os.environ["GT_NT8_MAX"])
    ctx.set_points(X)
    for k in (int(v) for v in os.environ.get("GT_KS", "16,96").split(",")):
        ctx.knn_search(k)
        t0 = time.perf_counter()
        d, i, fl = ctx.knn_search(k)
        wall = time.perf_counter() - t0
        print(json.dumps({"k": k, "wall_s": round(wall, 3), "main": ctx.last_knn_precision(),
                          "stage_ms": {s: round(ctx.stage_ms(s), 2) for s in ("query_order", "knn_select", "rerank", "fallback")}}))
