#!/usr/bin/env python
"""Development probe: ablation of the candidate kernel (knn only) at N=1e6."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip
from tools.gpu_perf import make_mix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = make_mix(n, 64, 1)
nq = min(n, 131072)
for prec in ("f16", "f32"):
    for dbg in (0, 1, 2):
        ctx = _hip.Context(0)
        ctx.set_option("knn_precision", prec)
        ctx.set_option("dbg_select", dbg)
        ctx.set_points(X)
        idx = np.empty((8, 16), dtype=np.int64); dist = np.empty((8, 16))
        import ctypes
        # run the candidate pass for nq query rows through gt_knn_search on a row block (outputs ignored for dbg != 0)
        try:
            t = time.time()
            d, i, fl = ctx.knn_search(16, rows=(0, nq))
            wall = time.time() - t
        except Exception as e:
            wall = -1
        ms = ctx.stage_ms("knn_select")
        flops = 2.0 * nq * n * 64
        print(json.dumps({"prec": prec, "dbg": dbg, "nq": nq, "select_ms": round(ms, 2), "TF_alg": round(flops / ms / 1e9, 1), "rerank_ms": round(ctx.stage_ms("rerank"), 2), "fallback_ms": round(ctx.stage_ms("fallback"), 2)}))
        ctx.close()
