#!/usr/bin/env python
"""Development probe: gt_graph_spmm (P @ X on the device) at benchmark size."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

if __name__ == "__main__":
    n = 1000000
    X = make_mix(n, 64, 1)
    ctx = _hip.Context(0)
    ctx.set_points(X)
    p, hold = ctx.make_params(15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    nnz, fl = ctx.graph_build(p)
    for c in (1, 16, 64, 100):
        Z = np.random.default_rng(c).standard_normal((n, c))
        ctx.graph_spmm(_hip.CSR_P, Z)
        t0 = time.perf_counter()
        out = ctx.graph_spmm(_hip.CSR_P, Z)
        wall = time.perf_counter() - t0
        ms = ctx.stage_ms("spmm")
        print(json.dumps({"ncols": c, "kernel_ms": round(ms, 3), "wall_s": round(wall, 3),
                          "gather_GBps": round(nnz * c * 8 / ms / 1e6, 1), "nnz": int(nnz)}), flush=True)
