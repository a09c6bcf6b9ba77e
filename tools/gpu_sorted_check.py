#!/usr/bin/env python
"""Development check: mix N=1e6 with the rows ORDERED BY CLUSTER (every neighbourhood contiguous in memory - the
adversarial layout for prefix-based thresholds): graph build statistics and the kNN tables of sampled rows against the
brute-force oracle."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from graphtools_amd import _hip  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

if __name__ == "__main__":
    n, d = 1000000, 64
    X = make_mix(n, d, 1)
    rng = np.random.default_rng(1)
    c = n // 2000
    rng.uniform(-10, 10, (c, d))
    labels = rng.integers(c, size=n)
    X = np.ascontiguousarray(X[np.argsort(labels, kind="stable")])
    ctx = _hip.Context(0)
    ctx.set_points(X)
    p, hold = ctx.make_params(15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    nnz, fl = ctx.graph_build(p)
    nnz, fl = ctx.graph_build(p)
    out = {"nnz": int(nnz), "stats": ctx.graph_stats(), "main": ctx.last_knn_precision(),
           "stage_ms": {s: round(ctx.stage_ms(s), 2) for s in ("query_order", "knn_select", "rerank", "radius", "symmetrize")}}
    rows = np.sort(np.random.default_rng(7).choice(n, 96, replace=False))
    dist, idx, _ = ctx.knn_search(40, Y=X[rows])
    d0, i0 = oracle.kneighbors(X, X[rows], 40)
    out["knn_idx_equal"] = bool(np.array_equal(idx, i0))
    out["knn_dist_equal"] = bool(np.array_equal(dist[:, 1:], d0[:, 1:]))
    blk = ctx.knn_search(16, rows=(500000, 500000 + 40000))
    d1, i1 = oracle.kneighbors(X, X[500000:500064], 16)
    out["block_idx_equal"] = bool(np.array_equal(blk[1][:64], i1))
    print(json.dumps(out))
