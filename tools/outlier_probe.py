#!/usr/bin/env python
"""C3 with a handful of rows that belong to no cluster: isolated points, points half way between two clusters.  Build time and
stages per case (the bound pass reasons about cells: a row far from every landmark used to inflate its cell's ball).
usage: outlier_probe.py [n] [d]     GT_OPTS=k=v,..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from bench import make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
X0 = make_mix(n, d, 1)
rng = np.random.default_rng(5)
for tag, nout in (("clean", 0), ("20 isolated points", 20), ("200 isolated points", 200), ("2000 isolated points", 2000),
                  ("1 % mid-way points", n // 100)):
    X = X0.copy()
    if nout and nout <= 2000:
        idx = rng.choice(len(X), nout, replace=False)
        X[idx] = rng.uniform(-12, 12, (nout, d)).astype(np.float32)      # nowhere near a cluster
    elif nout:
        idx = rng.choice(len(X), nout, replace=False)
        X[idx] = (0.5 * (X[idx] + X[rng.choice(len(X), nout)])).astype(np.float32)   # between two clusters
    res = {}
    for mode, opts in (("default", {}), ("classic", {"select_symmetric": "0"})):
        if mode == "classic" and os.environ.get("GT_NO_CLASSIC") == "1":
            continue
        ctx = _hip.Context(0)
        for o in [o for o in os.environ.get("GT_OPTS", "").split(",") if o]:
            ctx.set_option(*o.split("="))
        for k, v in opts.items():
            ctx.set_option(k, v)
        xb = ctx.dev_alloc(X.nbytes)
        ctx.dev_upload(xb, X)
        p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
        best = None
        for rep in range(3):
            ctx.sync()
            t = time.perf_counter()
            ctx.set_points_device(xb, n, d, np.float32)
            nnz, _ = ctx.graph_build(p)
            ctx.sync()
            ms = (time.perf_counter() - t) * 1e3
            if rep and (best is None or ms < best):
                best = ms
        deg = float(ctx.graph_fetch_vec(1).sum())
        res[mode] = (best, nnz, deg, {s: round(ctx.stage_ms(s), 2) for s in ("query_order", "sym_prepare", "sym_seed", "sym_bound", "sym_cold", "knn_select",
                                                                          "rerank", "fallback", "radius") if ctx.stage_ms(s) > 0})
        ctx.dev_free(xb)
        ctx.close()
    if "classic" in res:
        assert res["default"][1:3] == res["classic"][1:3], "the pruned pass built another graph than the classic pass"
    print("%-22s %6.2f ms (classic %s)  nnz %d  %s" % (tag, res["default"][0], "%.1f" % res["classic"][0] if "classic" in res else "-",
                                                      res["default"][1], res["default"][3]), flush=True)
