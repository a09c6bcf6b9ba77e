#!/usr/bin/env python
"""One-at-a-time sweep of library options on a point set: build time per value, next to the default.
usage: opt_sweep.py n d kind opt=v1,v2,... [opt=...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from bench import make_gauss, make_manifold, make_mix  # noqa: E402

n, d, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
X = {"mix": make_mix, "gauss": make_gauss, "manifold": make_manifold}[kind](n, d, 1)
STAGES = ("prep", "query_order", "sym_prepare", "sym_seed", "sym_bound", "sym_cold", "knn_select", "rerank", "fallback", "radius", "affinity", "symmetrize", "symm_bins", "symm_merge", "symm_huge", "symm_compact", "normalize")


def run(opts):
    ctx = _hip.Context(0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    xb = ctx.dev_alloc(X.nbytes)
    ctx.dev_upload(xb, X)
    p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
    best = None
    for rep in range(5):
        ctx.sync()
        t = time.perf_counter()
        ctx.set_points_device(xb, n, d, np.float32)
        nnz, _ = ctx.graph_build(p)
        ctx.sync()
        ms = (time.perf_counter() - t) * 1e3
        if rep and (best is None or ms < best[0]):
            best = (ms, {s: round(ctx.stage_ms(s), 2) for s in STAGES if ctx.stage_ms(s) > 0}, nnz)
    ctx.dev_free(xb)
    ctx.close()
    return best


base = run({})
print("%6.2f ms  default  %s" % (base[0], base[1]), flush=True)
for spec in sys.argv[4:]:
    name, vals = spec.split("=")
    for v in vals.split(","):
        r = run({name: v})
        assert r[2] == base[2], "another graph"
        print("%6.2f ms  %s=%s  %s" % (r[0], name, v, r[1]), flush=True)
