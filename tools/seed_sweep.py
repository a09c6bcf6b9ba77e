#!/usr/bin/env python
"""Sweep of the threshold-seeding neighbourhood (select_sym_cells / select_sym_max_nb / select_sym_stride) on C3: build time and the
stage times behind it.  usage: seed_sweep.py [n] [d] [kind]"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from bench import make_gauss, make_manifold, make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kind = sys.argv[3] if len(sys.argv) > 3 else "mix"
X = {"mix": make_mix, "gauss": make_gauss, "manifold": make_manifold}[kind](n, d, 1)
base_nnz = None
grid = [dict()] + [dict(select_sym_cells=c, select_sym_max_nb=m, select_sym_stride=s)
                   for c, m, s in itertools.product(tuple(int(v) for v in os.environ.get("GT_SWEEP_CELLS", "6,8,12,16").split(",")), tuple(int(v) for v in os.environ.get("GT_SWEEP_NB", "192,256,384").split(",")), tuple(int(v) for v in os.environ.get("GT_SWEEP_STRIDE", "384,768,0").split(",")))]
for g in grid:
    ctx = _hip.Context(0)
    for k, v in g.items():
        ctx.set_option(k, str(v))
    xb = ctx.dev_alloc(X.nbytes)
    ctx.dev_upload(xb, X)
    p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
    best = None
    for rep in range(4):
        ctx.sync()
        t = time.perf_counter()
        ctx.set_points_device(xb, n, d, np.float32)
        nnz, _ = ctx.graph_build(p)
        ctx.sync()
        ms = (time.perf_counter() - t) * 1e3
        if rep and (best is None or ms < best[0]):
            best = (ms, {s: round(ctx.stage_ms(s), 2) for s in ("sym_prepare", "sym_seed", "sym_bound", "sym_cold", "knn_select", "rerank", "affinity", "symmetrize")
                         if ctx.stage_ms(s) > 0})
    if base_nnz is None:
        base_nnz = nnz
    assert nnz == base_nnz
    print("%6.2f ms  %s  %s" % (best[0], g or "default", best[1]), flush=True)
    ctx.dev_free(xb)
    ctx.close()
