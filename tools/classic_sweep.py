#!/usr/bin/env python
"""Sweep of the classic candidate pass's threshold-seeding schedule (select_samp_*) on one point set: stage times of the whole kNN
side (candidate pass + re-rank + radius pass) per setting.  usage: classic_sweep.py [n] [d] [kind]"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from bench import make_gauss, make_manifold, make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kind = sys.argv[3] if len(sys.argv) > 3 else "gauss"
X = {"mix": make_mix, "gauss": make_gauss, "manifold": make_manifold}[kind](n, d, 1)
ctx = _hip.Context(0)
ctx.set_option("select_symmetric", "0")
p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
ctx.set_points(X)
base = None
rows = []
grid = [dict(select_samp_stride=s, select_samp_keep=k, select_samp2_level=l2, select_samp2_keep=k2)
        for s, k, l2, k2 in itertools.product((16, 32, 64, 128), (16, 24, 32), (2, 3, 4), (48, 64, 96))]
grid = [dict(select_samp_stride=32, select_samp_keep=0, select_samp2_level=3, select_samp2_keep=64)] + grid
for g in grid:
    for k, v in g.items():
        ctx.set_option(k, str(v))
    best = None
    for rep in range(2):
        nnz, _ = ctx.graph_build(p)
        st = {s: max(ctx.stage_ms(s), 0.0) for s in ("knn_select", "rerank", "radius", "fallback")}
        tot = sum(st.values())
        if best is None or tot < best[0]:
            best = (tot, st, nnz)
    if base is None:
        base = best
    rows.append((best[0], g, best[1], best[2]))
    print("%7.2f ms  %s  %s" % (best[0], g, {k: round(v, 2) for k, v in best[1].items()}), flush=True)
    assert best[2] == base[2], "another graph"
rows.sort(key=lambda r: r[0])
print("default: %.2f ms; best five:" % base[0])
for r in rows[:5]:
    print("  %.2f ms  %s" % (r[0], r[1]))
