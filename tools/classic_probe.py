#!/usr/bin/env python
"""Classic candidate pass (knn_select_kernel<DP, 8, 0, 2>) on its own: time of the `knn_select` stage under development switches.
usage: classic_probe.py [n] [d] [kind]     GT_DBG_LIST=4,5,22 (dbg_select values; 4 = stop behind the candidate pass, +1 = no admissions,
+2 / +16 = no list compaction), GT_OPTS=k=v,k=v"""
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
kind = sys.argv[3] if len(sys.argv) > 3 else "gauss"
X = {"mix": make_mix, "gauss": make_gauss, "manifold": make_manifold}[kind](n, d, 1)
for dbg in [int(v) for v in os.environ.get("GT_DBG_LIST", "0,4,5").split(",")]:
    ctx = _hip.Context(0)
    ctx.set_option("select_symmetric", "0")
    for o in [o for o in os.environ.get("GT_OPTS", "").split(",") if o]:
        k, v = o.split("=")
        ctx.set_option(k, v)
    ctx.set_option("dbg_select", str(dbg))
    p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
    ctx.set_points(X)
    best = None
    for rep in range(3):
        ctx.sync()
        t = time.perf_counter()
        try:
            ctx.graph_build(p)
        except Exception as e:  # (the experiments leave no valid tables: whatever comes after the stage may complain)
            err = str(e)[:60]
        ctx.sync()
        ms = ctx.stage_ms("knn_select")
        best = ms if best is None else min(best, ms)
    if dbg == 0:
        deg = ctx.graph_fetch_vec(1)
        print("      nnz %d  degree sum %.17g  stages %s" % (ctx.graph_build(p)[0], float(deg.sum()),
              {s_: round(ctx.stage_ms(s_), 2) for s_ in ("knn_select", "rerank", "radius", "fallback", "affinity", "symmetrize")}), flush=True)
    flop = 2.0 * n * n * d
    print("dbg %3d  knn_select %.2f ms  = %.3f PF (%.1f %% of 2.5 PF)   wall of last call %.1f ms" % (
        dbg, best, flop / best / 1e12, flop / best / 1e12 / 2.5 * 100, (time.perf_counter() - t) * 1e3), flush=True)
    ctx.close()
