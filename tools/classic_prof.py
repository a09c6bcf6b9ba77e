#!/usr/bin/env python
"""Where the classic candidate pass spends its wave cycles: the kernel's own cycle counters (dbg_select 64: admission path, list
compaction, tile barrier, per wave).  usage: classic_prof.py [n] [d] [kind]   GT_OPTS=k=v,..."""
import ctypes
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
for o in [o for o in os.environ.get("GT_OPTS", "").split(",") if o]:
    k, v = o.split("=")
    ctx.set_option(k, v)
ctx.set_option("dbg_select", str(64 + 4))
p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
ctx.set_points(X)
ctx.graph_build(p)
ctx.sync()
bq = 256
nwaves = ((n + bq - 1) // bq) * 4
out = np.zeros((nwaves, 8), dtype=np.uint64)
ctx.lib.gt_dbg_fetch_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
rc = ctx.lib.gt_dbg_fetch_prof(ctx.h, nwaves, out.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
o = out[out[:, 7] > 0].astype(np.float64)
tot = o[:, 7].sum()
print("waves %d  knn_select %.2f ms" % (len(o), ctx.stage_ms("knn_select")))
print("share of wave cycles: admission path %.1f %%  compaction %.1f %%  barrier wait %.1f %%  (level 0 of the sampling: %.1f %% of the cycles)" % (
    100 * o[:, 0].sum() / tot, 100 * o[:, 1].sum() / tot, 100 * o[:, 2].sum() / tot, 100 * o[:, 5].sum() / tot))
units = (n / 128.0) * 8      # units per wave: tiles x 4 sub-tiles x 2 query tiles
print("per wave: %.0f units, admission-path entries %.0f (%.2f per unit, %.0f of them in level 0), compactions %.0f (%.1f per query)" % (
    units, o[:, 4].mean(), o[:, 4].mean() / units, o[:, 6].mean(), o[:, 3].mean(), o[:, 3].mean() / 64))
print("cycles per admission-path entry %.0f, per compaction %.0f; wave cycles per unit %.0f" % (
    o[:, 0].sum() / max(o[:, 4].sum(), 1), o[:, 1].sum() / max(o[:, 3].sum(), 1), tot / len(o) / units))
ctx.close()
