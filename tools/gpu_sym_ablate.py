#!/usr/bin/env python
"""Development probe: time of the symmetric candidate launches alone (dbg_select 4: tables are not valid), for kernel
variants built with tools/build_variant.py (GRAPHTOOLS_AMD_LIB=<variant> python tools/gpu_sym_ablate.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from tools.gpu_sym_check import make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = make_mix(n, 64, 1)
ctx = _hip.Context(0)
for o in [o for o in os.environ.get("GT_OPTS", "").split(",") if o]:
    k, v = o.split("=")
    ctx.set_option(k, v)
ctx.set_option("knn_precision", "f16x1")
ctx.set_option("dbg_select", str(4 | int(os.environ.get("GT_DBG", "0"))))
ctx.set_points(X)
best = {}
for _ in range(3):
    ctx.knn_search(16)
    for s in ("sym_seed", "knn_select"):
        v = ctx.stage_ms(s)
        best[s] = min(best.get(s, 1e9), v)
print(json.dumps({"lib": os.path.basename(_hip.LIB_PATH), "opts": os.environ.get("GT_OPTS", ""), **{k: round(v, 3) for k, v in best.items()}}))
if int(os.environ.get("GT_DBG", "0")) & 64:
    import ctypes
    nw = ((n + 255) // 256) * 4
    buf = np.zeros((nw, 8), dtype=np.uint64)
    ctx.lib.gt_dbg_fetch_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    rc = ctx.lib.gt_dbg_fetch_prof(ctx.h, nw, buf.ctypes.data)
    m = buf.mean(axis=0)
    print(json.dumps({"rc": rc, "cycles_per_wave": {"admission": float(m[0]), "compaction": float(m[1]), "barrier": float(m[2]),
                                                    "total": float(m[7])}, "admission_entries_per_wave": float(m[4]),
                      "compactions_per_wave": float(m[3])}))
ctx.close()
