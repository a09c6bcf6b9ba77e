#!/usr/bin/env python
"""Development probe: stage times, list statistics and per-wave cycle counters of the symmetric candidate pass.
usage: gpu_sym_probe.py [n] [d] [kind]   env GT_OPTS="name=value,..." GT_DBG=<dbg_select bits>"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from tools.gpu_sym_check import STAGES, make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kind = sys.argv[3] if len(sys.argv) > 3 else "mix"
if kind == "mix":
    X = make_mix(n, d, 1)
elif kind == "manifold":
    _r = np.random.default_rng(1)
    X = (_r.standard_normal((n, 5)) @ _r.standard_normal((5, d)) + 0.01 * _r.standard_normal((n, d))).astype(np.float32)
elif kind == "sorted":      # mix with the rows stored cluster by cluster
    X = make_mix(n, d, 1)
    X = X[np.argsort(np.random.default_rng(1).integers(max(n // 2000, 1), size=n), kind="stable")]
else:
    X = np.random.default_rng(1).standard_normal((n, d)).astype(np.float32)
for variant in os.environ.get("GT_VARIANTS", "").split(";") or [""]:
    ctx = _hip.Context(0)
    opts = [o for o in (os.environ.get("GT_OPTS", "") + "," + variant).split(",") if o]
    for o in opts:
        k, v = o.split("=")
        ctx.set_option(k, v)
    dbg = int(os.environ.get("GT_DBG", "0"))
    if dbg:
        ctx.set_option("dbg_select", str(dbg))
    ctx.set_points(X)
    p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
    best = None
    for _ in range(int(os.environ.get("GT_REPS", "3"))):
        t = time.time()
        nnz, fl = ctx.graph_build(p)
        wall = time.time() - t
        if best is None or wall < best["wall_s"]:
            best = {"wall_s": round(wall, 4), "nnz": int(nnz), "flags": int(fl),
                    "stage_ms": {s: round(ctx.stage_ms(s), 3) for s in STAGES}, "knn": ctx.knn_stats(), "graph": ctx.graph_stats()}
    best["opts"] = opts
    print(json.dumps(best), flush=True)
    if dbg & 64:
        nw = ((n + 255) // 256) * 4 * best["knn"].get("sym_nseg", 1)
        buf = np.zeros((nw, 8), dtype=np.uint64)
        ctx.lib.gt_dbg_fetch_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        rc = ctx.lib.gt_dbg_fetch_prof(ctx.h, nw, buf.ctypes.data)
        m = buf.mean(axis=0)
        print(json.dumps({"rc": rc, "cycles_per_wave": {"admission": float(m[0]), "compaction": float(m[1]), "barrier": float(m[2]),
                                                        "total": float(m[7])},
                          "admission_entries_per_wave": float(m[4]), "compactions_per_wave": float(m[3]),
                          "total_min_max": [float(buf[:, 7].min()), float(buf[:, 7].max())]}), flush=True)
    ctx.close()
