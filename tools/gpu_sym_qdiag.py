#!/usr/bin/env python
"""Development probe: distribution of the queue entries of the two-stage collect over wave regions and thresholds."""
import ctypes
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
for o in [o for o in os.environ.get("GT_OPTS", "select_sym_nseg=1").split(",") if o]:
    k, v = o.split("=")
    ctx.set_option(k, v)
ctx.set_points(X)
p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
ctx.graph_build(p)
ctx.lib.gt_dbg_fetch_sym.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
n_pad = -(-n // 1024) * 1024
nw = n_pad // 512 * 4
q = np.zeros(nw, dtype=np.uint32)
assert ctx.lib.gt_dbg_fetch_sym(ctx.h, 8, nw, q.ctypes.data) == 0
thr = np.zeros(n_pad, dtype=np.float32)
rc = ctx.lib.gt_dbg_fetch_sym(ctx.h, 0, n, thr.ctypes.data)
assert rc == 0, (rc, ctx.lib.gt_last_error(ctx.h))
tc = np.zeros(n_pad, dtype=np.uint32)
assert ctx.lib.gt_dbg_fetch_sym(ctx.h, 3, n, tc.ctypes.data) == 0
print(json.dumps({"regions": int(nw), "entries": int(q.sum()), "pct": [int(v) for v in np.percentile(q, [50, 90, 99, 99.9, 100])],
                  "regions_over_512": int((q > 512).sum()), "entries_in_those": int(q[q > 512].sum())}))
big = np.argsort(q)[-8:]
for r in big:
    rows = np.arange(r * 128, min(r * 128 + 128, n))
    t = thr[rows]
    print(json.dumps({"region": int(r), "entries": int(q[r]), "thr_inf": int(np.isinf(t).sum()), "thr_min": float(t[np.isfinite(t)].min()) if np.isfinite(t).any() else None,
                      "thr_median": float(np.median(t[np.isfinite(t)])) if np.isfinite(t).any() else None, "tcounts_max": int(tc[rows].max()), "tcounts_med": int(np.median(tc[rows]))}))
thrh = np.zeros(n, dtype=np.float32); hh = np.zeros(n, dtype=np.float32); gh = np.zeros(n, dtype=np.float32)
assert ctx.lib.gt_dbg_fetch_sym(ctx.h, 10, n, thrh.ctypes.data) == 0
assert ctx.lib.gt_dbg_fetch_sym(ctx.h, 11, n, hh.ctypes.data) == 0
assert ctx.lib.gt_dbg_fetch_sym(ctx.h, 12, n, gh.ctypes.data) == 0
for r in big[-3:]:
    rows = np.arange(r * 128, min(r * 128 + 128, n))
    i = rows[np.argmin(thrh[rows])]
    print(json.dumps({"region": int(r), "thrh_min": float(thrh[rows].min()), "thrh_med": float(np.median(thrh[rows])), "row": int(i), "thr": float(thr[i]),
                      "hh": float(hh[i]), "gh": float(gh[i]), "gh_min_region": float(gh[rows].min()), "gh_med": float(np.median(gh[rows]))}))
print(json.dumps({"thrh_pct": [float(v) for v in np.percentile(thrh, [0, 0.01, 1, 50, 99])], "gh_pct": [float(v) for v in np.percentile(gh, [0, 0.01, 1, 50, 99])],
                  "hh_pct": [float(v) for v in np.percentile(hh, [0, 1, 50, 99, 100])]}))
fin = thr[:n][np.isfinite(thr[:n])]
print(json.dumps({"thr_pct": [float(v) for v in np.percentile(fin, [0.01, 0.1, 1, 50, 99])], "orphans": int(np.isinf(thr[:n]).sum())}))

rho = -gh.astype(np.float64); tc = tc[:n]; thr = thr[:n]
med = np.median(rho)
kept = np.zeros(n, dtype=np.uint32)
assert ctx.lib.gt_dbg_fetch_sym(ctx.h, 9, n, kept.ctypes.data) == 0
for f in (1.5, 2, 3, 5, 10):
    m = rho > f * med
    print(json.dumps({"rho_over_x_median": f, "rows": int(m.sum()), "tcounts_median": int(np.median(tc[m])) if m.any() else None,
                      "tcounts_over_512": int((tc[m] > 512).sum()), "thr_inf": int(np.isinf(thr[m]).sum())}))
print(json.dumps({"overflow_rows_total": int((tc > 512).sum()), "rho_of_overflow_rows_pct": [float(v / med) for v in np.percentile(rho[tc > 512], [0, 10, 50, 90, 100])] if (tc > 512).any() else None}))
