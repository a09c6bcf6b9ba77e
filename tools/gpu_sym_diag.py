#!/usr/bin/env python
"""Development probe: which rows of the symmetric pass collect long lists, and why (mix data)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 64
rng = np.random.default_rng(1)
c = max(n // 2000, 1)
centres = rng.uniform(-10, 10, (c, d))
labels = rng.integers(c, size=n)
X = (centres[labels] + rng.standard_normal((n, d))).astype(np.float32)
ctx = _hip.Context(0)
for o in [o for o in os.environ.get("GT_OPTS", "").split(",") if o]:
    k, v = o.split("=")
    ctx.set_option(k, v)
ctx.set_points(X)
p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
ctx.graph_build(p)
print(json.dumps(ctx.knn_stats()))
ctx.lib.gt_dbg_fetch_sym.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]


def fetch(which, count, dtype, shape=None):
    buf = np.zeros(count if shape is None else shape, dtype=dtype)
    rc = ctx.lib.gt_dbg_fetch_sym(ctx.h, which, count, buf.ctypes.data)
    assert rc == 0, rc
    return buf


thr = fetch(0, n, np.float32)
perm = fetch(1, n, np.int32)
c2 = fetch(2, n, np.uint32, (n, 2))
ct = fetch(3, n, np.uint32)
nb = (n + 255) // 256
tcnt = fetch(4, nb, np.int32)
cells = fetch(5, n, np.uint32)
tot = c2.sum(1).astype(np.int64) + ct
lab_sorted = labels[perm]
print("tile list lengths: min %d mean %.1f max %d" % (tcnt.min(), tcnt.mean(), tcnt.max()))
print("total per row: mean %.1f median %d p99 %d max %d" % (tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max()))
L = int(cells.max()) + 1
step = n // (L if L else 1)
# clusters without a landmark (landmarks = rows 0, step, 2 step, ...)
Lreal = min(4096, max(64, (n // 512) // 32 * 32))
lm_rows = np.arange(Lreal) * (n // Lreal)
has_lm = np.zeros(c, bool)
has_lm[labels[lm_rows]] = True
print("clusters without landmark: %d of %d; rows in them %d" % ((~has_lm).sum(), c, (~has_lm[labels]).sum()))
big = tot > 512
print("rows over 512:", int(big.sum()), " of them in landmark-less clusters:", int((~has_lm[lab_sorted[big]]).sum()))
# does the row's cell landmark belong to its own cluster?
cell_lab = labels[lm_rows][cells]
foreign = cell_lab != lab_sorted
print("rows whose cell landmark is of another cluster: %d ; among rows over 512: %d" % (foreign.sum(), (foreign & big).sum()))
# cells per cluster
med = np.median(tot[~big])
print("median total of normal rows", med)
# cell sizes
sizes = np.bincount(cells, minlength=Lreal)
print("cell sizes: min %d median %d max %d; cells > 2048 rows: %d" % (sizes.min(), np.median(sizes), sizes.max(), (sizes > 2048).sum()))
bigcell = sizes[cells] > 2048
print("rows over 512 in cells > 2048 rows:", int((big & bigcell).sum()))
ctx.close()
