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
c2 = np.zeros((n, 2), np.uint32)
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

# ---- why do the long rows see too little in launch A? ----
ctx2 = _hip.Context(0)
for o in [o for o in os.environ.get("GT_OPTS", "").split(",") if o]:
    k, v = o.split("=")
    ctx2.set_option(k, v)
ctx2.set_points(X)
ctx2.graph_build(p)
ctx = ctx2
T = (n + 127) // 128
stride_a = 64
tile_stride = ((T // stride_a + 1) + 384 + 2 + 63) // 64 * 64
tl = fetch(6, nb * tile_stride, np.int32).reshape(nb, tile_stride)
inv = np.empty(n, np.int64)
inv[perm] = np.arange(n)
bad = np.nonzero(tot > 512)[0]
rng2 = np.random.default_rng(0)
sample = rng2.choice(bad, size=min(12, len(bad)), replace=False)
M = 12
work = fetch(7, Lreal * M + 2 * Lreal, np.int32)
nbr = work[:Lreal * M].reshape(Lreal, M)
start = work[Lreal * M:Lreal * M + Lreal]
endp = work[Lreal * M + Lreal:]
xn = (X.astype(np.float64) ** 2).sum(1)
for pp in sample:
    r = perm[pp]
    members = np.nonzero(labels == labels[r])[0]
    d2 = xn[r] + xn[members] - 2.0 * (X[members].astype(np.float64) @ X[r].astype(np.float64))
    o = np.argsort(d2)[:16]
    nn_pos = inv[members[o]]
    blk = pp // 256
    tiles = set(tl[blk, :tcnt[blk]].tolist())
    seen = [int(q // 128) in tiles for q in nn_pos]
    print("pos %d row %d tot %d cluster_has_lm %s cell %d cellsize %d d2_16 %.1f nn_seen %d/16 ntiles %d  nn cells %s" % (
        pp, r, tot[pp], has_lm[labels[r]], cells[pp], sizes[cells[pp]], np.sort(d2)[15], sum(seen), tcnt[blk],
        sorted(set(cells[nn_pos].tolist()))))
    c = cells[pp]
    pos_c = np.nonzero(cells == c)[0]
    ct = set(range(pos_c.min() // 128, pos_c.max() // 128 + 1))
    blkcells = sorted(set(cells[blk * 256:blk * 256 + 256].tolist()))
    print("   own cell range [%d,%d) start/end arrays (%d,%d) tiles %d of which listed %d; block cells %s; nbr[c] %s; list head %s" % (
        pos_c.min(), pos_c.max() + 1, start[c], endp[c], len(ct), len(ct & tiles), blkcells, nbr[c].tolist(), tl[blk, :12].tolist()))
ctx2.close()
