#!/usr/bin/env python
"""Development probe: how local is the kernel matrix in the cell-sorted order?  For the K of one build, the share of
entries (i, j) whose sorted positions fall into the same block of 2^s rows, and the distribution of |pos i - pos j|."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_gauss, make_manifold, make_mix  # noqa: E402
from graphtools_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kind = sys.argv[3] if len(sys.argv) > 3 else "mix"
X = {"mix": make_mix, "manifold": make_manifold, "gauss": make_gauss}[kind](n, d, 1)
ctx = _hip.Context(0)
ctx.set_points(X)
p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
ctx.graph_build(p)
ctx.sync()
kd, ki, kp = ctx.graph_fetch_csr(_hip.CSR_K)
perm = np.zeros(n, dtype=np.int32)
import ctypes  # noqa: E402

ctx.lib.gt_dbg_fetch_sym.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
ctx.lib.gt_dbg_fetch_sym.restype = ctypes.c_int
assert ctx.lib.gt_dbg_fetch_sym(ctx.h, 1, n, perm.ctypes.data) == 0
pos = np.empty(n, dtype=np.int64)
pos[perm] = np.arange(n)
rows = np.repeat(np.arange(n), np.diff(kp))
pi, pj = pos[rows], pos[ki]
for s in (8, 9, 10, 11, 12, 14):
    print("same block of 2^%d rows: %.3f" % (s, np.mean((pi >> s) == (pj >> s))))
dist = np.abs(pi - pj)
print("quantiles of |pos i - pos j| (50, 75, 90, 99 %):", np.quantile(dist, [0.5, 0.75, 0.9, 0.99]).astype(int).tolist())
