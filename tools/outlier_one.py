#!/usr/bin/env python
"""One C3 build with 20 isolated points (for a kernel trace).  usage: outlier_one.py [n_outliers]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from bench import make_mix  # noqa: E402

nout = int(sys.argv[1]) if len(sys.argv) > 1 else 20
X = make_mix(1000000, 64, 1)
rng = np.random.default_rng(5)
if nout:
    idx = rng.choice(len(X), nout, replace=False)
    X[idx] = rng.uniform(-12, 12, (nout, 64)).astype(np.float32)
ctx = _hip.Context(0)
p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
for rep in range(3):
    ctx.set_points(X)
    ctx.graph_build(p)
ctx.sync()
print({s: round(ctx.stage_ms(s), 2) for s in ("sym_bound", "fallback", "sym_cold")})
