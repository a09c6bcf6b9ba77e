#!/usr/bin/env python
"""Development: where the host-complete wall time of one C3 graph goes (graphtools_amd.Graph(X).K / .P: pageable host X in, scipy
CSR out) - wall time per binding call and the Python time around them.  usage: gpu_host_complete_probe.py [n] [d]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_mix  # noqa: E402
import graphtools_amd  # noqa: E402
from graphtools_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
X = make_mix(n, d, 1)
acc = {}


def wrap(name):
    real = getattr(_hip.Context, name)

    def f(self, *a, **k):
        t = time.perf_counter()
        r = real(self, *a, **k)
        acc.setdefault(name, []).append(time.perf_counter() - t)
        return r
    setattr(_hip.Context, name, f)


for nm in [m for m in dir(_hip.Context) if not m.startswith("_") and callable(getattr(_hip.Context, m))]:
    wrap(nm)
for rep in range(int(os.environ.get("GT_REPS", "4"))):
    acc.clear()
    t0 = time.perf_counter()
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, verbose=0)
    t1 = time.perf_counter()
    K = G.K
    t2 = time.perf_counter()
    P = G.P
    t3 = time.perf_counter()
    tot = t3 - t0
    inner = sum(sum(v) for v in acc.values())
    print("run %d: total %.1f ms = Graph() %.1f + .K %.1f + .P %.1f; binding calls %.1f ms, Python around them %.1f ms" % (
        rep, tot * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, inner * 1e3, (tot - inner) * 1e3))
    print("      pool: %d blocks, %.2f GB; K.data at %#x (block of %s)" % (
        len(_hip._host_pool.blocks), sum(b.nbytes for b in _hip._host_pool.blocks) / 2**30, K.data.__array_interface__["data"][0],
        "the pool" if K.data.base is not None else "np.empty"))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) > 2e-4:
            print("      %-28s x%-2d %.2f ms" % (k, len(v), sum(v) * 1e3))
    del G, K, P
