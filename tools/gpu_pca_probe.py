#!/usr/bin/env python
"""Development probe: DevicePCA (gt_pca.hip) at scale - wall time, stage times, throughput of the tall products, and
sklearn's PCA(svd_solver="randomized") on the host next to it.  usage: gpu_pca_probe.py [n] [d] [k]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from graphtools_amd._pca import DevicePCA  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rng = np.random.default_rng(0)
r = 150
V = np.linalg.qr(rng.standard_normal((d, r)))[0].astype(np.float32)
sv = (10.0 * 0.97 ** np.arange(r)).astype(np.float32)
X = np.empty((n, d), dtype=np.float32)
for s in range(0, n, 100000):
    e = min(n, s + 100000)
    X[s:e] = (rng.standard_normal((e - s, r), dtype=np.float32) * sv) @ V.T + 0.05 * rng.standard_normal((e - s, d), dtype=np.float32)
out = {"n": n, "d": d, "k": k}
for rep in range(0 if os.environ.get('GT_DBG') else 2):
    t = time.perf_counter()
    dev = DevicePCA(k, random_state=0)
    T = dev.fit_transform(X)
    out["device_fit_transform_s"] = round(time.perf_counter() - t, 3)
    out["phases_s"] = getattr(dev, "phase_s_", None)
# stage times of the products alone (one context, X resident)
ctx = _hip.Context(0)
if os.environ.get('GT_DBG'):
    ctx.set_option('dbg_select', os.environ['GT_DBG'])
t = time.perf_counter()
mean, ssq = ctx.pca_begin(X)
out["upload_and_moments_s"] = round(time.perf_counter() - t, 3)
W = rng.standard_normal((d, k + 10))
ctx.pca_matmul(0, W, mean @ W, 1)
Z, cs = ctx.pca_tmatmul(1, k + 10)
C = ctx.pca_gram(1, k + 10)
out["matmul_ms"] = round(ctx.stage_ms("pca_matmul"), 3)
out["tmatmul_ms"] = round(ctx.stage_ms("pca_tmatmul"), 3)
out["gram_ms"] = round(ctx.stage_ms("pca_gram"), 3)
flop = 2.0 * n * d * 128
out["matmul_TF_padded"] = round(flop / (out["matmul_ms"] * 1e-3) / 1e12, 1)
out["tmatmul_TF_padded"] = round(flop / (out["tmatmul_ms"] * 1e-3) / 1e12, 1)
out["X_GBps_matmul"] = round(4.0 * n * d / (out["matmul_ms"] * 1e-3) / 1e9, 1)
ctx.pca_end()
ctx.close()
if os.environ.get("GT_PCA_CPU", "1") == "1":
    from sklearn.decomposition import PCA

    m = min(n, 200000)
    t = time.perf_counter()
    sk = PCA(k, svd_solver="randomized", random_state=0).fit(X[:m])
    sk.transform(X[:m])
    out["sklearn_s_on_%d_rows" % m] = round(time.perf_counter() - t, 3)
    if m == n:
        out["sv_rel_diff"] = float(np.max(np.abs(dev.singular_values_ - sk.singular_values_) / sk.singular_values_))
print(json.dumps(out))
