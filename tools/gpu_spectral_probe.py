#!/usr/bin/env python
"""Development probe: default (spectral) landmarking at scale - the device front end phase by phase.
usage: gpu_spectral_probe.py [n] [d] [n_landmark] [n_svd]"""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
from graphtools_amd import _spectral  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 50
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
n_svd = int(sys.argv[4]) if len(sys.argv) > 4 else 100
X = make_mix(n, d, 3)
out = {"n": n, "d": d, "n_landmark": L, "n_svd": n_svd}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    t = time.perf_counter()
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=L, n_svd=n_svd, random_state=42, verbose=0)
    G.K
    out["graph_K_s"] = round(time.perf_counter() - t, 3)
    G._ensure_device_graph()
    t = time.perf_counter()
    E, S = _spectral.spectral_embedding(G.hip, n, n_svd, 42)
    out["embedding_s"] = round(time.perf_counter() - t, 3)
    out["spmm_ms_total"] = round(G.hip.stage_ms("spmm"), 1)
    out["singular_values_head"] = [round(float(v), 6) for v in S[:4]]
    t = time.perf_counter()
    op = G.landmark_op
    out["landmark_op_total_s"] = round(time.perf_counter() - t, 3)
    out["n_clusters"] = int(len(np.unique(G.clusters)))
    out["op_row_sum_err"] = float(np.abs(op.sum(axis=1) - 1).max())
print(json.dumps(out))
