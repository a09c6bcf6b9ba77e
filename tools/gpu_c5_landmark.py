#!/usr/bin/env python
"""Development probe: BASELINE config 5 on one GPU - mix N=1e6 d=50, kNNGraph knn=15 decay=40 + landmark operator
(n_landmark=2000, random landmarking): wall times of the kernel build, the cluster assignment and the operator."""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    X = make_mix(n, 50, 3)
    out = []
    for rep in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=2000, random_landmarking=True,
                                     random_state=42, verbose=0)
            t1 = time.perf_counter()
            cl = G.clusters if False else None
            op = G.landmark_op
            t2 = time.perf_counter()
        out.append({"graph_K_s": round(t1 - t0, 3), "landmark_op_s": round(t2 - t1, 3), "L": int(op.shape[0]),
                    "op_row_sum_err": float(np.abs(op.sum(axis=1) - 1).max()), "transitions_nnz": int(G.transitions.nnz)})
        print(json.dumps(out[-1]), flush=True)
        del G
