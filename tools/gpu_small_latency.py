#!/usr/bin/env python
"""Development probe: wall time of small graphs (reference plumbing size, config 1) through graphtools_amd.Graph."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

if __name__ == "__main__":
    for n, d in ((1797, 64), (10000, 50), (100000, 50)):
        X = make_mix(n, d, 0)
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            G = graphtools_amd.Graph(X, knn=5, decay=40, n_pca=None, verbose=0)
            P = G.P
            ts.append(time.perf_counter() - t0)
            del G, P
        print(json.dumps({"n": n, "d": d, "first_s": round(ts[0], 4), "best_s": round(min(ts[1:]), 4)}), flush=True)
