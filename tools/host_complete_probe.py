#!/usr/bin/env python
"""Development probe: host-complete time of one C3 graph (numpy in, scipy CSR K and P out), for the copy-lane settings in
the environment (GT_COPY_LANES, GT_COPY_SLOT_MB)."""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_mix  # noqa: E402
import graphtools_amd  # noqa: E402

X = make_mix(1000000, 64, 1)
ts = []
for _ in range(4):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, verbose=0)
        K, P = G.K, G.P
        ts.append(time.perf_counter() - t0)
    del G, K, P
print("lanes %s slot %s MB: %s ms" % (os.environ.get("GT_COPY_LANES", "8"), os.environ.get("GT_COPY_SLOT_MB", "8"), [round(t * 1e3, 1) for t in ts]))
