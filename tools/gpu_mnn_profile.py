#!/usr/bin/env python
"""Development probe: cProfile of MNNGraph at N = 3e5 (3 batches) - where does the host time go?"""
import cProfile
import os
import pstats
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    X = make_mix(n, 50, 5)
    idx = np.random.default_rng(5).choice(3, size=n, p=[0.45, 0.35, 0.2])

    def run():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            G = graphtools_amd.Graph(X, sample_idx=idx, knn=10, decay=20, n_pca=None, verbose=0)
            return G.P

    run()
    pr = cProfile.Profile()
    pr.enable()
    P = run()
    pr.disable()
    print("nnz", P.nnz)
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
