#!/usr/bin/env python
"""Development probe: cProfile of the host-complete path at benchmark size (where does the host time go?)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    X = make_mix(n, 64, 1)
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, verbose=0)
    P = G.P
    del G, P
    pr = cProfile.Profile()
    pr.enable()
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, verbose=0)
    P = G.P
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
