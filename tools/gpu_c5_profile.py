#!/usr/bin/env python
"""Development probe: cProfile of BASELINE config 5 on one GPU (N=1e6, d=50, kNN kernel + landmark operator with 2000
random landmarks) through graphtools_amd.Graph - where does the host time go?"""
import cProfile
import os
import pstats
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402


def run(X):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=2000, random_landmarking=True,
                                 random_state=42, verbose=0)
        G.landmark_op
    return G


if __name__ == "__main__":
    X = make_mix(1000000, 50, 3)
    run(X)
    pr = cProfile.Profile()
    pr.enable()
    run(X)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
