#!/usr/bin/env python
"""Where the host-complete time of BASELINE config 5 (kNN kernel + random-landmark operator) goes: cProfile of the second build.
usage: c5_host_profile.py [n]"""
import cProfile
import os
import pstats
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
from bench import make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = make_mix(n, 50, 3)


def run():
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=2000, random_landmarking=True, random_state=42, verbose=0)
    return G.landmark_op


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    run()
    pr = cProfile.Profile()
    pr.enable()
    run()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
