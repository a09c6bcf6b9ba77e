#!/usr/bin/env python
"""Development probe: device memory that stays allocated after a context is closed and the cache released."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
from graphtools_amd import _hip  # noqa: E402
from tools.gpu_sym_check import make_mix  # noqa: E402


def free():
    return torch.cuda.mem_get_info(0)[0]


X = make_mix(int(sys.argv[1]) if len(sys.argv) > 1 else 60000, 16, 2)
graphtools_amd.release_cached_memory()
for what in ("points", "knn", "graph", "graph", "Graph", "Graph"):
    base = free()
    if what == "Graph":
        G = graphtools_amd.Graph(X, knn=10, decay=20, n_pca=None, verbose=0)
        G.K
        del G
    else:
        c = _hip.Context(0)
        c.set_points(X)
        if what == "knn":
            c.knn_search(11)
        if what == "graph":
            p, keep = c.make_params(10, 20, 1e-4, None, 1.0, None, "+", None, 0)
            c.graph_build(p)
        c.close()
    held = base - free()
    graphtools_amd.release_cached_memory()
    print(what, "parked MB", held >> 20, "after release MB", (base - free()) >> 20, flush=True)
