#!/usr/bin/env python
"""Development probe: graphs of varying size built one after the other in one process - wall time per graph and the
device memory the library holds (process-wide cache of released workspace, gt_devpool.cpp) after each."""
import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch, graphtools_amd
from tools.gpu_perf import make_mix
torch.cuda.init()
base = torch.cuda.mem_get_info(0)[0]
out = []
for n in (100000, 300000, 1000000, 200000, 1000000, 50000, 600000):
    X = make_mix(n, 64, 1)
    t0 = time.perf_counter()
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, verbose=0); nnz = G.P.nnz
    dt = time.perf_counter() - t0
    del G
    out.append({"n": n, "s": round(dt, 3), "held_GB": round((base - torch.cuda.mem_get_info(0)[0]) / 2**30, 2)})
    print(json.dumps(out[-1]), flush=True)
graphtools_amd.release_cached_memory()
print("after release held_GB", round((base - torch.cuda.mem_get_info(0)[0]) / 2**30, 2))
