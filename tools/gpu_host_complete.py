#!/usr/bin/env python
"""Development probe: 'host-complete' wall time of graphtools_amd.Graph at benchmark size - host float32 X in,
scipy CSR K and P out (H2D of X, device build, D2H of the results, CSR wrapping)."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    X = make_mix(n, 64, 1)
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, verbose=0)
        t1 = time.perf_counter()
        P = G.P
        t2 = time.perf_counter()
        out.append({"graph_K_s": round(t1 - t0, 3), "P_s": round(t2 - t1, 3), "total_s": round(t2 - t0, 3), "nnz": int(G.K.nnz)})
        print(json.dumps(out[-1]), flush=True)
        del G, P
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gpu_host_complete.json"), "w"))
