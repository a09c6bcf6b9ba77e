#!/usr/bin/env python
"""Where the host-complete build (host X in, scipy CSR K and P out: SURVEY 8d) spends its wall time.
usage: host_complete_probe.py [n] [d]"""
import json
import os
import sys
import time

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_mix  # noqa: E402
from graphtools_amd import _hip  # noqa: E402
import graphtools_amd  # noqa: E402

if os.environ.get("GT_PROBE_TORCH") == "1":   # the same with torch's HIP state in the process (bench.py has it)
    import torch

    torch.zeros(1000, device="cuda:0")
    torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
X = make_mix(n, d, 1)
res = []
for rep in range(4):
    t0 = time.perf_counter()
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, verbose=0, initialize=False)
    t1 = time.perf_counter()
    G._bind_points()
    t2 = time.perf_counter()
    nnz, flags = G._device_build(G.kernel_symm, G.theta, G.anisotropy)
    G.hip.sync()
    t3 = time.perf_counter()
    stages = {s_: round(G.hip.stage_ms(s_), 3) for s_ in ("prep", "query_order", "sym_prepare", "sym_seed", "sym_bound", "sym_cold", "knn_select",
                                                          "rerank", "fallback", "radius", "affinity", "symmetrize", "normalize") if G.hip.stage_ms(s_) > 0}
    kd, ki, kp, pd = G.hip.graph_fetch_kp()
    t4 = time.perf_counter()
    kp32 = kp.astype(np.int32)
    K = sparse.csr_matrix((kd, ki, kp32), shape=(n, n))
    P = sparse.csr_matrix((pd, K.indices, K.indptr), shape=(n, n))
    t5 = time.perf_counter()
    res.append({"ctor_ms": (t1 - t0) * 1e3, "set_points_ms": (t2 - t1) * 1e3, "build_ms": (t3 - t2) * 1e3, "fetch_kp_ms": (t4 - t3) * 1e3,
                "scipy_ms": (t5 - t4) * 1e3, "stage_sum_ms": round(sum(stages.values()), 3), "stages": stages, "total_ms": (t5 - t0) * 1e3, "fetch_GBs": (kd.nbytes + ki.nbytes + kp.nbytes) / (t4 - t3) / 1e9})
    del G, K, P, kd, ki, kp, pd
    t0 = time.perf_counter()
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, verbose=0)
    K, P = G.K, G.P
    res[-1]["graph_api_total_ms"] = (time.perf_counter() - t0) * 1e3
    del G, K, P
print(json.dumps(res, indent=1))
