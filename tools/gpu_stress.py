#!/usr/bin/env python
"""Development probe: the other BASELINE configs at (near) full size - timings and invariants."""
import json, os, sys, time
import numpy as np
from scipy import sparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd
from graphtools_amd import _hip
from tools.gpu_perf import make_mix

out = {}
which = sys.argv[1:] or ["gauss", "dense", "landmark"]

if "gauss" in which:
    n = 200000
    X = np.random.default_rng(1).standard_normal((n, 64)).astype(np.float32)
    ctx = _hip.Context(0)
    ctx.set_points(X)
    p, keep = ctx.make_params(15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    t = time.time(); nnz, fl = ctx.graph_build(p); wall = time.time() - t
    t = time.time(); nnz, fl = ctx.graph_build(p); wall2 = time.time() - t
    st = ctx.graph_stats()
    out["gauss_200k"] = {"wall_s": wall2, "first_wall_s": wall, "nnz": nnz, "nnz_per_row": nnz / n, "stats": st,
                         "stage_ms": {s: round(ctx.stage_ms(s), 2) for s in ("knn_select", "rerank", "radius", "affinity", "symmetrize", "normalize")}}
    Pd, Pi, Pp = ctx.graph_fetch_csr(_hip.CSR_P)
    P = sparse.csr_matrix((Pd, Pi, Pp), shape=(n, n))
    out["gauss_200k"]["row_sum_err"] = float(np.abs(np.asarray(P.sum(axis=1)).ravel() - 1).max())
    print(json.dumps(out["gauss_200k"])); ctx.close()

if "dense" in which:
    n, d = 30000, 100
    X = make_mix(n, d, 2)
    ctx = _hip.Context(0)
    t = time.time()
    K, P, fl = ctx.dense_graph_build(X, False, 15, 40, 1e-4, None, 1.0, "+", None, 0)
    wall = time.time() - t
    out["dense_from_data_30k"] = {"wall_s_incl_d2h": wall, "stage_ms": {s: round(ctx.stage_ms(s), 2) for s in ("knn_select", "rerank", "dense_bandwidth", "dense_kernel", "dense_normalize")},
                                  "row_sum_err": float(np.abs(P.sum(axis=1) - 1).max()), "sym_err": float(np.abs(K - K.T).max()), "diag_min": float(K.diagonal().min())}
    print(json.dumps(out["dense_from_data_30k"]))
    del K, P
    # precomputed float32 distances, device resident, in place
    from scipy.spatial.distance import cdist
    Xs = X[:20000]
    D = cdist(Xs, Xs).astype(np.float32)
    t = time.time()
    K, P, fl = ctx.dense_graph_build(D, True, 15, 40, 1e-4, None, 1.0, "+", None, 0)
    wall = time.time() - t
    out["dense_from_D32_20k"] = {"wall_s_incl_h2d_d2h": wall, "stage_ms": {s: round(ctx.stage_ms(s), 2) for s in ("dense_bandwidth", "dense_kernel", "dense_normalize")},
                                 "GBs_kernel": round(2 * 4 * 20000**2 / (ctx.stage_ms("dense_kernel") * 1e-3) / 1e9, 1),
                                 "row_sum_err": float(np.abs(P.sum(axis=1) - 1).max())}
    print(json.dumps(out["dense_from_D32_20k"])); ctx.close()

if "landmark" in which:
    n, d, L = 1000000, 50, 2000
    X = make_mix(n, d, 3)
    t = time.time()
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=L, random_landmarking=True, random_state=42)
    t_kernel = time.time() - t
    t = time.time(); cl = G._assign_clusters(); G._clusters = cl; t_as = time.time() - t
    as_ms = G.hip.stage_ms("landmark_assign")
    t = time.time(); op = G.landmark_op; t_lm = time.time() - t
    out["landmark_1e6"] = {"kernel_wall_s_incl_fetch": t_kernel, "assign_wall_s": t_as, "assign_stage_ms": as_ms, "landmark_wall_s": t_lm, "landmark_stage_ms": G.hip.stage_ms("landmark"),
                           "row_sum_err": float(np.abs(op.sum(axis=1) - 1).max()), "n_clusters": int(len(np.unique(G.clusters))),
                           "transitions_nnz_per_row": G.transitions.nnz / n}
    print(json.dumps(out["landmark_1e6"]))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gpu_stress.json"), "w"), indent=1)
