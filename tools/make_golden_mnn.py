#!/usr/bin/env python
"""Generate tests/golden/g9_mnn*.npz with the REAL reference's MNNGraph (build container only; see
tools/make_golden.py for the rules) and cross-check the numpy oracle against them."""
import os
import sys
import warnings

import numpy as np
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from make_golden import OUT, csr_parts, make_mix  # noqa: E402
from ref_import import import_reference  # noqa: E402


def batches(n, d, seed):
    """three batches of one mixture, the second and third shifted (a batch effect)"""
    X = make_mix(n, d, seed)
    rng = np.random.default_rng(seed + 100)
    idx = rng.choice(3, size=n, p=[0.45, 0.35, 0.2])
    shift = rng.standard_normal((3, d)).astype(np.float32) * 0.5
    shift[0] = 0
    return (X + shift[idx]).astype(np.float32), idx.astype(np.int64)


def main():
    gt = import_reference()
    import oracle

    X, idx = batches(600, 20, 11)
    cases = {
        "g9_mnn_decay": dict(knn=5, decay=20, thresh=1e-4, beta=1, kernel_symm="+", theta=None, anisotropy=0),
        "g9b_mnn_binary_theta": dict(knn=4, decay=None, thresh=1e-4, beta=0.5, kernel_symm="mnn", theta=0.7, anisotropy=0),
        "g9c_mnn_aniso": dict(knn=5, decay=20, thresh=1e-3, beta=0.8, kernel_symm="*", theta=None, anisotropy=0.5),
    }
    for name, kw in cases.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            G = gt.Graph(X, sample_idx=idx, n_pca=None, verbose=0, **kw)
            K0 = sparse.csr_matrix(G.build_kernel())
            K = sparse.csr_matrix(G.K)
            P = sparse.csr_matrix(G.P)
        out = {"X": X, "sample_idx": idx}
        for k, v in kw.items():
            out["param_" + k] = np.array(np.nan if v is None else v)
        out.update(csr_parts("K0", K0))
        out.update(csr_parts("K", K))
        Pc = P.copy()
        Pc.sort_indices()
        out["P_data"] = Pc.data
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        # oracle cross-check
        okw = dict(kw)
        O0, OK, OP = oracle.mnn_graph(X, idx, engine="sklearn", **okw)
        O0.sort_indices()
        d0 = abs(O0 - K0).max()
        dK = abs(OK - K).max()
        dP = abs(OP - P).max()
        same = np.array_equal(OK.indices, csr_sorted(K).indices) and np.array_equal(OK.indptr, K.indptr)
        print("%-24s %6.2f MB  oracle: |dK0| %.2e |dK| %.2e |dP| %.2e structure %s" % (
            name, os.path.getsize(path) / 1e6, d0, dK, dP, same))
    landmark_case(gt, oracle, X, idx)


def landmark_case(gt, oracle, X, idx):
    """MNNLandmarkGraph (graphs.py:1973-1974), random landmarking: clusters, landmark operator, transitions."""
    kw = dict(knn=5, decay=20, thresh=1e-4, beta=1, kernel_symm="+", theta=None, anisotropy=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = gt.Graph(X, sample_idx=idx, n_pca=None, verbose=0, n_landmark=60, random_landmarking=True, random_state=42,
                     **kw)
        assert type(G).__name__ == "MNNLandmarkGraph"
        op = np.asarray(G.landmark_op)
        T = sparse.csr_matrix(G.transitions)
        clusters = np.asarray(G.clusters)
    out = {"X": X, "sample_idx": idx, "n_landmark": np.array(60), "random_state": np.array(42), "clusters": clusters,
           "landmark_op": op}
    for k, v in kw.items():
        out["param_" + k] = np.array(np.nan if v is None else v)
    T.sort_indices()
    out.update(csr_parts("transitions", T))
    path = os.path.join(OUT, "g9d_mnn_landmark.npz")
    np.savez_compressed(path, **out)
    _, OK, _ = oracle.mnn_graph(X, idx, engine="sklearn", **kw)
    oc, _ = oracle.random_landmark_clusters(X, 60, 42)
    oop, otr = oracle.landmark_operator(OK, oc)
    print("%-24s %6.2f MB  oracle: clusters equal %s |dop| %.2e |dT| %.2e" % (
        "g9d_mnn_landmark", os.path.getsize(path) / 1e6, np.array_equal(oc, clusters), abs(oop - op).max(),
        abs(sparse.csr_matrix(otr) - T).max()))


def csr_sorted(M):
    M = sparse.csr_matrix(M).copy()
    M.sort_indices()
    return M


if __name__ == "__main__":
    main()
