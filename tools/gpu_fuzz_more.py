#!/usr/bin/env python
"""Randomised parity sweep for the other graph types: exact dense graphs, out-of-sample extension, MNN graphs.
usage: gpu_fuzz_more.py [n_cases] [seed]"""
import json
import os
import sys
import warnings

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
import oracle  # noqa: E402
from tools.gpu_fuzz import make_data  # noqa: E402


def csr_check(A, B, rtol=1e-5):
    A = sparse.csr_matrix(A)
    B = sparse.csr_matrix(B)
    A.sort_indices()
    B.sort_indices()
    if not (np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)):
        return False, -1.0
    if A.nnz == 0:
        return True, 0.0
    err = float(np.max(np.abs(A.data - B.data) / np.maximum(np.abs(B.data), 1e-300)))
    return err <= rtol, err


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    fails = []
    for case in range(n_cases):
        mode = str(rng.choice(["exact", "extend", "mnn"]))
        kind = str(rng.choice(["mix", "manifold", "gauss"]))
        d = int(rng.choice([3, 10, 20, 50, 64, 100]))
        knn = int(rng.integers(3, 16))
        decay = float(rng.choice([5.0, 15.0, 40.0]))
        cfg = dict(case=case, mode=mode, kind=kind, d=d, knn=knn, decay=decay)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if mode == "exact":
                    n = int(rng.integers(100, 1200))
                    X = make_data(rng, kind, n, d, np.float32)
                    symm = str(rng.choice(["+", "*", "mnn"]))
                    theta = float(rng.uniform(0, 1)) if symm == "mnn" else None
                    aniso = float(rng.choice([0.0, 0.5]))
                    thresh = float(rng.choice([0.0, 1e-4]))
                    cfg.update(n=n, symm=symm, thresh=thresh, aniso=aniso)
                    G = graphtools_amd.Graph(X, knn=knn, decay=decay, thresh=thresh, kernel_symm=symm, theta=theta,
                                             anisotropy=aniso, graphtype="exact", n_pca=None, verbose=0)
                    Ko, Po = oracle.exact_graph(X, knn=knn, decay=decay, thresh=thresh, kernel_symm=symm, theta=theta,
                                                anisotropy=aniso)
                    ok = np.allclose(G.K, Ko, rtol=1e-8, atol=1e-300) and np.allclose(G.P, Po, rtol=1e-8, atol=1e-300)
                    err = float(np.max(np.abs(G.K - Ko)))
                elif mode == "extend":
                    n = int(rng.choice([int(rng.integers(300, 3000)), int(rng.integers(4096, 8000))]))
                    m = int(rng.integers(50, 5000))
                    X = make_data(rng, kind, n + m, d, np.float32)
                    X, Y = X[:n], X[n:]
                    thresh = float(rng.choice([1e-4, 1e-2]))
                    cfg.update(n=n, m=m, thresh=thresh)
                    G = graphtools_amd.Graph(X, knn=knn, decay=decay, thresh=thresh, n_pca=None, verbose=0)
                    Kyx = G.build_kernel_to_data(Y)
                    Ko = oracle.knn_kernel(X, knn=knn, decay=decay, thresh=thresh, Y=Y)
                    ok, err = csr_check(Kyx, Ko)
                    if ok:
                        T = G.extend_to_data(Y)
                        ok, err2 = csr_check(T, oracle.kernel.diff_op_fast(sparse.csr_matrix(Ko)))
                        err = max(err, err2)
                else:
                    n = int(rng.integers(300, 2500))
                    X = make_data(rng, kind, n, d, np.float32)
                    nb = int(rng.integers(2, 5))
                    idx = rng.integers(0, nb, size=n)
                    X = (X + rng.standard_normal((nb, d)).astype(np.float32)[idx] * 0.3).astype(np.float32)
                    symm = str(rng.choice(["+", "mnn"]))
                    theta = float(rng.uniform(0, 1)) if symm == "mnn" else None
                    beta = float(rng.choice([1.0, 0.5]))
                    dec = None if rng.random() < 0.3 else decay
                    cfg.update(n=n, batches=nb, symm=symm, beta=beta, decay=dec)
                    G = graphtools_amd.Graph(X, sample_idx=idx, knn=knn, decay=dec, beta=beta, kernel_symm=symm, theta=theta,
                                             n_pca=None, verbose=0)
                    K0o, Ko, Po = oracle.mnn_graph(X, idx, knn=knn, decay=dec, beta=beta, kernel_symm=symm, theta=theta)
                    ok, err = csr_check(G.K, Ko)
                    if ok:
                        ok, err2 = csr_check(G.P, Po)
                        err = max(err, err2)
            status = "ok" if ok else "FAIL"
        except Exception as e:   # noqa: BLE001
            status, err = "ERROR: %s: %s" % (type(e).__name__, str(e)[:200]), -1.0
        rec = dict(status=status, err=err, **cfg)
        print(json.dumps(rec), flush=True)
        if status != "ok":
            fails.append(rec)
    print(json.dumps({"cases": n_cases, "failures": len(fails)}))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
