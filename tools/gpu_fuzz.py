#!/usr/bin/env python
"""Randomised parity sweep: graphtools_amd.Graph (kNN graphs) vs the oracle over random shapes and parameters.
usage: gpu_fuzz.py [n_cases] [seed]"""
import json
import os
import sys
import time
import warnings

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphtools_amd  # noqa: E402
import oracle  # noqa: E402


def make_data(rng, kind, n, d, dtype):
    if kind == "mix":
        c = max(n // 500, 1)
        centres = rng.uniform(-10, 10, (c, d))
        X = centres[rng.integers(c, size=n)] + rng.standard_normal((n, d))
    elif kind == "manifold":
        z = rng.standard_normal((n, min(5, d)))
        X = z @ rng.standard_normal((min(5, d), d)) + 0.01 * rng.standard_normal((n, d))
    elif kind == "gauss":
        X = rng.standard_normal((n, d))
    elif kind == "shifted":
        X = rng.standard_normal((n, d)) * 0.3 + 25.0
    else:   # lattice: many exact ties
        X = rng.integers(0, 4, size=(n, d)).astype(np.float64)
    return np.ascontiguousarray(X.astype(dtype))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    fails = []
    t_start = time.time()
    for case in range(n_cases):
        kind = rng.choice(["mix", "manifold", "gauss", "shifted", "lattice"], p=[0.35, 0.2, 0.2, 0.15, 0.1])
        n = int(rng.choice([int(rng.integers(200, 3000)), int(rng.integers(4096, 9000))]))
        d = int(rng.choice([2, 3, 7, 16, 20, 33, 50, 64, 65, 100, 128, 150, 260]))
        dtype = rng.choice([np.float32, np.float64])
        knn = int(rng.integers(2, 25))
        decay = rng.choice([None, 2.0, 10.0, 40.0])
        thresh = float(rng.choice([1e-4, 1e-3, 1e-2]))
        symm = rng.choice(["+", "*", "mnn", "none"])
        symm = None if symm == "none" else str(symm)
        theta = float(rng.uniform(0, 1)) if symm == "mnn" else None
        aniso = float(rng.choice([0.0, 0.0, 0.5, 1.0]))
        distance = str(rng.choice(["euclidean", "euclidean", "cosine"]))
        if distance == "cosine" and (kind == "lattice" or dtype == np.float32 or d < 3):
            # float32 cosine distances are float32 GEMM results in scikit-learn (summation order of the BLAS decides
            # near ties), the device orders by the float64 value: parity is only defined for float64 input (fixture G8)
            distance = "euclidean"
        bw_mode = rng.choice(["none", "none", "scalar", "vector"]) if decay is not None else "none"
        X = make_data(rng, kind, n, d, dtype)
        if distance == "cosine":
            X = X + dtype(0.0)   # keep rows non-zero for the mixtures used here
        bandwidth = None
        if bw_mode == "scalar":
            bandwidth = float(np.median(np.linalg.norm(X[:50] - X[50:100], axis=1)) * 0.5 + 1e-3)
        elif bw_mode == "vector":
            bandwidth = rng.uniform(0.5, 2.0, n) * float(np.median(np.linalg.norm(X[:50] - X[50:100], axis=1)) * 0.4 + 1e-3)
        knn_max = int(knn + rng.integers(0, 40)) if (decay is not None and rng.random() < 0.2) else None
        bw_scale = float(rng.choice([1.0, 1.0, 0.7, 1.5]))
        cfg = dict(kind=str(kind), n=n, d=d, dtype=np.dtype(dtype).name, knn=knn, decay=None if decay is None else float(decay),
                   thresh=thresh, symm=symm, theta=theta, aniso=aniso, distance=distance, bw=str(bw_mode), knn_max=knn_max,
                   bw_scale=bw_scale)
        if case < int(os.environ.get("GT_FUZZ_FIRST", "0")):
            continue    # (the random stream is drawn all the same: later cases keep their configuration)
        if os.environ.get("GT_FUZZ_ECHO"):
            print("case %d: %s" % (case, json.dumps(cfg)), file=sys.stderr, flush=True)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                G = graphtools_amd.Graph(X, knn=knn, decay=cfg["decay"], thresh=thresh, kernel_symm=symm, theta=theta,
                                         anisotropy=aniso, distance=distance, bandwidth=bandwidth, bandwidth_scale=bw_scale,
                                         knn_max=knn_max, n_pca=None, verbose=0)
                K, P = G.K, G.P
                Ko, Po = oracle.knn_graph(X, knn=knn, decay=cfg["decay"], thresh=thresh, bandwidth=bandwidth,
                                          bandwidth_scale=bw_scale, knn_max=knn_max, kernel_symm=symm, theta=theta,
                                          anisotropy=aniso, distance=distance)
            Ko = sparse.csr_matrix(Ko)
            Ko.sort_indices()
            if symm == "*":
                Ko.eliminate_zeros()
            ok_struct = np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
            err = 0.0
            if ok_struct and K.nnz:
                err = float(np.max(np.abs(K.data - Ko.data) / np.maximum(np.abs(Ko.data), 1e-300)))
                Po = sparse.csr_matrix(Po)
                Po.sort_indices()
                perr = float(np.max(np.abs(P.data - Po.data) / np.maximum(np.abs(Po.data), 1e-300)))
                err = max(err, perr)
            # bar: 1e-5 relative on the entries of the kernel; the "*" symmetrisation multiplies two of them
            tol = 2e-5 if symm == "*" else 1e-5
            status = "ok" if (ok_struct and err <= tol) else "FAIL"
            if not ok_struct and kind == "lattice":
                # exact distance ties at a neighbourhood boundary: scikit-learn's pick is unspecified; compare sets loosely
                status = "tie-structure"
            elif not ok_struct and cfg["decay"] is not None:
                # An affinity within rounding of `thresh` may land on either side of the cut: the float64 squared
                # distance depends on the summation order at the 1e-16 level, which flips its float32 rounding on a
                # ~1e-7 fraction of entries (oracle/knn.py header) and moves the affinity by ~decay * 1e-7 relative.
                # Accept a structural difference only if every differing entry of the UNSYMMETRISED kernel is such a
                # threshold tie (within the 1e-5 relative bar of the values).
                K0 = sparse.csr_matrix(G.build_kernel())
                K0o = sparse.csr_matrix(oracle.knn_kernel(X, knn=knn, decay=cfg["decay"], thresh=thresh, bandwidth=bandwidth,
                                                          bandwidth_scale=bw_scale, knn_max=knn_max, distance=distance))
                A = (K0 != 0).astype(np.int8)
                B = (K0o != 0).astype(np.int8)
                diff = (A - B).tocoo()
                vals = np.asarray((K0.maximum(K0o))[diff.row, diff.col]).ravel()
                if len(vals) and len(vals) <= 1e-5 * max(K0o.nnz, 1) + 8 and np.all(np.abs(vals - thresh) <= 1e-5 * thresh):
                    status = "threshold-tie"
        except Exception as e:   # noqa: BLE001
            status, err, ok_struct = "ERROR: %s: %s" % (type(e).__name__, str(e)[:200]), -1.0, False
        rec = dict(case=case, status=status, err=err, main=G.hip.last_knn_precision() if "G" in dir() else None, **cfg)
        print(json.dumps(rec), flush=True)
        if status not in ("ok", "tie-structure", "threshold-tie"):
            fails.append(rec)
    print(json.dumps({"cases": n_cases, "failures": len(fails), "seconds": round(time.time() - t_start, 1)}))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(fails, open(os.environ.get("GT_FUZZ_FAILURES") or os.path.join(ROOT, "gpurun_out", "gpu_fuzz_failures.json"), "w"), indent=1)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
