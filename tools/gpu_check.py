#!/usr/bin/env python
"""Development probe for the GPU box: runs the HIP path against the oracle on a few inputs and
prints detailed diagnostics (also written to gpurun_out/gpu_check.json).  Not a test - see tests/."""
import json
import os
import sys
import time
import traceback

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from graphtools_amd import _hip  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
report = {}


def make_mix(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    c = max(n // 2000, 1)
    centres = rng.uniform(-10, 10, (c, d))
    labels = rng.integers(c, size=n)
    return (centres[labels] + rng.standard_normal((n, d))).astype(dtype)


def make_gauss(n, d, seed, dtype=np.float32):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(dtype)


def cmp_knn(ctx, name, X, k):
    t = time.time()
    ctx.set_points(X)
    d, i, fl = ctx.knn_search(k)
    t_gpu = time.time() - t
    t = time.time()
    d0, i0 = oracle.kneighbors(X, None, k)
    t_cpu = time.time() - t
    same = i == i0
    rows_bad = int((~same.all(axis=1)).sum())
    # tie-aware: positions that differ must have equal oracle distances
    tie_ok = bool(np.all(d0[~same] == np.take_along_axis(d0, np.argsort(np.argsort(i0, axis=1), axis=1), axis=1)[~same])) if rows_bad else True
    dd = np.abs(d[:, 1:] - d0[:, 1:])
    rep = {
        "n": int(X.shape[0]), "d": int(X.shape[1]), "k": k, "dtype": str(X.dtype), "flags": int(fl),
        "idx_mismatch_rows": rows_bad, "idx_mismatch_entries": int((~same).sum()),
        "dist_max_abs_diff_cols1+": float(dd.max()), "dist_mismatch_entries": int((dd != 0).sum()),
        "dist_sorted_equal": bool(np.array_equal(np.sort(d, axis=1)[:, 1:], np.sort(d0, axis=1)[:, 1:])),
        "col0_max": float(np.abs(d[:, 0]).max()),
        "t_gpu_s": t_gpu, "t_oracle_s": t_cpu,
        "stage_ms": {s: ctx.stage_ms(s) for s in ("prep", "knn_select", "rerank", "fallback")},
    }
    if rows_bad:
        r = int(np.nonzero(~same.all(axis=1))[0][0])
        c = int(np.nonzero(~same[r])[0][0])
        rep["first_bad"] = {"row": r, "col": c, "gpu_idx": i[r, max(0, c - 2):c + 3].tolist(), "ref_idx": i0[r, max(0, c - 2):c + 3].tolist(),
                            "gpu_d": d[r, max(0, c - 2):c + 3].tolist(), "ref_d": d0[r, max(0, c - 2):c + 3].tolist()}
    report["knn_" + name] = rep
    print("knn", name, json.dumps(rep))


def cmp_graph(ctx, name, X, **kw):
    knn = kw.pop("knn", 15)
    decay = kw.pop("decay", 40)
    thresh = kw.pop("thresh", 1e-4)
    bandwidth = kw.pop("bandwidth", None)
    bandwidth_scale = kw.pop("bandwidth_scale", 1.0)
    knn_max = kw.pop("knn_max", None)
    kernel_symm = kw.pop("kernel_symm", "+")
    theta = kw.pop("theta", None)
    anisotropy = kw.pop("anisotropy", 0)
    t = time.time()
    ctx.set_points(X)
    p, keep = ctx.make_params(knn, decay, thresh, bandwidth, bandwidth_scale, knn_max, kernel_symm, theta, anisotropy)
    nnz, fl = ctx.graph_build(p)
    Kd, Ki, Kp = ctx.graph_fetch_csr(_hip.CSR_K)
    Pd, _, _ = ctx.graph_fetch_csr(_hip.CSR_P)
    t_gpu = time.time() - t
    n = X.shape[0]
    K = sparse.csr_matrix((Kd, Ki, Kp), shape=(n, n))
    P = sparse.csr_matrix((Pd, Ki, Kp), shape=(n, n))
    t = time.time()
    K0, P0 = oracle.knn_graph(X, knn=knn, decay=decay, thresh=thresh, bandwidth=bandwidth, bandwidth_scale=bandwidth_scale,
                              knn_max=knn_max, kernel_symm=kernel_symm, theta=theta, anisotropy=anisotropy)
    t_cpu = time.time() - t
    K0 = sparse.csr_matrix(K0); K0.sort_indices(); K0.eliminate_zeros()
    same_struct = (K.nnz == K0.nnz) and np.array_equal(K.indptr, K0.indptr) and np.array_equal(K.indices, K0.indices)
    rep = {"n": n, "nnz": int(nnz), "nnz_ref": int(K0.nnz), "flags": int(fl), "same_struct": bool(same_struct),
           "canonical": bool(K.has_canonical_format), "t_gpu_s": t_gpu, "t_oracle_s": t_cpu, "stats": ctx.graph_stats(),
           "stage_ms": {s: ctx.stage_ms(s) for s in ("knn_select", "rerank", "fallback", "radius", "affinity", "symmetrize", "normalize")}}
    if same_struct:
        rel = np.abs(K.data - K0.data) / np.abs(K0.data)
        relp = np.abs(P.data - sparse.csr_matrix(P0).data) / np.abs(sparse.csr_matrix(P0).data)
        rep["K_max_rel"] = float(rel.max())
        rep["P_max_rel"] = float(relp.max())
        rep["K_n_rel_gt_1e-5"] = int((rel > 1e-5).sum())
    else:
        D = abs(K - K0)
        rep["struct_diff_entries"] = int((K != K0).nnz)
        rep["max_abs_diff"] = float(D.max())
        Dc = D.tocoo()
        big = Dc.data > 2e-4
        rep["n_abs_diff_gt_2e-4"] = int(big.sum())
    report["graph_" + name] = rep
    print("graph", name, json.dumps(rep))



def probes(ctx):
    import ctypes
    lib = ctx.lib
    rng = np.random.default_rng(0)
    rep = {}
    # MFMA layout
    K = 8
    a = rng.standard_normal((32, K)).astype(np.float32)
    bt = rng.standard_normal((32, K)).astype(np.float32)
    c = np.zeros((32, 32), dtype=np.float32)
    lib.gt_dbg_mfma.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
    rc = lib.gt_dbg_mfma(ctx.h, a.ctypes.data, bt.ctypes.data, K, c.ctypes.data)
    rep["mfma_rc"] = rc
    rep["mfma_max_err"] = float(np.abs(c - a @ bt.T).max())
    # descending sort
    lib.gt_dbg_sort_desc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    ok = True
    for nt in (1, 2, 8, 32):
        for n in (1, 63, 64 * nt - 5, 64 * nt):
            n = max(1, min(n, 64 * nt))
            keys = rng.integers(1, 2**63, size=n, dtype=np.uint64)
            out = np.zeros(64 * nt, dtype=np.uint64)
            rc = lib.gt_dbg_sort_desc(ctx.h, keys.ctypes.data, n, nt, out.ctypes.data)
            exp = np.zeros(64 * nt, dtype=np.uint64)
            exp[:n] = np.sort(keys)[::-1]
            if rc != 0 or not np.array_equal(out, exp):
                ok = False
                rep["sort_desc_fail_%d_%d" % (nt, n)] = int((out != exp).sum())
    rep["sort_desc_ok"] = ok
    lib.gt_dbg_sort_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    ok = True
    for nt in (1, 2, 4, 8):
        for n in (1, 63, 64 * nt - 5, 64 * nt):
            n = max(1, min(n, 64 * nt))
            hi = rng.integers(0, 50, size=n, dtype=np.uint64)   # many ties -> exercises the lo tie-break
            lo = rng.permutation(n).astype(np.uint64)
            oh = np.zeros(64 * nt, dtype=np.uint64)
            ol = np.zeros(64 * nt, dtype=np.uint64)
            rc = lib.gt_dbg_sort_pair(ctx.h, hi.ctypes.data, lo.ctypes.data, n, nt, oh.ctypes.data, ol.ctypes.data)
            order = np.lexsort((lo, hi))
            if rc != 0 or not (np.array_equal(oh[:n], hi[order]) and np.array_equal(ol[:n], lo[order]) and np.all(oh[n:] == np.uint64(2**64 - 1))):
                ok = False
                rep["sort_pair_fail_%d_%d" % (nt, n)] = int((oh[:n] != hi[order]).sum())
    rep["sort_pair_ok"] = ok
    report["probes"] = rep
    print("probes", json.dumps(rep))

def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    ctx = _hip.Context(0)
    try:
        probes(ctx)
    except Exception as e:
        report['probes'] = {'error': repr(e), 'tb': traceback.format_exc()}
        print('probes ERROR', repr(e))
    tests = [
        ("knn", "mix1024_d50", lambda: (make_mix(1024, 50, 0), 96)),
        ("knn", "mix5000_d64", lambda: (make_mix(5000, 64, 1), 96)),
        ("knn", "gauss3000_d64", lambda: (make_gauss(3000, 64, 1), 96)),
        ("knn", "mix3000_d100", lambda: (make_mix(3000, 100, 2), 96)),
        ("knn", "mix2000_d20", lambda: (make_mix(2000, 20, 3), 30)),
        ("knn", "mix777_d50_f64", lambda: (make_mix(777, 50, 4, np.float64), 66)),
        ("knn", "mix20000_d64", lambda: (make_mix(20000, 64, 5), 96)),
        ("knn", "mix4000_d64_k300", lambda: (make_mix(4000, 64, 6), 300)),
    ]
    try:
        from sklearn import datasets
        digits = datasets.load_digits().data
        tests.append(("knn", "digits", lambda: (digits, 36)))
    except Exception:
        digits = None
    for kind, name, mk in tests:
        try:
            X, k = mk()
            cmp_knn(ctx, name, X, k)
        except Exception as e:
            report["knn_" + name] = {"error": repr(e), "tb": traceback.format_exc()}
            print("knn", name, "ERROR", repr(e))
    gtests = [
        ("mix1024_d50", lambda: make_mix(1024, 50, 0), {}),
        ("gauss1536_d64", lambda: make_gauss(1536, 64, 1), {}),
        ("mix5000_d64", lambda: make_mix(5000, 64, 1), {}),
        ("mix20000_d64", lambda: make_mix(20000, 64, 5), {}),
        ("mix2000_binary", lambda: make_mix(2000, 50, 6), {"decay": None}),
        ("mix384_mul", lambda: make_mix(384, 50, 5), {"knn": 10, "decay": 20, "kernel_symm": "*"}),
        ("mix384_mnn", lambda: make_mix(384, 50, 5), {"knn": 10, "decay": 20, "kernel_symm": "mnn", "theta": 0.7}),
        ("mix384_none", lambda: make_mix(384, 50, 5), {"knn": 10, "decay": 20, "kernel_symm": None}),
        ("mix384_aniso", lambda: make_mix(384, 50, 5), {"knn": 10, "decay": 20, "anisotropy": 0.5}),
        ("mix384_bw", lambda: make_mix(384, 50, 5), {"knn": 10, "decay": 20, "bandwidth": 7.0, "bandwidth_scale": 1.1}),
        ("mix384_knnmax", lambda: make_mix(384, 50, 5), {"knn": 10, "decay": 20, "knn_max": 40}),
    ]
    if digits is not None:
        gtests.append(("digits", lambda: digits, {"knn": 5, "decay": 40}))
    for name, mk, kw in gtests:
        try:
            cmp_graph(ctx, name, mk(), **dict(kw))
        except Exception as e:
            report["graph_" + name] = {"error": repr(e), "tb": traceback.format_exc()}
            print("graph", name, "ERROR", repr(e))
    with open(os.path.join(ROOT, "gpurun_out", "gpu_check.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
