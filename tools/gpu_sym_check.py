#!/usr/bin/env python
"""Development probe: the symmetric candidate pass (gt_sym.hip) against the classic pass and the oracle.
usage: gpu_sym_check.py [n_big]"""
import json
import os
import sys
import time

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from graphtools_amd import _hip  # noqa: E402

STAGES = ("prep", "query_order", "sym_prepare", "sym_seed", "sym_bound", "knn_select", "sym_cold", "rerank", "fallback", "radius", "affinity",
          "symmetrize", "normalize")


def make_mix(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    c = max(n // 2000, 1)
    centres = rng.uniform(-10, 10, (c, d))
    labels = rng.integers(c, size=n)
    return (centres[labels] + rng.standard_normal((n, d))).astype(dtype)


def build(ctx, X, sym, knn=15, decay=40.0, reps=1, **opts):
    ctx.set_option("select_symmetric", str(sym))
    for k, v in opts.items():
        ctx.set_option(k, str(v))
    ctx.set_points(X)
    p, keep = ctx.make_params(knn, decay, 1e-4, None, 1.0, None, "+", None, 0)
    best = None
    for _ in range(reps):
        t = time.time()
        nnz, fl = ctx.graph_build(p)
        wall = time.time() - t
        st = {s: round(ctx.stage_ms(s), 3) for s in STAGES}
        if best is None or wall < best["wall_s"]:
            best = {"wall_s": round(wall, 4), "nnz": int(nnz), "flags": int(fl), "stage_ms": st, "knn": ctx.knn_stats(),
                    "graph": ctx.graph_stats()}
    return best


def fetch(ctx, n):
    d_, i_, p_ = ctx.graph_fetch_csr(_hip.CSR_K)
    return d_, i_, p_


def same_csr(a, b):
    return bool(np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]))


def main():
    n_big = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rep = {}
    ctx = _hip.Context(0)
    # small cases with the pass forced on
    for name, X, kw in (("mix20k_d64", make_mix(20000, 64, 0), {}),
                        ("gauss30k_d20", np.random.default_rng(3).standard_normal((30000, 20)).astype(np.float32), {}),
                        ("mix50k_d50_f64", make_mix(50000, 50, 5, np.float64), {})):
        os.environ["GT_QUERY_ORDER_MIN_ROWS"] = "1"
        c2 = _hip.Context(0)
        r1 = build(c2, X, 1, select_sym_stride=4, select_sym_min_rows=1)
        k1 = fetch(c2, X.shape[0])
        r0 = build(c2, X, 0)
        k0 = fetch(c2, X.shape[0])
        rec = {"sym": r1, "classic": r0, "identical_K": same_csr(k1, k0)}
        if X.shape[0] <= 30000:
            Ko, Po = oracle.knn_graph(X, knn=15, decay=40.0)
            Ko = sparse.csr_matrix(Ko)
            Ko.sort_indices()
            ok = np.array_equal(k1[2], Ko.indptr) and np.array_equal(k1[1], Ko.indices)
            rec["oracle_structure"] = bool(ok)
            if ok:
                rec["oracle_max_rel"] = float(np.max(np.abs(k1[0] - Ko.data) / np.abs(Ko.data)))
        c2.close()
        rep[name] = rec
        print(name, json.dumps(rec), flush=True)
    # headline size
    X = make_mix(n_big, 64, 1)
    r1 = build(ctx, X, 1, reps=3)
    k1 = fetch(ctx, n_big)
    print("big sym", json.dumps(r1), flush=True)
    r0 = build(ctx, X, 0, reps=2)
    k0 = fetch(ctx, n_big)
    print("big classic", json.dumps(r0), flush=True)
    rep["big"] = {"n": n_big, "sym": r1, "classic": r0, "identical_K": same_csr(k1, k0)}
    print("big identical_K", rep["big"]["identical_K"], flush=True)
    ctx.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "gpu_sym_check.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
