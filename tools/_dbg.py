import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd())
from graphtools_amd import _hip
from bench import make_mix, make_manifold, make_gauss
for kind, mk in (("mix", make_mix), ("manifold", make_manifold)):
    X = mk(1000000, 64, 1)
    for coh in ("1", "0"):
        c = _hip.Context(0)
        c.set_option("query_order_coherent", coh)
        c.set_points(X)
        p, keep = c.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
        nnz, fl = c.graph_build(p)
        ts = []
        for _ in range(5):
            c.sync(); t = time.perf_counter(); c.set_points(X); c.graph_build(p); c.sync(); ts.append((time.perf_counter() - t) * 1e3)
        st = c.knn_stats()
        print(kind, "coherent", coh, "nnz", nnz, "ms", round(min(ts), 2), {k: st[k] for k in ("symmetric", "sym_far_kept", "sym_bound_pass", "sym_two_stage", "sym_cold_pairs") if k in st},
              {s: round(c.stage_ms(s), 2) for s in ("query_order", "sym_prepare", "sym_seed", "sym_bound", "sym_cold", "knn_select", "rerank", "affinity", "symmetrize")}, flush=True)
        c.close()
