import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_ladder as tl
from graphtools_amd import _hip
tl.N = 1000000
X = tl.FAMILIES["mix with 15 isolated points d=48"]()
for coh in ("1", "0"):
    c = _hip.Context(0)
    c.set_option("query_order_coherent", coh)
    c.set_option("dbg_select", "2048")
    c.set_points(X)
    p, keep = c.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
    print("---- coherent", coh, flush=True)
    c.graph_build(p)
    c.set_option("dbg_select", "0")
    c.set_points(X); c.sync(); t = time.perf_counter(); c.graph_build(p); c.sync(); ms = (time.perf_counter() - t) * 1e3
    st = c.knn_stats()
    print("ms %.1f" % ms, {k: st[k] for k in ("symmetric", "sym_far_kept", "sym_bound_pass", "sym_two_stage", "sym_cold_pairs", "sym_overflow_rows", "repaired_rows") if k in st},
          {s: round(c.stage_ms(s), 2) for s in tl.TOP_STAGES}, flush=True)
    c.close()
