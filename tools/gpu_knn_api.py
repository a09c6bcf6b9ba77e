import sys, time, json, numpy as np
sys.path.insert(0, '/root/repo')
from graphtools_amd import _hip
from tools.gpu_perf import make_mix
X = make_mix(1000000, 64, 1)
ctx = _hip.Context(0)
import os
if os.environ.get('GT_NT8_MAX'):
    ctx.set_option('select_nt8_max_need', os.environ['GT_NT8_MAX'])
ctx.set_points(X)
for k in (int(v) for v in os.environ.get('GT_KS', '16,96').split(',')):
    t0 = time.perf_counter(); d, i, fl = ctx.knn_search(k); t1 = time.perf_counter()
    t0 = time.perf_counter(); d, i, fl = ctx.knn_search(k); t1 = time.perf_counter()
    print(json.dumps({"k": k, "wall_s": round(t1 - t0, 3), "stage_ms": {s: round(ctx.stage_ms(s), 2) for s in ("query_order", "knn_select", "rerank", "fallback")}, "main": ctx.last_knn_precision()}))
