#!/usr/bin/env python
"""Per-rank time of the row-sharded build where the symmetric pass does not apply (isotropic data: the classic candidate
pass over the rank's rows), simulated on one GPU: every rank's gt_graph_begin + emit in turn, the all-to-all by hand,
then rank `who` timed back to back (begin, emit, finish).  usage: gpu_shard_classic_probe.py [n] [d] [world] [kind] [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from bench import make_gauss, make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kind = sys.argv[4] if len(sys.argv) > 4 else "gauss"
out_path = sys.argv[5] if len(sys.argv) > 5 else None
X = make_mix(n, d, 1) if kind == "mix" else make_gauss(n, d, 1)
splits = np.linspace(0, n, world + 1).astype(np.int64)
TRIP = np.dtype([("row", np.uint32), ("col", np.uint32), ("val", np.float64)])
STAGES = ["query_order", "knn_select", "rerank", "fallback", "radius", "affinity", "symmetrize", "normalize"]
who = int(os.environ.get("GT_WHO", "0"))

ctx = _hip.Context(0)
ctx.set_option("select_symmetric", os.environ.get("GT_SYM", "0"))
ctx.set_points(X)
p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
# single-rank reference time of the same build
for _ in range(2):
    ctx.sync()
    t = time.time()
    ctx.set_points(X)
    nnz1, _ = ctx.graph_build(p)
    ctx.sync()
    single_ms = (time.time() - t) * 1e3
single_stage = {s: round(ctx.stage_ms(s), 3) for s in STAGES if ctx.stage_ms(s) > 0}
recv_parts = []
for r in range(world):
    sc = ctx.graph_begin(p, world, r, splits)
    tot = int(sc.sum())
    buf = ctx.dev_alloc(max(tot, 1) * 16)
    ctx.graph_emit(buf)
    off = int(sc[:who].sum())
    host = np.zeros(int(sc[who]), dtype=TRIP)
    if len(host):
        ctx.dev_download(host, buf + off * 16)
    ctx.dev_free(buf)
    recv_parts.append(host)
recv = np.concatenate(recv_parts)
rb = ctx.dev_alloc(max(len(recv), 1) * 16)
ctx.dev_upload(rb, recv)
res = []
for rep in range(3):
    ctx.sync()
    t0 = time.time()
    sc = ctx.graph_begin(p, world, who, splits)
    tt = ctx.dev_alloc(max(int(sc.sum()), 1) * 16)
    ctx.graph_emit(tt)
    ctx.sync()
    t1 = time.time()
    nnz, flags = ctx.graph_finish(rb, len(recv))
    ctx.sync()
    t2 = time.time()
    ctx.dev_free(tt)
    res.append({"begin+emit_ms": round((t1 - t0) * 1e3, 2), "finish_ms": round((t2 - t1) * 1e3, 2),
                "total_ms": round((t2 - t0) * 1e3, 2), "stage_ms": {s: round(ctx.stage_ms(s), 3) for s in STAGES if ctx.stage_ms(s) > 0}})
best = min(res, key=lambda r: r["total_ms"])
out = {"workload": "%s N=%d d=%d knn=15 decay=40, world %d, rank %d (one GPU plays every rank in turn)" % (kind, n, d, world, who),
       "single_rank_ms": round(single_ms, 2), "single_rank_stage_ms": single_stage, "nnz_single": int(nnz1),
       "per_rank": best, "triplets_sent": int(sc.sum()), "triplets_received": int(len(recv)), "nnz_rows_of_rank": int(nnz),
       "speedup_before_collectives": round(single_ms / best["total_ms"], 2),
       "bytes_all_to_all_per_rank": int(sc.sum()) * 16, "bytes_allgather_points": int(X.nbytes)}
print(json.dumps(out), flush=True)
if out_path:
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
