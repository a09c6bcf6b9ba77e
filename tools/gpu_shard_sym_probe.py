#!/usr/bin/env python
"""Development probe: the row-sharded symmetric candidate pass simulated on one GPU (one context per rank, exchanges by
hand through the host), per-rank stage times.  usage: gpu_shard_sym_probe.py [n] [d] [world] [kind]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from graphtools_amd import _hip  # noqa: E402
from tools.gpu_sym_check import make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kind = sys.argv[4] if len(sys.argv) > 4 else "mix"
X = make_mix(n, d, 1) if kind == "mix" else np.random.default_rng(1).standard_normal((n, d)).astype(np.float32)
splits = np.linspace(0, n, world + 1).astype(np.int64)
REC = np.dtype([("row", np.uint32), ("pad", np.uint32), ("key", np.uint64)])
STAGES = ["query_order", "sym_prepare", "sym_seed", "knn_select", "sym_exchange", "rerank", "fallback", "radius", "affinity"]

ctx = _hip.Context(0)       # ONE context plays every rank in turn (the stages of a rank do not overlap another's)
for o in [o for o in os.environ.get("GT_OPTS", "").split(",") if o]:
    k, v = o.split("=")
    ctx.set_option(k, v)
if os.environ.get("GT_DBG"):
    ctx.set_option("dbg_select", os.environ["GT_DBG"])
ctx.set_points(X)
p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
# phase 1: every rank's seeds (the thresholds do not depend on who computes them)
parts, far, seed_ms = [], 0, []
for r in range(world):
    ok, n_pad, ss = ctx.graph_sym_plan(p, world, r, splits)
    assert ok, "plan refused"
    rows = int(ss[r + 1] - ss[r])
    buf = ctx.dev_alloc(max(rows, 1) * 4)
    far += ctx.graph_sym_seed(buf)
    host = np.zeros(rows, dtype=np.float32)
    ctx.dev_download(host, buf)
    ctx.dev_free(buf)
    parts.append(host)
    seed_ms.append({s: round(ctx.stage_ms(s), 3) for s in ("query_order", "sym_prepare", "sym_seed")})
thr_all = np.concatenate(parts)
print(json.dumps({"seed_stage_ms_rank0": seed_ms[0], "seed_stage_ms_last": seed_ms[-1], "far": far}), flush=True)
# phase 2: every rank's collect + emit; keep only what rank `who` will receive
who = int(os.environ.get("GT_WHO", "0"))
recv_parts, col_ms, counts_all = [], [], []
tb = ctx.dev_alloc(n_pad * 4)
ctx.dev_upload(tb, thr_all)
for r in range(world):
    ok, n_pad, ss = ctx.graph_sym_plan(p, world, r, splits)
    buf = ctx.dev_alloc(max(int(ss[r + 1] - ss[r]), 1) * 4)
    ctx.graph_sym_seed(buf)
    ctx.dev_free(buf)
    t = time.time()
    ok, cnt = ctx.graph_sym_collect(tb, far, world)
    assert ok, "collect refused"
    total = int(cnt.sum())
    sb = ctx.dev_alloc(max(total, 1) * 16)
    ctx.graph_sym_emit(sb)
    wall = time.time() - t
    host = np.zeros(total, dtype=REC)
    ctx.dev_download(host, sb)
    ctx.dev_free(sb)
    off = int(cnt[:who].sum())
    recv_parts.append(host[off: off + int(cnt[who])].copy())
    counts_all.append(cnt)
    col_ms.append(dict({s: round(ctx.stage_ms(s), 3) for s in ("sym_prepare", "knn_select", "sym_exchange")}, wall_ms=round(wall * 1e3, 2),
                       records=total))
    del host
print(json.dumps({"collect": col_ms}), flush=True)
# phase 3: rank `who` finishes: plan/seed/collect/emit once more to be in the right state, then the received records
ok, n_pad, ss = ctx.graph_sym_plan(p, world, who, splits)
buf = ctx.dev_alloc(max(int(ss[who + 1] - ss[who]), 1) * 4)
ctx.graph_sym_seed(buf)
ctx.dev_free(buf)
ok, cnt = ctx.graph_sym_collect(tb, far, world)
sb = ctx.dev_alloc(max(int(cnt.sum()), 1) * 16)
ctx.graph_sym_emit(sb)
ctx.dev_free(sb)
recv = np.concatenate(recv_parts)
rb = ctx.dev_alloc(max(len(recv), 1) * 16)
ctx.dev_upload(rb, recv)
t = time.time()
ctx.graph_sym_finish(rb, len(recv))
t_fin = time.time() - t
t = time.time()
sc = ctx.graph_begin(p, world, who, splits)
t_begin = time.time() - t
print(json.dumps({"rank": who, "received": int(len(recv)), "finish_wall_ms": round(t_fin * 1e3, 2), "begin_wall_ms": round(t_begin * 1e3, 2),
                  "stage_ms": {s: round(ctx.stage_ms(s), 3) for s in STAGES}, "knn": ctx.knn_stats(), "triplets_out": int(sc.sum())}), flush=True)
