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
from bench import make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kind = sys.argv[4] if len(sys.argv) > 4 else "mix"
X = make_mix(n, d, 1) if kind == "mix" else np.random.default_rng(1).standard_normal((n, d)).astype(np.float32)
splits = np.linspace(0, n, world + 1).astype(np.int64)
REC = np.dtype([("row", np.uint32), ("pad", np.uint32), ("key", np.uint64)])
TRIP = np.dtype([("row", np.uint32), ("col", np.uint32), ("val", np.float64)])
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
parts, far, seed_ms = [], np.zeros(5), []
for r in range(world):
    ok, n_pad, ss = ctx.graph_sym_plan(p, world, r, splits)
    assert ok, "plan refused"
    rows = int(ss[r + 1] - ss[r])
    buf = ctx.dev_alloc(max(rows, 1) * 8)
    far += ctx.graph_sym_seed(buf)
    host = np.zeros((rows, 2), dtype=np.float32)
    ctx.dev_download(host, buf)
    ctx.dev_free(buf)
    parts.append(host)
    seed_ms.append({s: round(ctx.stage_ms(s), 3) for s in ("query_order", "sym_prepare", "sym_seed")})
thr_all = np.concatenate(parts)
print(json.dumps({"seed_stage_ms_rank0": seed_ms[0], "seed_stage_ms_last": seed_ms[-1], "far": [float(v) for v in far]}), flush=True)
if os.environ.get("GT_COLLECT_ONLY"):
    tb = ctx.dev_alloc(n_pad * 8)
    ctx.dev_upload(tb, thr_all)
    for variant in os.environ.get("GT_VARIANTS", "").split(";"):
        for o in [o for o in variant.split(",") if o]:
            k, v = o.split("=")
            ctx.set_option(k, v)
        for r in [int(x) for x in os.environ.get("GT_RANKS", "0,3").split(",")]:
            ok, n_pad, ss = ctx.graph_sym_plan(p, world, r, splits)
            buf = ctx.dev_alloc(max(int(ss[r + 1] - ss[r]), 1) * 8)
            ctx.graph_sym_seed(buf)
            ctx.dev_free(buf)
            ok, cnt = ctx.graph_sym_collect(tb, far, world)
            print(json.dumps({"variant": variant, "rank": r, "knn_select_ms": round(ctx.stage_ms("knn_select"), 3),
                              "sym_seed_ms": round(ctx.stage_ms("sym_seed"), 3), "records": int(cnt.sum())}), flush=True)
    sys.exit(0)
# phase 2: every rank's collect + emit; keep only what rank `who` will receive
who = int(os.environ.get("GT_WHO", "0"))
recv_parts, col_ms, counts_all = [], [], []
tb = ctx.dev_alloc(n_pad * 8)
ctx.dev_upload(tb, thr_all)
for r in range(world):
    ok, n_pad, ss = ctx.graph_sym_plan(p, world, r, splits)
    buf = ctx.dev_alloc(max(int(ss[r + 1] - ss[r]), 1) * 8)
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
# phase 3: rank `who` runs its whole sequence back to back (the collectives replaced by the buffers prepared above):
# the per-rank compute time of the sharded build up to the triplet exchange
recv = np.concatenate(recv_parts)
rb = ctx.dev_alloc(max(len(recv), 1) * 16)
ctx.dev_upload(rb, recv)
ss0 = ctx.graph_sym_plan(p, world, who, splits)[2]
lb = ctx.dev_alloc(max(int(ss0[who + 1] - ss0[who]), 1) * 8)
sb = ctx.dev_alloc(max(int(max(c.sum() for c in counts_all)), 1) * 16)
for rep in range(int(os.environ.get("GT_REPS", "3"))):
    ctx.sync()
    t = time.time()
    ok, n_pad, ss = ctx.graph_sym_plan(p, world, who, splits)
    ctx.graph_sym_seed(lb)
    t1 = time.time()
    ok, cnt = ctx.graph_sym_collect(tb, far, world)
    ctx.graph_sym_emit(sb)
    t2 = time.time()
    ctx.graph_sym_finish(rb, len(recv))
    t3 = time.time()
    sc = ctx.graph_begin(p, world, who, splits)
    tt = ctx.dev_alloc(max(int(sc.sum()), 1) * 16)
    ctx.graph_emit(tt)
    ctx.sync()
    t4 = time.time()
    if rep == 0 and os.environ.get("GT_VERIFY"):
        trip_sym = np.zeros(int(sc.sum()), dtype=TRIP)
        ctx.dev_download(trip_sym, tt)
        sc_sym = sc.copy()
    ctx.dev_free(tt)
    print(json.dumps({"rank": who, "received": int(len(recv)),
                      "wall_ms": {"plan+seed": round((t1 - t) * 1e3, 2), "collect+emit": round((t2 - t1) * 1e3, 2),
                                  "finish": round((t3 - t2) * 1e3, 2), "begin+emit": round((t4 - t3) * 1e3, 2),
                                  "total": round((t4 - t) * 1e3, 2)},
                      "stage_ms": {s: round(ctx.stage_ms(s), 3) for s in STAGES}, "knn": ctx.knn_stats(), "triplets_out": int(sc.sum())}), flush=True)

if os.environ.get("GT_VERIFY"):
    # full-size check: the transposed triplets rank `who` emits (its rows of the unsymmetrised kernel, every value) must be
    # those of the classic sharded pass
    c2 = _hip.Context(0)
    c2.set_option("select_symmetric", "0")
    c2.set_points(X)
    sc2 = c2.graph_begin(p, world, who, splits)
    assert not c2.knn_stats()["symmetric"]
    t2 = c2.dev_alloc(max(int(sc2.sum()), 1) * 16)
    c2.graph_emit(t2)
    trip_cl = np.zeros(int(sc2.sum()), dtype=TRIP)
    c2.dev_download(trip_cl, t2)
    same_counts = bool(np.array_equal(sc2, sc_sym))
    a = np.sort(trip_sym.view(np.dtype([("k", np.uint64), ("v", np.uint64)])), order=["k", "v"])
    b = np.sort(trip_cl.view(np.dtype([("k", np.uint64), ("v", np.uint64)])), order=["k", "v"])
    print(json.dumps({"verify": {"send_counts_equal": same_counts, "triplets": int(len(a)),
                                 "triplets_identical": bool(len(a) == len(b) and np.array_equal(a, b))}}), flush=True)
