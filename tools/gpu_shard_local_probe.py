#!/usr/bin/env python
"""Per-rank critical path of the row-sharded build on renumbered points (gt_points_cell_sort / gt_graph_shard_local),
simulated on ONE GPU: a context plays every rank in turn (set_points on the gathered points, renumbering, local candidate
lists, affinities, triplet emit); the all-to-all is done by hand through the host; then rank `who` runs its whole sequence
back to back, timed - everything a rank computes between the points all-gather and its finished rows of K and P.  What is
NOT in the number: the two collectives themselves (sizes are reported).

usage: gpu_shard_local_probe.py [n] [d] [world] [kind] [out.json]      (GT_WHO = rank to time, GT_OPTS = k=v,k=v)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from bench import make_gauss, make_manifold, make_mix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kind = sys.argv[4] if len(sys.argv) > 4 else "mix"
out_path = sys.argv[5] if len(sys.argv) > 5 else None
who = int(os.environ.get("GT_WHO", "0"))
# c5: BASELINE config 5 - mix d = 50 seed 3 + the landmark operator (GT_LANDMARKS, default 2000; random landmarking, seed 42)
X = make_mix(n, d, 3) if kind == "c5" else {"mix": make_mix, "gauss": make_gauss, "manifold": make_manifold}[kind](n, d, 1)
n_landmark = int(os.environ.get("GT_LANDMARKS", "2000"))
TRIP = np.dtype([("row", np.uint32), ("col", np.uint32), ("val", np.float64)])
STAGES = ["prep", "query_order", "renumber", "sym_prepare", "sym_seed", "sym_bound", "sym_cold", "knn_select", "rerank", "fallback",
          "radius", "affinity", "symmetrize", "symm_merge", "symm_compact", "normalize"]

ctx = _hip.Context(0)
for o in [o for o in os.environ.get("GT_OPTS", "").split(",") if o]:
    k, v = o.split("=")
    ctx.set_option(k, v)
p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)

# the single-rank build of the same graph on this GPU (device-complete, points resident): the 1-GPU time of the scaling ratio
xb = ctx.dev_alloc(X.nbytes)
ctx.dev_upload(xb, X)
single = []
for _ in range(3):
    ctx.sync()
    t = time.perf_counter()
    ctx.set_points_device(xb, n, d, np.float32)
    nnz1, _ = ctx.graph_build(p)
    ctx.sync()
    single.append((time.perf_counter() - t) * 1e3)
single_ms = min(single)
single_stage = {s: round(ctx.stage_ms(s), 3) for s in STAGES if ctx.stage_ms(s) > 0}


in_splits = np.linspace(0, n, world + 1).astype(np.int64)   # the slices of the caller's rows the ranks contribute
split_assign = os.environ.get("GT_SPLIT_ASSIGN", "1") != "0"
cells_all_dev = None
if split_assign:
    # every rank's share of the cell assignment (what the cells all-gather delivers)
    cells_all_dev = ctx.dev_alloc(n * 4)
    for r in range(world):
        ok = ctx.points_cells_begin(xb, n, d, np.float32, in_splits[r], in_splits[r + 1], cells_all_dev + int(in_splits[r]) * 4)
        assert ok, "no cell order for these points"


want_pairs = os.environ.get("GT_PAIRS", "1") != "0"   # the pair-resolved tail on the ranks (gt_graph_bandwidth_local), where it applies
bw_all_dev = None                                      # the bandwidths of all rows: what the all-gather between the halves delivers


# Scratch the HARNESS hands the library (the cells of a rank's share, its bandwidths, its send buffer): allocated once, outside
# the timed sequence - dist.py takes them from torch's caching allocator, a hipMalloc / hipFree per call (50-100 us each, and a
# device synchronisation) is not part of a rank's critical path.  (Until round 6 the probe allocated inside: ~0.25 ms per rank.)
_scratch = {}


def scratch(name, nbytes):
    cur = _scratch.get(name)
    if cur is None or cur[1] < nbytes:
        if cur is not None:
            ctx.dev_free(cur[0])
        cur = (ctx.dev_alloc(int(nbytes * 1.25) + 256), int(nbytes * 1.25) + 256)
        _scratch[name] = cur
    return cur[0]


def rank_until_emit(r, timed=False, only_bandwidths=False):
    """set_points .. emit for rank r; -> (send_counts, device buffer of the triplets, wall ms by phase, stage ms, used)
    only_bandwidths: stop after the first half of graph_begin -> the rank's bandwidths (host), None where the tail does not apply"""
    ctx.sync()
    t0 = time.perf_counter()
    if split_assign:
        own = scratch("cells", int(in_splits[r + 1] - in_splits[r]) * 4)
        applied = ctx.points_cells_begin(xb, n, d, np.float32, in_splits[r], in_splits[r + 1], own)
        st_a = {s: ctx.stage_ms(s) for s in ("prep", "query_order")}
        ctx.points_cells_finish(cells_all_dev)
    else:
        ctx.set_points_device(xb, n, d, np.float32)
        applied = ctx.points_cell_sort()
    ctx.sync()
    t1 = time.perf_counter()
    st0 = {s: ctx.stage_ms(s) for s in ("prep", "query_order", "renumber")}
    if split_assign:   # (the stage timers are reset by the bind inside _begin only: both halves are in them)
        pass
    splits = ctx.points_shard_splits(world)
    used = ctx.graph_shard_local(p, world, r, splits)
    ctx.sync()
    t2 = time.perf_counter()
    if only_bandwidths or bw_all_dev is not None:
        nloc = int(splits[r + 1] - splits[r])
        bwb = scratch("bw", max(nloc, 1) * 8)
        ok = ctx.graph_bandwidth_local(p, world, r, splits, bwb)
        if only_bandwidths:
            bw = None
            if ok:
                bw = np.empty(nloc, dtype=np.float64)
                ctx.sync()
                ctx.dev_download(bw, bwb)
            return bw
        assert ok
        ctx.graph_set_bandwidths(bw_all_dev)   # (the all-gather itself is not in the time: 8 B per row, listed below)
    sc = ctx.graph_begin(p, world, r, splits)
    ctx.sync()
    t3 = time.perf_counter()
    buf = scratch("send", max(int(sc.sum()), 1) * 16)
    ctx.graph_emit(buf)
    ctx.sync()
    t4 = time.perf_counter()
    wall = {"set_points+renumber": (t1 - t0) * 1e3, "local_lists": (t2 - t1) * 1e3, "begin(rerank+affinity)": (t3 - t2) * 1e3,
            "emit": (t4 - t3) * 1e3}
    st = {s: max(ctx.stage_ms(s), 0.0) for s in STAGES}
    for s, v in st0.items():
        st[s] = max(v, 0.0)
    return sc, buf, wall, st, used, applied, splits


if want_pairs and hasattr(ctx, "graph_bandwidth_local"):
    parts = [rank_until_emit(r, only_bandwidths=True) for r in range(world)]
    if all(q is not None for q in parts):
        bw_host = np.concatenate(parts)
        bw_all_dev = ctx.dev_alloc(n * 8)
        ctx.dev_upload(bw_all_dev, bw_host)
    else:
        assert not any(q is not None for q in parts), "the ranks disagree on the pair-resolved tail"

# every rank once: what rank `who` will receive
recv_parts, send_totals, used_all = [], [], []
for r in range(world):
    sc, buf, wall, st, used, applied, splits = rank_until_emit(r)
    off = int(sc[:who].sum())
    host = np.zeros(int(sc[who]), dtype=TRIP)
    if len(host):
        ctx.dev_download(host, buf + off * 16)
    recv_parts.append(host)
    send_totals.append(int(sc.sum()))
    used_all.append(bool(used))
recv = np.concatenate(recv_parts)
rb = ctx.dev_alloc(max(len(recv), 1) * 16)
if len(recv):
    ctx.dev_upload(rb, recv)

labels_all = None
if kind == "c5":
    # what the labels all-gather delivers: the nearest landmark of EVERY row, by the caller's row numbers (not timed here: every
    # rank labels its own rows - timed below - and receives the others')
    landmarks = np.random.default_rng(42).choice(n, n_landmark, replace=False)
    lm = _hip.Context(0)
    lm.set_points(X[landmarks])
    labels_all = np.asarray(lm.knn_first_nearest(int(min(4, n_landmark)), Y=X), dtype=np.int32)
    lm.close()

runs = []
for rep in range(int(os.environ.get("GT_REPS", "4"))):
    sc, buf, wall, st, used, applied, splits = rank_until_emit(who)
    t0 = time.perf_counter()
    nnz, flags = ctx.graph_finish(rb, len(recv))
    ctx.sync()
    wall["finish(merge+P)"] = (time.perf_counter() - t0) * 1e3
    for s in ("symmetrize", "symm_merge", "symm_compact", "normalize"):
        st[s] = max(ctx.stage_ms(s), 0.0)
    if kind == "c5":
        # the rank's share of the landmark stage: labels of its own rows (1-NN against the L landmark rows on the MFMA path,
        # argmin's first-index rule on the device), partial L x L products of its rows (left on the device for the all-reduce),
        # the scaling of the summed operator
        r0_, r1_ = int(splits[who]), int(splits[who + 1])
        ctx.sync()
        t0 = time.perf_counter()
        lm = _hip.Context(0)
        lm.set_points(X[landmarks])
        own_labels = lm.knn_first_nearest(int(min(4, n_landmark)), y_dev_ptr=ctx.points_device(r0_), m=r1_ - r0_)
        lm.close()
        t1 = time.perf_counter()
        lbuf = ctx.dev_alloc((n_landmark * n_landmark + n_landmark) * 8)
        tnnz = ctx.landmark_build_device(labels_all, n_landmark, lbuf)
        ctx.sync()
        t2 = time.perf_counter()
        ctx.landmark_scale_device(lbuf, n_landmark)
        ctx.sync()
        t3 = time.perf_counter()
        ctx.dev_free(lbuf)
        assert np.array_equal(np.asarray(own_labels, dtype=np.int32), labels_all[ctx.points_row_ids(r0_, r1_)])
        wall["landmark_labels_own_rows"] = (t1 - t0) * 1e3
        wall["landmark_build_partial"] = (t2 - t1) * 1e3
        wall["landmark_scale"] = (t3 - t2) * 1e3
    wall["total"] = sum(wall.values())
    runs.append({"wall_ms": {k: round(v, 3) for k, v in wall.items()}, "stage_ms": {k: round(v, 3) for k, v in st.items() if v > 0},
                 "nnz_rows_of_rank": int(nnz)})
best = min(runs[1:] or runs, key=lambda r: r["wall_ms"]["total"])
rows_who = int(splits[who + 1] - splits[who])
out = {
    "what": "row-sharded build on renumbered points, per-rank critical path simulated on one MI355X (one context plays every "
            "rank in turn; the collectives are NOT in the time, their sizes are below)",
    "workload": "%s N=%d d=%d float32 seed=1, knn=15 decay=40 thresh=1e-4, world %d, rank %d (%d rows)" % (kind, n, d, world, who, rows_who),
    "renumbering_applied": bool(applied), "local_lists_used_by_rank": used_all,
    "pair_resolved_tail": bw_all_dev is not None,
    "single_rank_ms": round(single_ms, 3), "single_rank_stage_ms": single_stage, "nnz_single": int(nnz1),
    "per_rank": best, "per_rank_all_runs_total_ms": [r["wall_ms"]["total"] for r in runs],
    "speedup_before_collectives": round(single_ms / best["wall_ms"]["total"], 2),
    "collectives": {
        "all_gather_points_bytes_total": int(X.nbytes),
        "all_gather_cells_bytes_total": int(4 * n) if split_assign else 0,
        "all_to_all_counts_bytes": 8 * world,
        "all_to_all_triplets_bytes_sent_by_rank": int(send_totals[who]) * 16,
        "all_to_all_triplets_bytes_received_by_rank": int(len(recv)) * 16,
        "candidate_record_exchange_bytes": 0,
        "threshold_all_gather_bytes": 0,
        "all_gather_bandwidths_bytes_total": int(8 * n) if bw_all_dev is not None else 0,
        "number_of_collectives": (4 if split_assign else 3) + (2 if kind == "c5" else 0) + (1 if bw_all_dev is not None else 0),
        "all_gather_labels_bytes_total": int(4 * n) if kind == "c5" else 0,
        "all_reduce_landmark_bytes": int(8 * (n_landmark * n_landmark + n_landmark)) if kind == "c5" else 0,
    },
    "triplets_sent_by_every_rank": send_totals,
}
print(json.dumps(out), flush=True)
if out_path:
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
