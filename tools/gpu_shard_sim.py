#!/usr/bin/env python
"""Development probe: per-rank phase times of the row-sharded build, simulated on ONE GPU.

Every rank's gt_graph_begin / gt_graph_emit runs in turn on the same device (the database side is the full
matrix on every rank, exactly as after the all-gather), the all-to-all is done by hand through the host, then
gt_graph_finish runs per rank.  Wall times are per phase and rank; the RCCL collectives are not included.
usage: gpu_shard_sim.py [N] [WORLD]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from graphtools_amd.dist import even_row_splits  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    X = make_mix(n, 64, 1)
    splits = even_row_splits(n, world)
    trip = np.dtype([("row", np.uint32), ("col", np.uint32), ("val", np.float64)])
    ctx = _hip.Context(0)
    xbuf = ctx.dev_alloc(X.nbytes)
    ctx.dev_upload(xbuf, X)
    p, keep = ctx.make_params(15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    recs = []
    sends, counts = [], []
    for rep in range(2):          # first pass warms allocations up
        sends, counts, recs = [], [], []
        for r in range(world):
            t0 = time.perf_counter()
            ctx.set_points_device(xbuf, n, 64, np.float32)
            ctx.sync()
            t1 = time.perf_counter()
            cnt = ctx.graph_begin(p, world, r, splits)
            ctx.sync()
            t2 = time.perf_counter()
            total = int(cnt.sum())
            buf = ctx.dev_alloc(total * 16)
            t2b = time.perf_counter()
            ctx.graph_emit(buf)
            ctx.sync()
            t3 = time.perf_counter()
            st = {s: round(ctx.stage_ms(s), 3) for s in ("prep", "knn_select", "rerank", "radius", "affinity")}
            if r == 0 or rep == 0:
                host = np.zeros(total, dtype=trip)
                ctx.dev_download(host, buf)
                sends.append(host)
            else:
                sends.append(None)
            counts.append(cnt)
            ctx.dev_free(buf)
            recs.append({"rank": r, "set_points_ms": (t1 - t0) * 1e3, "begin_ms": (t2 - t1) * 1e3, "alloc_ms": (t2b - t2) * 1e3,
                         "emit_ms": (t3 - t2b) * 1e3, "stage_ms": st, "sent": total})
            if rep == 1 and r == 0:
                first_sends = sends[0]
        if rep == 0:
            all_sends, all_counts = sends, counts
    # finish for rank 0 with what every rank sent to it (from the first pass; the build is deterministic)
    r = 0
    ctx.set_points_device(xbuf, n, 64, np.float32)
    ctx.graph_begin(p, world, r, splits)
    parts = []
    for s in range(world):
        off = int(all_counts[s][:r].sum())
        parts.append(all_sends[s][off: off + int(all_counts[s][r])])
    recv = np.concatenate(parts)
    rb = ctx.dev_alloc(max(len(recv), 1) * 16)
    ctx.dev_upload(rb, recv)
    fin_ms = []
    for rep in range(3):
        if rep:
            ctx.graph_begin(p, world, r, splits)
        ctx.sync()
        t0 = time.perf_counter()
        nnz, fl = ctx.graph_finish(rb, len(recv))
        ctx.sync()
        t1 = time.perf_counter()
        fin_ms.append((t1 - t0) * 1e3)
    fin = {"finish_ms": fin_ms, "recv": int(len(recv)), "nnz_rank0": int(nnz),
           "stage_ms": {s: round(ctx.stage_ms(s), 3) for s in ("symmetrize", "normalize")}}
    if os.environ.get("GT_SIM_VERIFY"):
        # rank 0's row block must be bit-identical to the same rows of the single-rank build
        d0, i0, p0 = ctx.graph_fetch_csr(_hip.CSR_K)
        pd0, _, _ = ctx.graph_fetch_csr(_hip.CSR_P)
        ref = _hip.Context(0)
        ref.set_points_device(xbuf, n, 64, np.float32)
        ref.graph_build(p)
        d1, i1, p1 = ref.graph_fetch_csr(_hip.CSR_K)
        pd1, _, _ = ref.graph_fetch_csr(_hip.CSR_P)
        nl = int(splits[1])
        e = int(p1[nl])
        fin["identical_to_single_rank"] = bool(np.array_equal(p0, p1[: nl + 1]) and np.array_equal(i0, i1[:e]) and
                                               np.array_equal(d0, d1[:e]) and np.array_equal(pd0, pd1[:e]))
        fin["precision"] = ctx.last_knn_precision()
        ref.close()
    out = {"n": n, "world": world, "ranks": recs, "finish_rank0": fin}
    print(json.dumps({"rank0": recs[0], "finish_rank0": fin}))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gpu_shard_sim.json"), "w") as f:
        json.dump(out, f, indent=1)
