#!/usr/bin/env python
"""Development probe: effect of the threshold-seeding phase (select_samp_stride / select_samp_keep) on the
candidate pass at benchmark size.  usage: gpu_samp_ablate.py N "stride:keep[:end[:level2:keep2]],..." [dbg]
GT_SORTED=1: rows ordered by mixture component (every neighbourhood contiguous in memory - the adversarial order for
prefix-based thresholds)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    combos = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[2] if len(sys.argv) > 2 else "0:0,16:16").split(",")]
    dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    d = int(os.environ.get("GT_DIM", "64"))
    X = make_mix(n, d, 1)
    if os.environ.get("GT_SORTED", "").startswith("voronoi:"):
        # rows ordered by the nearest of L strided sample rows (what a device-side pre-ordering could do)
        L = int(os.environ["GT_SORTED"].split(":")[1])
        lm = X[:: max(n // L, 1)][:L].astype(np.float32)
        ln = (lm * lm).sum(1) * 0.5
        cell = np.empty(n, dtype=np.int32)
        for a in range(0, n, 65536):
            cell[a:a + 65536] = np.argmax(X[a:a + 65536] @ lm.T - ln[None, :], axis=1)
        X = np.ascontiguousarray(X[np.argsort(cell, kind="stable")])
    elif os.environ.get("GT_SORTED"):
        rng = np.random.default_rng(1)
        c = max(n // 2000, 1)
        rng.uniform(-10, 10, (c, d))
        labels = rng.integers(c, size=n)   # the label stream of make_mix(n, d, 1)
        X = np.ascontiguousarray(X[np.argsort(labels, kind="stable")])
    out = []
    for combo in combos:
        stride, keep = combo[0], combo[1]
        end = combo[2] if len(combo) > 2 else -1
        level2 = combo[3] if len(combo) > 3 else 0
        keep2 = combo[4] if len(combo) > 4 else 40
        ctx = _hip.Context(0)
        ctx.set_option("select_samp_stride", str(stride))
        ctx.set_option("select_samp_keep", str(keep))
        ctx.set_option("select_samp_end", str(end))
        ctx.set_option("select_samp2_level", str(level2))
        ctx.set_option("select_samp2_keep", str(keep2))
        if os.environ.get("GT_CELL_ROWS"):
            ctx.set_option("query_order_cell_rows", os.environ["GT_CELL_ROWS"])
        if os.environ.get("GT_QORDER"):
            ctx.set_option("query_order", os.environ["GT_QORDER"])
        if os.environ.get("GT_PREC"):
            ctx.set_option("knn_precision", os.environ["GT_PREC"])
        if dbg:
            ctx.set_option("dbg_select", str(dbg))
        ctx.set_points(X)
        p, hold = ctx.make_params(15, 40, 1e-4, None, 1.0, None, "+", None, 0)
        best = None
        for r in range(2):
            nnz, fl = ctx.graph_build(p)
            st = {s: round(ctx.stage_ms(s), 3) for s in ("query_order", "knn_select", "rerank", "fallback", "radius", "affinity", "symmetrize", "normalize")}
            if best is None or st["knn_select"] < best["knn_select"]:
                best = st
        rec = {"n": n, "stride": stride, "keep": keep, "end": end, "level2": level2, "keep2": keep2, "nnz": nnz, "flags": fl, "stage_ms": best, "stats": ctx.graph_stats(), "main": ctx.last_knn_precision()}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        ctx.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gpu_samp_ablate.json"), "w") as f:
        json.dump(out, f, indent=1)
