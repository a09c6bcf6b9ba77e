#!/usr/bin/env python
"""Development probe: stage timings of the graph build at benchmark sizes."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402


def make_mix(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    c = max(n // 2000, 1)
    centres = rng.uniform(-10, 10, (c, d))
    labels = rng.integers(c, size=n)
    return (centres[labels] + rng.standard_normal((n, d))).astype(dtype)


def run(ctx, n, d, seed, reps=2, **kw):
    X = make_mix(n, d, seed)
    t = time.time()
    ctx.set_points(X)
    t_set = time.time() - t
    p, keep = ctx.make_params(kw.get("knn", 15), kw.get("decay", 40), 1e-4, None, 1.0, None, "+", None, 0)
    out = []
    for r in range(reps):
        t = time.time()
        nnz, fl = ctx.graph_build(p)
        wall = time.time() - t
        st = {s: round(ctx.stage_ms(s), 3) for s in ("knn_select", "rerank", "fallback", "radius", "affinity", "symmetrize", "normalize")}
        out.append({"wall_s": round(wall, 4), "nnz": nnz, "flags": fl, "stage_ms": st, "stats": ctx.graph_stats()})
    flops = 2.0 * n * n * d
    best = min(o["stage_ms"]["knn_select"] for o in out)
    rep = {"n": n, "d": d, "set_points_s": round(t_set, 3), "runs": out, "select_TF_algorithmic": round(flops / (best * 1e-3) / 1e12, 2)}
    print(json.dumps(rep))
    return rep


if __name__ == "__main__":
    sizes = [int(s) for s in sys.argv[1:]] or [100000, 1000000]
    reps = []
    for prec in ("f16", "f32"):
        ctx = _hip.Context(0)
        ctx.set_option("knn_precision", prec)
        for n in sizes:
            d = 50 if n == 100000 else 64
            r = run(ctx, n, d, 0 if n == 100000 else 1)
            r["precision"] = prec
            reps.append(r)
        ctx.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gpu_perf.json"), "w") as f:
        json.dump(reps, f, indent=1)
