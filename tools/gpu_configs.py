#!/usr/bin/env python
"""Development probe: graph build across the BASELINE configurations and a few off-path shapes (stage times,
arithmetic chosen by 'auto', repair statistics)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402


def manifold(n, d, seed):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, 5))
    a = rng.standard_normal((5, d))
    return (z @ a + 0.01 * rng.standard_normal((n, d))).astype(np.float32)


def run(name, X, knn=15, decay=40.0, metric="euclidean", reps=2):
    ctx = _hip.Context(0)
    ctx.set_option("metric", metric)
    ctx.set_points(X)
    p, keep = ctx.make_params(knn, decay, 1e-4, None, 1.0, None, "+", None, 0)
    best = None
    for r in range(reps):
        t0 = time.perf_counter()
        nnz, fl = ctx.graph_build(p)
        ctx.sync()
        wall = (time.perf_counter() - t0) * 1e3
        st = {s: round(ctx.stage_ms(s), 2) for s in ("knn_select", "rerank", "fallback", "radius", "affinity", "symmetrize", "normalize")}
        if best is None or wall < best["wall_ms"]:
            best = {"cfg": name, "n": X.shape[0], "d": X.shape[1], "knn": knn, "wall_ms": round(wall, 2), "main": ctx.last_knn_precision(),
                    "nnz_per_row": round(nnz / X.shape[0], 1), "stats": ctx.graph_stats(), "stage_ms": st}
    print(json.dumps(best), flush=True)
    ctx.close()
    return best


if __name__ == "__main__":
    out = []
    if os.environ.get("GT_WIDE") == "only":
        sys.argv.append("wide-only")
    out.append(run("C2 mix 100k d50", make_mix(100000, 50, 0)))
    out.append(run("mix 300k d64", make_mix(300000, 64, 1)))
    out.append(run("mix 200k d100", make_mix(200000, 100, 2)))
    out.append(run("mix 200k d128", make_mix(200000, 128, 2)))
    out.append(run("mix 200k d32", make_mix(200000, 32, 2)))
    out.append(run("manifold 300k d64", manifold(300000, 64, 3)))
    out.append(run("gauss 200k d64", np.random.default_rng(4).standard_normal((200000, 64)).astype(np.float32)))
    out.append(run("mix 100k d64 knn100 (wide lists)", make_mix(100000, 64, 5), knn=100))
    out.append(run("mix 200k d50 cosine", make_mix(200000, 50, 6), metric="cosine"))
    out.append(run("mix 100k d64 float64", make_mix(100000, 64, 7).astype(np.float64)))
    out.append(run("mix 100k d64 binary", make_mix(100000, 64, 8), decay=None))
    if os.environ.get("GT_WIDE"):
        rng = np.random.default_rng(9)
        scales = 0.97 ** np.arange(300)
        Xw = (rng.standard_normal((100000, 300)) * scales + (rng.standard_normal((20, 300)) * scales * 3)[rng.integers(20, size=100000)])
        out.append(run("wide pca-like 100k d300", np.ascontiguousarray(Xw[:, rng.permutation(300)].astype(np.float32))))
        out.append(run("wide mix 100k d300", make_mix(100000, 300, 10)))
        out.append(run("wide gauss 30k d300 (isotropic)", rng.standard_normal((30000, 300)).astype(np.float32)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gpu_configs.json"), "w"), indent=1)
