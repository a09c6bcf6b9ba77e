#!/usr/bin/env python
"""Development probe: timing of the candidate kernel (kNN only) for a block of query rows at N=1e6."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip
from tools.gpu_perf import make_mix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
precs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f16", "f32"]
D = int(os.environ.get("GT_DIM", "64"))
X = make_mix(n, D, 1)
nq = min(n, int(os.environ.get("GT_NQ", "131072")))
import oracle
dbgs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for prec, dbg in [(p_, d_) for p_ in precs for d_ in dbgs]:
    ctx = _hip.Context(0)
    ctx.set_option("knn_precision", prec)
    ctx.set_option("dbg_select", dbg)
    if os.environ.get("GT_SAMP"):   # "stride:keep"
        st_, kp_, en_ = (os.environ["GT_SAMP"].split(":") + ["-1"])[:3]
        ctx.set_option("select_samp_end", en_)
        ctx.set_option("select_samp_stride", st_)
        ctx.set_option("select_samp_keep", kp_)
    if os.environ.get("GT_TRIG"):
        ctx.set_option("select_samp_trig", os.environ["GT_TRIG"])
    if os.environ.get("GT_THR0"):
        ctx.set_option("select_thr0", os.environ["GT_THR0"])
    if os.environ.get("GT_NARROW"):
        ctx.set_option("select_narrow", os.environ["GT_NARROW"])
    if os.environ.get("GT_QORDER"):
        ctx.set_option("query_order", os.environ["GT_QORDER"])
    if os.environ.get("GT_SAMP2"):  # "level:keep"
        lv_, k2_ = os.environ["GT_SAMP2"].split(":")
        ctx.set_option("select_samp2_level", lv_)
        ctx.set_option("select_samp2_keep", k2_)
    ctx.set_points(X)
    best = 1e9
    for rep in range(3):
        d, i, fl = ctx.knn_search(16, rows=(0, nq))
        best = min(best, ctx.stage_ms("knn_select"))
    flops = 2.0 * nq * n * D
    d0, i0 = oracle.kneighbors(X, X[:256], 16)
    print(json.dumps({"prec": prec, "dbg": dbg, "nq": nq, "select_ms": round(best, 2), "TF_alg": round(flops / best / 1e9, 1),
                      "rerank_ms": round(ctx.stage_ms("rerank"), 2), "fallback_ms": round(ctx.stage_ms("fallback"), 2),
                      "flags": fl, "idx_ok_256": bool(np.array_equal(i[:256], i0)), "main": ctx.last_knn_precision()}))
    if dbg & 64:
        import ctypes
        nw = (nq // 256) * 4
        buf = np.zeros((nw, 8), dtype=np.uint64)
        ctx.lib.gt_dbg_fetch_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        rc = ctx.lib.gt_dbg_fetch_prof(ctx.h, nw, buf.ctypes.data)
        tot = best * 1e-3
        m = buf.mean(axis=0)
        print(json.dumps({"rc": rc, "mean_cycles_per_wave": {"admission": float(m[0]), "compaction": float(m[1]), "barrier": float(m[2])},
                          "n_compactions_per_wave": float(m[3]), "n_admission_entries_per_wave": float(m[4]),
                          "max_barrier": float(buf[:, 2].max()), "kernel_ms": best,
                          "level0_cycles_per_wave": float(m[5]), "level0_admission_entries": float(m[6]),
                          "total_cycles_per_wave": float(m[7])}))
    ctx.close()
