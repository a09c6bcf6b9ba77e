#!/usr/bin/env python
"""Development probe: host<->device copy rates of gt_dev_upload / gt_dev_download for GT_COPY_LANES / GT_COPY_SLOT_MB
settings (run once per setting: the settings are read when the library loads)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402

if __name__ == "__main__":
    nbytes = 1 << 30
    ctx = _hip.Context(0)
    p = ctx.dev_alloc(nbytes)
    src = np.random.default_rng(0).integers(0, 256, size=nbytes, dtype=np.uint8)
    res = {"lanes": os.environ.get("GT_COPY_LANES", "8"), "slot_mb": os.environ.get("GT_COPY_SLOT_MB", "8")}
    ctx.dev_upload(p, src)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.dev_upload(p, src)
        t.append(time.perf_counter() - t0)
    res["h2d_GBps"] = round(nbytes / min(t) / 1e9, 1)
    warm = np.zeros(nbytes, dtype=np.uint8)
    ctx.dev_download(warm, p)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.dev_download(warm, p)
        t.append(time.perf_counter() - t0)
    res["d2h_warm_GBps"] = round(nbytes / min(t) / 1e9, 1)
    t = []
    for _ in range(3):
        fresh = np.empty(nbytes, dtype=np.uint8)
        t0 = time.perf_counter()
        ctx.dev_download(fresh, p)
        t.append(time.perf_counter() - t0)
        del fresh
    res["d2h_fresh_GBps"] = round(nbytes / min(t) / 1e9, 1)
    assert np.array_equal(warm, src)
    print(json.dumps(res), flush=True)
