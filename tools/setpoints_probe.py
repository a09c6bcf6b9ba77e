#!/usr/bin/env python
"""Development probe: wall time of gt_set_points on device-resident points (what every bench step pays before the build)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_mix  # noqa: E402
from graphtools_amd import _hip  # noqa: E402

n, d = 1000000, 64
X = make_mix(n, d, 1)
dev = torch.device("cuda", 0)
xd = torch.from_numpy(X).to(dev)
ctx = _hip.Context(0)
for _ in range(3):
    ctx.set_points_device(xd.data_ptr(), n, d, np.float32)
ctx.sync()
os.environ["GT_TRACE"] = os.environ.get("GT_TRACE", "")
t = time.perf_counter()
R = 20
for _ in range(R):
    ctx.set_points_device(xd.data_ptr(), n, d, np.float32)
ctx.sync()
print("set_points_device: %.3f ms" % ((time.perf_counter() - t) / R * 1e3), {s: round(ctx.stage_ms(s), 3) for s in ("prep",) if ctx.stage_ms(s) > 0})
