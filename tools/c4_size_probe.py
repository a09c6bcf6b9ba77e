#!/usr/bin/env python
"""C4 (exact dense graph from a resident float32 distance matrix) at several sizes: bytes per second of each stage - does the tile-pair
kernel's rate depend on the size of the matrix (address translation reach) or only on its access pattern?
usage: c4_size_probe.py [n ...]   GT_C4_OPTS=k=v,..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from graphtools_amd import _hip  # noqa: E402

device = torch.device("cuda", 0)
for n in [int(v) for v in sys.argv[1:]] or [50000, 100000, 200000]:
    r = bench.secondary_c4(_hip, torch, device, n=n)
    st = r["stage_ms"]
    nn = float(n) * n
    print("n %7d  total %.1f ms (%.2f TB/s of the bytes as run)  bandwidth %.1f ms (%.2f TB/s)  scan %.1f ms  kernel %.1f ms, list "
          "transposition %.1f ms of it (N^2 floats at %.2f TB/s)  normalise %.1f ms" % (
              n, r["ms_per_graph"], r["roofline"]["achieved"] / 1e3, st["dense_bandwidth"], 4 * nn / st["dense_bandwidth"] / 1e9,
              st["dense_rows_scan"], st["dense_kernel"], st["dense_rows_transpose"], 4 * nn / st["dense_kernel"] / 1e9,
              st["dense_normalize"]), flush=True)
