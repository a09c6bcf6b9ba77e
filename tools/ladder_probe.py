#!/usr/bin/env python
"""What the seeding sample sees on the data families of tests/test_gpu_ladder.py (stderr of the library with GT_DBG_SELECT=2048)."""
import os
import sys

os.environ["GT_DBG_SELECT"] = "2048"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ladder as L  # noqa: E402
from graphtools_amd import _hip  # noqa: E402

for name, make in L.FAMILIES.items():
    X = make()
    print("=== %s" % name, flush=True)
    sys.stderr.write("=== %s\n" % name)
    sys.stderr.flush()
    c = _hip.Context(0)
    c.set_points(X)
    p, keep = c.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
    c.graph_build(p)
    print("   symmetric:", bool(c.knn_stats()["symmetric"]), {k: v for k, v in c.knn_stats().items() if "far" in k}, flush=True)
    c.close()
