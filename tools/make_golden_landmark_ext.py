#!/usr/bin/env python
"""Golden vectors for the landmark graph's out-of-sample methods (reference graphs.py:1247-1317):
LandmarkGraph.extend_to_data(Y) and LandmarkGraph.interpolate(transform[, Y=]) of the REAL reference (imported from
/root/reference, build container only) on the G7 configuration.  Writes tests/golden/g7b_landmark_extend.npz."""
import os
import sys
import warnings

import numpy as np
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from ref_import import import_reference  # noqa: E402
from make_golden import make_mix  # noqa: E402


def main():
    gt = import_reference()
    X = make_mix(4096, 50, 0)
    rng = np.random.default_rng(123)
    Y = (X[rng.choice(4096, 96, replace=False)] + 0.3 * rng.standard_normal((96, 50))).astype(np.float32)
    transform = rng.standard_normal((50, 3))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = gt.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=50, random_landmarking=True, random_state=42, verbose=0)
        pnm = G.extend_to_data(Y)
        out = {"X": X, "Y": Y, "transform": transform, "n_landmark": np.int64(50), "random_state": np.int64(42),
               "clusters": np.asarray(G.clusters).astype(np.int32),
               "extend_pnm": np.asarray(sparse.csr_matrix(pnm).toarray()),
               "interp_self": np.asarray(G.interpolate(transform)),
               "interp_Y": np.asarray(G.interpolate(transform, Y=Y))}
    path = os.path.join(ROOT, "tests", "golden", "g7b_landmark_extend.npz")
    np.savez_compressed(path, **out)
    print("g7b_landmark_extend %.2f MB" % (os.path.getsize(path) / 1e6), out["extend_pnm"].shape, out["interp_self"].shape)


if __name__ == "__main__":
    main()
