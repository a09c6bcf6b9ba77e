import os
import sys

import numpy as np
import pytest
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # torch bundles its own copy of the HIP runtime; when it initialises AFTER libgraphtools_amd.so has been working in
    # the process for a while it can fail to see the GPU ("No HIP GPUs are available").  The tests that hand device
    # memory to torch (diff_op_torch) therefore bring torch's context up first, as bench.py and the sharded path do.
    try:
        import torch

        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:   # CPU-only box: nothing to initialise
        pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_csr(z, prefix):
    return sparse.csr_matrix(
        (z[prefix + "_data"], z[prefix + "_indices"], z[prefix + "_indptr"]), shape=tuple(z[prefix + "_shape"])
    )


def golden_params(z):
    """graph keyword arguments stored in a variant fixture as param_* entries"""
    kw = {}
    for key in ("bandwidth", "bandwidth_scale", "kernel_symm", "theta", "anisotropy", "knn_max", "thresh"):
        if "param_" + key in z.files:
            v = z["param_" + key]
            v = v.item() if v.shape == () else v
            if key == "kernel_symm":
                v = None if str(v) == "none" else str(v)
            kw[key] = v
    return kw


def make_mix(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    c = max(n // 2000, 1)
    centres = rng.uniform(-10, 10, (c, d))
    labels = rng.integers(c, size=n)
    return (centres[labels] + rng.standard_normal((n, d))).astype(dtype)


def make_manifold(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, 5))
    a = rng.standard_normal((5, d))
    return (z @ a + 0.01 * rng.standard_normal((n, d))).astype(dtype)


def make_gauss(n, d, seed, dtype=np.float32):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(dtype)


@pytest.fixture(scope="session")
def hip_ctx():
    """A HIP context on device 0 (GPU tests only); fails loudly if the extension is missing."""
    from graphtools_amd import _hip

    ctx = _hip.Context(0)
    yield ctx
    ctx.close()


def mnn_params(z):
    """keyword arguments of the MNN fixtures (tools/make_golden_mnn.py); NaN encodes None"""
    def val(k):
        v = z["param_" + k]
        return None if np.isnan(v) else v.item()
    kw = {k: val(k) for k in ("knn", "decay", "thresh", "beta", "theta", "anisotropy")}
    kw["knn"] = int(kw["knn"])
    kw["kernel_symm"] = str(z["param_kernel_symm"])
    return kw
