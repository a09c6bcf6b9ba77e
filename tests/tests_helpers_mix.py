"""tiny helper importable by subprocess workers (same generator as conftest.make_mix)"""
import numpy as np


def make_mix(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    c = max(n // 2000, 1)
    centres = rng.uniform(-10, 10, (c, d))
    labels = rng.integers(c, size=n)
    return (centres[labels] + rng.standard_normal((n, d))).astype(dtype)
