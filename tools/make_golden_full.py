#!/usr/bin/env python
"""Pin the BASELINE sizes to the REAL reference: build C2 (mix N=1e5, d=50, seed 0) and C3 (mix N=1e6, d=64, seed 1)
with the reference imported from /root/reference (tools/ref_import.py) and keep compact fixtures of the results:

  row_len      uint16[N]   entries per row of K (= of P); indptr is its running sum
  row_hash     uint16[N]   per-row checksum of the sorted column indices: upper half of sum_j (col_j + 1) * 2654435761 mod 2^32
  sha256       bytes       sha-256 of K.indices as little-endian int32 (whole-matrix structure hash)
  degree4      float64[ceil(N/4)]  kernel_degree (row sums of K, base.py:648-660) of the rows 0, 4, 8, ...
  degree_blocks float64[ceil(N/1024)]  sums of kernel_degree over blocks of 1024 rows (every row counts)
  sample_{i,j,K,P}         10^5 stored entries drawn uniformly (default_rng(12345)) with their K and P values
  nnz, time_s, versions

Only inputs' recipe (seeded generator) and outputs are written; no reference text travels.  Runs only in the build
container: `python tools/make_golden_full.py c2` (about 15 s) / `c3` (about 20 minutes, 8 cores, ~25 GB of RAM).

Round 5 - the production paths of BASELINE configs 4 and 5 at sizes where they engage:

  c4   TraditionalGraph(D, precomputed="distance", knn=15, decay=40) on a float32 distance matrix of n = 16384 points (the
       row-streaming form of gt_dense.hip is the default from 16384 rows).  The points are `mix` d = 100 seed 4 rounded to
       multiples of 1/4: every squared distance is then an exact multiple of 1/16 below 2^24 - the same float32 matrix D on
       any BLAS, any box (quantised_distance_matrix).  Kept: per-row non-zero counts and column hashes of K, kernel degrees,
       10^5 sampled (i, j, K_ij, P_ij) (float32, as the reference holds them).
  c5   kNNLandmarkGraph: mix N = 1e5, d = 50, seed 3, knn=15 decay=40, n_landmark=2000, random_landmarking=True,
       random_state=42: clusters (all), landmark_op (2000 x 2000, float64), transitions (row lengths, hashes, samples).
"""
import hashlib
import os
import sys
import time
import warnings

import numpy as np
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import import_reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

CONFIGS = {"c2": dict(n=100000, d=50, seed=0), "c3": dict(n=1000000, d=64, seed=1)}


def make_mix(n, d, seed, dtype=np.float32):
    # bench.py's generator (chunked; the same stream of draws as tests/conftest.make_mix)
    rng = np.random.default_rng(seed)
    c = max(n // 2000, 1)
    centres = rng.uniform(-10, 10, (c, d))
    labels = rng.integers(c, size=n)
    out = np.empty((n, d), dtype=dtype)
    step = 100000
    for s in range(0, n, step):
        e = min(n, s + step)
        out[s:e] = centres[labels[s:e]] + rng.standard_normal((e - s, d))
    return out


def row_hash(indices, indptr):
    h = ((indices.astype(np.uint64) + np.uint64(1)) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    cs = np.zeros(len(h) + 1, dtype=np.uint64)
    np.cumsum(h, dtype=np.uint64, out=cs[1:])            # (wraps modulo 2^64: differences stay exact modulo 2^32)
    return ((cs[indptr[1:]] - cs[indptr[:-1]]) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def build(tag):
    cfg = CONFIGS[tag]
    gt = import_reference()
    X = make_mix(cfg["n"], cfg["d"], cfg["seed"])
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = gt.Graph(X, knn=15, decay=40, n_pca=None, verbose=0, random_state=42)
        K = sparse.csr_matrix(G.K)
        t1 = time.perf_counter()
        P = sparse.csr_matrix(G.P)
    t2 = time.perf_counter()
    assert type(G).__name__ == "kNNGraph"
    K.sort_indices()
    P.sort_indices()
    assert np.array_equal(K.indices, P.indices) and np.array_equal(K.indptr, P.indptr)
    n = K.shape[0]
    row_len = np.diff(K.indptr)
    assert row_len.max() < 65536
    degree = np.asarray(K.sum(axis=1)).ravel()   # base.py:659 kernel_degree (matrix row sums)
    rng = np.random.default_rng(12345)
    pos = np.sort(rng.choice(K.nnz, size=100000, replace=False))
    rows = np.searchsorted(K.indptr, pos, side="right") - 1
    import scipy
    import sklearn

    out = {
        "n": np.int64(n), "d": np.int64(cfg["d"]), "seed": np.int64(cfg["seed"]), "knn": np.int64(15), "decay": np.float64(40),
        "thresh": np.float64(1e-4), "nnz": np.int64(K.nnz),
        "row_len": row_len.astype(np.uint16), "row_hash": (row_hash(K.indices, K.indptr) >> np.uint32(16)).astype(np.uint16),
        "sha256_indices": np.frombuffer(hashlib.sha256(K.indices.astype("<i4").tobytes()).digest(), dtype=np.uint8),
        "degree4": degree[::4].astype(np.float64),
        "degree_blocks": np.add.reduceat(degree.astype(np.float64), np.arange(0, n, 1024)),
        "sample_i": rows.astype(np.int32), "sample_j": K.indices[pos].astype(np.int32),
        "sample_K": K.data[pos].astype(np.float64), "sample_P": P.data[pos].astype(np.float64),
        "time_kernel_s": np.float64(t1 - t0), "time_total_s": np.float64(t2 - t0), "cores": np.int64(os.cpu_count()),
        "versions": np.array("graphtools %s numpy %s scipy %s sklearn %s" % (gt.__version__, np.__version__, scipy.__version__,
                                                                             sklearn.__version__)),
    }
    path = os.path.join(OUT, "full_%s_reference.npz" % tag)
    np.savez_compressed(path, **out)
    print("%s: N=%d nnz=%d  reference wall %.1f s (kernel %.1f s)  fixture %.2f MB" % (
        tag, n, K.nnz, t2 - t0, t1 - t0, os.path.getsize(path) / 1e6), flush=True)


def quantised_points(n, d, seed):
    """`mix` rounded to multiples of 1/4 (float64): squared distances are exact multiples of 1/16"""
    return np.round(make_mix(n, d, seed, dtype=np.float64) * 4.0) / 4.0


def quantised_distance_matrix(Xq):
    """float32 euclidean distance matrix of quantised points: d2 = |x|^2 + |y|^2 - 2 x.y is EXACT in float64 in any summation
    order (integers / 16 far below 2^53), sqrt is correctly rounded, so is the conversion to float32: one matrix everywhere"""
    sq = (Xq * Xq).sum(axis=1)
    d2 = sq[:, None] + sq[None, :] - 2.0 * (Xq @ Xq.T)
    assert d2.min() >= 0.0 and np.all(d2 * 16.0 == np.round(d2 * 16.0))
    return np.sqrt(d2).astype(np.float32)


def dense_row_stats(M):
    """non-zero count and 16-bit column checksum of every row of a dense matrix (row_hash's formula)"""
    nz = M != 0
    cnt = nz.sum(axis=1)
    cols = np.flatnonzero(nz.ravel()) % M.shape[1]
    ptr = np.concatenate([[0], np.cumsum(cnt)])
    return cnt, (row_hash(cols, ptr) >> np.uint32(16)).astype(np.uint16)


def build_c4(n=16384, d=100, seed=4):
    gt = import_reference()
    D = quantised_distance_matrix(quantised_points(n, d, seed))
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = gt.Graph(D, precomputed="distance", knn=15, decay=40, n_pca=None, verbose=0)
        K = np.asarray(G.K)
        P = np.asarray(G.P)
    t1 = time.perf_counter()
    assert type(G).__name__ == "TraditionalGraph" and K.shape == (n, n)
    cnt, h16 = dense_row_stats(K)
    degree = np.asarray(G.kernel_degree).ravel()
    rng = np.random.default_rng(12345)
    nzpos = np.flatnonzero(K.ravel())
    pos = np.sort(rng.choice(nzpos, size=100000, replace=False))
    import scipy
    import sklearn

    out = {"n": np.int64(n), "d": np.int64(d), "seed": np.int64(seed), "knn": np.int64(15), "decay": np.float64(40),
           "thresh": np.float64(1e-4), "nnz": np.int64(len(nzpos)), "dtype_K": np.array(str(K.dtype)), "dtype_P": np.array(str(P.dtype)),
           "row_nnz": cnt.astype(np.uint32), "row_hash": h16, "degree": degree.astype(np.float64),
           "P_row_sums": P.astype(np.float64).sum(axis=1),
           "sample_i": (pos // n).astype(np.int32), "sample_j": (pos % n).astype(np.int32),
           "sample_K": K.ravel()[pos].astype(np.float64), "sample_P": P.ravel()[pos].astype(np.float64),
           "time_total_s": np.float64(t1 - t0), "cores": np.int64(os.cpu_count()),
           "versions": np.array("graphtools %s numpy %s scipy %s sklearn %s" % (gt.__version__, np.__version__, scipy.__version__,
                                                                                sklearn.__version__))}
    path = os.path.join(OUT, "full_c4_n16384_reference.npz")
    np.savez_compressed(path, **out)
    print("c4: n=%d nnz=%d (%.1f per row) K %s P %s  reference wall %.1f s  fixture %.2f MB" % (
        n, len(nzpos), len(nzpos) / n, K.dtype, P.dtype, t1 - t0, os.path.getsize(path) / 1e6), flush=True)


def build_c5(n=100000, d=50, seed=3, L=2000):
    gt = import_reference()
    X = make_mix(n, d, seed)
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = gt.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=L, random_landmarking=True, random_state=42, verbose=0)
        op = np.asarray(G.landmark_op)
        T = sparse.csr_matrix(G.transitions)
        clusters = np.asarray(G.clusters)
    t1 = time.perf_counter()
    assert type(G).__name__ == "kNNLandmarkGraph" and op.shape == (L, L)
    T.sort_indices()
    T.eliminate_zeros()
    rng = np.random.default_rng(12345)
    pos = np.sort(rng.choice(T.nnz, size=100000, replace=False))
    rows = np.searchsorted(T.indptr, pos, side="right") - 1
    import scipy
    import sklearn

    out = {"n": np.int64(n), "d": np.int64(d), "seed": np.int64(seed), "knn": np.int64(15), "decay": np.float64(40),
           "n_landmark": np.int64(L), "random_state": np.int64(42),
           "clusters": clusters.astype(np.int32), "landmark_op": op.astype(np.float64),
           "t_nnz": np.int64(T.nnz), "t_row_len": np.diff(T.indptr).astype(np.uint16),
           "t_row_hash": (row_hash(T.indices, T.indptr) >> np.uint32(16)).astype(np.uint16),
           "t_sample_i": rows.astype(np.int32), "t_sample_j": T.indices[pos].astype(np.int32), "t_sample_v": T.data[pos].astype(np.float64),
           "time_total_s": np.float64(t1 - t0), "cores": np.int64(os.cpu_count()),
           "versions": np.array("graphtools %s numpy %s scipy %s sklearn %s" % (gt.__version__, np.__version__, scipy.__version__,
                                                                                sklearn.__version__))}
    path = os.path.join(OUT, "full_c5_n1e5_reference.npz")
    np.savez_compressed(path, **out)
    print("c5: N=%d L=%d transitions nnz=%d  reference wall %.1f s  fixture %.2f MB" % (n, L, T.nnz, t1 - t0,
                                                                                      os.path.getsize(path) / 1e6), flush=True)


if __name__ == "__main__":
    for tag in sys.argv[1:] or ["c2"]:
        if tag == "c4":
            build_c4()
        elif tag == "c5":
            build_c5()
        else:
            build(tag)
