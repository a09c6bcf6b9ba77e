"""Oracle: brute-force k-nearest-neighbour search exactly as the reference gets it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference does not implement kNN itself: ``kNNGraph.knn_tree``
(graphtools/graphs.py:748-769) fits ``sklearn.neighbors.NearestNeighbors(
algorithm="auto")`` and ``build_kernel_to_data`` calls ``kneighbors`` /
``radius_neighbors`` on it (graphtools/graphs.py:883, 922-924, 957-959, 966-973).
For n_features > 15 scikit-learn (pinned here: 1.7.2, unpinned ">=0.20" in the
reference's setup.py:6-14) picks the brute-force back end
(sklearn:neighbors/_base.py:622-633) which, for the euclidean metric, is the
Cython ``EuclideanArgKmin{32,64}`` / ``EuclideanRadiusNeighbors{32,64}`` pairwise
reduction.  Its published arithmetic, restated below in numpy:

* squared distances in the expanded "GEMM" form, entirely in float64, float32
  inputs being upcast chunk-wise:  d2[i, j] = ||x_i||^2 + (-2 x_i . y_j) + ||y_j||^2
  (sklearn:metrics/_pairwise_distances_reduction/_argkmin.pyx.tp:471-510,
  _middle_term_computer.pyx.tp:309-440, _base.pyx.tp:45-83 for the float64
  row norms), clamped at 0;
* the k smallest d2 per query kept (max-heap) and sorted ascending
  (_argkmin.pyx.tp:159-168, 186-196);
* the returned distance is ``_rdist_to_dist`` applied in the INPUT dtype
  (_argkmin.pyx.tp:283-295, sklearn:metrics/_dist_metrics.pyx.tp:1018-1019):
  float32 data -> float64(sqrtf(float32(d2))), float64 data -> sqrt(d2);
* radius search keeps d2 <= r_radius with r_radius = ``_dist_to_rdist(radius)``
  evaluated in the input dtype, i.e. float32(radius)*float32(radius) for
  float32 data (_radius_neighbors.pyx.tp:136-137, 505-512); results unsorted.

Ties (exactly equal d2) are resolved by the heap's insertion history in
scikit-learn, which is unspecified; this restatement orders ties by ascending
index.  The float64 value of d2 itself depends on the BLAS summation order at the
1e-16 relative level, so distances can differ from scikit-learn's by one float32
ulp on a ~1e-7 fraction of entries (float32 data) or by a few float64 ulps
(float64 data); the self-distance d2[i, i] is rounding noise in both.
"""
import numpy as np

__all__ = ["row_norms_sq", "kneighbors", "radius_neighbors", "rdist_to_dist", "cosine_kneighbors",
           "cosine_radius_neighbors"]


def row_norms_sq(X):
    """float64 squared row norms (sklearn:_base.pyx.tp:45-83, upcast then dot)."""
    X64 = np.asarray(X, dtype=np.float64)
    return np.einsum("ij,ij->i", X64, X64)


def rdist_to_dist(d2, dtype):
    """``_rdist_to_dist`` in the input dtype (sklearn:_dist_metrics.pyx.tp:1018-1019)."""
    d2 = np.maximum(d2, 0.0)
    if np.dtype(dtype) == np.float32:
        return np.sqrt(d2.astype(np.float32)).astype(np.float64)
    return np.sqrt(d2)


def _sq_dists_block(Q64, qn, Y64, yn):
    # (xn + middle) + yn, middle = -2 * (X @ Y.T): same association as
    # _argkmin.pyx.tp:497-501.
    mid = Q64 @ Y64.T
    mid *= -2.0
    mid += qn[:, None]
    mid += yn[None, :]
    np.maximum(mid, 0.0, out=mid)
    return mid


def _normalize_rows(X):
    """sklearn.preprocessing.normalize(X, "l2") as used by cosine_distances
    (sklearn:metrics/pairwise.py:1169-1172, preprocessing/_data.py:1985-2014): rows / sqrt(einsum(x, x)),
    zero rows untouched."""
    X = np.array(X, dtype=np.float64 if X.dtype != np.float32 else np.float32, copy=True)
    norms = np.sqrt(np.einsum("ij,ij->i", X, X))
    norms[norms == 0.0] = 1.0
    X /= norms[:, None]
    return X


def cosine_kneighbors(data, queries=None, n_neighbors=5, q_chunk=1024):
    """kNN under sklearn's cosine distance: D = clip(1 - xhat . yhat, 0, 2) in the input dtype
    (NearestNeighbors(metric="cosine") -> brute -> pairwise_distances_chunked -> cosine_distances,
    sklearn:metrics/pairwise.py:1130-1184; reduce sklearn:neighbors/_base.py:703-741).  Ties by index."""
    data = np.ascontiguousarray(data)
    qn = _normalize_rows(data if queries is None else np.ascontiguousarray(queries))
    yn = _normalize_rows(data)
    k = int(n_neighbors)
    m = qn.shape[0]
    out_d = np.empty((m, k), dtype=np.float64)
    out_i = np.empty((m, k), dtype=np.int64)
    for q0 in range(0, m, q_chunk):
        q1 = min(m, q0 + q_chunk)
        S = qn[q0:q1] @ yn.T
        S *= -1
        S += 1
        np.clip(S, 0, 2, out=S)
        order = np.lexsort((np.broadcast_to(np.arange(S.shape[1]), S.shape), S), axis=1)[:, :k]
        out_i[q0:q1] = order
        out_d[q0:q1] = np.take_along_axis(S, order, axis=1).astype(np.float64)
    return out_d, out_i


def cosine_radius_neighbors(data, queries, radius, q_chunk=512):
    data = np.ascontiguousarray(data)
    qn = _normalize_rows(np.ascontiguousarray(queries))
    yn = _normalize_rows(data)
    m = qn.shape[0]
    dist_out = np.empty(m, dtype=object)
    ind_out = np.empty(m, dtype=object)
    for q0 in range(0, m, q_chunk):
        q1 = min(m, q0 + q_chunk)
        S = qn[q0:q1] @ yn.T
        S *= -1
        S += 1
        np.clip(S, 0, 2, out=S)
        for r in range(q1 - q0):
            keep = np.nonzero(S[r] <= radius)[0]
            ind_out[q0 + r] = keep.astype(np.int64)
            dist_out[q0 + r] = S[r, keep].astype(np.float64)
    return dist_out, ind_out


def kneighbors(data, queries=None, n_neighbors=5, q_chunk=512, y_chunk=65536):
    """k nearest rows of ``data`` for every row of ``queries`` (default: data itself).

    Returns (distances float64 [m, k], indices int64 [m, k]) like
    ``NearestNeighbors.kneighbors`` (graphtools/graphs.py:883).
    """
    data = np.ascontiguousarray(data)
    in_dtype = data.dtype if data.dtype in (np.float32, np.float64) else np.float64
    if queries is None:
        queries = data
    queries = np.ascontiguousarray(queries)
    n = data.shape[0]
    m = queries.shape[0]
    k = int(n_neighbors)
    if k > n:
        raise ValueError("n_neighbors > n_samples")
    yn_all = row_norms_sq(data)
    out_d2 = np.empty((m, k), dtype=np.float64)
    out_ix = np.empty((m, k), dtype=np.int64)
    for q0 in range(0, m, q_chunk):
        q1 = min(m, q0 + q_chunk)
        Q64 = queries[q0:q1].astype(np.float64)
        qn = np.einsum("ij,ij->i", Q64, Q64)
        best_d2 = None
        best_ix = None
        for y0 in range(0, n, y_chunk):
            y1 = min(n, y0 + y_chunk)
            d2 = _sq_dists_block(Q64, qn, data[y0:y1].astype(np.float64), yn_all[y0:y1])
            kk = min(k, y1 - y0)
            if kk < y1 - y0:
                part = np.argpartition(d2, kk - 1, axis=1)[:, :kk]
                # ties at the cut: argpartition may pick any of the equal values;
                # pull in every column equal to the kk-th value to stay deterministic
                cand_d2 = np.take_along_axis(d2, part, axis=1)
                cut = cand_d2.max(axis=1)
                n_le = (d2 <= cut[:, None]).sum(axis=1)
                if np.any(n_le > kk):
                    # rare (exact ties at the boundary): rebuild those rows fully sorted
                    for r in np.nonzero(n_le > kk)[0]:
                        order = np.lexsort((np.arange(y1 - y0), d2[r]))[:kk]
                        part[r] = order
                    cand_d2 = np.take_along_axis(d2, part, axis=1)
            else:
                part = np.broadcast_to(np.arange(y1 - y0), (q1 - q0, y1 - y0)).copy()
                cand_d2 = d2
            cand_ix = part.astype(np.int64) + y0
            if best_d2 is None:
                best_d2, best_ix = cand_d2, cand_ix
            else:
                best_d2 = np.concatenate([best_d2, cand_d2], axis=1)
                best_ix = np.concatenate([best_ix, cand_ix], axis=1)
            if best_d2.shape[1] > k:
                # keep the k smallest by (d2, index)
                order = np.lexsort((best_ix, best_d2), axis=1)[:, :k]
                best_d2 = np.take_along_axis(best_d2, order, axis=1)
                best_ix = np.take_along_axis(best_ix, order, axis=1)
        order = np.lexsort((best_ix, best_d2), axis=1)
        out_d2[q0:q1] = np.take_along_axis(best_d2, order, axis=1)
        out_ix[q0:q1] = np.take_along_axis(best_ix, order, axis=1)
    return rdist_to_dist(out_d2, in_dtype), out_ix


def radius_neighbors(data, queries, radius, q_chunk=256):
    """All rows of ``data`` within ``radius`` of each query (unsorted, ragged).

    Follows ``NearestNeighbors.radius_neighbors`` as called at
    graphtools/graphs.py:966-973 (sort_results=False).  Returns two object arrays
    (distances float64, indices int64), one ragged entry per query.
    """
    data = np.ascontiguousarray(data)
    in_dtype = data.dtype if data.dtype in (np.float32, np.float64) else np.float64
    queries = np.ascontiguousarray(queries)
    if np.dtype(in_dtype) == np.float32:
        r32 = np.float32(radius)
        r_radius = np.float64(r32 * r32)  # float32 product (_dist_metrics.pyx.tp:1021-1022)
    else:
        r_radius = np.float64(radius) * np.float64(radius)
    yn = row_norms_sq(data)
    Y64 = data.astype(np.float64)
    m = queries.shape[0]
    dist_out = np.empty(m, dtype=object)
    ind_out = np.empty(m, dtype=object)
    for q0 in range(0, m, q_chunk):
        q1 = min(m, q0 + q_chunk)
        Q64 = queries[q0:q1].astype(np.float64)
        qn = np.einsum("ij,ij->i", Q64, Q64)
        d2 = _sq_dists_block(Q64, qn, Y64, yn)
        for r in range(q1 - q0):
            keep = np.nonzero(d2[r] <= r_radius)[0]
            ind_out[q0 + r] = keep.astype(np.int64)
            dist_out[q0 + r] = rdist_to_dist(d2[r, keep], in_dtype)
    return dist_out, ind_out
