"""Oracle: kNN affinity kernel, symmetrisation, anisotropy, diffusion operator.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in numpy/scipy, the non-numba float64 branch of

* ``kNNGraph.build_kernel`` / ``build_kernel_to_data``   graphtools/graphs.py:771-785, 819-982
* ``_build_csr_from_neighbors`` (python branch)           graphtools/graphs.py:450-559
* ``BaseGraph.symmetrize_kernel`` / ``apply_anisotropy``  graphtools/base.py:557-592
* ``BaseGraph.P`` (= sklearn ``normalize(K, "l1", axis=1)``) graphtools/base.py:629-646,
  sklearn:preprocessing/_data.py:1997-2014
* ``kernel_degree`` / ``diff_aff``                        graphtools/base.py:648-698
"""
import numbers

import numpy as np
from scipy import sparse

from . import knn as _knn

__all__ = [
    "build_csr_from_neighbors",
    "knn_kernel",
    "symmetrize_kernel",
    "apply_anisotropy",
    "diff_op",
    "kernel_degree",
    "diff_aff",
    "knn_graph",
]


class _NumpyEngine:
    """kneighbors / radius_neighbors through the numpy restatement (oracle/knn.py)."""

    def __init__(self, data, distance="euclidean"):
        self.data = data
        self.distance = distance

    def kneighbors(self, Y, k):
        if self.distance == "cosine":
            return _knn.cosine_kneighbors(self.data, Y, k)
        return _knn.kneighbors(self.data, Y, k)

    def radius_neighbors(self, Y, radius):
        if self.distance == "cosine":
            return _knn.cosine_radius_neighbors(self.data, Y, radius)
        return _knn.radius_neighbors(self.data, Y, radius)


class _SklearnEngine:
    """The reference's own third-party call sites (graphtools/graphs.py:763-768)."""

    def __init__(self, data, knn, distance="euclidean"):
        from sklearn.neighbors import NearestNeighbors

        self.tree = NearestNeighbors(n_neighbors=knn, algorithm="auto", metric=distance).fit(data)

    def kneighbors(self, Y, k):
        return self.tree.kneighbors(Y, n_neighbors=k)

    def radius_neighbors(self, Y, radius):
        return self.tree.radius_neighbors(Y, radius=radius)


def build_csr_from_neighbors(row_neighbors, row_distances, bandwidth, decay, thresh, shape):
    """graphtools/graphs.py:450-559 (python branch :491-551, then sum_duplicates :557).

    w = exp(-(d / bw)^decay), NaN -> 1, keep w >= thresh, columns sorted per row.
    """
    n_rows, _ = shape
    counts = np.empty(n_rows, dtype=np.int64)
    kept_idx = []
    kept_w = []
    scalar_bw = isinstance(bandwidth, numbers.Number)
    for i in range(n_rows):
        d_i = np.asarray(row_distances[i], dtype=np.float64)
        bw = bandwidth if scalar_bw else bandwidth[i]
        w = np.exp(-np.power(d_i / bw, decay))
        w = np.where(np.isnan(w), 1.0, w)
        mask = w >= thresh
        nb = np.asarray(row_neighbors[i])[mask]
        w = w[mask]
        if len(nb) > 1:
            order = np.argsort(nb)
            nb = nb[order]
            w = w[order]
        counts[i] = len(nb)
        kept_idx.append(nb)
        kept_w.append(w)
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    indices = np.concatenate(kept_idx).astype(np.int32) if n_rows else np.empty(0, np.int32)
    data = np.concatenate(kept_w).astype(np.float64) if n_rows else np.empty(0, np.float64)
    K = sparse.csr_matrix((data, indices, indptr), shape=shape)
    K.sum_duplicates()
    return K


def knn_kernel(
    data,
    knn=5,
    decay=40,
    thresh=1e-4,
    bandwidth=None,
    bandwidth_scale=1.0,
    knn_max=None,
    search_multiplier=6,
    Y=None,
    engine="numpy",
    return_search=False,
    distance="euclidean",
):
    """Unsymmetrised kernel of ``kNNGraph`` (graphtools/graphs.py:771-785, 819-982).

    ``Y is None`` is the ``build_kernel`` case: queries are the data themselves and
    the neighbour count is ``knn + 1`` (self included, graphs.py:783-784); otherwise
    it is ``build_kernel_to_data(Y)`` with ``knn`` as given (graphs.py:854-855).
    """
    data = np.ascontiguousarray(data)
    n = data.shape[0]
    if decay is not None and thresh < np.finfo(float).eps:
        thresh = np.finfo(float).eps  # graphs.py:628-629
    if Y is None:
        Y = data
        k_eff = knn + 1
        knn_max_eff = knn_max + 1 if knn_max else None
    else:
        Y = np.ascontiguousarray(Y)
        k_eff = knn
        knn_max_eff = knn_max
    if k_eff > n:
        k_eff = n
    if knn_max_eff is None:
        knn_max_eff = n
    eng = _SklearnEngine(data, knn + 1, distance) if engine == "sklearn" else _NumpyEngine(data, distance)
    m = Y.shape[0]

    if decay is None or thresh == 1:
        # binary connectivity kernel, graphs.py:872-877 -> kneighbors_graph(mode="connectivity"):
        # ones at the k_eff nearest columns, in distance order (sklearn:neighbors/_base.py:1039-1048)
        _, ind = eng.kneighbors(Y, k_eff)
        indptr = np.arange(0, m * k_eff + 1, k_eff)
        K = sparse.csr_matrix(
            (np.ones(m * k_eff, dtype=np.float64), ind.reshape(-1), indptr), shape=(m, n)
        )
        return (K, None) if return_search else K

    search_knn = min(k_eff * search_multiplier, knn_max_eff)
    distances, indices = eng.kneighbors(Y, search_knn)
    first = (distances.copy(), indices.copy())
    if bandwidth is None:
        bw = distances[:, k_eff - 1] * bandwidth_scale  # graphs.py:892
        bw = np.maximum(bw, np.finfo(float).eps)
    else:
        bw = bandwidth * bandwidth_scale
        bw = np.maximum(bw, np.finfo(float).eps)
    radius = bw * np.power(-1 * np.log(thresh), 1 / decay)  # graphs.py:902-904
    update_idx = np.argwhere(np.max(distances, axis=1) < radius).reshape(-1)
    if len(update_idx) > 0:
        distances = [d for d in distances]
        indices = [i for i in indices]
    scalar_bw = isinstance(bw, numbers.Number)
    search_knn = min(search_knn * search_multiplier, knn_max_eff)
    while len(update_idx) > m // 10 and search_knn < n / 2 and search_knn < knn_max_eff:
        dist_new, ind_new = eng.kneighbors(Y[update_idx], search_knn)
        for i, idx in enumerate(update_idx):
            distances[idx] = dist_new[i]
            indices[idx] = ind_new[i]
        update_idx = [
            i for i, d in enumerate(distances) if np.max(d) < (radius if scalar_bw else radius[i])
        ]
        search_knn = min(search_knn * search_multiplier, knn_max_eff)
    if len(update_idx) > 0:
        if search_knn == knn_max_eff:
            dist_new, ind_new = eng.kneighbors(Y[update_idx], search_knn)  # graphs.py:957-962
        else:
            dist_new, ind_new = eng.radius_neighbors(  # graphs.py:966-976
                Y[update_idx, :], radius if scalar_bw else np.max(radius[update_idx])
            )
        for i, idx in enumerate(update_idx):
            distances[idx] = dist_new[i]
            indices[idx] = ind_new[i]
    K = build_csr_from_neighbors(indices, distances, bw, decay, thresh, (m, n))
    if return_search:
        return K, {"distances": first[0], "indices": first[1], "bandwidth": bw, "radius": radius}
    return K


def symmetrize_kernel(K, kernel_symm="+", theta=None):
    """graphtools/base.py:557-577 (+ matrix.py:16-29 for the mnn min/max)."""
    if kernel_symm == "+":
        return (K + K.T) / 2
    if kernel_symm == "*":
        return K.multiply(K.T) if sparse.issparse(K) else np.multiply(K, K.T)
    if kernel_symm == "mnn":
        if sparse.issparse(K):
            lo, hi = K.minimum(K.T), K.maximum(K.T)
        else:
            lo, hi = np.minimum(K, K.T), np.maximum(K, K.T)
        return theta * lo + (1 - theta) * hi
    if kernel_symm is None:
        return K
    raise NotImplementedError(kernel_symm)


def apply_anisotropy(K, anisotropy=0):
    """graphtools/base.py:579-592."""
    if anisotropy == 0:
        return K
    if sparse.issparse(K):
        d = np.array(K.sum(1)).flatten()
        K = K.tocoo()
        K.data = K.data / ((d[K.row] * d[K.col]) ** anisotropy)
        return K.tocsr()
    d = K.sum(1)
    return K / (np.outer(d, d) ** anisotropy)


def diff_op(K):
    """``normalize(K, "l1", axis=1)`` (graphtools/base.py:645): rows / sum|row|, zero rows untouched.

    sklearn:utils/sparsefuncs_fast.pyx ``inplace_csr_row_normalize_l1`` accumulates
    |x| sequentially in storage order and divides each entry by the sum.
    """
    if sparse.issparse(K):
        P = sparse.csr_matrix(K, dtype=np.float64, copy=True)
        P.sort_indices()
        for i in range(P.shape[0]):
            s0, s1 = P.indptr[i], P.indptr[i + 1]
            tot = 0.0
            for v in P.data[s0:s1]:
                tot += abs(v)
            if tot != 0.0:
                P.data[s0:s1] /= tot
        return P
    K = np.asarray(K)
    norms = np.abs(K).sum(axis=1)
    norms[norms == 0.0] = 1.0
    return K / norms[:, None]


def diff_op_fast(K):
    """Vectorised equivalent of :func:`diff_op` (row sums via reduceat; same to ~1e-16)."""
    if not sparse.issparse(K):
        return diff_op(K)
    P = sparse.csr_matrix(K, dtype=np.float64, copy=True)
    sums = np.asarray(abs(P).sum(axis=1)).ravel()
    sums[sums == 0.0] = 1.0
    P.data /= np.repeat(sums, np.diff(P.indptr))
    return P


def kernel_degree(K):
    """graphtools/base.py:648-666: row sums as an (N, 1) array."""
    return np.asarray(K.sum(axis=1)).reshape(-1, 1)


def diff_aff(K):
    """graphtools/base.py:668-698: D^-1/2 K D^-1/2."""
    deg = kernel_degree(K)
    if sparse.issparse(K):
        n = len(deg)
        Dm = sparse.csr_matrix((1 / np.sqrt(deg.flatten()), np.arange(n), np.arange(n + 1)))
        return Dm @ K @ Dm
    return (K / np.sqrt(deg)) / np.sqrt(deg.T)


def knn_graph(
    data,
    knn=5,
    decay=40,
    thresh=1e-4,
    bandwidth=None,
    bandwidth_scale=1.0,
    knn_max=None,
    kernel_symm="+",
    theta=None,
    anisotropy=0,
    engine="numpy",
    distance="euclidean",
):
    """Kernel K and diffusion operator P of ``graphtools.Graph(data, ...)`` when it
    resolves to ``kNNGraph`` (graphtools/base.py:534-555 ``_build_kernel``, :629-646 ``P``)."""
    K0 = knn_kernel(
        data, knn=knn, decay=decay, thresh=thresh, bandwidth=bandwidth,
        bandwidth_scale=bandwidth_scale, knn_max=knn_max, engine=engine, distance=distance,
    )
    K = symmetrize_kernel(K0, kernel_symm, theta)
    K = apply_anisotropy(K, anisotropy)
    K = sparse.csr_matrix(K)
    P = diff_op_fast(K)
    return K, P
