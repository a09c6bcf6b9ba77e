"""Oracle: landmark operator algebra of ``LandmarkGraph``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates graphtools/graphs.py:1169-1182 (``_landmarks_to_data``: pmn = S^T K, one
row per unique cluster label), :1232-1246 (row-L1 normalise pmn and pnm = pmn^T,
landmark_op = pmn . pnm dense, transitions = pnm) and the deterministic random
landmark assignment :1200-1213 (``default_rng(seed).choice`` + nearest landmark by
``euclidean_distances`` for n > 5000, scipy ``cdist`` otherwise).  The spectral
front end (randomized SVD + MiniBatchKMeans, :1215-1230) is out of scope: cluster
labels are an input.
"""
import numpy as np
from scipy import sparse

__all__ = ["landmark_operator", "random_landmark_clusters", "landmark_extend"]


def _normalize_l1(M):
    if sparse.issparse(M):
        M = sparse.csr_matrix(M, dtype=np.float64, copy=True)
        sums = np.asarray(abs(M).sum(axis=1)).ravel()
        sums[sums == 0.0] = 1.0
        M.data /= np.repeat(sums, np.diff(M.indptr))
        return M
    M = np.asarray(M, dtype=np.float64)
    sums = np.abs(M).sum(axis=1)
    sums[sums == 0.0] = 1.0
    return M / sums[:, None]


def landmark_operator(K, clusters):
    """Returns (landmark_op dense [L, L], transitions [N, L])."""
    clusters = np.asarray(clusters)
    landmarks, inverse = np.unique(clusters, return_inverse=True)
    L, n = len(landmarks), K.shape[0]
    S = sparse.csr_matrix((np.ones(n), (inverse, np.arange(n))), shape=(L, n))
    if sparse.issparse(K):
        pmn = sparse.csr_matrix(S @ K)
    else:
        pmn = S @ np.asarray(K)
    pnm = pmn.transpose()
    pmn = _normalize_l1(pmn)
    pnm = _normalize_l1(pnm)
    op = pmn.dot(pnm)
    if sparse.issparse(op):
        op = op.toarray()
    return np.asarray(op), pnm


def landmark_extend(kernel_to_data, clusters):
    """LandmarkGraph.extend_to_data (graphtools/graphs.py:1247-1289): the columns of the kernel from new points to the
    data summed per cluster (``np.unique`` order), rows l1-normalised.  Returns a dense [m, L] array."""
    clusters = np.asarray(clusters)
    if sparse.issparse(kernel_to_data):
        K = sparse.csr_matrix(kernel_to_data)
        pnm = sparse.hstack([sparse.csr_matrix(K[:, clusters == i].sum(axis=1)) for i in np.unique(clusters)])
        return np.asarray(_normalize_l1(sparse.csr_matrix(pnm)).toarray())
    K = np.asarray(kernel_to_data)
    pnm = np.array([np.sum(K[:, clusters == i], axis=1).T for i in np.unique(clusters)]).transpose()
    return _normalize_l1(pnm)


def random_landmark_clusters(data, n_landmark, random_state):
    """graphtools/graphs.py:1200-1213.

    For n > 5000 the reference calls sklearn ``euclidean_distances`` whose float32
    path computes chunk-wise in float64 and returns float32
    (sklearn:metrics/pairwise.py:582-596); argmin is then taken on the float32
    values (first minimum wins).  For n <= 5000 it is scipy ``cdist`` (float64,
    difference form).
    """
    data = np.asarray(data)
    n = data.shape[0]
    rng = np.random.default_rng(random_state)
    idx = rng.choice(n, n_landmark, replace=False)
    X64 = data.astype(np.float64)
    L64 = X64[idx]
    if n > 5000:
        xn = np.einsum("ij,ij->i", X64, X64)
        ln = np.einsum("ij,ij->i", L64, L64)
        d2 = xn[:, None] - 2.0 * (X64 @ L64.T) + ln[None, :]
        np.maximum(d2, 0, out=d2)
        if data.dtype == np.float32:
            d2 = d2.astype(np.float32)
        dist = np.sqrt(d2)
    else:
        acc = np.zeros((n, n_landmark))
        for k in range(X64.shape[1]):
            diff = X64[:, k][:, None] - L64[:, k][None, :]
            acc += diff * diff
        dist = np.sqrt(acc)
    return np.argmin(dist, axis=1), idx
