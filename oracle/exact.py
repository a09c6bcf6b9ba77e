"""Oracle: exact dense kernel of ``TraditionalGraph``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates graphtools/graphs.py:1514-1610 (``TraditionalGraph.build_kernel``):
pairwise distances by scipy ``pdist``/``squareform`` (scipy 1.15.3, C
``euclidean_distance_double``: float64, direct difference form
sqrt(sum_k (u_k - v_k)^2) accumulated sequentially in k) or a precomputed
distance matrix (dtype preserved), bandwidth = (knn+1)-th smallest entry of each
row (self's 0 counted, graphs.py:1583-1587), K = exp(-(D/bw)^decay) with NaN -> 1
(graphs.py:1593-1596), entries < thresh zeroed (graphs.py:1609).
"""
import numpy as np
from scipy import sparse

from .kernel import apply_anisotropy, diff_op, symmetrize_kernel

__all__ = ["pairwise_distances_exact", "cross_distances_exact", "exact_kernel", "exact_graph", "exact_kernel_to_data",
           "exact_graph_rows"]


def pairwise_distances_exact(X):
    """float64 N x N euclidean distances, difference form, sequential in k (scipy pdist)."""
    X64 = np.asarray(X, dtype=np.float64)
    n, d = X64.shape
    acc = np.zeros((n, n), dtype=np.float64)
    for k in range(d):
        col = X64[:, k]
        diff = col[:, None] - col[None, :]
        acc += diff * diff
    return np.sqrt(acc)


def exact_kernel(
    data, knn=5, decay=40, thresh=1e-4, bandwidth=None, bandwidth_scale=1.0, precomputed=None
):
    """Unsymmetrised dense kernel (graphtools/graphs.py:1514-1610)."""
    if precomputed in ("affinity", "adjacency"):
        # graphs.py:1532-1545: the caller's matrix is the kernel (adjacency: diagonal set to 1), truncated at thresh like
        # every other kernel (:1596-1609); sparse input stays sparse
        if sparse.issparse(data):
            K = sparse.lil_matrix(data, dtype=np.float64, copy=True)
            if precomputed == "adjacency":
                K.setdiag(1)
            K = K.tocsr()
            K.data[K.data < thresh] = 0
            K.eliminate_zeros()
            return K
        K = np.array(data, copy=True)
        if precomputed == "adjacency":
            np.fill_diagonal(K, 1)
        K[K < thresh] = 0
        return K
    if precomputed == "distance":
        pdx = np.asarray(data)
    elif precomputed is None:
        pdx = pairwise_distances_exact(data)
    else:
        raise ValueError(precomputed)
    if bandwidth is None:
        knn_dist = np.partition(pdx, knn + 1, axis=1)[:, : knn + 1]
        bw = np.max(knn_dist, axis=1)
    elif callable(bandwidth):
        bw = bandwidth(pdx)
    else:
        bw = bandwidth
    bw = bw * bandwidth_scale
    pdx = (pdx.T / bw).T
    K = np.exp(-1 * np.power(pdx, decay))
    K = np.where(np.isnan(K), 1, K)
    K[K < thresh] = 0
    return K


def cross_distances_exact(Y, X):
    """float64 M x N euclidean distances, difference form, sequential in k (scipy cdist, graphs.py:1653)."""
    Y64, X64 = np.asarray(Y, dtype=np.float64), np.asarray(X, dtype=np.float64)
    acc = np.zeros((Y64.shape[0], X64.shape[0]), dtype=np.float64)
    for k in range(X64.shape[1]):
        diff = Y64[:, k][:, None] - X64[:, k][None, :]
        acc += diff * diff
    return np.sqrt(acc)


def exact_kernel_to_data(data, Y, knn=5, decay=40, thresh=1e-4, bandwidth=None, bandwidth_scale=1.0):
    """``TraditionalGraph.build_kernel_to_data`` (graphtools/graphs.py:1612-1678, non-numba branch): bandwidth = the
    knn-th smallest distance of each row of cdist(Y, data) unless given."""
    pdx = cross_distances_exact(Y, data)
    if bandwidth is None:
        knn_dist = np.partition(pdx, knn, axis=1)[:, :knn]
        bandwidth = np.max(knn_dist, axis=1)
    elif callable(bandwidth):
        bandwidth = bandwidth(pdx)
    bandwidth = bandwidth_scale * bandwidth
    pdx = (pdx.T / bandwidth).T
    K = np.exp(-1 * pdx**decay)
    K = np.where(np.isnan(K), 1, K)
    K[K < thresh] = 0
    return K


def exact_graph(
    data, knn=5, decay=40, thresh=1e-4, bandwidth=None, bandwidth_scale=1.0, precomputed=None,
    kernel_symm="+", theta=None, anisotropy=0,
):
    """K and P of ``graphtools.Graph(..., graphtype="exact")`` (base.py:534-555, 629-646)."""
    K0 = exact_kernel(data, knn, decay, thresh, bandwidth, bandwidth_scale, precomputed)
    K = symmetrize_kernel(K0, kernel_symm, theta)
    K = apply_anisotropy(K, anisotropy)
    if sparse.issparse(K):
        K = sparse.csr_matrix(K)
    return K, diff_op(K)


def exact_graph_rows(D_rows, D_cols, bw, rows, decay=40, thresh=1e-4):
    """Rows ``rows`` of K and P of :func:`exact_graph` (precomputed distances, '+' rule, no anisotropy) WITHOUT the N x N
    matrix: what a test at BASELINE config 4's size can afford (N = 2e5: the matrix is 160 GB and lives on the device).

    D_rows : D[rows, :]  [m, N];  D_cols : D[:, rows]  [N, m]  (D need not be symmetric);  bw : the bandwidth of EVERY row -
    the (knn+1)-th smallest entry of the row, graphs.py:1583-1587 (a selection: whoever picks it gets the same value).
    The arithmetic is exact_kernel's, entry by entry and in D's dtype like numpy's (graphs.py:1589-1609): K0_ij =
    exp(-(D_ij / bw_i)^decay), NaN -> 1, < thresh -> 0; then (K0 + K0^T) / 2 (base.py:557-561) and rows over their sums
    (base.py:645, :func:`diff_op`).  Returns (K_rows, P_rows, degree_rows)."""
    D_rows, D_cols = np.asarray(D_rows), np.asarray(D_cols)
    bw = np.asarray(bw)
    rows = np.asarray(rows)

    def k0(pdx):
        K = np.exp(-1 * np.power(pdx, decay))
        K = np.where(np.isnan(K), 1, K)
        K[K < thresh] = 0
        return K

    own = k0((D_rows.T / bw[rows]).T)            # K0[rows, :]
    other = k0((D_cols.T / bw).T.T)              # K0[:, rows]^T laid out [m, N]: entry (r, j) = K0[j, rows[r]]
    K = (own + other) / 2
    norms = np.abs(K).sum(axis=1)
    deg = K.sum(axis=1)
    norms[norms == 0.0] = 1.0
    return K, K / norms[:, None], deg
