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

from .kernel import apply_anisotropy, diff_op, symmetrize_kernel

__all__ = ["pairwise_distances_exact", "exact_kernel", "exact_graph"]


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
    if precomputed == "affinity":
        return np.asarray(data)
    if precomputed == "adjacency":
        K = np.array(data, copy=True)
        np.fill_diagonal(K, 1)
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


def exact_graph(
    data, knn=5, decay=40, thresh=1e-4, bandwidth=None, bandwidth_scale=1.0, precomputed=None,
    kernel_symm="+", theta=None, anisotropy=0,
):
    """K and P of ``graphtools.Graph(..., graphtype="exact")`` (base.py:534-555, 629-646)."""
    K0 = exact_kernel(data, knn, decay, thresh, bandwidth, bandwidth_scale, precomputed)
    K = symmetrize_kernel(K0, kernel_symm, theta)
    K = apply_anisotropy(K, anisotropy)
    return K, diff_op(K)
