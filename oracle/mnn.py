"""Oracle: mutual-nearest-neighbours (batch correction) kernel.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates ``MNNGraph.build_kernel`` (graphtools/graphs.py:1870-1946) on top of the kNN restatement:

* one ``kNNGraph`` per batch with ``kernel_symm="+"`` (graphs.py:1885-1903), K_ii = its symmetrised kernel;
* for every ordered pair (i, j), i != j, the rectangular kernel from batch i's points to batch j's graph,
  ``Y.build_kernel_to_data(X.data_nu, knn=knn)`` (graphs.py:1924), every row scaled by
  ``min(1, rowsum(K_ii) / rowsum(K_ij)) * beta`` (graphs.py:1925-1933);
* blocks placed at the batches' row / column positions (``matrix.set_submatrix``, graphs.py:1910-1941);
* then the generic tail: symmetrisation, anisotropy (base.py:557-592) and ``P`` (base.py:629-646).
"""
import numpy as np
from scipy import sparse

from . import kernel as _kernel

__all__ = ["mnn_kernel", "mnn_graph"]


def mnn_kernel(data, sample_idx, knn=5, decay=None, thresh=1e-4, bandwidth=None, beta=1, distance="euclidean",
               engine="numpy"):
    """Unsymmetrised MNN kernel (scipy CSR, n x n)."""
    data = np.ascontiguousarray(data)
    sample_idx = np.asarray(sample_idx)
    samples = np.unique(sample_idx)
    n = data.shape[0]
    masks = [sample_idx == s for s in samples]
    index = [np.nonzero(m)[0] for m in masks]
    parts = [data[m] for m in masks]
    rows, cols, vals = [], [], []
    within = []
    for i, Xi in enumerate(parts):
        Kii, _ = _kernel.knn_graph(Xi, knn=knn, decay=decay, thresh=thresh, bandwidth=bandwidth, kernel_symm="+",
                                   engine=engine, distance=distance)
        Kii = sparse.csr_matrix(Kii).tocoo()
        rows.append(index[i][Kii.row])
        cols.append(index[i][Kii.col])
        vals.append(Kii.data)
        within.append(np.array(np.sum(Kii.tocsr(), 1)).flatten())
    for i, Xi in enumerate(parts):
        for j, Xj in enumerate(parts):
            if i == j:
                continue
            Kij = _kernel.knn_kernel(Xj, knn=knn, decay=decay, thresh=thresh, bandwidth=bandwidth, Y=Xi, engine=engine,
                                     distance=distance)
            between = np.array(np.sum(Kij, 1)).flatten()
            scale = np.minimum(1, within[i] / between) * beta
            Kij = sparse.csr_matrix(Kij).multiply(scale[:, None]).tocoo()
            rows.append(index[i][Kij.row])
            cols.append(index[j][Kij.col])
            vals.append(Kij.data)
    return sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))


def mnn_graph(data, sample_idx, knn=5, decay=None, thresh=1e-4, bandwidth=None, beta=1, kernel_symm="+", theta=None,
              anisotropy=0, distance="euclidean", engine="numpy"):
    """Kernel K and diffusion operator P of ``graphtools.Graph(data, sample_idx=...)`` (MNNGraph)."""
    K0 = mnn_kernel(data, sample_idx, knn=knn, decay=decay, thresh=thresh, bandwidth=bandwidth, beta=beta,
                    distance=distance, engine=engine)
    K = _kernel.symmetrize_kernel(K0, kernel_symm, theta)
    K = _kernel.apply_anisotropy(K, anisotropy)
    K = sparse.csr_matrix(K)
    K.sort_indices()
    return K0, K, _kernel.diff_op_fast(K)
