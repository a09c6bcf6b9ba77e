"""float32 cosine: what the device returns against scikit-learn's float32 arithmetic (sgemm of the normalised rows) -
the measurements behind tests/test_gpu_graph.py::test_float32_cosine_agrees_with_scikit_learn_up_to_float32_near_ties."""
import sys, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from scipy import sparse
import graphtools_amd, oracle
from conftest import make_mix, make_gauss
from sklearn.neighbors import NearestNeighbors
for name, X in (("mix+3", make_mix(6000, 40, 17, np.float32) + 3.0), ("gauss+1", make_gauss(6000, 32, 5).astype(np.float32) + 1.0), ("mix", make_mix(6000, 64, 3, np.float32))):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = graphtools_amd.Graph(X, knn=12, decay=15, n_pca=None, distance="cosine", verbose=0)
        K0, P0 = oracle.knn_graph(X, knn=12, decay=15, distance="cosine")
        K64, P64 = oracle.knn_graph(X.astype(np.float64), knn=12, decay=15, distance="cosine")
        d_ref, i_ref = NearestNeighbors(n_neighbors=13, metric="cosine", algorithm="brute").fit(X).kneighbors(X)
        d_dev, i_dev = G.knn_tree.kneighbors(X, n_neighbors=13)
    Kd = sparse.csr_matrix(G.K)
    D = abs(Kd - sparse.csr_matrix(K0))
    same = (i_ref == i_dev)
    print(name, "max|K-Kref| %.3g" % D.max(), "nnz %d vs %d" % (G.K.nnz, K0.nnz), "index agreement %.5f" % same.mean(),
          "max|d-dref| %.3g (same index) %.3g (other), median d %.3g" % (np.abs(d_dev - d_ref)[same].max(),
          np.abs(d_dev - d_ref)[~same].max() if (~same).any() else 0.0, np.median(d_ref[:, 1:])))
    rows_ok = same.all(axis=1)
    Dr = D.tocsr()[np.nonzero(rows_ok)[0]]
    print("   rows whose 13 indices all agree: %d of %d; max|K-Kref| on them %.3g" % (rows_ok.sum(), len(rows_ok), Dr.max()))
    D64 = abs(Kd - sparse.csr_matrix(K64))
    print("   against the float64 oracle (exact distances, unrounded): max|K-K64| %.3g nnz %d vs %d" % (D64.max(), Kd.nnz, K64.nnz))
    # relative bound: |dK| <= K u decay (|dd|/d + |dbw|/bw) with u = -ln K
    A = Kd.tocoo()
    kref = np.asarray(sparse.csr_matrix(K0)[A.row, A.col]).ravel()
    both = kref > 0
    k, kr = A.data[both], kref[both]
    u = -np.log(np.minimum(np.maximum(kr, 1e-300), 1.0))
    rel = np.abs(k - kr) / np.maximum(kr * np.maximum(u, 1e-3), 1e-300)
    print("   |dK| / (K ln(1/K)) max %.3g  (= decay x relative distance noise)" % rel.max())
