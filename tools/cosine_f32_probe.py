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
        d_ref, i_ref = NearestNeighbors(n_neighbors=13, metric="cosine", algorithm="brute").fit(X).kneighbors(X)
        d_dev, i_dev = G.knn_tree.kneighbors(X, n_neighbors=13)
    D = abs(sparse.csr_matrix(G.K) - sparse.csr_matrix(K0))
    same = (i_ref == i_dev)
    print(name, "max|K-Kref| %.3g" % D.max(), "nnz %d vs %d" % (G.K.nnz, K0.nnz), "index agreement %.5f" % same.mean(),
          "max|d-dref| %.3g (where same index), median d %.3g" % (np.abs(d_dev - d_ref)[same].max(), np.median(d_ref[:, 1:])))
