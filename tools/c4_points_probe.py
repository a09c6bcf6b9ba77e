"""exact graph from points through the neighbour search: sparse form vs dense copy vs oracle (debug / timing probe)"""
import sys, time
import numpy as np
from scipy import sparse
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import make_mix
from graphtools_amd import _hip
import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4500
X = make_mix(n, 24, 33).astype(np.float64) + 40.0
Xc = np.ascontiguousarray(X - X.mean(axis=0, keepdims=True))
ctx = _hip.Context(0)
ctx.set_points(Xc)
params, keep = _hip.Context.make_params(6, 20, 1e-4, None, 1.0, None, "+", None, 0)
nnz, flags = ctx.graph_build(params)
d, i, p = ctx.graph_fetch_csr(_hip.CSR_K)
Ks = sparse.csr_matrix((d, i, p), shape=(n, n))
Kd = ctx.graph_to_dense(_hip.CSR_K, n)
print("nnz", nnz, "flags", flags, "dense nnz", (Kd != 0).sum(), "max diff", abs(Ks.toarray() - Kd).max())
K0, P0 = oracle.exact_graph(X, knn=6, decay=20, thresh=1e-4)
print("oracle nnz", (K0 != 0).sum())
m = (K0 != 0) & (Kd != 0)
print("rel", (abs(Kd - K0)[m] / K0[m]).max(), "flips", ((K0 != 0) != (Kd != 0)).sum())
from oracle.exact import pairwise_distances_exact
pdx = pairwise_distances_exact(X)
srt = np.sort(pdx, axis=1)
bw_dev = ctx.graph_fetch_vec(_hip.VEC_BANDWIDTH) if hasattr(_hip, "VEC_BANDWIDTH") else None
print("oracle bw (rank 6)", srt[:3, 6], "rank 7", srt[:3, 7], "device bw", None if bw_dev is None else bw_dev[:3])
row = 0
print("row0 dev entries", Ks[row].nnz, "oracle", (K0[row] != 0).sum())
# K0 is symmetrised: compare unsymmetrised
from oracle.exact import exact_kernel
Ku = exact_kernel(X, knn=6, decay=20, thresh=1e-4)
params.kernel_symm = _hip.SYMM[None]
ctx.graph_build(params)
Kdu = ctx.graph_to_dense(_hip.CSR_K, n)
m = (Ku != 0) & (Kdu != 0)
print("unsym: dev nnz", (Kdu != 0).sum(), "oracle", (Ku != 0).sum(), "rel", (abs(Kdu - Ku)[m] / Ku[m]).max())
import graphtools_amd
G = graphtools_amd.Graph(X, n_pca=None, graphtype="exact", knn=6, decay=20, thresh=1e-4)
print("Graph K nnz", (G.K != 0).sum(), "P nnz", (G.P != 0).sum())
