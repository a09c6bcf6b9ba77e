#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference, see tools/ref_import.py) on small seeded inputs.

Runs only in the build container (the reference is not on the GPU box).  Only
inputs + outputs are written - no reference source travels.  Re-run with
``python tools/make_golden.py``; it also cross-checks the numpy oracle
(``oracle/``) against every fixture it writes and prints a report.

Synthetic inputs follow SURVEY.md section 8(d):
  mix      C = max(N // 2000, 1) centres ~ U(-10, 10)^d, X = centre[label] + N(0, 1)
  manifold Z ~ N(0,1)^(N x 5), A ~ N(0,1)^(5 x d), X = Z A + 0.01 N(0,1)
  gauss    X ~ N(0,1)^(N x d)
all ``numpy.random.default_rng(seed)``, float32, C order.
"""
import os
import sys
import warnings

import numpy as np
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import import_reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def make_mix(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    c = max(n // 2000, 1)
    centres = rng.uniform(-10, 10, (c, d))
    labels = rng.integers(c, size=n)
    return (centres[labels] + rng.standard_normal((n, d))).astype(dtype)


def make_manifold(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, 5))
    a = rng.standard_normal((5, d))
    return (z @ a + 0.01 * rng.standard_normal((n, d))).astype(dtype)


def make_gauss(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, d)).astype(dtype)


def csr_parts(prefix, M):
    M = sparse.csr_matrix(M)
    M.sort_indices()
    return {
        prefix + "_data": M.data.astype(np.float64),
        prefix + "_indices": M.indices.astype(np.int32),
        prefix + "_indptr": M.indptr.astype(np.int64),
        prefix + "_shape": np.array(M.shape, dtype=np.int64),
    }


def knn_fixture(gt, name, X, knn, decay, search_k, extra=None, store_k0=True, **graph_kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = gt.Graph(X, knn=knn, decay=decay, n_pca=None, verbose=0, random_state=42, **graph_kw)
        K0 = G.build_kernel()
        K, P = G.K, G.P
        dist, ind = G.knn_tree.kneighbors(X, n_neighbors=search_k)
    out = {"X": X, "knn": np.int64(knn), "decay": np.float64(np.nan if decay is None else decay),
           "search_k": np.int64(search_k),
           "knn_dist": dist.astype(np.float32 if X.dtype == np.float32 else np.float64),
           "knn_idx": ind.astype(np.int32)}
    # float32 data: distances are float64(float32 value), the float32 store is lossless
    if X.dtype == np.float32:
        assert np.array_equal(out["knn_dist"].astype(np.float64), dist)
    if store_k0:
        out.update(csr_parts("K0", K0))
    out.update(csr_parts("K", K))
    P = sparse.csr_matrix(P)
    P.sort_indices()
    Kc = sparse.csr_matrix(K)
    Kc.sort_indices()
    assert np.array_equal(P.indices, Kc.indices) and np.array_equal(P.indptr, Kc.indptr)
    out["P_data"] = P.data.astype(np.float64)
    if extra:
        out.update(extra(G))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.2f MB  N=%d d=%d nnz(K0)=%d nnz(K)=%d" % (
        name, os.path.getsize(path) / 1e6, X.shape[0], X.shape[1], K0.nnz, Kc.nnz))
    return G


def check_oracle_knn(name):
    import oracle

    z = np.load(os.path.join(OUT, name + ".npz"))
    X = z["X"]
    knn = int(z["knn"])
    decay = None if np.isnan(z["decay"]) else float(z["decay"])
    kw = {}
    for key in ("bandwidth", "bandwidth_scale", "kernel_symm", "theta", "anisotropy", "knn_max", "thresh"):
        if "param_" + key in z.files:
            v = z["param_" + key]
            kw[key] = v.item() if v.shape == () else v
            if key == "kernel_symm":
                kw[key] = str(kw[key])
                if kw[key] == "none":
                    kw[key] = None
    d, i = oracle.kneighbors(X, None, int(z["search_k"]))
    ties = np.any(np.diff(z["knn_dist"].astype(np.float64), axis=1) == 0)
    same_idx = np.array_equal(i, z["knn_idx"])
    dd = np.abs(d[:, 1:] - z["knn_dist"].astype(np.float64)[:, 1:]).max()
    K, P = oracle.knn_graph(X, knn=knn, decay=decay, **kw)
    Kg = sparse.csr_matrix((z["K_data"], z["K_indices"], z["K_indptr"]), shape=tuple(z["K_shape"]))
    dK = abs(K - Kg)
    print("   oracle vs %-20s idx_equal=%s (ties=%s) max|dD|=%.2e  struct_diff=%d max|dK|=%.2e max|dP|=%.2e" % (
        name, same_idx, ties, dd, (K != Kg).nnz, dK.max() if dK.nnz else 0.0,
        np.abs(sparse.csr_matrix(P).data - z["P_data"]).max() if K.nnz == Kg.nnz else np.nan))


def main():
    os.makedirs(OUT, exist_ok=True)
    gt = import_reference()
    from sklearn import datasets

    digits = datasets.load_digits().data  # float64 1797 x 64

    # G1 / G2: digits, alpha-decay and connectivity kernels
    knn_fixture(gt, "g1_digits_decay40", digits, knn=5, decay=40, search_k=36)
    knn_fixture(gt, "g2_digits_binary", digits, knn=5, decay=None, search_k=6)
    knn_fixture(gt, "g2b_mix_binary", make_mix(768, 50, 6), knn=15, decay=None, search_k=16)

    # G3: mixture (one 1024-point blob at this size), float32
    Xmix = make_mix(1024, 50, 0)

    def lm_extra(G):
        return {}

    knn_fixture(gt, "g3_mix_f32", Xmix, knn=15, decay=40, search_k=96)
    # G4: isotropic gaussian, forces the k*6 re-search loop then the radius fallback
    knn_fixture(gt, "g4_gauss_f32", make_gauss(1536, 64, 1), knn=15, decay=40, search_k=96)
    # G5: 5-dim manifold in 64-d, no overflow rows
    knn_fixture(gt, "g5_manifold_f32", make_manifold(1024, 64, 2), knn=15, decay=40, search_k=96)

    # G3 variants: symmetrisation / bandwidth / anisotropy / knn_max (parameters stored as param_*)
    small = make_mix(384, 50, 5)
    variants = {
        "g3b_mix_symm_mul": dict(kernel_symm="*"),
        "g3c_mix_symm_mnn": dict(kernel_symm="mnn", theta=0.7),
        "g3d_mix_symm_none": dict(kernel_symm=None),
        "g3e_mix_aniso": dict(anisotropy=0.5),
        "g3f_mix_bwscalar": dict(bandwidth=7.0, bandwidth_scale=1.1),
        "g3g_mix_knnmax": dict(knn_max=40),
        "g3h_mix_thresh": dict(thresh=1e-2, bandwidth_scale=0.8),
    }
    for name, kw in variants.items():
        def extra(G, kw=kw):
            e = {}
            for k, v in kw.items():
                e["param_" + k] = np.array("none" if v is None else v)
            return e
        knn_fixture(gt, name, small, knn=10, decay=20, search_k=66, extra=extra, store_k0=False, **kw)
    rng = np.random.default_rng(7)
    bw_vec = rng.uniform(6.5, 8.5, size=small.shape[0])

    def extra_bw(G):
        return {"param_bandwidth": bw_vec}

    knn_fixture(gt, "g3i_mix_bwvector", small, knn=10, decay=20, search_k=66, extra=extra_bw, store_k0=False, bandwidth=bw_vec)

    # G6: exact (TraditionalGraph) from data and from precomputed distances (f64 / f32 D)
    Xe = make_mix(256, 100, 2)
    from scipy.spatial.distance import pdist, squareform

    D64 = squareform(pdist(Xe.astype(np.float64)))
    D32 = D64.astype(np.float32)
    out = {"X": Xe, "D64": D64, "D32": D32, "knn": np.int64(15), "decay": np.float64(40)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, data, kw in [
            ("data_t1e-4", Xe, dict(thresh=1e-4)),
            ("data_t0", Xe, dict(thresh=0)),
            ("d64_t1e-4", D64, dict(thresh=1e-4, precomputed="distance")),
            ("d32_t1e-4", D32, dict(thresh=1e-4, precomputed="distance")),
            ("d32_t0", D32, dict(thresh=0, precomputed="distance")),
        ]:
            G = gt.Graph(data, knn=15, decay=40, n_pca=None, graphtype="exact", verbose=0, **kw)
            assert type(G).__name__ == "TraditionalGraph"
            out["K_" + tag] = np.asarray(G.K)
            out["P_" + tag] = np.asarray(G.P)
    path = os.path.join(OUT, "g6_exact.npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.2f MB" % ("g6_exact", os.path.getsize(path) / 1e6))

    # G7: landmark operator, random landmarking (deterministic) on the G3 data
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = gt.Graph(Xmix, knn=15, decay=40, n_pca=None, n_landmark=50, random_landmarking=True,
                     random_state=42, verbose=0)
        lop = G.landmark_op
        out = {"X": Xmix, "n_landmark": np.int64(50), "random_state": np.int64(42),
               "clusters": np.asarray(G.clusters).astype(np.int32), "landmark_op": np.asarray(lop)}
        out.update(csr_parts("transitions", G.transitions))
        out.update(csr_parts("K", G.K))
        # spectral clusters as an INPUT (front end out of scope)
        G2 = gt.Graph(Xmix, knn=15, decay=40, n_pca=None, n_landmark=50, random_state=42, verbose=0)
        out["spectral_clusters"] = np.asarray(G2.clusters).astype(np.int32)
        out["spectral_landmark_op"] = np.asarray(G2.landmark_op)
        out.update(csr_parts("spectral_transitions", G2.transitions))
        # n > 5000 path of the random landmark assignment (sklearn euclidean_distances)
        Xbig = make_mix(6000, 20, 3)
        G3 = gt.Graph(Xbig, knn=5, decay=None, n_pca=None, n_landmark=64, random_landmarking=True,
                      random_state=7, verbose=0)
        out["big_X"] = Xbig
        out["big_clusters"] = np.asarray(G3.clusters).astype(np.int32)
    path = os.path.join(OUT, "g7_landmark.npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.2f MB" % ("g7_landmark", os.path.getsize(path) / 1e6))

    # G8: cosine metric (float64 input)
    Xc = make_mix(768, 50, 4, dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = gt.Graph(Xc, knn=10, decay=40, n_pca=None, distance="cosine", verbose=0)
        dist, ind = G.knn_tree.kneighbors(Xc, n_neighbors=66)
        out = {"X": Xc, "knn": np.int64(10), "decay": np.float64(40), "knn_dist": dist, "knn_idx": ind.astype(np.int32)}
        out.update(csr_parts("K", G.K))
        out["P_data"] = sparse.csr_matrix(G.P).data
    path = os.path.join(OUT, "g8_cosine.npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.2f MB" % ("g8_cosine", os.path.getsize(path) / 1e6))

    print("\noracle cross-check:")
    for name in ["g1_digits_decay40", "g2_digits_binary", "g2b_mix_binary", "g3_mix_f32", "g4_gauss_f32", "g5_manifold_f32",
                 *variants.keys(), "g3i_mix_bwvector"]:
        check_oracle_knn(name)


if __name__ == "__main__":
    main()
