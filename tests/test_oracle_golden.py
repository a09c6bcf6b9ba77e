"""CPU: the numpy oracle (oracle/) reproduces every golden vector generated from the reference."""
import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import golden_csr, golden_params, load_golden, mnn_params

KNN_FIXTURES = [
    "g1_digits_decay40", "g2b_mix_binary", "g3_mix_f32", "g4_gauss_f32", "g5_manifold_f32",
    "g3b_mix_symm_mul", "g3c_mix_symm_mnn", "g3d_mix_symm_none", "g3e_mix_aniso", "g3f_mix_bwscalar",
    "g3g_mix_knnmax", "g3h_mix_thresh", "g3i_mix_bwvector",
]


def _decay(z):
    return None if np.isnan(z["decay"]) else float(z["decay"])


@pytest.mark.parametrize("name", KNN_FIXTURES)
def test_knn_graph_matches_reference(name):
    z = load_golden(name)
    K, P = oracle.knn_graph(z["X"], knn=int(z["knn"]), decay=_decay(z), **golden_params(z))
    Kg = golden_csr(z, "K")
    K = sparse.csr_matrix(K)
    K.sort_indices()
    assert (K != Kg).nnz == 0  # identical structure and values (same float64 arithmetic)
    P = sparse.csr_matrix(P)
    P.sort_indices()
    np.testing.assert_allclose(P.data, z["P_data"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", ["g1_digits_decay40", "g3_mix_f32", "g4_gauss_f32", "g5_manifold_f32", "g2b_mix_binary"])
def test_unsymmetrised_kernel_matches_reference(name):
    z = load_golden(name)
    K0 = oracle.knn_kernel(z["X"], knn=int(z["knn"]), decay=_decay(z))
    K0 = sparse.csr_matrix(K0)
    K0.sort_indices()
    assert (K0 != golden_csr(z, "K0")).nnz == 0


@pytest.mark.parametrize("name", ["g3_mix_f32", "g4_gauss_f32", "g5_manifold_f32", "g2b_mix_binary"])
def test_kneighbors_matches_sklearn_vectors(name):
    z = load_golden(name)
    d, i = oracle.kneighbors(z["X"], None, int(z["search_k"]))
    assert np.array_equal(i, z["knn_idx"])  # bit-exact neighbour order (no exact ties in these inputs)
    # distances: float64(float32) rounding of scikit-learn; column 0 is self (rounding noise in sklearn)
    assert np.array_equal(d[:, 1:], z["knn_dist"].astype(np.float64)[:, 1:])


def test_kneighbors_digits_ties():
    """integer-valued data: exact distance ties; neighbour order within a tie group is unspecified in
    scikit-learn, distances must still agree exactly and indices as sets per distinct distance."""
    z = load_golden("g1_digits_decay40")
    d, i = oracle.kneighbors(z["X"], None, int(z["search_k"]))
    assert np.array_equal(d, z["knn_dist"])
    gi = z["knn_idx"]
    for r in range(0, d.shape[0], 37):
        # all but the last tie group (which may be cut by k) must hold the same members
        last = d[r, -1]
        keep = d[r] < last
        assert set(i[r, keep]) == set(gi[r, keep])


def test_digits_binary_kernel_ties():
    z = load_golden("g2_digits_binary")
    K, _ = oracle.knn_graph(z["X"], knn=int(z["knn"]), decay=None)
    Kg = golden_csr(z, "K")
    # tie-breaking at the k-th neighbour may swap equidistant members: only a handful of entries differ
    assert (sparse.csr_matrix(K) != Kg).nnz <= 64
    assert abs(K.sum() - Kg.sum()) < 1e-9


def test_radius_neighbors_rounding():
    z = load_golden("g3_mix_f32")
    X = z["X"]
    from sklearn.neighbors import NearestNeighbors

    nn = NearestNeighbors(algorithm="brute").fit(X)
    d_ref, i_ref = nn.radius_neighbors(X[:64], radius=9.3)
    d, i = oracle.radius_neighbors(X, X[:64], 9.3)
    for r in range(64):
        o_ref, o = np.argsort(i_ref[r]), np.argsort(i[r])
        assert np.array_equal(i_ref[r][o_ref], i[r][o])
        keep = i[r][o] != r
        assert np.array_equal(d_ref[r][o_ref][keep], d[r][o][keep])


def test_kneighbors_matches_live_sklearn():
    from sklearn.neighbors import NearestNeighbors

    rng = np.random.default_rng(11)
    X = rng.standard_normal((700, 33)).astype(np.float32)
    Y = rng.standard_normal((50, 33)).astype(np.float32)
    nn = NearestNeighbors(algorithm="brute").fit(X)
    d_ref, i_ref = nn.kneighbors(Y, 40)
    d, i = oracle.kneighbors(X, Y, 40)
    assert np.array_equal(i, i_ref)
    assert np.array_equal(d, d_ref)


@pytest.mark.parametrize("tag,src,kw", [
    ("data_t1e-4", "X", dict(thresh=1e-4)),
    ("data_t0", "X", dict(thresh=0)),
    ("d64_t1e-4", "D64", dict(thresh=1e-4, precomputed="distance")),
    ("d32_t1e-4", "D32", dict(thresh=1e-4, precomputed="distance")),
    ("d32_t0", "D32", dict(thresh=0, precomputed="distance")),
])
def test_exact_graph_matches_reference(tag, src, kw):
    z = load_golden("g6_exact")
    K, P = oracle.exact_graph(z[src], knn=int(z["knn"]), decay=float(z["decay"]), **kw)
    assert K.dtype == z["K_" + tag].dtype
    assert np.array_equal(K, z["K_" + tag])
    assert np.array_equal(P, z["P_" + tag])


def test_pairwise_distances_match_scipy_vector():
    z = load_golden("g6_exact")
    assert np.array_equal(oracle.pairwise_distances_exact(z["X"]), z["D64"])


def test_landmark_operator_matches_reference():
    z = load_golden("g7_landmark")
    K = golden_csr(z, "K")
    cl, _ = oracle.random_landmark_clusters(z["X"], int(z["n_landmark"]), int(z["random_state"]))
    assert np.array_equal(cl, z["clusters"])
    op, tr = oracle.landmark_operator(K, z["clusters"])
    np.testing.assert_allclose(op, z["landmark_op"], rtol=0, atol=1e-14)
    T = golden_csr(z, "transitions")
    assert abs(sparse.csr_matrix(tr) - T).max() < 1e-14
    op2, tr2 = oracle.landmark_operator(K, z["spectral_clusters"])
    np.testing.assert_allclose(op2, z["spectral_landmark_op"], rtol=0, atol=1e-14)
    cl_big, _ = oracle.random_landmark_clusters(z["big_X"], 64, 7)
    assert np.array_equal(cl_big, z["big_clusters"])


def test_landmark_extend_and_interpolate_match_reference():
    """LandmarkGraph.extend_to_data / interpolate (graphs.py:1247-1317) on the G7 configuration"""
    z = load_golden("g7b_landmark_extend")
    X, Y = z["X"], z["Y"]
    cl, _ = oracle.random_landmark_clusters(X, int(z["n_landmark"]), int(z["random_state"]))
    assert np.array_equal(cl, z["clusters"])
    # build_kernel_to_data(Y) of a kNN graph built with knn=15: knn neighbours (no self), bandwidth from them
    Ky = oracle.knn_kernel(X, knn=15, decay=40, Y=Y)
    pnm = oracle.landmark_extend(Ky, cl)
    np.testing.assert_allclose(pnm, z["extend_pnm"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(pnm.dot(z["transform"]), z["interp_Y"], rtol=0, atol=1e-13)
    K, _ = oracle.knn_graph(X, knn=15, decay=40)
    _, tr = oracle.landmark_operator(sparse.csr_matrix(K), cl)
    np.testing.assert_allclose(tr.dot(z["transform"]), z["interp_self"], rtol=0, atol=1e-13)


def test_cosine_matches_reference():
    z = load_golden("g8_cosine")
    d, i = oracle.knn.cosine_kneighbors(z["X"], None, 66)
    assert np.array_equal(i, z["knn_idx"])
    assert np.array_equal(d, z["knn_dist"])
    K, P = oracle.knn_graph(z["X"], knn=int(z["knn"]), decay=float(z["decay"]), distance="cosine")
    Kg = golden_csr(z, "K")
    assert (sparse.csr_matrix(K) != Kg).nnz == 0


MNN_FIXTURES = ["g9_mnn_decay", "g9b_mnn_binary_theta", "g9c_mnn_aniso"]


@pytest.mark.parametrize("name", MNN_FIXTURES)
def test_mnn_graph_matches_reference(name):
    """MNNGraph.build_kernel (graphs.py:1870-1946) + symmetrisation + P, bit-identical to the reference."""
    z = load_golden(name)
    K0, K, P = oracle.mnn_graph(z["X"], z["sample_idx"], **mnn_params(z))
    K0.sort_indices()
    assert (K0 != golden_csr(z, "K0")).nnz == 0
    assert (K != golden_csr(z, "K")).nnz == 0
    np.testing.assert_allclose(P.data, z["P_data"], rtol=0, atol=1e-15)


def test_mnn_landmark_graph_matches_reference():
    """MNNLandmarkGraph (graphs.py:1973-1974): random landmark clusters, landmark operator and transitions."""
    z = load_golden("g9d_mnn_landmark")
    _, K, _ = oracle.mnn_graph(z["X"], z["sample_idx"], **mnn_params(z))
    clusters, _ = oracle.random_landmark_clusters(z["X"], int(z["n_landmark"]), int(z["random_state"]))
    assert np.array_equal(clusters, z["clusters"])
    op, tr = oracle.landmark_operator(K, clusters)
    np.testing.assert_allclose(op, z["landmark_op"], rtol=0, atol=1e-15)
    assert abs(sparse.csr_matrix(tr) - golden_csr(z, "transitions")).max() < 1e-15


# ---- round 3: diff_aff, precomputed affinity / adjacency, exact-graph extension, per-row bandwidth of the kNN extension ----
def test_diff_aff_matches_reference():
    """BaseGraph.kernel_degree / diff_aff (base.py:648-698) on the reference's own K"""
    z = load_golden("g10_diff_aff")
    K = golden_csr(z, "K")
    assert np.array_equal(oracle.kernel_degree(K).ravel(), z["degree"])
    A = sparse.csr_matrix(oracle.diff_aff(K))
    A.sort_indices()
    Ar = golden_csr(z, "A")
    assert np.array_equal(A.indptr, Ar.indptr) and np.array_equal(A.indices, Ar.indices)
    assert np.array_equal(A.data, Ar.data)
    assert np.array_equal(oracle.diff_aff(z["exact_K"]), z["exact_A"])


@pytest.mark.parametrize("tag,src,kw", [
    ("aff64", "A", dict(precomputed="affinity")),
    ("aff64_mnn", "A", dict(precomputed="affinity", kernel_symm="mnn", theta=0.7)),
    ("aff64_aniso", "A", dict(precomputed="affinity", anisotropy=0.5)),
    ("aff32", "A32", dict(precomputed="affinity")),
    ("adj64", "Adj", dict(precomputed="adjacency")),
])
def test_precomputed_affinity_and_adjacency_match_reference(tag, src, kw):
    z = load_golden("g11_exact_passthrough")
    K, P = oracle.exact_graph(z[src], knn=5, decay=40, **kw)
    assert K.dtype == z["K_" + tag].dtype
    assert np.array_equal(K, z["K_" + tag])
    assert np.array_equal(P, z["P_" + tag])


@pytest.mark.parametrize("tag,src,mode", [("adj_sparse", "Adj", "adjacency"), ("aff_sparse", "A", "affinity")])
def test_sparse_precomputed_matches_reference(tag, src, mode):
    z = load_golden("g11_exact_passthrough")
    K, P = oracle.exact_graph(sparse.csr_matrix(z[src]), knn=5, decay=40, precomputed=mode)
    K = sparse.csr_matrix(K)
    K.sort_indices()
    Kr, Pr = golden_csr(z, "K_" + tag), golden_csr(z, "P_" + tag)
    assert np.array_equal(K.indptr, Kr.indptr) and np.array_equal(K.indices, Kr.indices) and np.array_equal(K.data, Kr.data)
    P = sparse.csr_matrix(P)
    P.sort_indices()
    np.testing.assert_allclose(P.data, Pr.data, rtol=1e-15, atol=0)


@pytest.mark.parametrize("tag,kw", [
    ("K_default", {}),
    ("K_knn4", dict(knn=4)),
    ("K_bw_scalar", dict(bandwidth=4.5)),
    ("K_bw_vector", dict(bandwidth="vector", bandwidth_scale=1.25)),
])
def test_exact_extension_matches_reference(tag, kw):
    z = load_golden("g12_exact_extend")
    kw = dict(kw)
    if kw.get("bandwidth") == "vector":
        kw["bandwidth"] = z["bw_vector"]
    kw.setdefault("knn", int(z["knn"]))
    K = oracle.exact_kernel_to_data(z["X"], z["Y"], decay=float(z["decay"]), **kw)
    assert np.array_equal(K, z[tag])
    if tag == "K_default":
        T = K / np.abs(K).sum(axis=1, keepdims=True)
        np.testing.assert_allclose(T, z["T_default"], rtol=1e-15, atol=0)
        K64 = oracle.exact_kernel_to_data(z["X"].astype(np.float64), z["Y"].astype(np.float64), knn=int(z["knn"]),
                                          decay=float(z["decay"]))
        assert np.array_equal(K64, z["K_f64"])


@pytest.mark.parametrize("tag,scale", [("K_bwvec", 1.0), ("K_bwvec_scaled", 0.8)])
def test_knn_extension_with_vector_bandwidth_matches_reference(tag, scale):
    z = load_golden("g13_knn_extend_bwvec")
    K = oracle.knn_kernel(z["X"], knn=int(z["knn"]), decay=float(z["decay"]), Y=z["Y"], bandwidth=z["bw_vector"],
                          bandwidth_scale=scale, engine="sklearn")
    K = sparse.csr_matrix(K)
    K.sort_indices()
    Kr = golden_csr(z, tag)
    assert np.array_equal(K.indptr, Kr.indptr) and np.array_equal(K.indices, Kr.indices)
    assert np.array_equal(K.data, Kr.data)


def test_exact_graph_rows_is_exact_graph_restricted_to_rows():
    """oracle.exact_graph_rows (the checker of BASELINE config 4 at full size, where the N x N matrix lives on the device only)
    gives the rows of oracle.exact_graph bit for bit - also on a matrix that is not symmetric"""
    from scipy.spatial.distance import pdist, squareform

    import oracle

    rng = np.random.default_rng(0)
    n = 300
    D = squareform(pdist(rng.standard_normal((n, 6)))).astype(np.float32)
    D = (D * (1 + 0.05 * rng.random((n, n)))).astype(np.float32)
    np.fill_diagonal(D, 0)
    for decay, knn in ((15, 5), (40, 15)):
        K, P = oracle.exact_graph(D, knn=knn, decay=decay, precomputed="distance")
        bw = np.max(np.partition(D, knn + 1, axis=1)[:, : knn + 1], axis=1)
        rows = np.array([3, 17, 299, 0])
        Kr, Pr, deg = oracle.exact_graph_rows(D[rows], D[:, rows], bw, rows, decay=decay)
        assert Kr.dtype == K.dtype == np.float32
        assert np.array_equal(K[rows], Kr) and np.array_equal(P[rows], Pr)
        np.testing.assert_allclose(deg, K[rows].sum(axis=1), rtol=1e-6)
