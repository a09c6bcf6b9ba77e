"""GPU: the drop-in holes closed in round 3, against fixtures generated from the reference (tools/make_golden_r3.py):
``diff_aff`` on the device, ``TraditionalGraph`` with ``precomputed="affinity"`` / ``"adjacency"`` (dense and sparse),
``TraditionalGraph.build_kernel_to_data`` / ``extend_to_data``, the per-row bandwidth of ``kNNGraph.build_kernel_to_data``
and exact graphs under a non-euclidean metric.  Bars: CSR structure identical; float64 values within 1e-9 relative (the
device's exp / pow against numpy's), float32 within 1e-5."""
import warnings

import numpy as np
import pytest
from scipy import sparse

import graphtools_amd
from conftest import golden_csr, load_golden

pytestmark = pytest.mark.gpu


def _graph(*a, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return graphtools_amd.Graph(*a, n_pca=None, verbose=0, **kw)


def _same_sparse(M, Mr, rtol):
    M = sparse.csr_matrix(M)
    M.sort_indices()
    assert np.array_equal(M.indptr, Mr.indptr) and np.array_equal(M.indices, Mr.indices)
    np.testing.assert_allclose(M.data, Mr.data, rtol=rtol, atol=0)


def test_diff_aff_on_the_device_matches_reference():
    z = load_golden("g10_diff_aff")
    G = _graph(z["X"], knn=int(z["knn"]), decay=float(z["decay"]))
    assert type(G).__name__ == "kNNGraph"
    np.testing.assert_allclose(np.asarray(G.kernel_degree).ravel(), z["degree"], rtol=1e-13, atol=0)
    A = G.diff_aff
    assert sparse.issparse(A) and A.shape == G.K.shape
    _same_sparse(A, golden_csr(z, "A"), 1e-12)
    Ga = _graph(z["X"], knn=int(z["knn"]), decay=float(z["decay"]), anisotropy=0.5)
    np.testing.assert_allclose(np.asarray(Ga.kernel_degree).ravel(), z["aniso_degree"], rtol=1e-12, atol=0)
    _same_sparse(Ga.diff_aff, golden_csr(z, "aniso_A"), 1e-11)
    Ge = _graph(z["exact_X"], knn=7, decay=10, graphtype="exact")
    np.testing.assert_allclose(Ge.diff_aff, z["exact_A"], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("tag,src,kw,rtol", [
    ("aff64", "A", dict(precomputed="affinity"), 1e-13),
    ("aff64_mnn", "A", dict(precomputed="affinity", kernel_symm="mnn", theta=0.7), 1e-13),
    ("aff64_aniso", "A", dict(precomputed="affinity", anisotropy=0.5), 1e-12),
    ("aff32", "A32", dict(precomputed="affinity"), 1e-6),
    ("adj64", "Adj", dict(precomputed="adjacency"), 1e-13),
])
def test_precomputed_affinity_and_adjacency_dense(tag, src, kw, rtol):
    """graphs.py:1532-1545: the caller's matrix is the kernel (adjacency: diagonal 1); truncation, symmetrisation,
    anisotropy and P on the device"""
    z = load_golden("g11_exact_passthrough")
    G = _graph(z[src].copy(), **kw)
    assert type(G).__name__ == "TraditionalGraph"
    Kr, Pr = z["K_" + tag], z["P_" + tag]
    assert G.K.dtype == Kr.dtype and G.P.dtype == Pr.dtype
    assert np.array_equal(G.K == 0, Kr == 0)
    np.testing.assert_allclose(G.K, Kr, rtol=rtol, atol=0)
    np.testing.assert_allclose(G.P, Pr, rtol=max(rtol, 1e-12) if Kr.dtype == np.float64 else 1e-5, atol=0)
    np.testing.assert_allclose(np.asarray(G.kernel_degree).ravel(), Kr.sum(axis=1), rtol=1e-6)


@pytest.mark.parametrize("tag,src,mode", [("adj_sparse", "Adj", "adjacency"), ("aff_sparse", "A", "affinity")])
def test_precomputed_affinity_and_adjacency_sparse(tag, src, mode):
    z = load_golden("g11_exact_passthrough")
    G = _graph(sparse.csr_matrix(z[src]), precomputed=mode)
    assert type(G).__name__ == "TraditionalGraph" and sparse.issparse(G.K) and sparse.issparse(G.P)
    _same_sparse(G.K, golden_csr(z, "K_" + tag), 1e-14)
    _same_sparse(G.P, golden_csr(z, "P_" + tag), 1e-13)


@pytest.mark.parametrize("tag,kw", [
    ("K_default", {}),
    ("K_knn4", dict(knn=4)),
    ("K_bw_scalar", dict(bandwidth=4.5)),
    ("K_bw_vector", dict(bandwidth="vector", bandwidth_scale=1.25)),
])
def test_exact_graph_extension(tag, kw):
    """TraditionalGraph.build_kernel_to_data (graphs.py:1612-1678): float64 difference-form cdist on the device"""
    z = load_golden("g12_exact_extend")
    kw = dict(kw)
    if kw.get("bandwidth") == "vector":
        kw["bandwidth"] = z["bw_vector"]
    G = _graph(z["X"], knn=int(z["knn"]), decay=float(z["decay"]), graphtype="exact")
    K = G.build_kernel_to_data(z["Y"], **kw)
    Kr = z[tag]
    assert K.shape == Kr.shape and K.dtype == np.float64
    flip = (K == 0) != (Kr == 0)
    assert flip.sum() <= 2 and np.all(np.maximum(K, Kr)[flip] < 1e-4 * (1 + 1e-6))
    np.testing.assert_allclose(K[~flip], Kr[~flip], rtol=1e-9, atol=0)
    if tag == "K_default":
        T = G.extend_to_data(z["Y"])
        np.testing.assert_allclose(T[~flip], z["T_default"][~flip], rtol=1e-9, atol=0)
        G64 = _graph(z["X"].astype(np.float64), knn=int(z["knn"]), decay=float(z["decay"]), graphtype="exact")
        K64 = G64.build_kernel_to_data(z["Y"].astype(np.float64))
        m = (K64 == 0) == (z["K_f64"] == 0)
        assert (~m).sum() <= 2
        np.testing.assert_allclose(K64[m], z["K_f64"][m], rtol=1e-9, atol=0)
    with pytest.raises(ValueError, match="Cannot extend kernel on precomputed graph"):
        D = np.abs(np.random.default_rng(0).standard_normal((40, 40)))
        _graph(D + D.T, precomputed="distance", knn=3, decay=5).build_kernel_to_data(z["Y"][:, :40])


@pytest.mark.parametrize("tag,scale", [("K_bwvec", None), ("K_bwvec_scaled", 0.8)])
def test_knn_extension_with_a_bandwidth_per_row(tag, scale):
    """kNNGraph.build_kernel_to_data accepts one bandwidth per row of Y (graphs.py:819-982)"""
    z = load_golden("g13_knn_extend_bwvec")
    G = _graph(z["X"], knn=int(z["knn"]), decay=float(z["decay"]))
    kw = {} if scale is None else {"bandwidth_scale": scale}
    K = G.build_kernel_to_data(z["Y"], bandwidth=z["bw_vector"], **kw)
    _same_sparse(K, golden_csr(z, tag), 1e-12)
    with pytest.raises(ValueError):
        G.build_kernel_to_data(z["Y"], bandwidth=z["bw_vector"][:7])


def test_exact_graph_with_the_cosine_metric():
    """metrics beyond euclidean: scipy pdist on the host as in the reference (graphs.py:1552), the rest on the device"""
    z = load_golden("g14_exact_cosine")
    G = _graph(z["X"], knn=int(z["knn"]), decay=float(z["decay"]), graphtype="exact", distance="cosine")
    assert np.array_equal(G.K == 0, z["K"] == 0)
    np.testing.assert_allclose(G.K, z["K"], rtol=1e-9, atol=0)
    np.testing.assert_allclose(G.P, z["P"], rtol=1e-9, atol=0)


def test_landmark_operator_of_an_exact_graph():
    """TraditionalLandmarkGraph (reference graphs.py:1169-1246 on a dense kernel): the non-zeros of K go through the device's
    landmark products; transitions come back dense like the reference's"""
    z = load_golden("g15_exact_landmark")
    G = _graph(z["X"], knn=int(z["knn"]), decay=float(z["decay"]), graphtype="exact", n_landmark=int(z["n_landmark"]),
               random_landmarking=True, random_state=int(z["random_state"]))
    assert type(G).__name__ == "TraditionalLandmarkGraph"
    assert np.array_equal(G.clusters, z["clusters"])
    np.testing.assert_allclose(G.landmark_op, z["landmark_op"], rtol=1e-9, atol=1e-15)
    assert isinstance(G.transitions, np.ndarray) and G.transitions.shape == z["transitions"].shape
    np.testing.assert_allclose(G.transitions, z["transitions"], rtol=1e-9, atol=1e-15)


def test_pickle_round_trip_drops_the_device_context_and_rebuilds_lazily(tmp_path, capsys):
    """reference graphs are plain picklable objects (base.py:887-902 to_pickle); here the device context is left behind and
    rebuilt on first use: fetched results travel, device-side calls (extend_to_data) work again after loading"""
    import pickle

    from conftest import make_mix

    X = make_mix(3000, 20, 3)
    G = graphtools_amd.Graph(X, knn=8, decay=20, n_pca=None, verbose=True)
    out = capsys.readouterr().out
    assert "Calculated KNN search in" in out and "Calculated affinities in" in out     # the reference's task names (graphs.py:873-885)
    K, P = G.K.copy(), G.P.copy()
    blob = pickle.dumps(G)
    G2 = pickle.loads(blob)
    assert not hasattr(G2, "_hip_ctx")
    assert (G2.K != K).nnz == 0 and (G2.P != P).nnz == 0           # host results travelled
    Y = X[:50] + 0.01
    T1 = G.extend_to_data(Y)
    T2 = G2.extend_to_data(Y)                                         # rebinds the points, rebuilds on the device
    assert (T1 != T2).nnz == 0
    np.testing.assert_array_equal(np.asarray(G2.kernel_degree), np.asarray(G.kernel_degree))
    path = tmp_path / "g.pkl"
    G.to_pickle(str(path))
    with open(path, "rb") as f:
        G3 = pickle.load(f)
    assert (G3.K != K).nnz == 0
    # a graph that was never initialised pickles too and builds after loading
    G4 = pickle.loads(pickle.dumps(graphtools_amd.Graph(X, knn=8, decay=20, n_pca=None, initialize=False)))
    assert not hasattr(G4, "_kernel")
    assert (G4.K != K).nnz == 0
    quiet = graphtools_amd.Graph(X, knn=8, decay=20, n_pca=None, verbose=0)
    assert "Calculated" not in capsys.readouterr().out


def test_a_knn_kernel_is_built_on_construction_and_copied_to_the_host_when_asked_for():
    """initialize=True builds the kernel (base.py:77-83) - on the device; the host copy of a kNN kernel is made on the first access
    of K / P.  The warnings of the build come at construction as in the reference, the consumers that read the kernel where it is
    (landmark operator, diffuse) do not fetch it, a pickle made before any access holds K like the reference's."""
    import pickle
    import warnings

    from conftest import make_mix

    X = make_mix(4000, 16, 5)
    X[7] = X[3]         # a duplicate: the reference warns while it builds (graphs.py:887-914)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        G = graphtools_amd.Graph(X, knn=6, decay=15, n_pca=None, n_landmark=50, random_landmarking=True, random_state=3)
    assert any("zero distance between samples 3 and 7" in str(x.message) for x in w)
    assert not hasattr(G, "_kernel") and G.build_stats["stage_ms"]["rerank"] > 0        # built, not fetched
    op = G.landmark_op
    Z = G.diffuse(X[:, :3])
    assert not hasattr(G, "_kernel")
    G2 = pickle.loads(pickle.dumps(G))
    assert hasattr(G2, "_kernel") and hasattr(G, "_kernel")                              # (pickling fetched it)
    eager = graphtools_amd.Graph(X, knn=6, decay=15, n_pca=None, n_landmark=50, random_landmarking=True, random_state=3, verbose=0)
    K = eager.K
    assert (G.K != K).nnz == 0 and (G2.K != K).nnz == 0 and (G.P != eager.P).nnz == 0
    np.testing.assert_allclose(op, eager.landmark_op, rtol=1e-12, atol=1e-300)      # (float64 atomics: the summation order varies)
    np.testing.assert_array_equal(Z, eager.P.dot(X[:, :3]))


def test_host_copy_of_P_derived_on_the_way_equals_the_device_P():
    """gt_graph_fetch_kp: K and the structure cross the link, P = K / degree is formed by the copy threads on the host -
    it must equal the P the device wrote (base.py:645) bit for bit, with and without anisotropy, on a graph large enough
    for the pipelined copy (> 32 MB of values)"""
    from conftest import make_mix
    from graphtools_amd import _hip

    X = make_mix(60000, 32, 17)
    for aniso in (0.0, 0.5):
        c = _hip.Context(0)
        c.set_points(X)
        p, keep = c.make_params(15, 10, 1e-4, None, 1.0, None, "+", None, aniso)
        nnz, _ = c.graph_build(p)
        assert nnz * 8 > (32 << 20)
        kd, ki, kp, pd = c.graph_fetch_kp()
        kd2, ki2, kp2 = c.graph_fetch_csr(_hip.CSR_K)
        pd2, _, _ = c.graph_fetch_csr(_hip.CSR_P, structure=False)
        c.close()
        assert np.array_equal(kd, kd2) and np.array_equal(ki, ki2) and np.array_equal(kp, kp2)
        assert np.array_equal(pd, pd2), "host-derived P differs from the device's (anisotropy %g)" % aniso


def test_every_option_the_header_documents_is_accepted_and_nothing_else():
    """include/graphtools_amd.h OPTIONS list <-> gt_set_option on the device (the CPU half: test_abi.py)"""
    from graphtools_amd import _hip
    from test_abi import header_options

    values = {"knn_precision": "auto", "metric": "euclidean", "distance_dtype": "data", "query_order": "auto",
              "symmetrize_bin_shift": "9"}
    c = _hip.Context(0)
    try:
        for name in header_options():
            c.set_option(name, values.get(name, "0" if name == "dbg_select" else "1"))
        for stale in ("select_thr0", "symmetrize_fused", "select_sym_cold_split", "xcd_chunk", "rerank_waves_per_block",
                      "row_waves_per_block", "select_samp2_level", "select_samp_trig", "no_such_option"):
            with pytest.raises(Exception, match="unknown option"):
                c.set_option(stale, "1")
    finally:
        c.close()
