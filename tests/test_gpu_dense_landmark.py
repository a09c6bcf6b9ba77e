"""GPU: exact dense graph (TraditionalGraph) and landmark operator vs golden vectors / oracle."""
import warnings

import numpy as np
import pytest
from scipy import sparse

import graphtools_amd
import oracle
from conftest import golden_csr, load_golden, make_mix

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,src,kw,rtol", [
    ("data_t1e-4", "X", dict(thresh=1e-4), 1e-9),
    ("data_t0", "X", dict(thresh=0), 1e-9),
    ("d64_t1e-4", "D64", dict(thresh=1e-4, precomputed="distance"), 1e-9),
    ("d32_t1e-4", "D32", dict(thresh=1e-4, precomputed="distance"), 1e-5),
    ("d32_t0", "D32", dict(thresh=0, precomputed="distance"), 1e-5),
])
def test_exact_graph_matches_reference_vectors(tag, src, kw, rtol):
    z = load_golden("g6_exact")
    G = graphtools_amd.Graph(z[src], knn=int(z["knn"]), decay=float(z["decay"]), n_pca=None, graphtype="exact", **kw)
    assert type(G).__name__ == "TraditionalGraph"
    Kg, Pg = z["K_" + tag], z["P_" + tag]
    assert G.K.dtype == Kg.dtype and G.P.dtype == Pg.dtype
    # same sparsity pattern except entries within tolerance of the threshold
    thresh = kw["thresh"]
    flip = (G.K == 0) != (Kg == 0)
    assert np.all(np.abs(np.where(flip, np.maximum(G.K, Kg), thresh) - thresh) <= 4 * rtol * max(thresh, 1e-30) + 1e-30)
    m = ~flip
    # float32 results: values in the float32 subnormal range (< 1.2e-38) carry no relative precision
    atol = 1e-37 if Kg.dtype == np.float32 else 1e-300
    np.testing.assert_allclose(G.K[m], Kg[m], rtol=rtol, atol=atol)
    np.testing.assert_allclose(G.P[m], Pg[m], rtol=max(rtol, 1e-6), atol=atol)
    np.testing.assert_allclose(G.kernel_degree.ravel(), Kg.sum(axis=1), rtol=max(rtol, 1e-6))


@pytest.mark.parametrize("kw", [
    dict(kernel_symm="*"), dict(kernel_symm="mnn", theta=0.4), dict(kernel_symm=None), dict(anisotropy=0.5),
    dict(bandwidth=9.0, bandwidth_scale=1.2), dict(knn=3, decay=5),
])
def test_exact_graph_variants_vs_oracle(kw):
    X = make_mix(333, 30, 21)     # not a multiple of the tile size
    args = dict(knn=8, decay=12, thresh=1e-4)
    args.update(kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = graphtools_amd.Graph(X, n_pca=None, graphtype="exact", **args)
    K0, P0 = oracle.exact_graph(X, **args)
    np.testing.assert_allclose(G.K, K0, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(G.P, P0, rtol=1e-9, atol=1e-300)


def test_exact_graph_vector_bandwidth_precomputed():
    z = load_golden("g6_exact")
    D = z["D64"]
    bw = np.random.default_rng(5).uniform(8, 12, size=D.shape[0])
    G = graphtools_amd.Graph(D, precomputed="distance", bandwidth=bw, decay=10, knn=5, n_pca=None)
    K0, P0 = oracle.exact_graph(D, knn=5, decay=10, bandwidth=bw, precomputed="distance")
    np.testing.assert_allclose(G.K, K0, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(G.P, P0, rtol=1e-9, atol=1e-300)


def test_landmark_operator_matches_reference_vectors():
    z = load_golden("g7_landmark")
    X = z["X"]
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=int(z["n_landmark"]), random_landmarking=True,
                             random_state=int(z["random_state"]))
    assert type(G).__name__ == "kNNLandmarkGraph"
    assert np.array_equal(G.clusters, z["clusters"])
    np.testing.assert_allclose(G.landmark_op, z["landmark_op"], rtol=1e-9, atol=1e-15)
    T = golden_csr(z, "transitions")
    Tg = sparse.csr_matrix(G.transitions)
    Tg.sort_indices()
    assert np.array_equal(Tg.indptr, T.indptr) and np.array_equal(Tg.indices, T.indices)
    np.testing.assert_allclose(Tg.data, T.data, rtol=1e-9)
    np.testing.assert_allclose(G.landmark_op.sum(axis=1), 1.0, rtol=0, atol=1e-12)


def test_landmark_operator_given_spectral_labels():
    """the spectral front end is out of scope: its labels are an input to the device algebra"""
    z = load_golden("g7_landmark")
    G = graphtools_amd.Graph(z["X"], knn=15, decay=40, n_pca=None, n_landmark=int(z["n_landmark"]), random_state=42)
    G._clusters = z["spectral_clusters"].astype(np.int64)
    np.testing.assert_allclose(G.landmark_op, z["spectral_landmark_op"], rtol=1e-9, atol=1e-15)
    T = golden_csr(z, "spectral_transitions")
    assert abs(sparse.csr_matrix(G.transitions) - T).max() < 1e-12


def test_random_landmark_assignment_large_n_path():
    """n > 5000: scikit-learn euclidean_distances arithmetic (float32 rounding of the float64 GEMM form)"""
    z = load_golden("g7_landmark")
    X = z["big_X"]
    G = graphtools_amd.Graph(X, knn=5, decay=None, n_pca=None, n_landmark=64, random_landmarking=True, random_state=7)
    assert np.array_equal(G.clusters, z["big_clusters"])
    K = sparse.csr_matrix(G.K)
    op, tr = oracle.landmark_operator(K, z["big_clusters"])
    np.testing.assert_allclose(G.landmark_op, op, rtol=1e-9, atol=1e-15)


def test_transitions_of_long_rows_are_reproducible_and_ordered():
    """Kernel rows of more than 512 entries are aggregated by a workgroup over a dense accumulator (gt_landmark.hip
    aggregate_big_rows_kernel): every cluster's sum is formed by one thread in column order - what scipy's `K @ onehot` does
    (graphs.py:1232-1236) - so two builds give the same bits (LDS atomics, the first form, did not) and the sums equal the
    host's column-ordered sums exactly."""
    X = make_mix(3000, 12, 6)
    args = dict(knn=400, decay=None, n_pca=None, n_landmark=60, random_landmarking=True, random_state=5, verbose=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G1 = graphtools_amd.Graph(X, **args)
        G2 = graphtools_amd.Graph(X, **args)
        T1, T2 = sparse.csr_matrix(G1.transitions), sparse.csr_matrix(G2.transitions)
    K = sparse.csr_matrix(G1.K)
    assert np.diff(K.indptr).max() > 512
    assert (T1 != T2).nnz == 0
    # column-ordered sums on the host: row by row, cluster by cluster, in the order of the CSR columns
    cl = np.asarray(G1.clusters)
    _, inv = np.unique(cl, return_inverse=True)
    rows = np.flatnonzero(np.diff(K.indptr) > 512)[:20]
    for i in rows:
        cols, vals = K.indices[K.indptr[i]:K.indptr[i + 1]], K.data[K.indptr[i]:K.indptr[i + 1]]
        assert np.all(np.diff(cols) > 0)
        acc = np.zeros(int(inv.max()) + 1)
        for j, v in zip(cols, vals):
            acc[inv[j]] += v
        want = acc / acc.sum()
        got = np.asarray(T1[i].todense()).ravel()
        np.testing.assert_allclose(got, want, rtol=4e-16, atol=0)      # (the normalising sum is formed in another order: an ulp)
        assert np.array_equal(got != 0, want != 0)


@pytest.mark.parametrize("where", ["host rows", "device rows"])
def test_first_nearest_on_the_device_follows_argmin(where):
    """gt_knn_first_nearest (random landmarking at scale, graphs.py:1200-1213): the label formed on the device is numpy's
    argmin over the distances gt_knn_search returns - the FIRST index among the nearest that tie (landmark rows that are copies
    of each other tie for every row; their own rows tie at distance 0)."""
    from graphtools_amd import _hip

    rng = np.random.default_rng(11)
    X = make_mix(40000, 24, 4)      # (from 32768 query rows against fewer points the bound points seed the queries' thresholds)
    lm = rng.choice(X.shape[0], 300, replace=False)
    L = X[lm].copy()
    L[250:] = L[:50]       # fifty landmarks twice: rows near one of them see a tie of the two nearest
    c = _hip.Context(0)
    c.set_points(L)
    c.set_option("query_order", "0")
    dist, idx, _ = c.knn_search(4, Y=X)
    c.set_option("query_order", "1")
    unseeded_ms = c.stage_ms("query_order")
    want = np.where(dist == dist[:, :1], idx, np.iinfo(np.int64).max).min(axis=1)
    if where == "host rows":
        got = c.knn_first_nearest(4, Y=X)
    else:
        g = _hip.Context(0)
        g.set_points(X)
        got = c.knn_first_nearest(4, y_dev_ptr=g.points_device(0), m=X.shape[0])
        g.close()
    assert c.stage_ms("query_order") > 10 * unseeded_ms      # (the seeded search ran: the stage is not an empty span)
    d2, i2, _ = c.knn_search(4, Y=X)
    assert np.array_equal(d2, dist) and np.array_equal(i2, idx)      # ... and returns the tables of the unseeded one
    c.close()
    assert got.dtype == np.int64 and np.array_equal(got, want)
    assert (dist[:, 0] == dist[:, 1]).sum() > 500 and got.max() < 250     # (ties happened; the copies never win)
    assert np.array_equal(got[lm[:250]], np.arange(250))                   # a landmark is its own nearest


@pytest.mark.parametrize("n", [1022, 1024, 257])
@pytest.mark.parametrize("kw", [dict(kernel_symm="+"), dict(kernel_symm="*"), dict(kernel_symm="mnn", theta=0.3),
                                dict(kernel_symm="+", thresh=0)])
def test_exact_graph_from_float32_distances_vs_oracle(n, kw):
    """float32 precomputed distances: results stay float32 like numpy's; n % 4 == 0 takes the 16-byte-vector tile path,
    the other sizes the scalar one; thresh = 0 exercises the exp-underflow cut."""
    X = make_mix(n, 12, 9)
    D = np.sqrt(((X[:, None, :].astype(np.float64) - X[None, :, :]) ** 2).sum(-1)).astype(np.float32)
    args = dict(knn=6, decay=8, thresh=1e-4)
    args.update(kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = graphtools_amd.Graph(D, precomputed="distance", n_pca=None, **args)
        K0, P0 = oracle.exact_graph(D, precomputed="distance", **args)
    assert G.K.dtype == np.float32
    thresh = args["thresh"]
    # entries within float32 rounding of the threshold may fall either way; with thresh = 0 the "threshold" is the
    # float32 underflow of exp: subnormal results (< 1.2e-38) carry no precision and may be flushed to zero
    flip = (G.K == 0) != (K0 == 0)
    assert np.all(np.abs(np.maximum(G.K, K0)[flip] - thresh) <= 1e-5 * max(thresh, 1e-30) + 1e-37)
    assert flip.sum() <= (4 if thresh > 0 else 1e-4 * n * n)
    m = ~flip
    np.testing.assert_allclose(G.K[m], K0[m], rtol=1e-5, atol=1e-37)
    np.testing.assert_allclose(G.P[m], P0[m], rtol=2e-5, atol=1e-37)


def test_landmark_graph_extend_and_interpolate_match_reference():
    """kNNLandmarkGraph.extend_to_data(Y) -> [m, n_landmark] cluster-aggregated, l1-normalised transitions and
    interpolate(transform) with no Y -> the graph's own landmark transitions (reference graphs.py:1247-1317; the
    landmark versions must win over kNNGraph's in the MRO)."""
    import warnings

    z = load_golden("g7b_landmark_extend")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = graphtools_amd.Graph(z["X"], knn=15, decay=40, n_pca=None, n_landmark=int(z["n_landmark"]),
                                 random_landmarking=True, random_state=int(z["random_state"]), verbose=0)
    assert type(G).__name__ == "kNNLandmarkGraph"
    assert np.array_equal(np.asarray(G.clusters), z["clusters"])
    pnm = G.extend_to_data(z["Y"])
    assert pnm.shape == (z["Y"].shape[0], int(z["n_landmark"]))
    np.testing.assert_allclose(np.asarray(pnm.todense()), z["extend_pnm"], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(G.interpolate(z["transform"]), z["interp_self"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(G.interpolate(z["transform"], Y=z["Y"]), z["interp_Y"], rtol=1e-9, atol=1e-12)
    with pytest.raises(Exception):
        G.interpolate(z["transform"][:5], Y=z["Y"])       # shape mismatch surfaces like in the reference (dot fails)


def test_exact_graph_names_the_duplicate_pairs():
    """TraditionalGraph from data: identical rows are reported pair by pair like the reference (graphs.py:1553-1574)"""
    X = make_mix(300, 12, 3).astype(np.float64)
    X[17] = X[250]
    X[40] = X[5]
    X[41] = X[5]
    with pytest.warns(RuntimeWarning, match="Detected zero distance between samples 5 and 40, 5 and 41, 17 and 250, 40 and 41. Consider"):
        graphtools_amd.Graph(X, knn=5, decay=10, graphtype="exact", n_pca=None, verbose=0).K


def _compare_dense_to_oracle(G, K0, P0, thresh, rtol):
    assert G.K.shape == K0.shape and G.K.dtype == np.float64 and not sparse.issparse(G.K)
    flip = (G.K == 0) != (K0 == 0)
    assert np.all(np.abs(np.where(flip, np.maximum(G.K, K0), thresh) - thresh) <= 1e-6 * thresh)
    assert flip.sum() <= 4
    m = ~flip
    np.testing.assert_allclose(G.K[m], K0[m], rtol=rtol, atol=1e-300)
    np.testing.assert_allclose(G.P[m], P0[m], rtol=max(rtol, 1e-7), atol=1e-300)
    np.testing.assert_allclose(G.kernel_degree.ravel(), K0.sum(axis=1), rtol=1e-7)


@pytest.mark.parametrize("kw", [
    dict(), dict(kernel_symm="mnn", theta=0.3), dict(anisotropy=1.0), dict(bandwidth=7.5, bandwidth_scale=0.9),
    dict(knn=4, decay=2, thresh=1e-3), dict(dtype=np.float32), dict(kernel_symm="*"), dict(kernel_symm=None, anisotropy=0.5),
])
def test_exact_graph_from_points_through_the_neighbour_search(kw, monkeypatch):
    """above _NEIGHBOUR_ROUTE_MIN points the exact graph is built by the kNN path's radius search and written out densely
    (graphs.py:1546-1609: everything below thresh is exactly 0); tolerance 1e-9 relative on K like the all-pairs path (measured:
    2e-12 - the search forms |x|^2 - 2 x.y + |y|^2 in float64 on centred points where pdist sums squared differences)"""
    kw = dict(kw)
    dtype = kw.pop("dtype", np.float64)
    X = (make_mix(4500, 24, 33) + 40.0).astype(dtype)    # (far from the origin: the route centres the points)
    args = dict(knn=6, decay=20, thresh=1e-4)
    args.update(kw)
    calls = []
    from graphtools_amd import _hip
    real = _hip.Context.graph_to_dense
    monkeypatch.setattr(_hip.Context, "graph_to_dense", lambda self, *a, **k: (calls.append(a[0]), real(self, *a, **k))[1])
    G = graphtools_amd.Graph(X, n_pca=None, graphtype="exact", **args)
    G.K
    assert calls == [_hip.CSR_K, _hip.CSR_P]
    K0, P0 = oracle.exact_graph(X.astype(np.float64), **args)
    _compare_dense_to_oracle(G, K0, P0, args["thresh"], 1e-9)
    np.testing.assert_allclose(G.diff_op, G.P)


def test_exact_graph_from_points_keeps_the_all_pairs_path_for_duplicates_and_small_sets(monkeypatch):
    from graphtools_amd import _hip
    calls = []
    real = _hip.Context.graph_to_dense
    monkeypatch.setattr(_hip.Context, "graph_to_dense", lambda self, *a, **k: (calls.append(a[0]), real(self, *a, **k))[1])
    X = make_mix(4400, 12, 5)
    X[17] = X[4100]
    with pytest.warns(RuntimeWarning, match="Detected zero distance between samples 17 and 4100"):
        G = graphtools_amd.Graph(X, n_pca=None, graphtype="exact", knn=5, decay=10)
        G.K
    assert calls == []
    K0, P0 = oracle.exact_graph(X, knn=5, decay=10)
    np.testing.assert_allclose(G.K, K0, rtol=1e-9, atol=1e-300)
    G2 = graphtools_amd.Graph(X[:500], n_pca=None, graphtype="exact", knn=5, decay=10, thresh=0)
    G2.K
    assert calls == []


def test_exact_graph_from_points_hands_what_the_search_cannot_hold_to_the_all_pairs_path(monkeypatch):
    """(round-3 advisor finding) a threshold below float64's eps is NOT clamped by TraditionalGraph (graphs.py:628-629 is a
    kNNGraph rule): entries in [thresh, eps) stay; and inputs the search refuses (more than 2048 features with a caller-given
    bandwidth: GT_E_LIMIT) are built by the all-pairs kernel instead of raising"""
    from graphtools_amd import _hip
    calls = []
    real = _hip.Context.graph_to_dense
    monkeypatch.setattr(_hip.Context, "graph_to_dense", lambda self, *a, **k: (calls.append(a[0]), real(self, *a, **k))[1])
    X = make_mix(4200, 16, 21).astype(np.float64)
    G = graphtools_amd.Graph(X, n_pca=None, graphtype="exact", knn=5, decay=10, thresh=1e-20)
    G.K
    assert calls == []                       # the all-pairs path
    K0, P0 = oracle.exact_graph(X, knn=5, decay=10, thresh=1e-20)
    tiny = (K0 > 0) & (K0 < np.finfo(float).eps)
    assert tiny.sum() > 0, "the case should hold entries between thresh and eps"
    np.testing.assert_allclose(G.K, K0, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(G.P, P0, rtol=1e-7, atol=1e-300)
    # wide data + explicit bandwidth: the neighbour search refuses, the build still succeeds
    # (1300 rows with the route's minimum lowered for the test: the oracle's pdist over 2100 columns took 80 s at 4100 rows -
    #  a sixth of the suite - and what is under test is the refusal, which does not depend on the row count)
    from graphtools_amd.graphs import TraditionalGraph
    monkeypatch.setattr(TraditionalGraph, "_NEIGHBOUR_ROUTE_MIN", 1024)
    refused = []
    real_build = _hip.Context.graph_build

    def build_spy(self, *a, **k):
        try:
            return real_build(self, *a, **k)
        except _hip.HipError as e:
            refused.append(str(e))
            raise
    real_points = _hip.Context.set_points

    def points_spy(self, *a, **k):
        try:
            return real_points(self, *a, **k)
        except _hip.HipError as e:
            refused.append(str(e))
            raise
    monkeypatch.setattr(_hip.Context, "graph_build", build_spy)
    monkeypatch.setattr(_hip.Context, "set_points", points_spy)
    rng = np.random.default_rng(3)
    Xw = (rng.standard_normal((1300, 5)) @ rng.standard_normal((5, 2100))).astype(np.float64)
    Gw = graphtools_amd.Graph(Xw, n_pca=None, graphtype="exact", knn=5, decay=10, bandwidth=60.0)
    Gw.K
    assert refused, "the neighbour route should have been tried and refused (more than 2048 features)"
    Kw0, Pw0 = oracle.exact_graph(Xw, knn=5, decay=10, bandwidth=60.0)
    flip = (Gw.K == 0) != (Kw0 == 0)
    assert flip.sum() <= 4
    np.testing.assert_allclose(Gw.K[~flip], Kw0[~flip], rtol=1e-9, atol=1e-300)


def test_distance_dtype_option_gives_float32_points_float64_distances():
    """'distance_dtype' = 'float64': a float32 point set's distances come out of the float64 keys unrounded (scipy's pdist
    semantics) - the same graph as the float64 copy of the points gives"""
    from graphtools_amd import _hip
    X32 = make_mix(6000, 16, 3).astype(np.float32)
    out = []
    for X, opt in ((X32, "float64"), (X32.astype(np.float64), "data"), (X32, "data")):
        ctx = _hip.Context(0)
        ctx.set_option("distance_dtype", opt)
        ctx.set_points(X)
        params, keep = _hip.Context.make_params(8, 30, 1e-4, None, 1.0, None, "+", None, 0)
        ctx.graph_build(params)
        d, i, p = ctx.graph_fetch_csr(_hip.CSR_K)
        out.append(sparse.csr_matrix((d, i, p), shape=(6000, 6000)))
        ctx.close()
    a, b, c = out
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    np.testing.assert_allclose(a.data, b.data, rtol=1e-11)
    assert abs(a - c).max() > 1e-9       # (the float32 rounding of the distances is visible without the option)


def test_dense_copy_on_the_device_in_float32():
    """gt_graph_to_dense into a caller's device buffer (float32): the sparse K of the same build, densified"""
    import torch
    from graphtools_amd import _hip

    torch.cuda.init()
    X = make_mix(5000, 16, 9).astype(np.float32)
    ctx = _hip.Context(0)
    try:
        ctx.set_points(X)
        params, keep = _hip.Context.make_params(7, 30, 1e-4, None, 1.0, None, "+", None, 0)
        ctx.graph_build(params)
        d, i, p = ctx.graph_fetch_csr(_hip.CSR_K)
        K = sparse.csr_matrix((d, i, p), shape=(5000, 5000))
        out = torch.full((5000, 5000), -1.0, dtype=torch.float32, device="cuda:0")
        ctx.graph_to_dense(_hip.CSR_K, 5000, np.float32, out_device=out)
        assert np.array_equal(out.cpu().numpy(), K.toarray().astype(np.float32))
        pd_, _, _ = ctx.graph_fetch_csr(_hip.CSR_P, structure=False)
        ctx.graph_to_dense(_hip.CSR_P, 5000, np.float32, out_device=out)       # (the same buffer: rows are zero-filled first)
        P = sparse.csr_matrix((pd_, i, p), shape=(5000, 5000))
        assert np.array_equal(out.cpu().numpy(), P.toarray().astype(np.float32))
    finally:
        ctx.close()


@pytest.mark.parametrize("n,symm", [(2500, "+"), (1028, "*"), (2052, "mnn")])
def test_one_pass_bandwidth_and_fused_row_sums_equal_the_separate_passes(n, symm):
    """round 4: the exact graph from a float32 distance matrix reads the matrix ONCE for the bandwidths
    (dense_bandwidth_1pass_kernel) and accumulates the row sums in the tile-pair kernel (no pass of their own).  Against the
    round-3 kernels (options) K is bit-identical; the row sums differ only by their summation order; against the oracle
    (graphs.py:1583-1609, base.py:645) the usual tolerances hold.  Sizes with ragged 64 x 64 edge tiles."""
    from graphtools_amd import _hip
    rng = np.random.default_rng(n)
    X = make_mix(n, 12, 3).astype(np.float64)
    D = np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)).astype(np.float32) if n <= 1100 else None
    if D is None:
        from scipy.spatial.distance import pdist, squareform
        D = squareform(pdist(X)).astype(np.float32)
    D[rng.integers(0, n, 50), rng.integers(0, n, 50)] += 0.0     # (no-op: keeps the generator in the signature of the case)
    theta = 0.3 if symm == "mnn" else None
    res = {}
    # (the round-3 two-pass bandwidth kernel was removed in round 5: the comparison is the separate row-sum pass; the bandwidths are
    #  pinned by the oracle below)
    for tag, opts in (("new", {}), ("old", {"dense_fused_rowsum": "0"})):
        c = _hip.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        K, P, flags = c.dense_graph_build(D, "distance", 7, 12.0, 1e-4, None, 1.0, symm, theta, 0.0, want_P=True)
        deg = c.dense_fetch_vec(_hip.VEC_DEGREE, n)
        bw = c.dense_fetch_vec(_hip.VEC_BANDWIDTH, n) if hasattr(_hip, "VEC_BANDWIDTH") else None
        c.close()
        res[tag] = (K, P, deg, bw)
    assert res["new"][0].dtype == np.float32
    assert np.array_equal(res["new"][0], res["old"][0])                      # K: bit-identical (the bandwidths are)
    if res["new"][3] is not None:
        assert np.array_equal(res["new"][3], res["old"][3])
    np.testing.assert_allclose(res["new"][2], res["old"][2], rtol=1e-12)    # degrees: float64 sums in another order
    np.testing.assert_allclose(res["new"][1], res["old"][1], rtol=2e-7, atol=0)   # P: float32 K / float32(sum)
    K0, P0 = oracle.exact_graph(D, knn=7, decay=12.0, thresh=1e-4, precomputed="distance", kernel_symm=symm, theta=theta)
    m = (res["new"][0] == 0) == (K0 == 0)
    assert (~m).sum() <= 4
    np.testing.assert_allclose(res["new"][0][m], K0[m], rtol=1e-5, atol=0)
    np.testing.assert_allclose(res["new"][1][m], P0[m], rtol=1e-5, atol=0)


def test_operator_alone_in_place_never_stores_K():
    """BASELINE config 4 as bench.py runs it: a device-resident float32 distance matrix becomes diff_op IN PLACE (K and P together
    would not fit the HBM at N = 2e5; graphs.py:1583-1609 + base.py:645) - K in place + an in-place normalisation (below 16 384
    rows, where the tile pairs serve; the variant that never stored K - two tile passes - was measured slower and removed in round 5).
    Equal to the P of the ordinary build; the degrees are K's row sums."""
    import ctypes

    import torch

    from graphtools_amd import _hip

    n = 2116                      # ragged edge tiles, a multiple of 4
    X = make_mix(n, 10, 8).astype(np.float64)
    from scipy.spatial.distance import pdist, squareform
    D = squareform(pdist(X)).astype(np.float32)
    c = _hip.Context(0)
    K, P, flags = c.dense_graph_build(D, "distance", 6, 15.0, 1e-4, None, 1.0, "+", None, 0.0, want_P=True)
    deg = c.dense_fetch_vec(_hip.VEC_DEGREE, n)
    c.close()
    for _ in range(1):
        Dd = torch.from_numpy(D).cuda()
        c = _hip.Context(0)
        fl = ctypes.c_uint32(0)
        rc = c.lib.gt_dense_graph_build(c.h, ctypes.c_void_p(Dd.data_ptr()), n, 0, 0, 1, 1, 6, 15.0, 1e-4, None, 0, 1.0, _hip.SYMM["+"],
                                        1.0, 0.0, 1, None, ctypes.c_void_p(Dd.data_ptr()), 1, ctypes.byref(fl))
        c._check(rc, "gt_dense_graph_build")
        c.sync()
        P2 = Dd.cpu().numpy()
        deg2 = c.dense_fetch_vec(_hip.VEC_DEGREE, n)
        c.close()
        assert np.array_equal(P2 == 0, P == 0)
        np.testing.assert_allclose(P2, P, rtol=3e-7, atol=0)
        np.testing.assert_allclose(deg2, deg, rtol=1e-12)
        np.testing.assert_allclose(P2.astype(np.float64).sum(axis=1), 1.0, rtol=1e-5)


@pytest.mark.parametrize("case", ["sparse", "asymmetric", "heavy rows", "some heavy rows", "not sparse", "zero diagonal"])
def test_row_streaming_form_equals_the_tile_pairs(case):
    """round 4: float32 distances under the '+' rule (graphs.py:1583-1609, base.py:557-561, 645): the matrix is read and written
    as whole rows, the transposed half of the thresholded kernel travels as a list (dense_rows_scan / dense_rows_write; default
    from 16384 rows, forced here); the write pass does not read the matrix: zeros are streamed, the listed entries placed.  Against the tile-pair kernels: K bit-identical - also for a matrix that is NOT symmetric,
    for rows with more non-zeros than the LDS list holds (appended directly) and for a kernel that is not sparse at all (the list
    overflows: the tile-pair form takes over) -, degrees to the summation order, P = K / float32(sum).  In place, P alone: equal
    to the P of the ordinary build; the matrix is consumed."""
    import ctypes

    import torch

    from graphtools_amd import _hip
    from scipy.spatial.distance import pdist, squareform

    n, decay, knn = 2116, 15.0, 6
    X = make_mix(n, 10, 8).astype(np.float64)
    if case == "heavy rows":
        knn = 1200      # every row keeps what lies within its 1200-th neighbour's distance: more than the LDS list holds
    if case == "some heavy rows":
        X[:1100] = X[0]     # 1100 copies of one point: their rows keep 1100 entries (more than the LDS list holds), the others a few
    D = squareform(pdist(X)).astype(np.float32)
    if case == "asymmetric":
        D = (D * (1.0 + 0.05 * np.random.default_rng(2).random((n, n)))).astype(np.float32)
        np.fill_diagonal(D, 0.0)
    if case == "zero diagonal":
        D[5, 5] = 1e3        # a "distance" of a point to itself beyond every radius: K_55 = 0 (base.py:553 warns)
    if case == "not sparse":
        decay, knn = 1.0, 200    # exp(-d / bw) with a wide bandwidth: nearly every entry survives 1e-4
    res = {}
    cap = {"dense_rows_cap": "100000"} if case == "not sparse" else {}      # (the list overflows: the tile pairs take over)
    for tag, opts in (("rows", dict(dense_rows="1", **cap)), ("rows, scan of its own", dict(dense_rows="1", dense_rows_fused="0", **cap)),
                      ("tiles", {"dense_rows": "0"})):
        c = _hip.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        K, P, flags = c.dense_graph_build(D, "distance", knn, decay, 1e-4, None, 1.0, "+", None, 0.0, want_P=True)
        deg = c.dense_fetch_vec(_hip.VEC_DEGREE, n)
        st = (c.stage_launches("dense_rows_scan"), c.stage_launches("dense_kernel"), c.stage_launches("dense_rows_listed"),
              c.stage_launches("dense_rows_placed"))
        c.close()
        res[tag] = (K, P, deg, flags, st)
    assert res["rows"][4][0] == 1 and res["tiles"][4][0] <= 0                # (the row-streaming form ran / did not)
    # (its list came out of the one-pass bandwidth kernel - which serves knn + 1 <= 256 - / a scan of its own)
    assert (res["rows"][4][2] == 1) == (knn + 1 <= 256) and res["rows, scan of its own"][4][2] <= 0
    assert np.array_equal(res["rows, scan of its own"][0], res["rows"][0]) and np.array_equal(res["rows, scan of its own"][1], res["rows"][1])
    assert res["rows, scan of its own"][3] == res["rows"][3]
    np.testing.assert_allclose(res["rows, scan of its own"][2], res["rows"][2], rtol=1e-12)
    # the write pass places the listed entries over streamed zeros (rows whose entries are not in one piece - "heavy rows" - are
    # read again, one by one; the variant that read EVERY row again was removed in round 5)
    assert (res["rows"][4][3] == 1) == (case != "not sparse")
    if case == "not sparse":
        assert (res["rows"][0] != 0).sum() > 100000 and res["rows"][4][1] == 1
    assert np.array_equal(res["rows"][0], res["tiles"][0]), "K differs"
    assert res["rows"][3] == res["tiles"][3], "flags differ"
    assert bool(res["rows"][3] & _hip.FLAG_ZERO_DIAGONAL) == (case == "zero diagonal")
    np.testing.assert_allclose(res["rows"][2], res["tiles"][2], rtol=1e-12)
    np.testing.assert_allclose(res["rows"][1], res["tiles"][1], rtol=2e-7, atol=0)
    if case == "heavy rows":
        assert (res["rows"][0] != 0).sum(axis=1).min() > 1024
    if case == "some heavy rows":
        nzr = (res["rows"][0] != 0).sum(axis=1)
        assert nzr[:1100].min() > 1024 and np.median(nzr[1100:]) < 200      # (K is symmetric: the copies' neighbours receive 1100 entries)
    # in place, the operator alone (BASELINE config 4 as bench.py runs it)
    Dd = torch.from_numpy(D).cuda()
    c = _hip.Context(0)
    c.set_option("dense_rows", "1")
    for k, v in cap.items():
        c.set_option(k, v)
    fl = ctypes.c_uint32(0)
    rc = c.lib.gt_dense_graph_build(c.h, ctypes.c_void_p(Dd.data_ptr()), n, 0, 0, 1, 1, knn, decay, 1e-4, None, 0, 1.0, _hip.SYMM["+"],
                                    1.0, 0.0, 1, None, ctypes.c_void_p(Dd.data_ptr()), 1, ctypes.byref(fl))
    c._check(rc, "gt_dense_graph_build")
    c.sync()
    P2 = Dd.cpu().numpy()
    deg2 = c.dense_fetch_vec(_hip.VEC_DEGREE, n)
    c.close()
    assert np.array_equal(P2 == 0, res["tiles"][1] == 0)
    np.testing.assert_allclose(P2, res["tiles"][1], rtol=3e-7, atol=0)
    np.testing.assert_allclose(deg2, res["tiles"][2], rtol=1e-12)
    assert fl.value == res["tiles"][3]
