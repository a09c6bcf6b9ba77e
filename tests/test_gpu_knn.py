"""GPU: brute-force kNN through the C ABI vs the oracle / golden vectors.

Bar: neighbour indices bit-exact (row argsort order); distances exact for float32 data (they carry
scikit-learn's float32 rounding) and within a few float64 ulps for float64 data."""
import ctypes

import numpy as np
import pytest

import oracle
from conftest import load_golden, make_gauss, make_manifold, make_mix

pytestmark = pytest.mark.gpu


def _check(ctx, X, k, Y=None):
    ctx.set_points(X)
    d, i, flags = ctx.knn_search(k, Y=Y)
    d0, i0 = oracle.kneighbors(X, Y, k)
    assert np.array_equal(i, i0), "kNN indices differ in %d rows" % int((i != i0).any(axis=1).sum())
    first = 1 if Y is None else 0   # column 0 of a self query is the self distance (rounding noise in sklearn)
    if X.dtype == np.float32:
        assert np.array_equal(d[:, first:], d0[:, first:])
    else:
        np.testing.assert_allclose(d[:, first:], d0[:, first:], rtol=1e-12, atol=0)
    if Y is None:
        assert np.all(d[:, 0] <= 1e-5 * (1 + np.abs(X).max()))
    return flags


@pytest.mark.parametrize("name", ["g3_mix_f32", "g4_gauss_f32", "g5_manifold_f32", "g2b_mix_binary"])
def test_knn_matches_reference_vectors(hip_ctx, name):
    z = load_golden(name)
    X, k = z["X"], int(z["search_k"])
    hip_ctx.set_points(X)
    d, i, _ = hip_ctx.knn_search(k)
    assert np.array_equal(i, z["knn_idx"])
    assert np.array_equal(d[:, 1:], z["knn_dist"].astype(np.float64)[:, 1:])


def test_knn_digits_float64_with_ties(hip_ctx):
    z = load_golden("g1_digits_decay40")
    X, k = z["X"], int(z["search_k"])
    hip_ctx.set_points(X)
    d, i, _ = hip_ctx.knn_search(k)
    assert np.array_equal(d, z["knn_dist"])          # integer data: float64 arithmetic is exact
    d0, i0 = oracle.kneighbors(X, None, k)
    assert np.array_equal(i, i0)                     # ties broken by index, like the oracle
    gi = z["knn_idx"]
    for r in range(0, X.shape[0], 13):               # vs scikit-learn: equal as sets below the last tie group
        keep = d[r] < d[r, -1]
        assert set(i[r, keep]) == set(gi[r, keep])


@pytest.mark.parametrize("n,d,k,maker,seed", [
    (1024, 50, 96, make_mix, 0), (5000, 64, 96, make_mix, 1), (3000, 64, 96, make_gauss, 1),
    (3000, 100, 96, make_mix, 2), (2000, 20, 30, make_mix, 3), (2500, 8, 16, make_gauss, 4),
    (4000, 64, 300, make_mix, 6), (3000, 128, 50, make_manifold, 7), (1000, 3, 10, make_gauss, 8),
])
def test_knn_random_inputs(hip_ctx, n, d, k, maker, seed):
    _check(hip_ctx, maker(n, d, seed), k)


def test_knn_float64_input(hip_ctx):
    _check(hip_ctx, make_mix(777, 50, 4, np.float64), 66)


def test_knn_external_queries(hip_ctx):
    X = make_mix(3000, 40, 9)
    Y = make_mix(500, 40, 10)
    _check(hip_ctx, X, 25, Y=Y)


def test_knn_small_and_ragged_sizes(hip_ctx):
    for n in (2, 5, 63, 64, 65, 127, 129, 257, 300):
        X = make_gauss(n, 7, n)
        _check(hip_ctx, X, min(n, 20))
    # k == n: every point is a neighbour
    X = make_gauss(50, 5, 1)
    _check(hip_ctx, X, 50)


def test_knn_row_block(hip_ctx):
    X = make_mix(4000, 32, 11)
    hip_ctx.set_points(X)
    d, i, _ = hip_ctx.knn_search(20, rows=(1000, 1700))
    d0, i0 = oracle.kneighbors(X, X[1000:1700], 20)
    assert np.array_equal(i, i0)


def test_knn_duplicates_flag(hip_ctx):
    X = make_mix(600, 16, 12)
    X[17] = X[400]
    hip_ctx.set_points(X)
    d, i, flags = hip_ctx.knn_search(10)
    assert flags & 1
    assert d[17, 1] == 0 and d[400, 1] == 0


def test_forced_exact_fallback(hip_ctx):
    """Massive exact ties (points on a tiny integer lattice): the candidate table cannot be proven complete,
    the exhaustive float64 fallback must take over and still return the exact (d2, index) order."""
    rng = np.random.default_rng(3)
    X = rng.integers(0, 2, size=(1500, 10)).astype(np.float32)   # 1024 distinct points, many duplicates
    hip_ctx.set_option("select_nt8_max_need", 112)   # the 128-entry list budget (512 entries would prove this table)
    hip_ctx.set_points(X)
    d, i, flags = hip_ctx.knn_search(96)
    assert flags & 4, "expected the exact fallback to trigger"
    d0, i0 = oracle.kneighbors(X, None, 96)
    assert np.array_equal(d, d0)
    assert np.array_equal(i, i0)


def test_device_primitives(hip_ctx):
    """wave-level sorting networks and the MFMA result layout, in isolation"""
    lib = hip_ctx.lib
    if not hasattr(lib, "gt_dbg_mfma"):
        pytest.skip("library built without the gt_dbg_* hooks (GT_BUILD_DEBUG_HOOKS=0)")
    rng = np.random.default_rng(0)
    a = rng.standard_normal((32, 8)).astype(np.float32)
    bt = rng.standard_normal((32, 8)).astype(np.float32)   # asymmetric on purpose (transpose-detecting)
    c = np.zeros((32, 32), dtype=np.float32)
    lib.gt_dbg_mfma.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
    assert lib.gt_dbg_mfma(hip_ctx.h, a.ctypes.data, bt.ctypes.data, 8, c.ctypes.data) == 0
    np.testing.assert_allclose(c, a @ bt.T, rtol=0, atol=1e-5)
    lib.gt_dbg_sort_desc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for nt in (1, 2, 8, 32):
        for n in (1, 63, 64 * nt - 5, 64 * nt):
            n = max(1, min(n, 64 * nt))
            keys = rng.integers(1, 2**63, size=n, dtype=np.uint64)
            out = np.zeros(64 * nt, dtype=np.uint64)
            assert lib.gt_dbg_sort_desc(hip_ctx.h, keys.ctypes.data, n, nt, out.ctypes.data) == 0
            exp = np.zeros(64 * nt, dtype=np.uint64)
            exp[:n] = np.sort(keys)[::-1]
            assert np.array_equal(out, exp)


def test_tie_heavy_lattice_collected_fallback(hip_ctx):
    """Integer lattice: squared distances are small integers, so the k-th neighbour ties with many others and
    the candidate table cannot be proven complete for most rows.  They are repaired by the radius-mode
    collection + exact (key, index) selection at MFMA speed; results must equal the oracle's (index order on ties)."""
    rng = np.random.default_rng(5)
    X = rng.integers(0, 4, size=(12000, 12)).astype(np.float32)
    hip_ctx.set_points(X)
    d, i, flags = hip_ctx.knn_search(100)
    assert flags & 4
    d0, i0 = oracle.kneighbors(X, None, 100)
    assert np.array_equal(d, d0)
    assert np.array_equal(i, i0)
    # and the graph on top of it
    p, keep = hip_ctx.make_params(60, 10, 1e-4, None, 1.0, None, "+", None, 0)
    import warnings
    from scipy import sparse
    nnz, fl = hip_ctx.graph_build(p)
    Kd, Ki, Kp = hip_ctx.graph_fetch_csr(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        K0, P0 = oracle.knn_graph(X, knn=60, decay=10)
    K0 = sparse.csr_matrix(K0); K0.sort_indices()
    assert np.array_equal(Kp, K0.indptr) and np.array_equal(Ki, K0.indices)
    np.testing.assert_allclose(Kd, K0.data, rtol=1e-9)


@pytest.mark.parametrize("prec", ["f16x1", "f16", "f32", "auto"])
@pytest.mark.parametrize("maker,n,d,k", [(make_mix, 20000, 64, 16), (make_manifold, 12000, 50, 30), (make_gauss, 9000, 24, 96)])
def test_knn_every_candidate_arithmetic_is_exact(prec, maker, n, d, k):
    """The arithmetic of the candidate pass only changes speed: single-chain float16 (wide error bound, rows it cannot
    prove complete are repaired on the split chains), split float16, float32 - identical neighbours and distances."""
    from graphtools_amd import _hip

    X = maker(n, d, 21)
    ctx = _hip.Context(0)
    ctx.set_option("knn_precision", prec)
    Y = X[::7]                                   # external queries exercise the query-side residual bound too
    ctx.set_points(X)
    dd, ii, _ = ctx.knn_search(k, Y=Y)
    d0, i0 = oracle.kneighbors(X, Y, k)
    assert np.array_equal(ii, i0)
    assert np.array_equal(dd[:, 1:], d0[:, 1:])   # column 0: a point's distance to itself is rounding noise
    dd, ii, _ = ctx.knn_search(k, rows=(0, 6000))
    d0, i0 = oracle.kneighbors(X, X[:6000], k)
    assert np.array_equal(ii, i0)
    assert np.array_equal(dd[:, 1:], d0[:, 1:])
    want = {"f16x1": "f16x1", "f16": "f16", "f32": "f32"}.get(prec)
    if want:
        assert ctx.last_knn_precision() == want
    ctx.close()


def test_auto_precision_verdict_follows_the_data():
    """'auto' keeps the single float16 chain where its error bound leaves the rows provable (well separated
    neighbour shells: mixture data) and settles on the split chains where it does not (a low-dimensional manifold
    far from the origin: neighbour gaps far below 2^-11 |x|^2) - with identical graphs either way."""
    import graphtools_amd
    from graphtools_amd import _hip

    for maker, shift, expect in ((make_mix, 0.0, "f16x1"), (make_manifold, 60.0, "f16")):
        X = (maker(30000, 64, 5) + np.float32(shift)).astype(np.float32)
        ctx = _hip.Context(0)
        ctx.set_points(X)
        p, keep = ctx.make_params(15, 40, 1e-4, None, 1.0, None, "+", None, 0)
        nnz, _ = ctx.graph_build(p)
        assert ctx.last_knn_precision() == expect
        Kd, Ki, Kp = ctx.graph_fetch_csr(_hip.CSR_K)
        ref = _hip.Context(0)
        ref.set_option("knn_precision", "f32")
        ref.set_points(X)
        ref.graph_build(p)
        Rd, Ri, Rp = ref.graph_fetch_csr(_hip.CSR_K)
        assert np.array_equal(Kp, Rp) and np.array_equal(Ki, Ri) and np.array_equal(Kd, Rd)
        ctx.close()
        ref.close()


def _pca_like(n, d, seed, decay=0.93):
    """anisotropic data: column scales fall off geometrically (what a few hundred principal components look like),
    columns shuffled so that the informative ones are not the leading ones"""
    rng = np.random.default_rng(seed)
    scales = decay ** np.arange(d)
    X = rng.standard_normal((n, d)) * scales
    centres = rng.standard_normal((8, d)) * scales * 3
    X += centres[rng.integers(8, size=n)]
    return np.ascontiguousarray(X[:, rng.permutation(d)].astype(np.float32))


@pytest.mark.parametrize("n,d,k,maker", [
    (3000, 200, 20, _pca_like), (6000, 300, 16, _pca_like), (2500, 129, 30, _pca_like),
    (1200, 400, 10, lambda n, d, s: make_gauss(n, d, s)),      # isotropic: the 128-column filter is weak, repairs carry it
])
def test_knn_wide_data_is_exact(hip_ctx, n, d, k, maker):
    """More than 128 features: candidates come from the 128 columns of largest variance (a lower bound of the distance),
    the float64 stages see all columns - same neighbours and distances as the brute-force oracle."""
    X = maker(n, d, 17)
    _check(hip_ctx, X, k)
    _check(hip_ctx, X, k, Y=maker(300, d, 18))


def test_wide_graph_matches_oracle():
    import graphtools_amd
    from scipy import sparse

    X = _pca_like(4500, 250, 3)
    G = graphtools_amd.Graph(X, knn=10, decay=20, n_pca=None, verbose=0)
    Ko, Po = oracle.knn_graph(X, knn=10, decay=20)
    Ko = sparse.csr_matrix(Ko)
    Ko.sort_indices()
    assert np.array_equal(G.K.indptr, Ko.indptr) and np.array_equal(G.K.indices, Ko.indices)
    np.testing.assert_allclose(G.K.data, Ko.data, rtol=1e-5, atol=0)
    np.testing.assert_allclose(G.P.data, sparse.csr_matrix(Po).data, rtol=1e-5, atol=0)


@pytest.mark.parametrize("nbytes", [1, 4096 + 3, (32 << 20) - 8, (32 << 20) + 8, (100 << 20) + 12345])
def test_host_copies_round_trip(hip_ctx, nbytes):
    """gt_dev_upload / gt_dev_download: small copies take the runtime path, large ones the pipelined path through
    pinned slots (gt_hostcopy.cpp) - byte-exact either way, including a ragged last chunk."""
    rng = np.random.default_rng(nbytes % 1000)
    src = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
    p = hip_ctx.dev_alloc(nbytes)
    try:
        hip_ctx.dev_upload(p, src)
        out = np.zeros(nbytes, dtype=np.uint8)
        hip_ctx.dev_download(out, p)
        assert np.array_equal(out, src)
        # second pass through the same pinned slots with different content
        src2 = src[::-1].copy()
        hip_ctx.dev_upload(p, src2)
        hip_ctx.dev_download(out, p)
        assert np.array_equal(out, src2)
    finally:
        hip_ctx.dev_free(p)


def _hip_ctx_with(**options):
    from graphtools_amd import _hip

    ctx = _hip.Context(0)
    for name, value in options.items():
        ctx.set_option(name, value)
    return ctx


def test_query_order_does_not_change_results():
    """Launches of >= 32768 rows deal the queries to workgroups grouped by nearest landmark (gt_order.hip); tables,
    distances and graphs are those of the plain row order, and a row block of a larger point set (the multi-GPU
    shard shape) maps back to the right rows."""
    X = make_mix(40000, 24, 21)
    res = {}
    for mode in ("off", "auto"):
        ctx = _hip_ctx_with(query_order=mode)
        ctx.set_points(X)
        res[mode] = ctx.knn_search(20)
        res[mode + "_block"] = ctx.knn_search(20, rows=(3000, 3000 + 33000))
        ctx.close()
    for key in ("", "_block"):
        d0, i0, _ = res["off" + key]
        d1, i1, _ = res["auto" + key]
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    d_or, i_or = oracle.kneighbors(X, X[3000:3256], 20)
    assert np.array_equal(res["auto_block"][1][:256], i_or)
    assert np.array_equal(res["auto"][1][3000:3256], i_or)


def test_query_order_for_external_queries_and_deeper_tables():
    """external query matrices are grouped the same way (cells of the bound points); tables of up to 32 neighbours
    start from the landmark bound, deeper ones from -inf - results identical either way"""
    X = make_mix(36000, 20, 5)
    Y = make_mix(34000, 20, 6)
    res = {}
    for mode in ("off", "auto"):
        ctx = _hip_ctx_with(query_order=mode)
        ctx.set_points(X)
        res[mode] = [ctx.knn_search(k, Y=Y) for k in (10, 30, 50)] + [ctx.knn_search(30)]
        ctx.close()
    for (d0, i0, _), (d1, i1, _) in zip(res["off"], res["auto"]):
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    d_or, i_or = oracle.kneighbors(X, Y[:200], 30)
    assert np.array_equal(res["auto"][1][1][:200], i_or)


def test_lane_exchanges_and_wave_reductions_match_ds_bpermute(hip_ctx):
    """gt_device.h exchanges lanes through DPP row operations and the gfx950 row / half swaps instead of ds_bpermute; every form
    (lane ^ 1 ... ^ 32, 32- and 64-bit) and the reductions built on them (wave_sum_f64 / wave_max_f32 / wave_sum_i32) must give
    what the __shfl_xor tree gives, bit for bit, with all 64 lanes active (their stated precondition)"""
    import ctypes

    lib = hip_ctx.lib
    if not hasattr(lib, "gt_dbg_lane_ops"):
        pytest.skip("library built without the gt_dbg_* hooks (GT_BUILD_DEBUG_HOOKS=0)")
    lib.gt_dbg_lane_ops.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    lib.gt_dbg_lane_ops.restype = ctypes.c_int
    for seed in (0, 12345, 0xFFFFFFFF):
        bad = np.ones(10, dtype=np.uint32)
        assert lib.gt_dbg_lane_ops(hip_ctx.h, seed, bad.ctypes.data) == 0
        assert not bad.any(), bad
