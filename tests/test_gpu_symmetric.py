"""GPU: the symmetric candidate pass (gt_sym.hip: every unordered pair of rows scored once, tested for both rows) against
the oracle and against the classic pass.  The pass engages by itself from 65536 rows; here it is forced on for small
point sets through the library's options.  Bar: kNN indices bit-exact, float32 distances exact, kernels identical to
the classic pass bit for bit."""
import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import make_gauss, make_manifold, make_mix

pytestmark = pytest.mark.gpu


@pytest.fixture()
def sym_ctx():
    from graphtools_amd import _hip

    ctx = _hip.Context(0)
    ctx.set_option("query_order_min_rows", "1")
    ctx.set_option("select_symmetric", "1")
    ctx.set_option("select_sym_stride", "4")
    yield ctx
    ctx.close()


def _knn(ctx, X, k):
    ctx.set_points(X)
    d, i, flags = ctx.knn_search(k)
    assert ctx.knn_stats()["symmetric"], "the symmetric pass did not run"
    d0, i0 = oracle.kneighbors(X, None, k)
    assert np.array_equal(i, i0), "kNN indices differ in %d rows" % int((i != i0).any(axis=1).sum())
    if X.dtype == np.float32:
        assert np.array_equal(d[:, 1:], d0[:, 1:])
    else:
        np.testing.assert_allclose(d[:, 1:], d0[:, 1:], rtol=1e-12, atol=0)
    return ctx.knn_stats()


@pytest.mark.parametrize("n,d,k,maker,seed,dtype", [
    (6000, 64, 16, make_mix, 0, np.float32),      # 256-row query blocks, 128-row tiles
    (5003, 64, 31, make_mix, 1, np.float32),      # ragged last block
    (4096, 20, 16, make_gauss, 2, np.float32),    # unclustered: the strided sample sets the thresholds
    (7000, 100, 16, make_mix, 3, np.float32),     # 112 padded features: 128-row query blocks, 64-row tiles
    (4500, 50, 20, make_mix, 4, np.float64),      # float64 points
    (4100, 64, 60, make_manifold, 5, np.float32),
    (9000, 33, 8, make_mix, 6, np.float32),
])
def test_symmetric_knn_matches_oracle(sym_ctx, n, d, k, maker, seed, dtype):
    _knn(sym_ctx, maker(n, d, seed, dtype), k)


def test_symmetric_knn_on_a_lattice_goes_through_the_repairs(sym_ctx):
    """exact distance ties straddle every threshold: rows that cannot be proven complete are repaired"""
    X = np.random.default_rng(0).integers(0, 4, size=(4500, 6)).astype(np.float32)
    sym_ctx.set_points(X)
    d, i, flags = sym_ctx.knn_search(12)
    assert sym_ctx.knn_stats()["symmetric"]
    d0, i0 = oracle.kneighbors(X, None, 12)
    assert np.array_equal(d, d0)
    assert np.array_equal(i, i0)


def test_symmetric_knn_with_duplicate_rows(sym_ctx):
    X = make_mix(5000, 32, 7)
    X[100:140] = X[50]          # 41 copies of one point: more exact ties than neighbours asked for
    X[4000] = X[3999]
    sym_ctx.set_points(X)
    d, i, flags = sym_ctx.knn_search(10)
    d0, i0 = oracle.kneighbors(X, None, 10)
    assert flags & 1                      # duplicates reported
    assert np.array_equal(i, i0)
    # distances between identical rows are rounding noise of the float64 GEMM form in scikit-learn (~1e-7), exactly 0 here
    far = d0 > 1e-5
    assert np.array_equal(d[far], d0[far]) and np.all(d[~far] <= 1e-5)


def test_symmetric_lists_overflow_is_repaired(sym_ctx):
    """tiny transposed lists: most rows overflow and must come back through the repair path unchanged"""
    sym_ctx.set_option("select_sym_tcap", "64")
    X = make_mix(6000, 64, 8)
    st = _knn(sym_ctx, X, 40)
    assert st["sym_overflow_rows"] > 0 and st["repaired_rows"] >= st["sym_overflow_rows"]


@pytest.mark.parametrize("kw", [
    dict(knn=15, decay=40.0), dict(knn=5, decay=10.0, kernel_symm="*"), dict(knn=10, decay=None),
    dict(knn=8, decay=20.0, kernel_symm="mnn", theta=0.3, anisotropy=0.5), dict(knn=12, decay=40.0, bandwidth=3.0),
])
def test_symmetric_graph_matches_oracle(sym_ctx, kw):
    X = make_mix(6000, 48, 11)
    symm = kw.get("kernel_symm", "+")
    sym_ctx.set_points(X)
    p, keep = sym_ctx.make_params(kw["knn"], kw["decay"], 1e-4, kw.get("bandwidth"), 1.0, None, symm, kw.get("theta"),
                                  kw.get("anisotropy", 0))
    nnz, flags = sym_ctx.graph_build(p)
    assert sym_ctx.knn_stats()["symmetric"]
    from graphtools_amd import _hip

    data, indices, indptr = sym_ctx.graph_fetch_csr(_hip.CSR_K)
    pdata, _, _ = sym_ctx.graph_fetch_csr(_hip.CSR_P)
    Ko, Po = oracle.knn_graph(X, knn=kw["knn"], decay=kw["decay"], bandwidth=kw.get("bandwidth"), kernel_symm=symm,
                              theta=kw.get("theta"), anisotropy=kw.get("anisotropy", 0))
    Ko = sparse.csr_matrix(Ko)
    Ko.sort_indices()
    if symm == "*":
        Ko.eliminate_zeros()
    assert np.array_equal(indptr, Ko.indptr) and np.array_equal(indices, Ko.indices)
    np.testing.assert_allclose(data, Ko.data, rtol=2e-5 if symm == "*" else 1e-5, atol=0)
    Po = sparse.csr_matrix(Po)
    Po.sort_indices()
    np.testing.assert_allclose(pdata, Po.data, rtol=2e-5 if symm == "*" else 1e-5, atol=0)


def test_symmetric_and_classic_pass_build_identical_kernels():
    """N = 150 000 (the pass engages by itself): K and P equal the classic pass bit for bit"""
    from graphtools_amd import _hip

    X = make_mix(150000, 64, 12)
    out = {}
    for mode in ("auto", "0"):
        ctx = _hip.Context(0)
        ctx.set_option("select_symmetric", mode)
        ctx.set_points(X)
        p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
        ctx.graph_build(p)
        assert ctx.knn_stats()["symmetric"] == (mode == "auto")
        out[mode] = ctx.graph_fetch_csr(_hip.CSR_K) + ctx.graph_fetch_csr(_hip.CSR_P)[:1]
        ctx.close()
    for a, b in zip(out["auto"], out["0"]):
        assert np.array_equal(a, b)


def _digest(ctx):
    """sha1 of indptr | indices | K data | P data of the finished graph (whole matrix, host copies)"""
    import hashlib

    from graphtools_amd import _hip

    kd, ki, kp = ctx.graph_fetch_csr(_hip.CSR_K)
    pd, _, _ = ctx.graph_fetch_csr(_hip.CSR_P, structure=False)
    return [hashlib.sha1(a.tobytes()).hexdigest() for a in (kp, ki, kd, pd)], int(kd.shape[0])


@pytest.mark.parametrize("kind", ["mix", "manifold"])
def test_headline_size_pruned_pass_equals_classic_pass_over_the_whole_matrix(kind):
    """N = 1e6, d = 64, knn = 15, decay = 40 (BASELINE.json's headline shape): the default build - dense seeding launch,
    cell bounds / two-stage collect, cold launch, four-lane re-rank - against the classic pass (every query row against
    every point): indptr, indices, K and P of the WHOLE matrix bit for bit.  The symmetric pass proves completeness from
    thresholds; a bound that wrongly ruled out a pair would drop a neighbour that only this comparison can see.
    `manifold` (5 dimensions embedded in 64) is the input on which the cell bounds give way to the collect launch."""
    from graphtools_amd import _hip

    n = 1000000
    if kind == "mix":
        X = make_mix(n, 64, 1)
    else:
        rng = np.random.default_rng(1)
        a = rng.standard_normal((5, 64))
        X = np.empty((n, 64), dtype=np.float32)
        for s0 in range(0, n, 100000):
            X[s0:s0 + 100000] = rng.standard_normal((100000, 5)) @ a + 0.01 * rng.standard_normal((100000, 64))
    out = {}
    for mode in ("auto", "0"):
        ctx = _hip.Context(0)
        ctx.set_option("select_symmetric", mode)
        ctx.set_points(X)
        p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
        nnz, _ = ctx.graph_build(p)
        assert ctx.knn_stats()["symmetric"] == (mode == "auto")
        out[mode] = _digest(ctx)
        assert out[mode][1] == nnz
        ctx.close()
        _hip.release_cached_memory()
    assert out["auto"] == out["0"]


# ---- two-stage collect (partial distances first, deferred cold pass) ---------------------------------------------------
def _build(X, opts, knn=15, decay=40.0):
    from graphtools_amd import _hip

    c = _hip.Context(0)
    for k, v in opts.items():
        c.set_option(k, str(v))
    c.set_points(X)
    p, keep = c.make_params(knn, decay, 1e-4, None, 1.0, None, "+", None, 0)
    c.graph_build(p)
    Kd, Ki, Kp = c.graph_fetch_csr(_hip.CSR_K)
    st, gs = c.knn_stats(), c.graph_stats()
    c.close()
    return (Kd, Ki, Kp), st, gs


def _same_csr(a, b):
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])


def test_two_stage_collect_equals_the_one_stage_collect_and_the_classic_pass():
    X = make_mix(150000, 64, 11)
    two, st2, _ = _build(X, {"select_sym_two_stage": 1})
    one, st1, _ = _build(X, {"select_sym_two_stage": 0})
    classic, st0, _ = _build(X, {"select_symmetric": 0})
    assert st2["symmetric"] and st2["sym_two_stage"] and st2["sym_cold_pairs"] > 0
    assert st1["symmetric"] and not st1["sym_two_stage"]
    assert not st0["symmetric"]
    _same_csr(two, one)
    _same_csr(two, classic)


@pytest.mark.parametrize("d", [32, 40, 64])
def test_two_stage_collect_on_forced_small_cases(sym_ctx, d):
    """32 / 48 / 64 padded features, ragged last 1024-row block; kNN against the oracle"""
    sym_ctx.set_option("select_sym_two_stage", "1")
    st = _knn(sym_ctx, make_mix(5003, d, 12), 15)
    assert st["sym_two_stage"] and st["sym_cold_pairs"] > 0


def test_two_stage_forecast_declines_where_partial_distances_do_not_separate():
    """a 5-dimensional manifold under a random projection: 16 of 64 features carry a quarter of every distance, a fifth of
    the sampled pairs would pass stage one - the one-stage kernel runs, no pass is wasted, the graph is the classic one"""
    X = make_manifold(100000, 64, 13)
    auto, st, _ = _build(X, {"select_sym_min_rows": 1, "select_sym_bounds": 0})
    classic, _, _ = _build(X, {"select_symmetric": 0})
    assert st["symmetric"] and not st["sym_two_stage"]
    _same_csr(auto, classic)
    # with the bound pass in front (the default) the cells may or may not decide enough - the graph is the same
    withb, st, _ = _build(X, {"select_sym_min_rows": 1})
    assert st["symmetric"]
    _same_csr(withb, classic)


def test_two_stage_queue_spills_and_its_overflow_starts_over_with_the_one_stage_kernel():
    """wave regions of two entries: nearly everything goes through the shared spill area; with that cut to 16 entries the
    launch drops pairs, says so, and the one-stage kernel runs instead"""
    X = make_mix(70000, 64, 14)
    ref, _, _ = _build(X, {"select_symmetric": 0})
    spilled, st, _ = _build(X, {"select_sym_two_stage": 1, "select_sym_bounds": 0, "select_sym_queue_cap": 2})
    assert st["symmetric"] and st["sym_two_stage"] and st["sym_cold_pairs"] > 0 and not st["sym_bound_pass"]
    _same_csr(spilled, ref)
    small, st, _ = _build(X, {"select_sym_two_stage": 1, "select_sym_bounds": 0, "select_sym_queue_cap": 2,
                              "select_sym_spill_cap": 16})
    assert st["symmetric"] and not st["sym_two_stage"]
    _same_csr(small, ref)


def test_bound_pass_lists_the_units_of_the_cold_launch_without_a_collect_launch():
    """clustered points: whole landmark cells rule out nearly every (64 queries, 32 rows) unit by the triangle inequality;
    what is left goes straight to the cold launch.  Same graph as with the collect launch and as the classic pass."""
    X = make_mix(150000, 64, 16)
    bound, st, _ = _build(X, {"select_sym_two_stage": 1})
    assert st["symmetric"] and st["sym_two_stage"] and st["sym_bound_pass"] and st["sym_cold_pairs"] > 0
    n_bound = st["sym_cold_pairs"]
    coll, st, _ = _build(X, {"select_sym_two_stage": 1, "select_sym_bounds": 0})
    assert st["sym_two_stage"] and not st["sym_bound_pass"]
    ref, _, _ = _build(X, {"select_symmetric": 0})
    _same_csr(bound, coll)
    _same_csr(bound, ref)
    # the bounds are no tighter than stage one by much, and far from the 150000^2 / 4096 units there are
    assert n_bound < 150000 ** 2 / 4096 / 20


def test_bound_pass_gives_way_to_the_collect_launch_when_the_cells_decide_too_little():
    """more units than the queue of the cold launch holds (here: a queue of 100).  The cell masks still rule out most TILES of
    a mixture, so the one-stage collect over listed walks takes over (round 6, gt_sym.hip collect_lists_kernel); with that
    switched off, the two-stage collect does, as it does when the lists would hold most of the walks.  Same graph either way."""
    X = make_mix(70000, 64, 17)
    ref, _, _ = _build(X, {"select_symmetric": 0})
    listed, st, _ = _build(X, {"select_sym_two_stage": 1, "select_sym_bound_cap": 100})
    assert st["symmetric"] and st["sym_listed"] and not st["sym_two_stage"] and not st["sym_bound_pass"]
    assert 0 < st["sym_cold_pairs"] < (70000 / 64) * (70000 / 32) / 2 / 4      # units scored: under a quarter of all pairs
    _same_csr(listed, ref)
    few, st, _ = _build(X, {"select_sym_two_stage": 1, "select_sym_bound_cap": 100, "select_sym_listed": 0})
    assert st["symmetric"] and st["sym_two_stage"] and not st["sym_listed"] and not st["sym_bound_pass"]
    _same_csr(few, ref)


@pytest.mark.parametrize("maker,d,n", [(make_manifold, 64, 70000), (make_gauss, 32, 66000), (make_mix, 48, 65536 + 777)])
def test_one_stage_collect_over_listed_walks_whatever_the_lists_hold(maker, d, n):
    """select_sym_listed = 1: the listed walks are used however long they are (a sheet in 64 dimensions: nearly every tile;
    isotropic points: every tile; a mixture with a ragged last block: a few) - graph equal to the classic pass's bit for bit"""
    X = maker(n, d, 29)
    ref, _, _ = _build(X, {"select_symmetric": 0})
    got, st, _ = _build(X, {"select_symmetric": 1, "select_sym_two_stage": 1, "select_sym_bound_cap": 1, "select_sym_listed": 1})
    assert st["symmetric"] and st["sym_listed"] and not st["sym_two_stage"]
    _same_csr(got, ref)


@pytest.mark.parametrize("maker,d", [(make_gauss, 64), (make_manifold, 64), (make_mix, 40)])
def test_bound_pass_on_small_forced_cases(sym_ctx, maker, d):
    """unclustered / low-dimensional / 48 padded features, ragged last block: whatever the cells decide, the kNN are exact"""
    sym_ctx.set_option("select_sym_two_stage", "1")
    sym_ctx.set_option("select_sym_bounds", "1")
    st = _knn(sym_ctx, maker(5003, d, 18), 15)
    assert st["sym_two_stage"]


def test_orphan_rows_are_repaired_from_their_seeds():
    """every row with a far-kept seed is declared an orphan (select_sym_orphan_far large): it collects nothing, starts its
    list with the rows launch A kept, and must come back through the repair pass with the exact neighbours"""
    X = make_mix(70000, 64, 15)
    orph, st, gs = _build(X, {"select_sym_two_stage": 1, "select_sym_orphan_far": 1000, "select_sym_stride": 16,
                              "select_sym_cells": 3})
    ref, _, _ = _build(X, {"select_symmetric": 0})
    assert st["symmetric"] and st["sym_two_stage"]
    assert st["repaired_rows"] > 5
    _same_csr(orph, ref)


@pytest.mark.parametrize("n,d,maker,seed,dtype,off", [
    (6000, 48, make_mix, 21, np.float64, 3.0),       # off-centre: the angular structure is not trivial
    (5003, 64, make_mix, 22, np.float32, 0.0),
    (4500, 20, make_gauss, 23, np.float64, 1.0),     # unclustered
    (8000, 33, make_manifold, 24, np.float32, 0.5),
])
def test_symmetric_pass_with_the_cosine_metric(n, d, maker, seed, dtype, off):
    """cosine distance on the symmetric pass: the rows are normalised, so the candidate stages are the euclidean ones; the
    thresholds come from the largest key 1 - x.y of the seeds, the re-rank's completeness bound is 1 - s - ymax^2 / 2.  Bar:
    neighbour table and kernel identical to the classic pass bit for bit (which the cosine tests of test_gpu_graph.py hold
    to the reference), float64 kernel within 1e-9 of the oracle."""
    from graphtools_amd import _hip

    X = (maker(n, d, seed, dtype) if maker is not make_manifold else maker(n, d, seed).astype(dtype)) + dtype(off)
    out = {}
    for mode in ("1", "0"):
        ctx = _hip.Context(0)
        for k, v in (("metric", "cosine"), ("query_order_min_rows", "1"), ("select_symmetric", mode), ("select_sym_stride", "4")):
            ctx.set_option(k, v)
        ctx.set_points(X)
        dist, idx, _ = ctx.knn_search(13)
        assert ctx.knn_stats()["symmetric"] == (mode == "1")
        p, keep = ctx.make_params(12, 15.0, 1e-4, None, 1.0, None, "+", None, 0)
        ctx.graph_build(p)
        assert ctx.knn_stats()["symmetric"] == (mode == "1")
        out[mode] = (dist, idx) + ctx.graph_fetch_csr(_hip.CSR_K) + ctx.graph_fetch_csr(_hip.CSR_P)[:1]
        ctx.close()
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a, b)
    if dtype == np.float64:
        Ko, Po = oracle.knn_graph(X, knn=12, decay=15, distance="cosine")
        Ko = sparse.csr_matrix(Ko)
        Ko.sort_indices()
        kd, ki, kp = out["1"][2], out["1"][3], out["1"][4]
        assert np.array_equal(kp, Ko.indptr) and np.array_equal(ki, Ko.indices)
        np.testing.assert_allclose(kd, Ko.data, rtol=1e-9, atol=0)


def test_cosine_graph_at_the_size_where_the_pass_engages_by_itself():
    """N = 150 000, cosine: the pruned symmetric pass equals the classic pass bit for bit"""
    from graphtools_amd import _hip

    X = make_mix(150000, 64, 31) + np.float32(1.0)
    out = {}
    for mode in ("auto", "0"):
        ctx = _hip.Context(0)
        ctx.set_option("metric", "cosine")
        ctx.set_option("select_symmetric", mode)
        ctx.set_points(X)
        p, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
        ctx.graph_build(p)
        assert ctx.knn_stats()["symmetric"] == (mode == "auto")
        out[mode] = ctx.graph_fetch_csr(_hip.CSR_K) + ctx.graph_fetch_csr(_hip.CSR_P)[:1]
        ctx.close()
    for a, b in zip(out["auto"], out["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kind", ["isolated points", "points between clusters", "both, float64"])
def test_rows_that_belong_to_no_cluster_do_not_change_the_graph(kind):
    """Real data holds rows far from everything.  They used to inflate the ball of the landmark cell they fell into (the bound
    pass then left millions of units and gave up) and, as hubs of the transpose, to refute the pair-resolved tail (the build
    was done twice).  Round 4: such rows get a cell of their own (gt_order.hip), union rows beyond the register sorts are
    finished by a segmented sort (gt_sparse.hip).  The graph must be what the classic pass over the whole matrix with the
    general tail builds - K (structure, values) and P bit for bit (graphs.py:771-982, base.py:557-646)."""
    from graphtools_amd import _hip

    n, d = 200000, 32
    rng = np.random.default_rng(11)
    X = make_mix(n, d, 7)
    if kind != "points between clusters":
        idx = rng.choice(n, 12, replace=False)
        X[idx] = rng.uniform(-12, 12, (12, d)).astype(np.float32)
    if kind != "isolated points":
        idx = rng.choice(n, 300, replace=False)
        X[idx] = (0.5 * (X[idx] + X[rng.choice(n, 300)])).astype(np.float32)
    if kind == "both, float64":
        X = X.astype(np.float64)
    res = {}
    for tag, opts in (("default", {}), ("classic", {"select_symmetric": "0", "symmetrize_pairs": "0"}),
                      ("no outlier cell", {"query_order_outliers": "0"})):
        c = _hip.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_points(X)
        p, keep = c.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
        nnz, flags = c.graph_build(p)
        K = c.graph_fetch_csr(_hip.CSR_K)
        P = c.graph_fetch_csr(_hip.CSR_P, structure=False)[0]
        st = c.knn_stats()
        res[tag] = (K, P, nnz, bool(st["symmetric"]), c.stage_launches("symm_huge"))
        c.close()
    assert res["default"][3] and not res["classic"][3]
    for tag in ("default", "no outlier cell"):
        for a, b in zip(res[tag][0], res["classic"][0]):
            assert np.array_equal(a, b), tag
        assert np.array_equal(res[tag][1], res["classic"][1]), tag
    # (a union row beyond the register sorts - the isolated points' rows usually are - took the segmented sort)
    # (float64 points take the general tail: no pair-resolved merge, no symm_huge stage - its register sorts reach 2048 entries)
    if kind != "both, float64":
        assert (res["default"][4] == 1) == (np.diff(res["default"][0][2]).max() > 1024)   # (kPairHugeRow, gt_sparse.hip)
    if kind == "isolated points":
        assert res["default"][4] == 1


def _cell_order_stats(X, coherent, stride=4):
    """-> (K csr, per cell: the largest distance IN NUMBER to one of its 8 nearest cells, rows per cell) of a build whose
    symmetric pass ran, with the cells numbered coherently or in the landmarks' own order"""
    import ctypes

    from graphtools_amd import _hip

    c = _hip.Context(0)
    c.set_option("select_symmetric", "1")
    c.set_option("select_sym_stride", str(stride))
    c.set_option("query_order_coherent", str(coherent))
    c.set_points(X)
    p, keep = c.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
    c.graph_build(p)
    assert c.knn_stats()["symmetric"]
    Kd, Ki, Kp = c.graph_fetch_csr(_hip.CSR_K)
    n = X.shape[0]
    c.lib.gt_dbg_fetch_sym.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]

    def fetch(which, count, dtype):
        out = np.empty(count, dtype=dtype)
        assert c.lib.gt_dbg_fetch_sym(c.h, which, count, out.ctypes.data_as(ctypes.c_void_p)) == 0
        return out

    cells = fetch(5, n, np.uint32).astype(np.int64)           # cell of every sorted position
    assert np.all(np.diff(cells) >= 0)
    L = int(cells.max()) + 1
    sizes = np.bincount(cells, minlength=L)
    M = 8
    nbr = fetch(7, L * M, np.int32).reshape(L, M).astype(np.int64)
    gap = np.abs(nbr - np.arange(L)[:, None]).max(axis=1)[: L - 1]   # (the last cell may be the outlier cell: not a landmark's)
    c.close()
    return (Kd, Ki, Kp), gap, sizes


def test_coherent_cell_order_groups_neighbouring_cells_and_changes_no_bit():
    """gt_order.hip coherent_landmark_order: the landmark cells numbered so that neighbours in space are neighbours in number
    (three levels of groups, one composite-key sort).  Any order is correct - the graph is the same bit for bit - and the
    order does what it is for: the eight cells around a cell lie within a few dozen numbers of it instead of anywhere."""
    X = make_mix(100000, 32, 17)
    on, gap_on, sizes_on = _cell_order_stats(X, 1)
    off, gap_off, sizes_off = _cell_order_stats(X, 0)
    _same_csr(on, off)
    # the same cells, renumbered
    assert np.array_equal(np.sort(sizes_on), np.sort(sizes_off))
    L = len(gap_off)
    assert L >= 256, "too few cells for a coherent order: %d" % L
    med_on, med_off = float(np.median(gap_on)), float(np.median(gap_off))
    print("cells %d: largest distance in number to one of the 8 nearest cells, median: coherent %.0f, landmark order %.0f" % (L, med_on, med_off))
    assert med_off > L / 4          # the landmarks' own order: anywhere
    assert med_on < med_off / 3     # coherent: close by (416 cells here, two levels of groups: 53 against 243)
    # below the size the strided sample of launch A exists at, the order is left alone (gt_order.hip explains)
    _, gap_small, _ = _cell_order_stats(X, 1, stride=768)
    assert np.array_equal(gap_small, gap_off)
