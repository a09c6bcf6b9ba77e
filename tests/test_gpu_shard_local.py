"""GPU: the row-sharded build on renumbered points (gt_points_cell_sort / gt_graph_shard_local, gt_knn_shard.cpp).

Every rank binds all points and renumbers them by landmark cell; a rank owns a run of whole cells, collects the
candidate lists of its own rows by itself (no candidate exchange), and only the transposed triplets of the
symmetrisation travel.  The ranks are played by one context each on one GPU, the all-to-all is done by hand.  The rows of
every rank, put back at the caller's row numbers (gt_points_row_ids), must equal the single-rank build bit for bit -
structure, K, P - which the other tests pin to the oracle and to the reference's fixtures."""
import numpy as np
import pytest
from scipy import sparse

from conftest import make_gauss, make_manifold, make_mix

pytestmark = pytest.mark.gpu

TRIP = np.dtype([("row", np.uint32), ("col", np.uint32), ("val", np.float64)])


def _ctx(opts):
    from graphtools_amd import _hip

    c = _hip.Context(0)
    c.set_option("query_order_min_rows", "1")
    c.set_option("select_symmetric", "1")
    c.set_option("select_sym_stride", "4")
    for k, v in opts.items():
        c.set_option(k, str(v))
    return c


def _exchange(sends, counts, world):
    out = []
    for r in range(world):
        parts = []
        for s in range(world):
            off = int(counts[s][:r].sum())
            parts.append(sends[s][off: off + int(counts[s][r])])
        out.append(np.concatenate(parts))
    return out


def sharded_local_build(X, world, pargs, opts=None, want_local=True, pairs=True):
    """-> (K csr, P csr in the caller's numbering, per-rank flags 'the local lists were used', stats)
    pairs: the ranks take the pair-resolved tail where the library offers it (the all-gather of the bandwidths between the
    halves of gt_graph_begin is played by hand); sharded_local_build.pairs_used says whether they did"""
    from graphtools_amd import _hip

    n = X.shape[0]
    # (opts: one dict for every rank, or a list with one dict per rank)
    ctxs = [_ctx((opts[r] if isinstance(opts, list) else opts) or {}) for r in range(world)]
    for c in ctxs:
        c.set_points(X)
        assert c.points_cell_sort(), "renumbering declined"
    p, keep = ctxs[0].make_params(*pargs)
    splits = ctxs[0].points_shard_splits(world)
    assert splits[0] == 0 and splits[-1] == n and np.all(np.diff(splits) >= 0)
    for c in ctxs[1:]:
        assert np.array_equal(c.points_shard_splits(world), splits)
    used = [c.graph_shard_local(p, world, r, splits) for r, c in enumerate(ctxs)]
    sharded_local_build.pairs_used = False
    bw_bufs = []
    if pairs:
        parts = []
        for r, c in enumerate(ctxs):
            nloc = int(splits[r + 1] - splits[r])
            buf = c.dev_alloc(max(nloc, 1) * 8)
            ok = c.graph_bandwidth_local(p, world, r, splits, buf)
            bw = np.empty(nloc, dtype=np.float64)
            if ok:
                c.sync()
                c.dev_download(bw, buf)
            c.dev_free(buf)
            parts.append(bw if ok else None)
        answers = [q is not None for q in parts]
        assert all(answers) or not any(answers), "the ranks disagree on the pair-resolved tail: %r" % (answers,)
        if all(answers):
            bw_all = np.concatenate(parts)
            assert len(bw_all) == n and np.all(bw_all > 0)
            for c in ctxs:
                b = c.dev_alloc(n * 8)
                c.dev_upload(b, bw_all)
                c.graph_set_bandwidths(b)
                bw_bufs.append((c, b))
            sharded_local_build.pairs_used = True
    sends, counts = [], []
    for r, c in enumerate(ctxs):
        cnt = c.graph_begin(p, world, r, splits)
        assert bool(c.knn_stats()["symmetric"]) == used[r], (r, used[r], c.knn_stats())
        total = int(cnt.sum())
        host = np.zeros(total, dtype=TRIP)
        if total:
            buf = c.dev_alloc(total * 16)
            c.graph_emit(buf)
            c.dev_download(host, buf)
            c.dev_free(buf)
        sends.append(host)
        counts.append(cnt)
    for c, b in bw_bufs:
        c.dev_free(b)
    sharded_local_build.triplets = int(sum(int(cnt.sum()) for cnt in counts))
    rows_all, blocks_K, blocks_P, stats = [], [], [], []
    for r, (c, recv) in enumerate(zip(ctxs, _exchange(sends, counts, world))):
        assert np.all((recv["row"] >= splits[r]) & (recv["row"] < splits[r + 1]))
        buf = c.dev_alloc(max(len(recv), 1) * 16)
        if len(recv):
            c.dev_upload(buf, recv)
        c.graph_finish(buf if len(recv) else 0, len(recv))
        c.dev_free(buf)
        r0, r1, nnz = c.graph_rows()
        assert (r0, r1) == (splits[r], splits[r + 1])
        d_, i_, p_ = c.graph_fetch_csr(_hip.CSR_K)
        pd_, _, _ = c.graph_fetch_csr(_hip.CSR_P)
        rows_all.append(c.points_row_ids(r0, r1))
        blocks_K.append(sparse.csr_matrix((d_, i_, p_), shape=(r1 - r0, n)))
        blocks_P.append(sparse.csr_matrix((pd_, i_, p_), shape=(r1 - r0, n)))
        stats.append((c.knn_stats(), c.graph_stats()))
    rows = np.concatenate(rows_all)
    assert np.array_equal(np.sort(rows), np.arange(n)), "the ranks' rows are not a partition of the caller's rows"
    # every rank arrives at the same numbering
    full = ctxs[0].points_row_ids(0, n)
    assert np.array_equal(full, rows)
    for c in ctxs:
        c.close()
    inv = np.empty(n, dtype=np.int64)
    inv[rows] = np.arange(n)
    K = sparse.vstack(blocks_K).tocsr()[inv]
    P = sparse.vstack(blocks_P).tocsr()[inv]
    return K, P, used, stats


def single_build(X, pargs):
    from graphtools_amd import _hip

    c = _hip.Context(0)
    c.set_points(X)
    p, keep = c.make_params(*pargs)
    c.graph_build(p)
    Kd, Ki, Kp = c.graph_fetch_csr(_hip.CSR_K)
    Pd, _, _ = c.graph_fetch_csr(_hip.CSR_P)
    n = X.shape[0]
    c.close()
    return sparse.csr_matrix((Kd, Ki, Kp), shape=(n, n)), sparse.csr_matrix((Pd, Ki, Kp), shape=(n, n))


def _same(A, B):
    assert A.has_sorted_indices or True
    A.sort_indices()
    B.sort_indices()
    assert np.array_equal(A.indptr, B.indptr)
    assert np.array_equal(A.indices, B.indices)
    assert np.array_equal(A.data, B.data)


@pytest.mark.parametrize("n,d,world,symm,seed,thresh", [
    (20000, 64, 2, "+", 0, 1e-4),
    (33001, 32, 3, "*", 1, 1e-4),      # ragged last block
    (40000, 48, 4, None, 2, 1e-4),
    (24000, 64, 8, "mnn", 3, 1e-3),    # three blocks per rank
])
def test_renumbered_sharded_build_equals_the_single_rank_build(n, d, world, symm, seed, thresh):
    X = make_mix(n, d, seed)
    pargs = (12, 30, thresh, None, 1.0, None, symm, 0.7 if symm == "mnn" else None, 0)
    K, P, used, stats = sharded_local_build(X, world, pargs)
    assert all(used), "the local candidate pass did not apply on every rank: %r" % (used,)
    K1, P1 = single_build(X, pargs)
    _same(K, K1)
    _same(P, P1)


def test_rows_that_belong_to_no_cluster_on_a_sharded_build():
    """isolated points ("next" 3b): the renumbering gives rows far from every landmark a cell of their own, the
    last one (gt_order.hip outlier cell, in the split assignment of gt_points_cells_begin as in gt_points_cell_sort), so the rank
    that receives them keeps its bound pass inside its queue instead of falling to the classic pass for all its rows; the graph is
    the single-rank build's bit for bit - with and without the outlier cell"""
    n, world = 64000, 4
    rng = np.random.default_rng(21)
    X = make_mix(n, 48, 9)
    idx = rng.choice(n, 15, replace=False)
    X[idx] = rng.uniform(-12, 12, (15, 48)).astype(np.float32)
    pargs = (15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    K1, P1 = single_build(X, pargs)
    K, P, used, stats = sharded_local_build(X, world, pargs)
    assert all(used), "a rank fell to the classic pass: %r" % (used,)
    _same(K, K1)
    _same(P, P1)
    K0, P0, used0, _ = sharded_local_build(X, world, pargs, opts={"query_order_outliers": 0})
    _same(K0, K1)
    _same(P0, P1)
    # the isolated rows are hubs of nothing: their rows hold little more than themselves and their own neighbours
    assert np.diff(K1.indptr)[idx].max() < 4096


@pytest.mark.parametrize("dtype,metric,d", [(np.float64, "euclidean", 40), (np.float32, "cosine", 64), (np.float64, "cosine", 30),
                                             (np.float32, "euclidean", 50)])
def test_renumbered_sharded_build_other_dtypes_and_metrics(dtype, metric, d):
    """float64 points (lane-per-row re-rank on the renumbered points), the cosine metric (the context's points are the
    normalised rows: they are what gets renumbered), a feature count that is no multiple of 4"""
    X = make_mix(26000, d, 21).astype(dtype)
    pargs = (10, 20, 1e-4, None, 1.0, None, "+", None, 0)
    K, P, used, stats = sharded_local_build(X, 3, pargs, opts={"metric": metric})
    assert all(used)
    from graphtools_amd import _hip
    c = _hip.Context(0)
    c.set_option("metric", metric)
    c.set_points(X)
    p, keep = c.make_params(*pargs)
    c.graph_build(p)
    Kd, Ki, Kp = c.graph_fetch_csr(_hip.CSR_K)
    Pd, _, _ = c.graph_fetch_csr(_hip.CSR_P)
    c.close()
    n = X.shape[0]
    _same(K, sparse.csr_matrix((Kd, Ki, Kp), shape=(n, n)))
    _same(P, sparse.csr_matrix((Pd, Ki, Kp), shape=(n, n)))


def test_one_rank_build_on_renumbered_points_returns_the_callers_columns():
    """gt_graph_build on a renumbered context: rows in the new order, the caller's column numbers, same bits"""
    from graphtools_amd import _hip

    X = make_mix(30000, 64, 4)
    pargs = (15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    c = _ctx({})
    c.set_points(X)
    assert c.points_cell_sort()
    p, keep = c.make_params(*pargs)
    nnz, flags = c.graph_build(p)
    d_, i_, p_ = c.graph_fetch_csr(_hip.CSR_K)
    pd_, _, _ = c.graph_fetch_csr(_hip.CSR_P)
    rows = c.points_row_ids(0, X.shape[0])
    deg = c.graph_fetch_vec(1)
    c.close()
    inv = np.empty(len(rows), dtype=np.int64)
    inv[rows] = np.arange(len(rows))
    n = X.shape[0]
    K = sparse.csr_matrix((d_, i_, p_), shape=(n, n))[inv]
    P = sparse.csr_matrix((pd_, i_, p_), shape=(n, n))[inv]
    K1, P1 = single_build(X, pargs)
    _same(K, K1)
    _same(P, P1)
    assert np.array_equal(np.asarray(K1.sum(axis=1)).ravel()[rows], deg) or np.allclose(np.asarray(K1.sum(axis=1)).ravel()[rows], deg, rtol=1e-14)


def test_anisotropy_and_vector_bandwidth_follow_the_renumbering():
    """per-row inputs (a bandwidth per row) and per-column ones (the degrees behind the anisotropy) are the CALLER's"""
    X = make_mix(20000, 32, 6)
    bw = np.random.default_rng(6).uniform(5.5, 7.5, size=X.shape[0])
    pargs = (10, 20, 1e-4, bw, 1.0, None, "+", None, 0.5)
    # two ranks: the degrees of all rows in the caller's numbering, as dist.py hands them over
    from graphtools_amd import _hip

    world, n = 2, X.shape[0]
    ctxs = [_ctx({}) for _ in range(world)]
    for c in ctxs:
        c.set_points(X)
        assert c.points_cell_sort()
    p, keep = ctxs[0].make_params(*pargs)
    splits = ctxs[0].points_shard_splits(world)
    sends, counts = [], []
    for r, c in enumerate(ctxs):
        c.graph_shard_local(p, world, r, splits)
        cnt = c.graph_begin(p, world, r, splits)
        host = np.zeros(int(cnt.sum()), dtype=TRIP)
        buf = c.dev_alloc(max(len(host), 1) * 16)
        c.graph_emit(buf)
        if len(host):
            c.dev_download(host, buf)
        c.dev_free(buf)
        sends.append(host)
        counts.append(cnt)
    deg_caller = np.zeros(n)
    for r, (c, recv) in enumerate(zip(ctxs, _exchange(sends, counts, world))):
        buf = c.dev_alloc(max(len(recv), 1) * 16)
        c.dev_upload(buf, recv)
        c.graph_finish(buf, len(recv))
        c.dev_free(buf)
        deg_caller[c.points_row_ids(splits[r], splits[r + 1])] = c.graph_fetch_vec(1)
    blocks_K, blocks_P, rows_all = [], [], []
    for r, c in enumerate(ctxs):
        dbuf = c.dev_alloc(n * 8)
        c.dev_upload(dbuf, deg_caller)
        c.graph_anisotropy(dbuf)
        c.dev_free(dbuf)
        d_, i_, p_ = c.graph_fetch_csr(_hip.CSR_K)
        pd_, _, _ = c.graph_fetch_csr(_hip.CSR_P)
        rows_all.append(c.points_row_ids(splits[r], splits[r + 1]))
        blocks_K.append(sparse.csr_matrix((d_, i_, p_), shape=(splits[r + 1] - splits[r], n)))
        blocks_P.append(sparse.csr_matrix((pd_, i_, p_), shape=(splits[r + 1] - splits[r], n)))
        c.close()
    rows = np.concatenate(rows_all)
    inv = np.empty(n, dtype=np.int64)
    inv[rows] = np.arange(n)
    K = sparse.vstack(blocks_K).tocsr()[inv]
    P = sparse.vstack(blocks_P).tocsr()[inv]
    K1, P1 = single_build(X, pargs)
    _same(K, K1)
    _same(P, P1)


def test_a_rank_may_take_the_classic_pass_on_its_own():
    """nothing is shared before the triplets: a rank that declines the local pass (here: by option) builds its rows with the
    classic candidate pass while its peers use their local lists - the graph is the same; isotropic points (whatever each rank
    decides) likewise"""
    X = make_mix(24000, 64, 7)
    pargs = (12, 30, 1e-4, None, 1.0, None, "+", None, 0)
    K1, P1 = single_build(X, pargs)
    K, P, used, stats = sharded_local_build(X, 3, pargs, opts=[{}, {"select_sym_two_stage": 0}, {}])
    assert used == [True, False, True]
    _same(K, K1)
    _same(P, P1)
    K, P, used, stats = sharded_local_build(X, 2, pargs, opts={"select_sym_two_stage": 0})
    assert not any(used)
    _same(K, K1)
    _same(P, P1)
    X = make_gauss(20000, 24, 7)
    pargs = (8, 30, 1e-4, None, 1.0, None, "+", None, 0)
    K, P, used, stats = sharded_local_build(X, 2, pargs, opts={"select_symmetric": "auto", "select_sym_min_rows": 1})
    K1, P1 = single_build(X, pargs)
    _same(K, K1)
    _same(P, P1)


@pytest.mark.parametrize("maker,n,d,world", [(make_mix, 70000, 64, 3), (make_mix, 90000, 32, 4), (make_mix, 66001, 48, 2)])
def test_cell_bounds_that_decide_too_little_send_a_rank_to_its_own_two_stage_collect(maker, n, d, world):
    """round 6: a rank whose cell bounds leave more units than the queue of the cold launch holds (here: a queue of 50) no
    longer falls back to the classic pass - its own 1024-row query blocks stream every tile through stage one of the two-stage
    collect (forward test only), the survivors go to the cold launch, filed under the queries.  Same graph; one rank of the
    first case alone (its peers within the bounds), every rank of the others; a ragged last block in the third.  (Sizes with a
    few dozen clusters: with a dozen, one random pair in twelve passes stage one and its forecast rightly declines.  The data
    the path is FOR - a sheet in 64 dimensions - passes the forecast from ~10^6 rows only: tests/test_gpu_shard_full.py.)"""
    X = maker(n, d, 7)
    pargs = (12, 30, 1e-4, None, 1.0, None, "+", None, 0)
    K1, P1 = single_build(X, pargs)
    opts = [{}, {"select_sym_bound_cap": 50}, {}] if world == 3 else {"select_sym_bound_cap": 50}
    K, P, used, stats = sharded_local_build(X, world, pargs, opts=opts)
    assert all(used)
    for r, (ks, gs) in enumerate(stats):
        capped = world != 3 or r == 1
        assert ks["sym_two_stage"] and ks["sym_bound_pass"] == (not capped), (r, ks)
    _same(K, K1)
    _same(P, P1)


def test_manifold_rows_and_small_lists():
    """overflowing lists (repairs on the owner) and a curled manifold (the bound pass leaves more units: still the same graph)"""
    X = make_manifold(30000, 64, 8)
    pargs = (15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    K, P, used, stats = sharded_local_build(X, 3, pargs)
    K1, P1 = single_build(X, pargs)
    _same(K, K1)
    _same(P, P1)
    X = make_mix(20000, 32, 9)
    K, P, used, stats = sharded_local_build(X, 2, pargs, opts={"select_sym_tcap": 64})
    assert all(used)
    assert sum(s[0]["sym_overflow_rows"] for s in stats) > 100
    K1, P1 = single_build(X, pargs)
    _same(K, K1)
    _same(P, P1)


def test_landmark_operator_of_row_blocks_on_renumbered_points():
    """gt_landmark_build on the row blocks of a sharded build (graphs.py:1169-1246 split over ranks): the stacked partial
    products M and row sums R of the ranks equal the single-rank ones - and the operator equals the reference's fixture
    g7 (tests/golden, written by the imported reference) - whatever the numbering of the rows; gt_nearest_landmark on a
    rank's row range assigns the caller's rows (graphs.py:1200-1213)"""
    from conftest import load_golden
    from graphtools_amd import _hip

    z = load_golden("g7_landmark")
    X = z["X"]
    n = X.shape[0]
    L = int(z["n_landmark"])
    clusters = z["clusters"].astype(np.int32)
    pargs = (15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    # single rank
    c1 = _hip.Context(0)
    c1.set_points(X)
    p, keep = c1.make_params(*pargs)
    c1.graph_build(p)
    M1, R1, t1 = c1.landmark_build(clusters, L)
    op1 = c1.landmark_scale(M1, R1)
    c1.close()
    np.testing.assert_allclose(op1, z["landmark_op"], rtol=1e-9, atol=1e-300)
    # two ranks on row blocks; small point sets keep the caller's numbering (the renumbering needs cells), larger ones do not
    for Xs, cl, Lx in ((X, clusters, L), (None, None, 64)):
        if Xs is None:
            Xs = make_mix(20000, 32, 12)
            cl = np.random.default_rng(12).integers(0, Lx, size=Xs.shape[0]).astype(np.int32)
            cl[:Lx] = np.arange(Lx)
            c1 = _hip.Context(0)
            c1.set_points(Xs)
            p, keep = c1.make_params(*pargs)
            c1.graph_build(p)
            M1, R1, t1 = c1.landmark_build(cl, Lx)
            op1 = c1.landmark_scale(M1, R1)
            # random landmarking on the caller's rows (graphs.py:1200-1213)
            lm = np.sort(np.random.default_rng(5).choice(Xs.shape[0], Lx, replace=False)).astype(np.int64)
            near1 = c1.nearest_landmark(lm, 0)
            c1.close()
        world, nn = 2, Xs.shape[0]
        ctxs = [_ctx({}) for _ in range(world)]
        renum = []
        for c in ctxs:
            c.set_points(Xs)
            renum.append(c.points_cell_sort())
        assert len(set(renum)) == 1
        if renum[0]:
            splits = ctxs[0].points_shard_splits(world)
        else:
            splits = np.array([0, nn // 2, nn], dtype=np.int64)
        p, keep = ctxs[0].make_params(*pargs)
        sends, counts = [], []
        for r, c in enumerate(ctxs):
            if renum[0]:
                c.graph_shard_local(p, world, r, splits)
            cnt = c.graph_begin(p, world, r, splits)
            host = np.zeros(int(cnt.sum()), dtype=TRIP)
            buf = c.dev_alloc(max(len(host), 1) * 16)
            c.graph_emit(buf)
            if len(host):
                c.dev_download(host, buf)
            c.dev_free(buf)
            sends.append(host)
            counts.append(cnt)
        M, R = np.zeros((Lx, Lx)), np.zeros(Lx)
        near = np.full(nn, -1, dtype=np.int64)
        for r, (c, recv) in enumerate(zip(ctxs, _exchange(sends, counts, world))):
            buf = c.dev_alloc(max(len(recv), 1) * 16)
            if len(recv):
                c.dev_upload(buf, recv)
            c.graph_finish(buf if len(recv) else 0, len(recv))
            c.dev_free(buf)
            Mr, Rr, tr = c.landmark_build(cl, Lx)     # (cluster labels by the caller's row numbers = the CSR's columns)
            M += np.asarray(Mr)
            R += np.asarray(Rr)
            if Xs is not X:
                ids = c.points_row_ids(splits[r], splits[r + 1])
                near[ids] = c.nearest_landmark(lm, 0, rows=(int(splits[r]), int(splits[r + 1])))
        op = ctxs[0].landmark_scale(M, R)
        for c in ctxs:
            c.close()
        np.testing.assert_allclose(op, op1, rtol=1e-12, atol=1e-300)
        if Xs is X:
            np.testing.assert_allclose(op, z["landmark_op"], rtol=1e-9, atol=1e-300)
        else:
            assert renum[0], "20 000 points should have been renumbered"
            assert np.array_equal(near, near1)


def test_split_cell_assignment_gives_the_same_numbering():
    """gt_points_cells_begin / _finish (every rank assigns 1 / world of the rows, the cells are all-gathered) arrive at the
    numbering gt_points_cell_sort finds alone - and the build on it is the same graph"""
    from graphtools_amd import _hip

    X = make_mix(30000, 64, 13)
    n, d = X.shape
    a = _ctx({})
    a.set_points(X)
    assert a.points_cell_sort()
    ids_a = a.points_row_ids(0, n)
    a.close()
    b = _ctx({})
    xb = b.dev_alloc(X.nbytes)
    b.dev_upload(xb, X)
    cells = b.dev_alloc(n * 4)
    cuts = [0, 7000, 7001, 22000, n]          # uneven shares, one of a single row
    for r in range(len(cuts) - 1):
        assert b.points_cells_begin(xb, n, d, np.float32, cuts[r], cuts[r + 1], cells + cuts[r] * 4)
    b.points_cells_finish(cells)
    ids_b = b.points_row_ids(0, n)
    assert np.array_equal(ids_a, ids_b)
    pargs = (15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    p, keep = b.make_params(*pargs)
    b.graph_build(p)
    d_, i_, p_ = b.graph_fetch_csr(_hip.CSR_K)
    b.dev_free(cells)
    b.dev_free(xb)
    b.close()
    inv = np.empty(n, dtype=np.int64)
    inv[ids_b] = np.arange(n)
    K = sparse.csr_matrix((d_, i_, p_), shape=(n, n))[inv]
    K1, P1 = single_build(X, pargs)
    _same(K, K1)
    # too few points for a cell order: bound as gt_set_points binds them, nothing to finish
    c = _hip.Context(0)
    small = make_mix(3000, 16, 1)
    xs = c.dev_alloc(small.nbytes)
    c.dev_upload(xs, small)
    assert not c.points_cells_begin(xs, 3000, 16, np.float32, 0, 1500, c.dev_alloc(1500 * 4))
    with pytest.raises(_hip.HipError):
        c.points_cells_finish(xs)
    p, keep = c.make_params(5, 20, 1e-4, None, 1.0, None, "+", None, 0)
    nnz, _ = c.graph_build(p)
    assert nnz > 0
    c.close()


# ---- the pair-resolved tail on the ranks of a sharded build (round 6: gt_graph_bandwidth_local / gt_graph_set_bandwidths) --------


@pytest.mark.parametrize("maker,n,d,world,knn,decay,thresh", [
    (make_mix, 30000, 64, 3, 12, 30, 1e-4),      # tables with transposed keys on every rank
    (make_mix, 20000, 8, 2, 10, 4, 1e-4),        # radii that reach past the tables: rows of the radius pass, long union rows
    (make_gauss, 20000, 24, 2, 8, 30, 1e-4),     # isotropic: the ranks fall to the classic pass - keys from the dot products
    (make_manifold, 40000, 32, 4, 15, 40, 1e-4),
])
def test_pair_resolved_tail_on_sharded_ranks_equals_the_general_tail_and_the_single_rank_build(maker, n, d, world, knn, decay, thresh):
    """'+' rule: every rank settles its mutual pairs itself with the gathered bandwidths, only one-sided entries travel, the
    union rows are sorted once and written straight into the CSR.  K and P equal the general tail's (every kept entry travels,
    union rows merged) and the single-rank build's bit for bit; fewer triplets cross."""
    X = maker(n, d, 11)
    pargs = (knn, decay, thresh, None, 1.0, None, "+", None, 0)
    K1, P1 = single_build(X, pargs)
    K, P, used, stats = sharded_local_build(X, world, pargs)
    assert sharded_local_build.pairs_used, "the ranks did not take the pair-resolved tail"
    sent_bins = sharded_local_build.triplets
    _same(K, K1)
    _same(P, P1)
    # ... without destination bins of the rank's own: every one-sided entry goes through the exchange, a rank's own too
    Kn, Pn, _, _ = sharded_local_build(X, world, pargs, opts={"symmetrize_bins": 0})
    assert sharded_local_build.pairs_used
    sent = sharded_local_build.triplets
    assert sent_bins <= sent
    _same(Kn, K1)
    _same(Pn, P1)
    Kg, Pg, _, stats_g = sharded_local_build(X, world, pargs, pairs=False)
    assert not sharded_local_build.pairs_used
    _same(Kg, K1)
    _same(Pg, P1)
    # the general tail sends every kept entry (nnz of the unsymmetrised kernel), this one the one-sided entries only:
    # K's entries = mutual pairs counted once per side + one-sided entries counted on both sides
    sent_g = sharded_local_build.triplets
    assert sent < sent_g and K1.nnz == sent_g + sent, (sent, sent_g, K1.nnz)
    print("triplets: %d one-sided of %d kept (%.0f %%), %d of them for other ranks' rows" % (sent, sent_g, 100.0 * sent / sent_g, sent_bins))


def test_pair_resolved_tail_is_an_option_and_only_serves_the_plus_rule():
    X = make_mix(20000, 32, 12)
    for symm, opts, want in (("+", {}, True), ("+", {"symmetrize_pairs_shard": 0}, False), ("+", {"symmetrize_pairs": 0}, False),
                             ("*", {}, False), (None, {}, False)):
        pargs = (10, 20, 1e-4, None, 1.0, None, symm, None, 0)
        K, P, used, _ = sharded_local_build(X, 2, pargs, opts=opts)
        assert sharded_local_build.pairs_used == want, (symm, opts)
        K1, P1 = single_build(X, pargs)
        _same(K, K1)
        _same(P, P1)


def test_pair_resolved_tail_with_a_bandwidth_per_row_and_a_rank_that_declines_its_local_pass():
    """the caller's bandwidths (by the caller's row numbers) reach the partners through the gather like the derived ones; a
    rank whose tables came from the classic pass forms the transposed keys from the dot products while its peers read theirs"""
    X = make_mix(24000, 64, 7)
    bw = np.random.default_rng(8).uniform(5.5, 7.5, size=X.shape[0])
    pargs = (12, 30, 1e-4, bw, 1.0, None, "+", None, 0)
    K1, P1 = single_build(X, pargs)
    K, P, used, _ = sharded_local_build(X, 3, pargs, opts=[{}, {"select_sym_two_stage": 0}, {}])
    assert sharded_local_build.pairs_used and used == [True, False, True]
    _same(K, K1)
    _same(P, P1)


def test_begin_after_a_bandwidth_half_must_be_the_same_build():
    from graphtools_amd import _hip

    X = make_mix(16000, 32, 3)
    c = _ctx({})
    c.set_points(X)
    assert c.points_cell_sort()
    p, keep = c.make_params(10, 20, 1e-4, None, 1.0, None, "+", None, 0)
    q, keep_q = c.make_params(11, 20, 1e-4, None, 1.0, None, "+", None, 0)
    splits = c.points_shard_splits(2)
    nloc = int(splits[1] - splits[0])
    buf = c.dev_alloc(nloc * 8)
    with pytest.raises(_hip.HipError):
        c.graph_set_bandwidths(buf)                      # nothing is half begun
    c.graph_shard_local(p, 2, 0, splits)
    assert c.graph_bandwidth_local(p, 2, 0, splits, buf)
    with pytest.raises(_hip.HipError):
        c.graph_begin(q, 2, 0, splits)                   # other parameters
    # (the refused call ended the half-begun build: a begin from the start is what follows, the general way)
    cnt = c.graph_begin(p, 2, 0, splits)
    assert int(cnt.sum()) > 0
    c.dev_free(buf)
    c.close()


def _fuzz_cases(count, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        maker = [make_mix, make_mix, make_manifold, make_gauss][int(rng.integers(4))]
        n = int(rng.integers(4200, 30000))
        d = int(rng.choice([8, 16, 24, 33, 48, 64, 100]))
        world = int(rng.integers(2, 7))
        knn = int(rng.integers(3, 40))
        decay = float(rng.choice([2, 4, 10, 20, 40, 2.5]))
        thresh = float(rng.choice([1e-2, 1e-3, 1e-4, 1e-6]))
        scale = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
        dtype = np.float64 if rng.random() < 0.2 else np.float32
        cases.append((maker, n, d, world, knn, decay, thresh, scale, dtype, int(rng.integers(1 << 30))))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(14, 20261004), ids=lambda c: "%s-n%d-d%d-w%d-k%d-a%g-t%g-s%g-%s" % (
    c[0].__name__[5:], c[1], c[2], c[3], c[4], c[5], c[6], c[7], np.dtype(c[8]).name))
def test_pair_resolved_tail_on_random_sharded_configurations(case):
    """random sizes, widths (also no multiple of 4, also beyond 64 features), rank counts, neighbour counts, decays (small ones:
    radii that reach far past the tables - rows of the radius pass, union rows of every length class), thresholds, bandwidth
    scales, float64 points: the ranks' pair-resolved tail gives the single-rank build's K and P bit for bit"""
    maker, n, d, world, knn, decay, thresh, scale, dtype, seed = case
    X = maker(n, d, seed % 1000, dtype)
    pargs = (knn, decay, thresh, None, scale, None, "+", None, 0)
    try:
        K1, P1 = single_build(X, pargs)
    except Exception as ex:     # (a kernel that reaches every point is declined with GT_E_LIMIT: not this test's subject)
        pytest.skip("single-rank build declined: %s" % str(ex)[:80])
    K, P, used, _ = sharded_local_build(X, world, pargs)
    assert sharded_local_build.pairs_used
    _same(K, K1)
    _same(P, P1)
