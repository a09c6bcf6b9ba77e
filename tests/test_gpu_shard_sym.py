"""GPU: the symmetric candidate pass split over ranks (gt_knn_shard.cpp, gt_graph_sym_*), simulated with one context per
rank on one GPU and every collective done by hand: thresholds all-gathered, candidate records moved to the owners of
the rows, then the ordinary sharded build (gt_graph_begin / emit / finish).  The stacked row blocks must equal the
single-rank build bit for bit (which the other tests pin to the oracle), and every rank must report that its tables
came from the symmetric lists."""
import numpy as np
import pytest

from conftest import make_gauss, make_mix

pytestmark = pytest.mark.gpu

REC = np.dtype([("row", np.uint32), ("pad", np.uint32), ("key", np.uint64)])
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
    """what the all-to-all does: rank r receives bucket r of every rank, in rank order"""
    out = []
    for r in range(world):
        parts = []
        for s in range(world):
            off = int(counts[s][:r].sum())
            parts.append(sends[s][off: off + int(counts[s][r])])
        out.append(np.concatenate(parts))
    return out


def sharded_build(X, splits, pargs, opts=None):
    from graphtools_amd import _hip

    world = len(splits) - 1
    ctxs = [_ctx(opts or {}) for _ in range(world)]
    for c in ctxs:
        c.set_points(X)
    p, keep = ctxs[0].make_params(*pargs)
    plans = [c.graph_sym_plan(p, world, r, splits) for r, c in enumerate(ctxs)]
    ran = all(pl[0] for pl in plans)
    if ran:
        n_pad = plans[0][1]
        ss = plans[0][2]
        assert all(np.array_equal(pl[2], ss) and pl[1] == n_pad for pl in plans)
        assert ss[0] == 0 and ss[-1] == n_pad
        parts, far = [], np.zeros(5)
        for r, c in enumerate(ctxs):
            rows = int(ss[r + 1] - ss[r])
            buf = c.dev_alloc(max(rows, 1) * 8)
            far += c.graph_sym_seed(buf)
            host = np.zeros((rows, 2), dtype=np.float32)     # {threshold, far-kept seeds} per position
            if rows:
                c.dev_download(host, buf)
            c.dev_free(buf)
            parts.append(host)
        thr_all = np.concatenate(parts)
        assert thr_all.shape == (n_pad, 2)
        sends, counts = [], []
        for r, c in enumerate(ctxs):
            buf = c.dev_alloc(n_pad * 8)
            c.dev_upload(buf, thr_all)
            ok, cnt = c.graph_sym_collect(buf, far, world)
            c.dev_free(buf)
            ran = ran and ok
            counts.append(cnt)
        if ran:
            for r, c in enumerate(ctxs):
                total = int(counts[r].sum())
                host = np.zeros(total, dtype=REC)
                buf = c.dev_alloc(max(total, 1) * 16)
                c.graph_sym_emit(buf if total else 0)
                if total:
                    c.dev_download(host, buf)
                c.dev_free(buf)
                sends.append(host)
            for r, (c, recv) in enumerate(zip(ctxs, _exchange(sends, counts, world))):
                assert np.all(recv["row"] < splits[r + 1] - splits[r])
                buf = c.dev_alloc(max(len(recv), 1) * 16)
                if len(recv):
                    c.dev_upload(buf, recv)
                c.graph_sym_finish(buf if len(recv) else 0, len(recv))
                c.dev_free(buf)
    # the ordinary sharded build; it consumes the lists when they are there
    sends, counts = [], []
    for r, c in enumerate(ctxs):
        cnt = c.graph_begin(p, world, r, splits)
        assert bool(c.knn_stats()["symmetric"]) == ran, (r, c.knn_stats())
        total = int(cnt.sum())
        host = np.zeros(total, dtype=TRIP)
        if total:
            buf = c.dev_alloc(total * 16)
            c.graph_emit(buf)
            c.dev_download(host, buf)
            c.dev_free(buf)
        sends.append(host)
        counts.append(cnt)
    for r, (c, recv) in enumerate(zip(ctxs, _exchange(sends, counts, world))):
        buf = c.dev_alloc(max(len(recv), 1) * 16)
        if len(recv):
            c.dev_upload(buf, recv)
        c.graph_finish(buf if len(recv) else 0, len(recv))
        c.dev_free(buf)
    datas, inds, ptrs, pdatas, base = [], [], [], [], 0
    stats = []
    for c in ctxs:
        d_, i_, p_ = c.graph_fetch_csr(_hip.CSR_K)
        pd_, _, _ = c.graph_fetch_csr(_hip.CSR_P)
        datas.append(d_); inds.append(i_); pdatas.append(pd_)
        ptrs.append(p_[:-1] + base)
        base += p_[-1]
        stats.append((c.knn_stats(), c.graph_stats()))
        c.close()
    return (np.concatenate(datas), np.concatenate(inds), np.concatenate(ptrs + [[base]]), np.concatenate(pdatas)), ran, stats


def single_build(X, pargs, symmetric):
    from graphtools_amd import _hip

    c = _ctx({}) if symmetric else _hip.Context(0)
    c.set_points(X)
    p, keep = c.make_params(*pargs)
    c.graph_build(p)
    Kd, Ki, Kp = c.graph_fetch_csr(_hip.CSR_K)
    Pd, _, _ = c.graph_fetch_csr(_hip.CSR_P)
    c.close()
    return Kd, Ki, Kp, Pd


def _same(a, b):
    assert np.array_equal(a[2], b[2])
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], b[0])
    np.testing.assert_allclose(a[3], b[3], rtol=1e-14)


@pytest.mark.parametrize("n,d,world,symm,seed,thresh", [
    (6000, 64, 2, "+", 0, 1e-4),
    (7001, 32, 3, "*", 1, 1e-4),       # ragged last block, unequal row split
    (9000, 50, 4, None, 2, 1e-4),
    # 112 padded features: 128-row query blocks.  (With thresh = 1e-4 about a fifth of these rows cannot be proven out to
    # the kernel's radius in the single-chain arithmetic - right at the limit where a rank redoes its rows classically)
    (5000, 100, 2, "+", 3, 1e-2),
])
def test_sharded_symmetric_pass_equals_the_single_rank_build(n, d, world, symm, seed, thresh):
    X = make_mix(n, d, seed)
    cuts = np.sort(np.random.default_rng(seed).choice(np.arange(200, n - 200), size=world - 1, replace=False))
    splits = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    pargs = (12, 30, thresh, None, 1.0, None, symm, None, 0)
    got, ran, stats = sharded_build(X, splits, pargs)
    assert ran, "the sharded symmetric pass did not apply"
    _same(got, single_build(X, pargs, True))
    _same(got, single_build(X, pargs, False))       # and the classic single-rank build


def test_sharded_symmetric_pass_with_overflowing_lists():
    """tiny lists: most rows overflow on some rank, the marker records hand them to the repairs of their owners"""
    X = make_mix(6000, 32, 5)
    splits = np.array([0, 2500, 6000], dtype=np.int64)
    pargs = (12, 30, 1e-4, None, 1.0, None, "+", None, 0)
    got, ran, stats = sharded_build(X, splits, pargs, opts={"select_sym_tcap": 64})
    assert ran
    assert sum(s[0]["sym_overflow_rows"] for s in stats) > 100
    _same(got, single_build(X, pargs, False))


def test_sharded_symmetric_pass_declines_consistently():
    """the plan refuses what the single-rank pass refuses (size below the engagement threshold, knn_max); the far-kept
    predictor refuses unstructured points at the collect stage, on the summed count; the build goes on classically"""
    from graphtools_amd import _hip

    X = make_gauss(5000, 16, 6)
    splits = np.array([0, 2000, 5000], dtype=np.int64)
    pargs = (5, 30, 1e-4, None, 1.0, None, "+", None, 0)
    c = _hip.Context(0)     # default options: 5000 rows are below the size the pass engages at
    c.set_points(X)
    p, keep = c.make_params(*pargs)
    assert not c.graph_sym_plan(p, 2, 0, splits)[0]
    c.close()
    c = _ctx({})
    c.set_points(X)
    p, keep = c.make_params(5, 30, 1e-4, None, 1.0, 60, "+", None, 0)       # knn_max
    assert not c.graph_sym_plan(p, 2, 0, splits)[0]
    c.close()
    got, ran, stats = sharded_build(X, splits, pargs, opts={"select_symmetric": "auto", "select_sym_min_rows": 1,
                                                            "select_sym_tcap": 64})
    assert not ran
    _same(got, single_build(X, pargs, False))


def test_stage_order_is_enforced():
    from graphtools_amd import _hip

    c = _ctx({})
    c.set_points(make_mix(5000, 16, 7))
    with pytest.raises(_hip.HipError):
        c.graph_sym_seed(0)
    with pytest.raises(_hip.HipError):
        c.graph_sym_emit(0)
    with pytest.raises(_hip.HipError):
        c.graph_sym_finish(0, 0)
    c.close()


@pytest.mark.parametrize("opts,what", [
    # the cell bounds leave more units than the cap: EVERY rank must move on to the collect launch - the verdict is taken on
    # the units of all ranks' pieces (a rank deciding on its own share would run a kernel that cuts the pair space
    # differently from its peers': pairs unscored, neighbours silently missing)
    ({"select_sym_two_stage": 1, "select_sym_bound_cap": 50}, "bound pass over its cap on the global count"),
    # queue regions of one entry per wave: everything goes through the spill area, which holds every unit of a rank's pieces
    ({"select_sym_two_stage": 1, "select_sym_bounds": 0, "select_sym_queue_cap": 1}, "two-stage queue through the spill area"),
    ({"select_sym_two_stage": 1, "select_sym_bounds": 0}, "two-stage collect"),
    ({"select_sym_two_stage": 0}, "one-stage collect"),
])
def test_every_rank_runs_the_same_collect_kernel(opts, what):
    n, world = 12000, 3
    X = make_mix(n, 64, 11)
    splits = np.array([0, 3100, 8000, n], dtype=np.int64)
    pargs = (12, 30, 1e-4, None, 1.0, None, "+", None, 0)
    got, ran, stats = sharded_build(X, splits, pargs, opts=opts)
    assert ran, what
    kinds = {(bool(s[0].get("sym_two_stage")), bool(s[0].get("sym_bound_pass"))) for s in stats}
    assert len(kinds) == 1, (what, kinds)
    _same(got, single_build(X, pargs, False))


def test_host_flow_over_rccl_world_one():
    """graphtools_amd/dist.py itself on the GPU: a one-rank RCCL group runs every collective of the sharded build (points
    all-gather, threshold all-gather / record all-to-all of the staged symmetric pass, triplet all-to-all) on torch's
    streams, ordered against the library's stream by gt_stream_order instead of host synchronisation.  The rows must equal
    the direct single-rank build."""
    import os
    import socket

    import torch
    import torch.distributed as dist

    from graphtools_amd import _hip
    from graphtools_amd import dist as gdist

    X = make_mix(60000, 24, 5)
    pargs = (10, 20.0, 1e-4, None, 1.0, None, "+", None, 0)
    ref = _ctx({})
    ref.set_points(X)
    p, keep = ref.make_params(*pargs)
    ref.graph_build(p)
    kd, ki, kp = ref.graph_fetch_csr(_hip.CSR_K)
    pd, _, _ = ref.graph_fetch_csr(_hip.CSR_P, structure=False)
    ref.close()

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["GT_SHARD_SYM_FORCE"] = "1"      # the staged / local symmetric pass although there is one rank
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0)
    try:
        from scipy import sparse

        device = torch.device("cuda", 0)
        n = X.shape[0]
        K_ref = sparse.csr_matrix((kd, ki, kp), shape=(n, n))
        clusters = np.random.default_rng(3).integers(0, 40, size=n).astype(np.int32)
        clusters[:40] = np.arange(40)
        op_ref = None
        for renumber in (True, False):     # the default flow (cell-sorted renumbering) and the staged pass of rounds 2-3
            ctx = _ctx({})
            g = gdist.ShardedKnnGraph(ctx, n, renumber=renumber)
            g.gather_points(torch.from_numpy(X).to(device))
            assert g.renumbered == renumber
            p2, keep2 = ctx.make_params(*pargs)
            nnz, _ = g.build(p2)
            assert g.symmetric_used
            assert g.pairs_used, "the pair-resolved tail (bandwidth all-gather between the halves of graph_begin) did not run"
            d2, i2, p2_ = ctx.graph_fetch_csr(_hip.CSR_K)
            pd2, _, _ = ctx.graph_fetch_csr(_hip.CSR_P, structure=False)
            ids = g.row_ids()
            assert np.array_equal(np.sort(ids), np.arange(n))
            want = K_ref[ids]                                   # the caller's rows in the rank's order, the caller's columns
            assert np.array_equal(p2_, want.indptr) and np.array_equal(i2, want.indices)
            assert np.array_equal(d2, want.data)
            want_p = sparse.csr_matrix((pd, ki, kp), shape=(n, n))[ids]
            assert np.array_equal(pd2, want_p.data)
            # landmark operator over the same group (one all-reduce of the L x L partials): independent of the numbering
            op, tnnz = g.landmark_operator(clusters, 40)
            np.testing.assert_allclose(op.sum(axis=1), 1.0, rtol=1e-12)
            if op_ref is None:
                op_ref = op
            else:
                np.testing.assert_allclose(op, op_ref, rtol=1e-12, atol=1e-300)
            ctx.close()
        # the package's own boundary over the same group: Graph(X, ..., distributed=True) == Graph(X, ...) - full K and P, degrees,
        # and (landmark graphs) clusters, operator and transitions
        import graphtools_amd

        G1 = graphtools_amd.Graph(X, knn=10, decay=20, n_pca=None, verbose=False, n_landmark=40, random_landmarking=True,
                                  random_state=7)
        Gd = graphtools_amd.Graph(X, knn=10, decay=20, n_pca=None, verbose=False, n_landmark=40, random_landmarking=True,
                                  random_state=7, distributed=True)
        assert Gd._dist_ranks() is not None and Gd._sharded.renumbered
        for a, b in ((G1.K, Gd.K), (G1.P, Gd.P)):
            assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data)
        assert np.array_equal(G1.K.data, kd)
        assert np.array_equal(G1.kernel_degree, Gd.kernel_degree)
        assert (Gd.K_local != Gd.K[Gd.local_rows]).nnz == 0
        assert np.array_equal(np.asarray(G1.clusters), np.asarray(Gd.clusters))
        np.testing.assert_allclose(Gd.landmark_op, G1.landmark_op, rtol=1e-12, atol=1e-300)
        assert np.array_equal(G1.transitions.indices, Gd.transitions.indices)
        np.testing.assert_allclose(Gd.transitions.data, G1.transitions.data, rtol=1e-13)
        np.testing.assert_allclose(sparse.csr_matrix(Gd.diff_aff).data, sparse.csr_matrix(G1.diff_aff).data, rtol=1e-13)
        with pytest.raises(NotImplementedError):
            Gd.extend_to_data(X[:5])
        # "auto" with a single rank stays on the single-GPU path
        Ga = graphtools_amd.Graph(X[:5000], knn=10, decay=20, n_pca=None, verbose=False, distributed="auto")
        assert Ga._dist_ranks() is None and not hasattr(Ga, "_sharded")
    finally:
        os.environ.pop("GT_SHARD_SYM_FORCE", None)
        dist.destroy_process_group()
