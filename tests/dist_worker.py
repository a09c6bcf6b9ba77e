"""Worker for tests/test_dist_cpu.py: runs on every rank of a world_size-2 gloo group (CPU only).

The HIP library is replaced by ``FakeCtx`` - a numpy stand-in for gt_graph_begin/emit/finish built on the
oracle (test infrastructure) - so that the HOST orchestration of the row-sharded build
(graphtools_amd/dist.py: splits, point all-gather, threshold all-gather / record all-to-all of the symmetric pass,
triplet all-to-all, merge order, degree all-gather of the anisotropic kernel, all-reduce of the landmark operator)
is exercised end to end without a GPU.  The device kernels themselves are covered by the -m gpu tests."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from graphtools_amd import dist as gdist  # noqa: E402

TRIPLET = np.dtype([("row", np.uint32), ("col", np.uint32), ("val", np.float64)])
SYMREC = np.dtype([("row", np.uint32), ("pad", np.uint32), ("key", np.uint64)])


def _view(ptr, count):
    buf = (ctypes.c_char * (count * 16)).from_address(ptr)
    return np.frombuffer(buf, dtype=TRIPLET)


class FakeParams(object):
    def __init__(self, knn, decay, thresh, kernel_symm, anisotropy=0.0, theta=1.0, knn_max=-1):
        self.knn, self.decay, self.thresh, self.kernel_symm = knn, decay, thresh, kernel_symm
        self.anisotropy, self.theta, self.knn_max = anisotropy, theta, knn_max


def _plain_params(p):
    """the ctypes parameter block the product hands to the library, in the stand-in's terms"""
    if isinstance(p, FakeParams):
        return p
    symm = {0: None, 1: "+", 2: "*"}[int(p.kernel_symm)]
    decay = None if p.decay != p.decay else float(p.decay)
    return FakeParams(int(p.knn), decay, float(p.thresh), symm, anisotropy=float(p.anisotropy), theta=float(p.theta),
                      knn_max=int(p.knn_max))


class FakeCtx(object):
    """numpy/oracle stand-in with the call sequence of graphtools_amd._hip.Context"""

    # ---- what graphtools_amd.graphs asks of a context around a sharded build ----
    def set_option(self, name, value):
        pass

    def stage_ms(self, stage):
        return 0.0

    def graph_fetch_csr(self, which, structure=True):
        M = self.K
        if which == 1:   # P: rows of K over their sums
            M = sparse.csr_matrix(sparse.diags(1.0 / np.asarray(self.K.sum(axis=1)).ravel()) @ self.K)
            M.sort_indices()
        return M.data.copy(), M.indices.astype(np.int32), M.indptr.astype(np.int64)

    def graph_fetch_vec(self, which):
        assert which == 1
        return np.asarray(self.K.sum(axis=1)).ravel()

    def nearest_landmark(self, landmarks, mode, rows=None):
        from scipy.spatial.distance import cdist

        r0, r1 = rows
        X0 = self.X0 if self.perm is None else self.X0   # landmark rows are the caller's row numbers
        return np.argmin(cdist(self.X[r0:r1], X0[np.asarray(landmarks)]), axis=1).astype(np.int32)

    def landmark_fetch_transitions(self, tnnz):
        T = sparse.csr_matrix(self._T / self._T.sum(axis=1)[:, None])
        T.sort_indices()
        return T.data.copy(), T.indices.astype(np.int32), T.indptr.astype(np.int64)

    def set_points_device(self, ptr, n, d, dtype):
        self.X = np.frombuffer((ctypes.c_char * (n * d * np.dtype(dtype).itemsize)).from_address(ptr),
                               dtype=dtype).reshape(n, d).copy()
        self.X0 = self.X   # by the caller's row numbers

    # ---- the sharded symmetric candidate pass (gt_graph_sym_*): a miniature with the same call sequence and buffer
    # contracts - thresholds are position numbers, every rank contributes a known set of records per row - so that the
    # host side (flag agreement, threshold all-gather, far-count all-reduce, record all-to-all) is checked end to end
    sym_applies = True      # what this rank's plan answers
    sym_refuse_at_collect = False
    sym_ready = False
    sym_consumed = False
    calls = ()

    @staticmethod
    def _sym_records(sender, n, world):
        """(row, key) pairs rank `sender` collected: row j gets (sender + j) % 3 of them"""
        rows = np.repeat(np.arange(n), (sender + np.arange(n)) % 3)
        keys = (np.uint64(sender) << np.uint64(40)) | (rows.astype(np.uint64) << np.uint64(8)) | \
            (np.arange(len(rows), dtype=np.uint64) & np.uint64(0xFF))
        return rows, keys

    def graph_sym_plan(self, params, world, rank, splits):
        self.calls = self.calls + ("plan",)
        self.world, self.rank, self.splits = world, rank, np.asarray(splits)
        n = self.X.shape[0]
        self.n_pad = -(-n // 8) * 8
        nb = self.n_pad // 8
        self.sorted_splits = np.array([(nb * r // world) * 8 for r in range(world + 1)], dtype=np.int64)
        return self.sym_applies, self.n_pad, self.sorted_splits

    def graph_sym_seed(self, ptr):
        self.calls = self.calls + ("seed",)
        p0, p1 = int(self.sorted_splits[self.rank]), int(self.sorted_splits[self.rank + 1])
        out = np.frombuffer((ctypes.c_char * ((p1 - p0) * 8)).from_address(ptr), dtype=np.float32).reshape(-1, 2)
        out[:, 0] = 0.5 * np.arange(p0, p1)
        out[:, 1] = np.arange(p0, p1) % 5
        return np.array([10.0 + self.rank, 1.5 * (self.rank + 1), 2.0, 0.25, 1.0], dtype=np.float64)

    def graph_sym_collect(self, ptr, stats_total, world):
        far_total = int(round(stats_total[0]))
        assert abs(stats_total[1] - sum(1.5 * (r + 1) for r in range(world))) < 1e-12 and stats_total[2] == 2.0 * world
        assert stats_total[3] == 0.25 * world and stats_total[4] == 1.0 * world
        self.calls = self.calls + ("collect",)
        thr = np.frombuffer((ctypes.c_char * (self.n_pad * 8)).from_address(ptr), dtype=np.float32).reshape(-1, 2)
        assert np.array_equal(thr[:, 0], 0.5 * np.arange(self.n_pad, dtype=np.float32)), "thresholds were not gathered in order"
        assert np.array_equal(thr[:, 1], np.arange(self.n_pad) % 5)
        assert far_total == sum(10 + r for r in range(world)), far_total
        if self.sym_refuse_at_collect:
            return False, np.zeros(world, dtype=np.int64)
        rows, keys = self._sym_records(self.rank, self.X.shape[0], world)
        owner = np.searchsorted(self.splits, rows, side="right") - 1
        order = np.argsort(owner, kind="stable")
        self._sym_send = (rows[order] - self.splits[owner[order]], keys[order])
        return True, np.bincount(owner, minlength=world).astype(np.int64)

    def graph_sym_emit(self, ptr):
        self.calls = self.calls + ("emit",)
        rl, keys = self._sym_send
        out = np.frombuffer((ctypes.c_char * (len(rl) * 16)).from_address(ptr), dtype=SYMREC) if len(rl) else np.zeros(0, SYMREC)
        out["row"][:] = rl
        out["pad"][:] = 0
        out["key"][:] = keys

    def graph_sym_finish(self, ptr, n_recv):
        self.calls = self.calls + ("finish",)
        got = np.frombuffer((ctypes.c_char * (n_recv * 16)).from_address(ptr), dtype=SYMREC) if n_recv else np.zeros(0, SYMREC)
        r0, r1 = int(self.splits[self.rank]), int(self.splits[self.rank + 1])
        assert np.all(got["row"] < r1 - r0), "received a record for a foreign row"
        want = []
        for s in range(self.world):
            rows, keys = self._sym_records(s, self.X.shape[0], self.world)
            sel = (rows >= r0) & (rows < r1)
            want.append(np.stack([(rows[sel] - r0).astype(np.uint64), keys[sel]], axis=1))
        want = np.concatenate(want)
        have = np.stack([got["row"].astype(np.uint64), got["key"]], axis=1)
        assert np.array_equal(want[np.lexsort(want.T[::-1])], have[np.lexsort(have.T[::-1])]), "records lost or duplicated"
        self.sym_ready = True

    # ---- cell-sorted renumbering (gt_points_cell_sort / gt_points_shard_splits / gt_points_row_ids /
    # gt_graph_shard_local): a miniature with the library's contracts - the bound points become a deterministic
    # permutation of the caller's, rows and triplets are in the new numbering, the finished rows carry the caller's columns
    renumber_applies = True
    local_applies = True
    perm = None

    def points_cell_sort(self):
        self.calls = self.calls + ("cell_sort",)
        if not self.renumber_applies:
            return False
        key = np.floor(self.X[:, 0].astype(np.float64) * 2.0)   # "cells": coarse bins of the first coordinate
        self.perm = np.argsort(key, kind="stable")
        self.X = self.X[self.perm]
        return True

    def points_shard_splits(self, world):
        n = self.X.shape[0]
        nb = -(-n // 8)
        s = np.array([min(n, (nb * r // world) * 8) for r in range(world + 1)], dtype=np.int64)
        s[-1] = n
        return s

    def points_row_ids(self, r0, r1):
        return (self.perm[r0:r1] if self.perm is not None else np.arange(r0, r1)).astype(np.int32)

    def graph_shard_local(self, params, world, rank, splits):
        params = _plain_params(params)
        self.calls = self.calls + ("shard_local",)
        assert self.perm is not None and np.array_equal(splits, self.points_shard_splits(world))
        self.sym_ready = self.local_applies
        return self.local_applies

    # ---- the pair-resolved tail of a sharded build (gt_graph_bandwidth_local / gt_graph_set_bandwidths): the rank's
    # bandwidths go out, the bandwidths of ALL rows come back in the order of the context's rows; the rank then settles its
    # mutual pairs itself and sends the one-sided entries only.  The stand-in checks what dist.py delivered against the
    # bandwidths of a whole-set build and looks the partners' entries up in that build's kernel.
    pairs_enabled = True      # (the library's option symmetrize_pairs_shard)
    pairs_calls = ()
    bw_all = None

    def graph_bandwidth_local(self, params, world, rank, splits, ptr):
        p = _plain_params(params)
        self.pairs_calls = self.pairs_calls + ("bandwidth_local",)
        self.bw_all = None
        if not self.pairs_enabled or p.kernel_symm != "+" or p.decay is None or p.anisotropy != 0 or p.knn_max > 0:
            return False      # (the parameters alone decide: the same answer on every rank)
        r0, r1 = int(splits[rank]), int(splits[rank + 1])
        _, info = oracle.knn_kernel(self.X, knn=p.knn + 1, decay=p.decay, thresh=p.thresh, Y=self.X[r0:r1], return_search=True)
        out = np.frombuffer((ctypes.c_char * ((r1 - r0) * 8)).from_address(ptr), dtype=np.float64)
        out[:] = info["bandwidth"]
        self._half = (p.knn, p.decay, p.thresh, world, rank, tuple(int(v) for v in splits))
        return True

    def graph_set_bandwidths(self, ptr):
        self.pairs_calls = self.pairs_calls + ("set_bandwidths",)
        n = self.X.shape[0]
        self.bw_all = np.frombuffer((ctypes.c_char * (n * 8)).from_address(ptr), dtype=np.float64).copy()

    def graph_begin(self, params, world, rank, splits):
        params = _plain_params(params)
        self.sym_consumed, self.sym_ready = self.sym_ready, False
        self.p, self.world, self.rank, self.splits = params, world, rank, np.asarray(splits)
        r0, r1 = int(splits[rank]), int(splits[rank + 1])
        self.r0, self.r1 = r0, r1
        K0 = oracle.knn_kernel(self.X, knn=params.knn + 1, decay=params.decay, thresh=params.thresh, Y=self.X[r0:r1])
        self.K0 = sparse.coo_matrix(K0)
        self.pairs = self.bw_all is not None
        if self.pairs:
            assert self._half == (params.knn, params.decay, params.thresh, world, rank, tuple(int(v) for v in splits)), \
                "graph_begin is not the build graph_bandwidth_local started"
            full, info = oracle.knn_kernel(self.X, knn=params.knn, decay=params.decay, thresh=params.thresh, return_search=True)
            assert np.array_equal(self.bw_all, info["bandwidth"]), "the gathered bandwidths are not those of the context's rows, in order"
            self.bw_all = None
            full = sparse.csr_matrix(full)
            partner = np.asarray(full[self.K0.col, self.K0.row + r0]).ravel()   # K0[j, i] of every own entry (i, j)
            mutual = partner != 0
            # settled here: (K[i, j] + K[j, i]) / 2, never sent; the one-sided entries travel (own rows' too)
            self.settled = sparse.csr_matrix(((self.K0.data[mutual] + partner[mutual]) / 2,
                                              (self.K0.row[mutual], self.K0.col[mutual])), shape=(r1 - r0, self.X.shape[0]))
            self.K0 = sparse.coo_matrix((self.K0.data[~mutual], (self.K0.row[~mutual], self.K0.col[~mutual])),
                                        shape=self.K0.shape)
        owner = np.searchsorted(self.splits, self.K0.col, side="right") - 1
        self.owner = owner
        if params.kernel_symm is None:
            return np.zeros(world, dtype=np.int64)
        return np.bincount(owner, minlength=world).astype(np.int64)

    def graph_emit(self, ptr):
        order = np.argsort(self.owner, kind="stable")
        out = _view(ptr, len(order))
        out["row"][:] = self.K0.col[order]
        out["col"][:] = self.K0.row[order] + self.r0
        out["val"][:] = self.K0.data[order]

    def graph_finish(self, ptr, n_recv):
        n = self.X.shape[0]
        nloc = self.r1 - self.r0
        A = sparse.csr_matrix((self.K0.data, (self.K0.row, self.K0.col)), shape=(nloc, n))
        if n_recv > 0:
            t = _view(ptr, n_recv)
            assert np.all((t["row"] >= self.r0) & (t["row"] < self.r1)), "received a triplet for a foreign row"
            B = sparse.csr_matrix((t["val"], (t["row"].astype(np.int64) - self.r0, t["col"].astype(np.int64))),
                                  shape=(nloc, n))
        else:
            B = sparse.csr_matrix((nloc, n))
        s = self.p.kernel_symm
        if s == "+" and self.pairs:
            # no pair meets its partner in a union row: own one-sided, received and settled entries are disjoint
            assert A.multiply(B).nnz == 0 and A.multiply(self.settled).nnz == 0 and B.multiply(self.settled).nnz == 0, \
                "a pair met its partner in a union row"
            K = self.settled + (A + B) / 2
        elif s == "+":
            K = (A + B) / 2
        elif s == "*":
            K = A.multiply(B)
        elif s is None:
            K = A
        else:
            raise NotImplementedError(s)
        if self.perm is not None:   # the caller's column numbers (relabelled in the final sort of the tail)
            K = sparse.coo_matrix(K)
            K = sparse.csr_matrix((K.data, (K.row, self.perm[K.col])), shape=K.shape)
        self.K = sparse.csr_matrix(K)
        self.K.sort_indices()
        return self.K.nnz, 0

    # ---- knn_max: the counts of the search-expansion loop are summed over the ranks before the build ----
    stage_totals = None

    def graph_stage_counts(self, params, world, rank, splits):
        self.calls = self.calls + ("stage_counts",)
        return np.array([100 + rank, 7 * (rank + 1)], dtype=np.int64)

    def graph_set_stage_totals(self, totals):
        self.calls = self.calls + ("stage_totals",)
        self.stage_totals = np.asarray(totals, dtype=np.int64).copy()

    # ---- anisotropy: the degrees of ALL rows are needed (dist.py all-gathers the owned slices) ----
    def graph_fetch_vec_device(self, which, ptr):
        assert which == 1
        self.calls = self.calls + ("fetch_degree",)
        nloc = self.r1 - self.r0
        out = np.frombuffer((ctypes.c_char * (nloc * 8)).from_address(ptr), dtype=np.float64)
        out[:] = np.asarray(self.K.sum(axis=1)).ravel()

    def graph_anisotropy(self, ptr):
        self.calls = self.calls + ("anisotropy",)
        n = self.X.shape[0]
        d = np.frombuffer((ctypes.c_char * (n * 8)).from_address(ptr), dtype=np.float64).copy()
        K = self.K.tocoo()
        own = self.perm[K.row + self.r0] if self.perm is not None else K.row + self.r0   # (d: by the caller's row numbers)
        K.data = K.data / ((d[own] * d[K.col]) ** self.p.anisotropy)
        self.K = sparse.csr_matrix(K)
        self.K.sort_indices()

    # ---- landmark operator: partial L x L products of the owned rows (gt_landmark.hip) ----
    def landmark_build(self, clusters, n_landmark):
        self.calls = self.calls + ("landmark_build",)
        n = self.X.shape[0]
        S = sparse.csr_matrix((np.ones(n), (np.asarray(clusters), np.arange(n))), shape=(n_landmark, n))
        T = np.asarray((self.K @ S.T).todense())            # [nloc, L]: row i of K summed per cluster
        self._T = T
        c = T.sum(axis=1)
        M = T.T @ (T / c[:, None])
        return M, T.sum(axis=0), int((T != 0).sum())

    def landmark_scale(self, M, R):
        self.calls = self.calls + ("landmark_scale",)
        return np.asarray(M) / np.asarray(R)[:, None]


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(123)

    # 1. splits
    sp = gdist.even_row_splits(1001, world)
    assert sp[0] == 0 and sp[-1] == 1001 and np.all(np.diff(sp) >= 1001 // world)

    # 2. all-gather of unequal row slices
    X = rng.standard_normal((1001, 7)).astype(np.float32)   # same on every rank (same seed)
    full = gdist.allgather_rows(torch.from_numpy(X[sp[rank]:sp[rank + 1]].copy()), sp)
    assert np.array_equal(full.numpy(), X)
    sp_even = gdist.even_row_splits(1000, world)
    full = gdist.allgather_rows(torch.from_numpy(X[sp_even[rank]:sp_even[rank + 1]].copy()), sp_even)
    assert np.array_equal(full.numpy(), X[:1000])

    # 3. triplet all-to-all: every triplet reaches the owner of its row, nothing is lost or duplicated
    lrng = np.random.default_rng(1000 + rank)
    m = 5000 + 37 * rank
    rows = lrng.integers(0, 1001, size=m).astype(np.uint32)
    trip = np.zeros(m, dtype=TRIPLET)
    trip["row"], trip["col"], trip["val"] = rows, lrng.integers(0, 1001, size=m), lrng.standard_normal(m)
    owner = np.searchsorted(sp, rows, side="right") - 1
    order = np.argsort(owner, kind="stable")
    send = torch.from_numpy(trip[order].view(np.int64).copy())
    counts = np.bincount(owner, minlength=world)
    recv, rc = gdist.exchange_triplets(send, counts)
    got = recv.numpy().view(TRIPLET)
    assert len(got) == rc.sum()
    assert np.all((got["row"] >= sp[rank]) & (got["row"] < sp[rank + 1]))
    tot = torch.tensor([float(trip["val"].sum()), float(got["val"].sum()), float(m), float(len(got))], dtype=torch.float64)
    dist.all_reduce(tot)
    assert abs(tot[0] - tot[1]) < 1e-9 and tot[2] == tot[3]
    # the same exchange in rounds of at most 100 triplets per ordered pair (the path that keeps every message
    # below RCCL's 1 GiB limit) delivers the identical buffer
    recv2, rc2 = gdist.exchange_triplets(send, counts, max_peer_bytes=100 * 16)
    assert np.array_equal(rc2, rc) and torch.equal(recv2, recv)

    # 4. end-to-end sharded build through ShardedKnnGraph with the numpy stand-in context
    from tests_helpers_mix import make_mix  # noqa: E402  (injected by the launcher)

    Xg = make_mix(900, 20, 5)
    for symm in ("+", "*", None):
        g = gdist.ShardedKnnGraph(FakeCtx(), Xg.shape[0], renumber=False)   # (the staged pass of the caller's numbering)
        local = torch.from_numpy(Xg[g.splits[rank]:g.splits[rank + 1]].copy())
        g.gather_points(local)
        nnz, _ = g.build(FakeParams(10, 20, 1e-4, symm))
        K_full, _ = oracle.knn_graph(Xg, knn=10, decay=20, kernel_symm=symm)
        K_full = sparse.csr_matrix(K_full)
        K_full.sort_indices()
        block = K_full[g.splits[rank]:g.splits[rank + 1]]
        assert (g.ctx.K != block).nnz == 0, "sharded rows differ from the single-process oracle (%s)" % symm
        assert g.symmetric_used and g.ctx.sym_consumed, "the symmetric candidate stages did not run"
        assert g.ctx.calls == ("plan", "seed", "collect", "emit", "finish"), g.ctx.calls

    # 5. the symmetric stages are all-or-nothing: one rank's plan declines -> nobody seeds; the predictor refuses at the
    #    collect stage (same summed count everywhere) -> nobody exchanges; the build itself is unaffected
    for scenario in ("plan", "collect", "off"):
        c = FakeCtx()
        c.sym_applies = not (scenario == "plan" and rank == 1)
        c.sym_refuse_at_collect = scenario == "collect"
        g = gdist.ShardedKnnGraph(c, Xg.shape[0], renumber=False)
        g.gather_points(torch.from_numpy(Xg[g.splits[rank]:g.splits[rank + 1]].copy()))
        g.build(FakeParams(10, 20, 1e-4, "+"), symmetric=False if scenario == "off" else "auto")
        assert not g.symmetric_used and not c.sym_consumed
        assert c.calls == {"plan": ("plan",), "collect": ("plan", "seed", "collect"), "off": ()}[scenario], (scenario, c.calls)
        K_full = sparse.csr_matrix(oracle.knn_graph(Xg, knn=10, decay=20, kernel_symm="+")[0])
        K_full.sort_indices()
        assert (c.K != K_full[g.splits[rank]:g.splits[rank + 1]]).nnz == 0
    # 5b. knn_max: one all-reduce of the ranks' loop counts in front of the build (sixth collective)
    c = FakeCtx()
    g = gdist.ShardedKnnGraph(c, Xg.shape[0], renumber=False)
    g.gather_points(torch.from_numpy(Xg[g.splits[rank]:g.splits[rank + 1]].copy()))
    g.build(FakeParams(10, 20, 1e-4, "+", knn_max=60), symmetric=False)
    assert c.calls[:2] == ("stage_counts", "stage_totals"), c.calls
    assert np.array_equal(c.stage_totals, [sum(100 + r for r in range(world)), sum(7 * (r + 1) for r in range(world))])

    # 5c. the DEFAULT flow: cell-sorted renumbering - three collectives (points all-gather, triplet counts, triplets), the
    #     rank's rows are rows of the new numbering, row_ids() gives the caller's; a rank may decline the local pass alone
    for symm, pairs_on in (("+", True), ("+", False), ("*", True), (None, True)):
        for scenario in ("local", "rank1 declines", "renumbering declines"):
            c = FakeCtx()
            c.pairs_enabled = pairs_on
            c.local_applies = not (scenario == "rank1 declines" and rank == 1)
            c.renumber_applies = scenario != "renumbering declines"
            g = gdist.ShardedKnnGraph(c, Xg.shape[0])
            ins = g.input_splits
            g.gather_points(torch.from_numpy(Xg[ins[rank]:ins[rank + 1]].copy()))
            assert g.renumbered == c.renumber_applies
            g.build(FakeParams(10, 20, 1e-4, symm))
            K_full = sparse.csr_matrix(oracle.knn_graph(Xg, knn=10, decay=20, kernel_symm=symm)[0])
            K_full.sort_indices()
            ids = g.row_ids()
            assert len(ids) == g.splits[rank + 1] - g.splits[rank]
            assert (c.K != K_full[ids]).nnz == 0, "rows of the renumbered build differ from the single-process oracle (%s, %s)" % (symm, scenario)
            # the pair-resolved tail: one more collective (the bandwidths) for the '+' rule, whatever the candidate pass was;
            # any other rule (or the option off) answers "no" on every rank and the build runs the general way
            want_pairs = symm == "+" and pairs_on
            assert c.pairs_calls == (("bandwidth_local", "set_bandwidths") if want_pairs else ("bandwidth_local",)), c.pairs_calls
            assert g.pairs_used == want_pairs and c.pairs == want_pairs
            if c.renumber_applies:
                assert c.calls == ("cell_sort", "shard_local"), c.calls
                assert g.symmetric_used == c.local_applies and c.sym_consumed == c.local_applies
                # the ranks' row sets partition the caller's rows
                mine = torch.zeros(Xg.shape[0], dtype=torch.int64)
                mine[torch.from_numpy(ids)] = 1
                dist.all_reduce(mine)
                assert bool((mine == 1).all())
            else:
                assert c.calls == ("cell_sort", "plan", "seed", "collect", "emit", "finish"), c.calls

    # 6. anisotropy: the owned degrees are all-gathered (fourth collective), then applied to the owned block
    g = gdist.ShardedKnnGraph(FakeCtx(), Xg.shape[0])
    g.gather_points(torch.from_numpy(Xg[g.input_splits[rank]:g.input_splits[rank + 1]].copy()))
    g.build(FakeParams(10, 20, 1e-4, "+", anisotropy=0.5))
    assert g.ctx.calls[-2:] == ("fetch_degree", "anisotropy"), g.ctx.calls
    K_an = sparse.csr_matrix(oracle.knn_graph(Xg, knn=10, decay=20, kernel_symm="+", anisotropy=0.5)[0])
    K_an.sort_indices()
    blk = K_an[g.row_ids()]
    assert np.array_equal(blk.indptr, g.ctx.K.indptr) and np.array_equal(blk.indices, g.ctx.K.indices)
    np.testing.assert_allclose(g.ctx.K.data, blk.data, rtol=1e-13, atol=0)

    # 7. landmark operator: all-reduce (fifth collective) of the partial L x L products and row sums
    g = gdist.ShardedKnnGraph(FakeCtx(), Xg.shape[0])
    g.gather_points(torch.from_numpy(Xg[g.input_splits[rank]:g.input_splits[rank + 1]].copy()))
    g.build(FakeParams(10, 20, 1e-4, "+"))
    L = 12
    clusters = np.random.default_rng(77).integers(0, L, size=Xg.shape[0])
    clusters[:L] = np.arange(L)     # every label occurs
    op, tnnz = g.landmark_operator(clusters, L)
    assert g.ctx.calls[-2:] == ("landmark_build", "landmark_scale")
    K_full = sparse.csr_matrix(oracle.knn_graph(Xg, knn=10, decay=20, kernel_symm="+")[0])
    op_ref, _ = oracle.landmark_operator(K_full, clusters)
    np.testing.assert_allclose(op, op_ref, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(op.sum(axis=1), 1.0, rtol=1e-12)
    # 8. the same through the package's own boundary: every rank calls graphtools_amd.Graph(X, ..., distributed=True) with the
    #    same data and gets the FULL K / P of a single-process build (blocks all-gathered, rows back in the caller's order),
    #    its own rows as K_local / local_rows, the gathered degrees, and - landmark graphs - the operator and transitions
    import graphtools_amd
    from graphtools_amd import graphs as ggraphs

    for symm, aniso in (("+", 0), ("*", 0), ("+", 0.5)):
        G = graphtools_amd.Graph(Xg, knn=10, decay=20, kernel_symm=symm, anisotropy=aniso, distributed=True, initialize=False,
                                 n_pca=None, verbose=False)
        assert isinstance(G, ggraphs.kNNGraph)
        G._hip_ctx = FakeCtx()
        K_ref = sparse.csr_matrix(oracle.knn_graph(Xg, knn=10, decay=20, kernel_symm=symm, anisotropy=aniso)[0])
        K_ref.sort_indices()
        K = G.K
        assert K.shape == K_ref.shape and np.array_equal(K.indptr, K_ref.indptr) and np.array_equal(K.indices, K_ref.indices)
        np.testing.assert_allclose(K.data, K_ref.data, rtol=1e-13, atol=0)
        P_ref = sparse.csr_matrix(sparse.diags(1.0 / np.asarray(K_ref.sum(axis=1)).ravel()) @ K_ref)
        P_ref.sort_indices()
        np.testing.assert_allclose(G.P.data, P_ref.data, rtol=1e-12, atol=0)
        assert np.array_equal(G.diff_op.indices, K_ref.indices)
        np.testing.assert_allclose(G.kernel_degree.ravel(), np.asarray(K_ref.sum(axis=1)).ravel(), rtol=1e-12)
        assert (G.K_local != K[G.local_rows]).nnz == 0 and G.K_local.shape == (len(G.local_rows), Xg.shape[0])
        mine = torch.zeros(Xg.shape[0], dtype=torch.int64)
        mine[torch.from_numpy(np.asarray(G.local_rows))] = 1
        dist.all_reduce(mine)
        assert bool((mine == 1).all()), "the ranks' rows do not partition the data"
        da = G.diff_aff
        d = np.asarray(K_ref.sum(axis=1)).ravel()
        Kc = K_ref.tocoo()
        np.testing.assert_allclose(sparse.csr_matrix(da).data, sparse.csr_matrix((Kc.data / np.sqrt(d[Kc.row] * d[Kc.col]), (Kc.row, Kc.col)), shape=K_ref.shape).data, rtol=1e-12)
        for call in (G.build_kernel, G.diff_op_torch, lambda: G.extend_to_data(Xg[:5])):
            try:
                call()
            except NotImplementedError:
                pass
            else:
                raise AssertionError("a sharded graph answered a single-device request")
    Gl = graphtools_amd.Graph(Xg, knn=10, decay=20, n_landmark=12, random_landmarking=True, random_state=3, distributed=True,
                              initialize=False, n_pca=None, verbose=False)
    Gl._hip_ctx = FakeCtx()
    K_full = sparse.csr_matrix(oracle.knn_graph(Xg, knn=10, decay=20, kernel_symm="+")[0])
    lm = np.random.default_rng(3).choice(Xg.shape[0], 12, replace=False)
    from scipy.spatial.distance import cdist

    want_clusters = np.argmin(cdist(Xg, Xg[lm]), axis=1)
    assert np.array_equal(np.asarray(Gl.clusters), want_clusters)
    op_ref, _ = oracle.landmark_operator(K_full, want_clusters)
    np.testing.assert_allclose(Gl.landmark_op, op_ref, rtol=1e-12, atol=1e-300)
    S = sparse.csr_matrix((np.ones(Xg.shape[0]), (np.arange(Xg.shape[0]), want_clusters)), shape=(Xg.shape[0], 12))
    T_ref = np.asarray((K_full @ S).todense())
    T_ref = T_ref / T_ref.sum(axis=1)[:, None]
    np.testing.assert_allclose(np.asarray(Gl.transitions.todense()), T_ref, rtol=1e-12, atol=1e-300)
    try:
        graphtools_amd.Graph(Xg, knn=10, decay=20, n_landmark=12, distributed=True, initialize=False, n_pca=None)._assign_clusters()
    except NotImplementedError:
        pass
    else:
        raise AssertionError("spectral landmarking answered on a sharded graph")
    # 9. random_state=None (Graph's default): ONE landmark draw for the job - every rank labels its rows against the same
    #    landmark rows (rank 0's draw, broadcast), so the gathered labels are the argmin against that one set
    Gn = graphtools_amd.Graph(Xg, knn=10, decay=20, n_landmark=12, random_landmarking=True, distributed=True,
                              initialize=False, n_pca=None, verbose=False)
    assert Gn.random_state is not None, "the job seed was not drawn"
    seeds = [None] * world
    dist.all_gather_object(seeds, int(Gn.random_state))
    assert len(set(seeds)) == 1, "the ranks hold different seeds: %r" % (seeds,)
    Gn._hip_ctx = FakeCtx()
    Gn.K
    drawn = Gn._sharded.draw_landmarks(12, None)        # (None on purpose: rank 0's OS-seeded draw must reach everybody)
    every = [None] * world
    dist.all_gather_object(every, drawn.tolist())
    assert all(e == every[0] for e in every), "the ranks drew different landmark sets"
    cl = np.asarray(Gn._sharded.random_landmark_clusters(12, None))
    lm_used = [None] * world
    dist.all_gather_object(lm_used, cl.tolist())
    assert all(c == lm_used[0] for c in lm_used), "the ranks disagree on the labels"
    # coherent clusters: every label is the argmin against ONE 12-row landmark set; a landmark is its own nearest, so the
    # set can be read back from the labels (the row of cluster j whose distance to itself is 0)
    assert len(np.unique(cl)) == 12
    # 10. n_pca with random_state=None: the randomized projection is the same on every rank (one seed), so the gathered
    #     point set is ONE projection - the data_nu of the ranks agree bit for bit
    Xw = np.random.default_rng(5).standard_normal((300, 40)).astype(np.float32)   # same on every rank
    Gp = graphtools_amd.Graph(Xw, knn=5, decay=20, n_pca=6, distributed=True, initialize=False, verbose=False)
    sums = [None] * world
    dist.all_gather_object(sums, np.asarray(Gp.data_nu, dtype=np.float64).tobytes())
    assert all(s == sums[0] for s in sums), "the ranks reduced the data with different random projections"
    # 11. a row-sharded graph pickles: the device side (context, ShardedKnnGraph, process group) stays behind, K and P travel
    import pickle

    blob = pickle.dumps(Gl)
    Gr = pickle.loads(blob)
    assert not hasattr(Gr, "_sharded") and not hasattr(Gr, "group")
    assert (Gr.K != Gl.K).nnz == 0 and (Gr.P != Gl.P).nnz == 0
    np.testing.assert_array_equal(Gr.landmark_op, Gl.landmark_op)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
