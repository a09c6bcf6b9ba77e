"""Worker for tests/test_dist_cpu.py: runs on every rank of a world_size-2 gloo group (CPU only).

The HIP library is replaced by ``FakeCtx`` - a numpy stand-in for gt_graph_begin/emit/finish built on the
oracle (test infrastructure) - so that the HOST orchestration of the row-sharded build
(graphtools_amd/dist.py: splits, point all-gather, triplet all-to-all, merge order) is exercised end to end
without a GPU.  The device kernels themselves are covered by the -m gpu tests."""
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


def _view(ptr, count):
    buf = (ctypes.c_char * (count * 16)).from_address(ptr)
    return np.frombuffer(buf, dtype=TRIPLET)


class FakeParams(object):
    def __init__(self, knn, decay, thresh, kernel_symm, anisotropy=0.0, theta=1.0):
        self.knn, self.decay, self.thresh, self.kernel_symm = knn, decay, thresh, kernel_symm
        self.anisotropy, self.theta = anisotropy, theta


class FakeCtx(object):
    """numpy/oracle stand-in with the call sequence of graphtools_amd._hip.Context"""

    def set_points_device(self, ptr, n, d, dtype):
        self.X = np.frombuffer((ctypes.c_char * (n * d * np.dtype(dtype).itemsize)).from_address(ptr),
                               dtype=dtype).reshape(n, d).copy()

    def graph_begin(self, params, world, rank, splits):
        self.p, self.world, self.rank, self.splits = params, world, rank, np.asarray(splits)
        r0, r1 = int(splits[rank]), int(splits[rank + 1])
        self.r0, self.r1 = r0, r1
        K0 = oracle.knn_kernel(self.X, knn=params.knn + 1, decay=params.decay, thresh=params.thresh, Y=self.X[r0:r1])
        self.K0 = sparse.coo_matrix(K0)
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
        if s == "+":
            K = (A + B) / 2
        elif s == "*":
            K = A.multiply(B)
        elif s is None:
            K = A
        else:
            raise NotImplementedError(s)
        self.K = sparse.csr_matrix(K)
        self.K.sort_indices()
        return self.K.nnz, 0


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
        g = gdist.ShardedKnnGraph(FakeCtx(), Xg.shape[0])
        local = torch.from_numpy(Xg[g.splits[rank]:g.splits[rank + 1]].copy())
        g.gather_points(local)
        nnz, _ = g.build(FakeParams(10, 20, 1e-4, symm))
        K_full, _ = oracle.knn_graph(Xg, knn=10, decay=20, kernel_symm=symm)
        K_full = sparse.csr_matrix(K_full)
        K_full.sort_indices()
        block = K_full[g.splits[rank]:g.splits[rank + 1]]
        assert (g.ctx.K != block).nnz == 0, "sharded rows differ from the single-process oracle (%s)" % symm
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
