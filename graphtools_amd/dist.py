"""Row-sharded multi-GPU build: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The N x N affinity build shards by row blocks; every rank needs all N points as the database side.

  1. all-gather of the points        (each rank contributes its row slice; RCCL all-gather) - collective 1
  2. renumbering                     gt_points_cell_sort: every rank renumbers the gathered points by landmark cell (the
                                      same deterministic numbering everywhere, no communication) and owns a run of whole
                                      cells: rows ``[splits[g], splits[g+1])`` of the NEW numbering (``row_ids()`` gives
                                      the caller's row numbers; the CSR's columns are the caller's numbers already)
  3. local work                      gt_graph_shard_local: the candidate lists of the rank's own rows, collected by the
                                      rank itself - no thresholds or candidate records travel (a rank may decline and run
                                      the classic pass for its rows: nothing is shared); gt_graph_begin: re-rank ->
                                      bandwidth -> radius pass -> affinities
  3b. '+' rule: all-gather of the bandwidths (8 B per row) between the two halves of gt_graph_begin
                                      (gt_graph_bandwidth_local / gt_graph_set_bandwidths): every rank settles its mutual
                                      pairs itself, only the one-sided entries travel in step 4
  4. all-to-all of transposed triplets {row, col, value} (16 B each) bucketed by the owner of ``row`` - collectives 3
                                      (counts, 8 B per peer) and 4 (the triplets); collective 2 is the all-gather of the
                                      cell numbers inside step 2 (4 B per row: each rank assigns 1 / world of the rows)
  5. local merge + normalisation     gt_graph_finish  (anisotropy: + one all-gather of the degrees)
  landmark operator: all-reduce(sum) of the L x L partial products and the L partial row sums.

Where the renumbering does not apply (a few thousand points, more than 128 features, ``renumber=False``) the rows keep the
caller's numbering and the staged symmetric pass of rounds 2-3 runs instead (gt_graph_sym_*: 1/world of the pair scores per
rank, an all-gather of thresholds and an all-to-all of candidate records in front of step 3).

torch is used for device buffers and collectives only; all numerics are in libgraphtools_amd.so.
The communication helpers work on CPU tensors with the gloo backend too (tests/test_dist_cpu.py).
"""
import os

import numpy as np

WORDS_PER_TRIPLET = 2  # a triplet {uint32 row, uint32 col, float64 value} travels as two int64 words


def even_row_splits(n, world):
    """world+1 ascending row boundaries covering [0, n] as evenly as possible."""
    base, rem = divmod(int(n), int(world))
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def _dist():
    import torch.distributed as dist

    return dist


def exchange_counts(send_counts, device, group=None):
    """all-to-all of one int64 per peer: how many triplets each peer will send us."""
    import torch

    dist = _dist()
    world = dist.get_world_size(group)
    send = torch.as_tensor(np.asarray(send_counts, dtype=np.int64), device=device)
    recv = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(recv, send, group=group)
    return recv.cpu().numpy()


# Largest per-peer message handed to one collective call.  RCCL 2.26 as shipped with torch 2.10+rocm7.0 delivers only
# half of a send/recv pair once it exceeds 1 GiB (tools/gpu_a2a_probe.py: 1024 MB intact, 1150 MB half zeros), so
# bigger segments are moved in rounds.
MAX_PEER_BYTES = 512 << 20


def exchange_triplets(send_words, send_counts, group=None, max_peer_bytes=None):
    """all-to-all of the bucketed triplet buffer.

    send_words : int64 tensor of 2 * sum(send_counts) words, buckets in rank order
    returns (recv_words int64 tensor, recv_counts numpy int64[world])
    """
    import torch

    dist = _dist()
    world = dist.get_world_size(group)
    send_counts = np.asarray(send_counts, dtype=np.int64)
    recv_counts = exchange_counts(send_counts, send_words.device, group)
    recv = torch.empty(int(recv_counts.sum()) * WORDS_PER_TRIPLET, dtype=torch.int64, device=send_words.device)
    limit = int(max_peer_bytes or MAX_PEER_BYTES) // (8 * WORDS_PER_TRIPLET)      # triplets per peer and round
    biggest = int(max(send_counts.max(initial=0), recv_counts.max(initial=0)))
    if biggest <= limit:
        dist.all_to_all_single(
            recv, send_words,
            output_split_sizes=[int(c) * WORDS_PER_TRIPLET for c in recv_counts],
            input_split_sizes=[int(c) * WORDS_PER_TRIPLET for c in send_counts],
            group=group,
        )
        return recv, recv_counts
    # rounds of at most `limit` triplets per ordered pair; every rank derives the same number of rounds
    rounds = torch.tensor([-(-biggest // limit)], dtype=torch.int64, device=send_words.device)
    dist.all_reduce(rounds, op=dist.ReduceOp.MAX, group=group)
    send_off = np.concatenate([[0], np.cumsum(send_counts)])
    recv_off = np.concatenate([[0], np.cumsum(recv_counts)])
    for k in range(int(rounds.item())):
        s_len = np.clip(send_counts - k * limit, 0, limit)
        r_len = np.clip(recv_counts - k * limit, 0, limit)
        s_in = torch.cat([send_words[(send_off[p] + k * limit) * WORDS_PER_TRIPLET:
                                     (send_off[p] + k * limit + s_len[p]) * WORDS_PER_TRIPLET] for p in range(world)])
        r_out = torch.empty(int(r_len.sum()) * WORDS_PER_TRIPLET, dtype=torch.int64, device=send_words.device)
        dist.all_to_all_single(
            r_out, s_in,
            output_split_sizes=[int(c) * WORDS_PER_TRIPLET for c in r_len],
            input_split_sizes=[int(c) * WORDS_PER_TRIPLET for c in s_len],
            group=group,
        )
        pos = 0
        for p in range(world):
            n = int(r_len[p]) * WORDS_PER_TRIPLET
            a = (int(recv_off[p]) + k * limit) * WORDS_PER_TRIPLET
            recv[a: a + n] = r_out[pos: pos + n]
            pos += n
    return recv, recv_counts


def allgather_rows(x_local, splits, group=None):
    """all-gather of row slices of unequal length into the full [n, d] matrix (every rank gets all rows)."""
    import torch

    dist = _dist()
    world = dist.get_world_size(group)
    splits = np.asarray(splits, dtype=np.int64)
    sizes = np.diff(splits)
    d = x_local.shape[1]
    maxrows = int(sizes.max())
    if int(sizes.min()) == maxrows:
        full = torch.empty((int(splits[-1]), d), dtype=x_local.dtype, device=x_local.device)
        dist.all_gather_into_tensor(full, x_local.contiguous(), group=group)
        return full
    pad = torch.zeros((maxrows, d), dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.shape[0]] = x_local
    gathered = torch.empty((world * maxrows, d), dtype=x_local.dtype, device=x_local.device)
    dist.all_gather_into_tensor(gathered, pad, group=group)
    return torch.cat([gathered[r * maxrows: r * maxrows + int(sizes[r])] for r in range(world)], dim=0)


def allgather_vector(v_local, splits, group=None):
    return allgather_rows(v_local.reshape(-1, 1), splits, group).reshape(-1)


def allgather_csr_blocks(indptr, indices, values, row_ids, shape, device, group=None):
    """Every rank holds a CSR row block - ``row_ids`` say which rows of the full matrix, in the block's order; the blocks of
    all ranks partition the rows - and every rank receives the full matrices with their rows in natural order.

    indptr, indices : the block's structure;  values : list of value arrays on that structure (K's and P's data share one)
    returns (indptr_full, indices_full, [values_full ...]) as numpy arrays (int64 / indices' dtype / float64)
    Four + len(values) all-gathers (sizes, row numbers, row lengths, column indices, each value array)."""
    import torch

    dist = _dist()
    world = dist.get_world_size(group)
    lens = np.diff(np.asarray(indptr, dtype=np.int64))
    sizes = torch.as_tensor(np.array([len(lens), int(lens.sum())], dtype=np.int64), device=device)
    all_sizes = torch.empty(2 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    all_sizes = all_sizes.cpu().numpy().reshape(world, 2)
    row_splits = np.concatenate([[0], np.cumsum(all_sizes[:, 0])]).astype(np.int64)
    nnz_splits = np.concatenate([[0], np.cumsum(all_sizes[:, 1])]).astype(np.int64)
    if int(row_splits[-1]) != int(shape[0]):
        raise RuntimeError("allgather_csr_blocks: the ranks' blocks hold %d rows of %d" % (int(row_splits[-1]), int(shape[0])))

    def gather(a, splits, dtype):
        t = torch.as_tensor(np.ascontiguousarray(a, dtype=dtype), device=device)
        return allgather_vector(t, splits, group).cpu().numpy()

    ids = gather(row_ids, row_splits, np.int64)
    lens_all = gather(lens, row_splits, np.int64)
    idx_dtype = np.asarray(indices).dtype
    cols = gather(indices, nnz_splits, idx_dtype)
    vals = [gather(v, nnz_splits, np.float64) for v in values]
    # rows into natural order: entry k of output row r comes from position start_of(block row holding r) + k
    where = np.empty(int(shape[0]), dtype=np.int64)
    where[ids] = np.arange(len(ids), dtype=np.int64)
    if len(np.unique(ids)) != len(ids):
        raise RuntimeError("allgather_csr_blocks: a row is held by more than one rank")
    src_ptr = np.concatenate([[0], np.cumsum(lens_all)]).astype(np.int64)
    out_lens = lens_all[where]
    out_ptr = np.concatenate([[0], np.cumsum(out_lens)]).astype(np.int64)
    take = np.arange(int(out_ptr[-1]), dtype=np.int64) - np.repeat(out_ptr[:-1] - src_ptr[where], out_lens)
    return out_ptr, cols[take], [v[take] for v in vals]


def _order_after_collectives(ctx, tensor):
    """The library runs on its own stream: what a collective wrote into ``tensor`` (complete with respect to torch's
    current stream) has to be ordered before the library's next launch - on the device, without stalling the host."""
    if tensor.is_cuda and hasattr(ctx, "wait_for_stream"):
        import torch

        ctx.wait_for_stream(torch.cuda.current_stream(tensor.device).cuda_stream)


def _order_before_collectives(ctx, tensor):
    """... and the other way round: what the library has queued into ``tensor`` comes before the collective that sends it"""
    if tensor.is_cuda and hasattr(ctx, "stream_waits_for_me"):
        import torch

        ctx.stream_waits_for_me(torch.cuda.current_stream(tensor.device).cuda_stream)


class ShardedKnnGraph(object):
    """One rank's share of a row-sharded kNN graph build.

    ``ctx`` is a :class:`graphtools_amd._hip.Context` on this rank's GPU; ``x_local`` a CUDA tensor with
    this rank's rows of the data (float32/float64, C order).  After :meth:`build`, the owned CSR row block
    of K and P is resident on the device (``ctx.graph_fetch_csr`` copies it to the host).
    """

    def __init__(self, ctx, n_total, group=None, renumber=True):
        dist = _dist()
        self.ctx = ctx
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n = int(n_total)
        self.renumber = bool(renumber)
        self.renumbered = False
        # the split of the caller's rows (who contributes which slice to the all-gather; also the ownership when the points
        # keep the caller's numbering)
        self.input_splits = even_row_splits(self.n, self.world)
        self.splits = self.input_splits
        self._row_ids_all = None

    def gather_points(self, x_local):
        """RCCL all-gather of the row slices; binds the full matrix to the context and renumbers it by landmark cell
        (``self.renumbered``; ``self.splits`` then are rows of the new numbering, ``row_ids()`` maps them back)."""
        import torch

        full = allgather_rows(x_local, self.input_splits, self.group)
        _order_after_collectives(self.ctx, full)
        self._points = full
        self._device = full.device
        np_dtype = np.float32 if full.dtype == torch.float32 else np.float64
        self.renumbered = False
        self.splits = self.input_splits
        self._row_ids_all = None
        if self.renumber and hasattr(self.ctx, "points_cells_begin") and full.is_cuda:
            # the cell assignment behind the renumbering, 1 / world of it per rank: every rank assigns its slice of the
            # gathered rows, one all-gather of the cells (4 B per row), every rank sorts (deterministic: one numbering)
            r0, r1 = int(self.input_splits[self.rank]), int(self.input_splits[self.rank + 1])
            cells = torch.empty(max(r1 - r0, 1), dtype=torch.int32, device=full.device)
            if self.ctx.points_cells_begin(full.data_ptr(), full.shape[0], full.shape[1], np_dtype, r0, r1, cells.data_ptr()):
                cells_all = allgather_vector(cells[: r1 - r0], self.input_splits, self.group).contiguous()
                _order_after_collectives(self.ctx, cells_all)
                self.ctx.points_cells_finish(cells_all.data_ptr())
                self.renumbered = True
        else:
            self.ctx.set_points_device(full.data_ptr(), full.shape[0], full.shape[1], np_dtype)
            if self.renumber and hasattr(self.ctx, "points_cell_sort"):
                # (every rank assigns every row: deterministic, the same numbering everywhere without a collective)
                self.renumbered = bool(self.ctx.points_cell_sort())
        if self.renumbered:
            # (the context holds its own, renumbered copy; the gathered matrix stays for row look-ups by the caller's numbers)
            self.splits = np.asarray(self.ctx.points_shard_splits(self.world), dtype=np.int64)
        return full

    def row_ids(self):
        """the caller's row numbers of this rank's rows of K and P, in the order of the rows (numpy int64)"""
        r0, r1 = int(self.splits[self.rank]), int(self.splits[self.rank + 1])
        if not self.renumbered:
            return np.arange(r0, r1, dtype=np.int64)
        return np.asarray(self.ctx.points_row_ids(r0, r1), dtype=np.int64)

    def _all_row_ids(self):
        import torch

        if self._row_ids_all is None:
            ids = np.asarray(self.ctx.points_row_ids(0, self.n), dtype=np.int64) if self.renumbered else np.arange(self.n)
            self._row_ids_all = torch.as_tensor(ids, device=self._device)
        return self._row_ids_all

    def symmetric_candidates(self, params):
        """The symmetric candidate pass split over the ranks (gt_knn_shard.cpp): each rank scores 1/world of the
        unordered row pairs; two small collectives (thresholds, far-kept count) and one all-to-all of candidate
        records.  Returns True when the owned rows' candidate lists are ready for ``graph_begin``, False when the
        pass does not apply (every rank agrees) and ``graph_begin`` will run the classic pass."""
        import torch

        dist = _dist()
        ctx = self.ctx
        device = self._device
        if not hasattr(ctx, "graph_sym_plan"):
            return False
        ok, n_pad, sorted_splits = ctx.graph_sym_plan(params, self.world, self.rank, self.splits)
        # all-or-nothing, and only on ONE geometry: the ranks must agree on the padded row count and on the split of the
        # sorted positions (MIN over {ok, n_pad, -n_pad, split hash, -split hash}: equal iff min(x) == -min(-x))
        sig = int(np.asarray(sorted_splits, dtype=np.int64).sum() % (1 << 40)) if ok else 0
        flag = torch.tensor([1 if ok else 0, n_pad, -n_pad, sig, -sig], dtype=torch.int64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        f = [int(v) for v in flag.cpu().tolist()]
        if f[0] == 0 or f[1] != -f[2] or f[3] != -f[4]:
            return False
        rows = int(sorted_splits[self.rank + 1] - sorted_splits[self.rank])
        thr_local = torch.empty((max(rows, 1), 2), dtype=torch.float32, device=device)   # {threshold, far-kept seeds}
        stats = ctx.graph_sym_seed(thr_local.data_ptr())      # [far-kept rows, four sums behind the orphan cut]
        _order_before_collectives(ctx, thr_local)
        thr_all = allgather_rows(thr_local[:rows], sorted_splits, self.group).contiguous()
        stats_t = torch.as_tensor(np.asarray(stats, dtype=np.float64), device=device)
        dist.all_reduce(stats_t, group=self.group)            # every rank receives the same sums
        _order_after_collectives(ctx, thr_all)
        ok, send_counts = ctx.graph_sym_collect(thr_all.data_ptr(), stats_t.cpu().numpy(), self.world)
        if not ok:      # the predictor saw the summed count: the same verdict on every rank
            return False
        total = int(send_counts.sum())
        send = torch.empty(max(total, 1) * WORDS_PER_TRIPLET, dtype=torch.int64, device=device)
        ctx.graph_sym_emit(send.data_ptr() if total > 0 else 0)
        _order_before_collectives(ctx, send)
        recv, recv_counts = exchange_triplets(send[: total * WORDS_PER_TRIPLET], send_counts, self.group)
        _order_after_collectives(ctx, recv)
        n_recv = int(recv_counts.sum())
        ctx.graph_sym_finish(recv.data_ptr() if n_recv > 0 else 0, n_recv)
        return True

    def build(self, params, symmetric="auto"):
        """kernel + diffusion operator for the owned rows; returns (nnz_local, flags)."""
        import torch

        dist = _dist()
        ctx = self.ctx
        device = self._device
        # (a single rank has the whole pass in gt_graph_begin; GT_SHARD_SYM_FORCE=1 runs the staged form anyway - development)
        staged = self.world > 1 or os.environ.get("GT_SHARD_SYM_FORCE") == "1"
        if self.renumbered:
            # the rank's own rows' candidate lists, no communication (a rank that declines runs the classic pass on its own)
            self.symmetric_used = bool(symmetric) and staged and bool(ctx.graph_shard_local(params, self.world, self.rank,
                                                                                            self.splits))
        else:
            self.symmetric_used = bool(symmetric) and staged and self.symmetric_candidates(params)
        if self.world > 1 and getattr(params, "knn_max", -1) > 0 and hasattr(ctx, "graph_stage_counts"):
            # knn_max: the reference's search-expansion loop looks at the rows of the whole point set - the ranks' counts
            # for its steps are summed (one small all-reduce) and handed back before the build
            local = np.zeros(4, dtype=np.int64)
            got = np.asarray(ctx.graph_stage_counts(params, self.world, self.rank, self.splits), dtype=np.int64)
            local[: len(got)] = got
            # {counts, number of steps, -number of steps}: the sums of the counts, and MIN over the ranks of the last two -
            # every rank reported the same number of steps iff min(len) == -min(-len) (decided alike on every rank)
            tot = torch.as_tensor(local, device=device)
            dist.all_reduce(tot, group=self.group)
            steps = torch.as_tensor(np.array([len(got), -len(got)], dtype=np.int64), device=device)
            dist.all_reduce(steps, op=dist.ReduceOp.MIN, group=self.group)
            tot = tot.cpu().numpy()
            steps = steps.cpu().numpy()
            if int(steps[0]) == -int(steps[1]) and int(steps[0]) > 0:
                ctx.graph_set_stage_totals(tot[: int(steps[0])])
        bw_all = None
        self.pairs_used = False
        if staged and hasattr(ctx, "graph_bandwidth_local"):
            # the pair-resolved tail on every rank ('+' rule): the ranks' bandwidths are gathered (collective: 8 B per row)
            # between the two halves of graph_begin - a rank then settles its mutual pairs itself, only one-sided entries
            # travel.  Whether it applies follows from the parameters alone: the same answer on every rank, no vote.
            nloc = int(self.splits[self.rank + 1] - self.splits[self.rank])
            bw = torch.empty(max(nloc, 1), dtype=torch.float64, device=device)
            if ctx.graph_bandwidth_local(params, self.world, self.rank, self.splits, bw.data_ptr()):
                _order_before_collectives(ctx, bw)
                bw_all = allgather_vector(bw[:nloc], self.splits, self.group).contiguous()
                _order_after_collectives(ctx, bw_all)
                ctx.graph_set_bandwidths(bw_all.data_ptr())
                self.pairs_used = True
        send_counts = ctx.graph_begin(params, self.world, self.rank, self.splits)
        total = int(send_counts.sum())
        send = torch.empty(max(total, 1) * WORDS_PER_TRIPLET, dtype=torch.int64, device=device)
        if total > 0:
            ctx.graph_emit(send.data_ptr())
        _order_before_collectives(ctx, send)
        recv, recv_counts = exchange_triplets(send[: total * WORDS_PER_TRIPLET], send_counts, self.group)
        _order_after_collectives(ctx, recv)
        n_recv = int(recv_counts.sum())
        nnz, flags = ctx.graph_finish(recv.data_ptr() if n_recv > 0 else 0, n_recv)
        if params.anisotropy != 0.0 and self.world > 1:
            nloc = int(self.splits[self.rank + 1] - self.splits[self.rank])
            deg = torch.empty(nloc, dtype=torch.float64, device=device)
            ctx.graph_fetch_vec_device(1, deg.data_ptr())      # the owned rows' kernel degrees, device to device
            _order_before_collectives(ctx, deg)
            deg_all = allgather_vector(deg, self.splits, self.group).contiguous()
            if self.renumbered:
                # the library wants the degrees by the CALLER's row numbers (the columns of its CSR)
                by_caller = torch.empty_like(deg_all)
                by_caller[self._all_row_ids()] = deg_all
                deg_all = by_caller
            _order_after_collectives(ctx, deg_all)
            ctx.graph_anisotropy(deg_all.data_ptr())
        self._keep = (send, recv, bw_all)
        return nnz, flags

    def draw_landmarks(self, n_landmark, random_state):
        """The landmark rows of a random landmarking, ONE draw for the whole job: rank 0 draws
        (``default_rng(random_state).choice(N, L, replace=False)``, the caller's row numbers) and broadcasts the rows it drew.
        ``random_state=None`` - Graph's default - seeds from the OS, a different set in every process: ranks that each drew
        their own would label their rows against different landmarks and gather the labels as if they shared one numbering."""
        import torch

        dist = _dist()
        L = int(n_landmark)
        if self.rank == 0:
            drawn = np.random.default_rng(random_state).choice(self.n, L, replace=False).astype(np.int64)
        else:
            drawn = np.zeros(L, dtype=np.int64)
        if self.world > 1:
            t = torch.as_tensor(drawn, device=self._device)
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(t, src=src, group=self.group)
            drawn = t.cpu().numpy()
        return drawn

    def random_landmark_clusters(self, n_landmark, random_state):
        """LandmarkGraph's random landmarking (graphs.py:1200-1213) over the ranks: the landmark rows are drawn once for the
        job (:meth:`draw_landmarks`: rank 0's draw, broadcast - the caller's row numbers), every rank assigns
        its own rows to their nearest landmark (gt_nearest_landmark), one all-gather of the labels (4 B per row).
        Returns the cluster label of every row, by the caller's row numbers (numpy int32)."""
        import torch

        landmarks = self.draw_landmarks(n_landmark, random_state)
        r0, r1 = int(self.splits[self.rank]), int(self.splits[self.rank + 1])
        if self.n > 5000 and hasattr(self.ctx, "points_device") and self._points.is_cuda:
            # sklearn's euclidean_distances arithmetic (graphs.py:1210): the rank's rows as queries of a 1-NN search against the
            # L landmark rows on the MFMA path, argmin's first-index rule among the nearest that tie after the float32
            # rounding (as graphtools_amd.graphs.LandmarkGraph does on one GPU)
            from . import _hip

            lm_rows = self._points[torch.as_tensor(landmarks, device=self._device)].contiguous()
            torch.cuda.current_stream(self._device).synchronize()
            lm_ctx = _hip.Context(self._device.index or 0)
            try:
                lm_ctx.set_points_device(lm_rows.data_ptr(), lm_rows.shape[0], lm_rows.shape[1],
                                         np.float32 if lm_rows.dtype == torch.float32 else np.float64)
                k = int(min(4, n_landmark))
                d_, idx, _ = lm_ctx.knn_search_device(k, self.ctx.points_device(r0), r1 - r0)
            finally:
                lm_ctx.close()
            tie = d_ == d_[:, :1]
            own = np.where(tie, idx, np.iinfo(np.int64).max).min(axis=1).astype(np.int32)
        else:
            # small sets: scipy cdist arithmetic (mode 0); larger ones on the exact float64 kernel (mode 1)
            own = np.asarray(self.ctx.nearest_landmark(landmarks, 1 if self.n > 5000 else 0, rows=(r0, r1)), dtype=np.int32)
        labels = allgather_vector(torch.as_tensor(own, device=self._device), self.splits, self.group)
        labels = labels.cpu().numpy()
        if not self.renumbered:
            return labels
        by_caller = np.empty(self.n, dtype=np.int32)
        by_caller[self._all_row_ids().cpu().numpy()] = labels
        return by_caller

    def landmark_operator(self, clusters, n_landmark):
        """all-reduce of the partial L x L products; returns the finished landmark operator (host array).

        On the GPU the partial products never leave the device: gt_landmark_build writes M and R into ONE device buffer, RCCL
        sums it over the ranks in place, gt_landmark_scale divides in place, and the finished operator crosses PCIe once
        (round 4 fetched M and R to the host, uploaded them for the all-reduce and fetched the sum again: three trips of
        32 MB at L = 2000)."""
        import torch

        dist = _dist()
        device = self._device
        L = int(n_landmark)
        if device.type == "cuda" and hasattr(self.ctx, "landmark_build_device"):
            buf = torch.empty(L * L + L, dtype=torch.float64, device=device)
            torch.cuda.current_stream(device).synchronize()     # (the allocation is visible to the library's stream)
            tnnz = self.ctx.landmark_build_device(clusters, L, buf.data_ptr())   # (clusters: by the caller's row numbers)
            _order_before_collectives(self.ctx, buf)
            dist.all_reduce(buf, group=self.group)               # ONE all-reduce: M with the L partial row sums appended
            _order_after_collectives(self.ctx, buf)
            self.ctx.landmark_scale_device(buf.data_ptr(), L)
            self.ctx.sync()
            return buf[: L * L].reshape(L, L).cpu().numpy(), tnnz
        M, R, tnnz = self.ctx.landmark_build(clusters, n_landmark)   # (CPU stand-ins, gloo: host arrays)
        buf = torch.as_tensor(np.concatenate([np.asarray(M, dtype=np.float64).ravel(), np.asarray(R, dtype=np.float64)]),
                              device=device)
        dist.all_reduce(buf, group=self.group)
        tot = buf.cpu().numpy()
        return self.ctx.landmark_scale(tot[: L * L].reshape(L, L), tot[L * L:]), tnnz
