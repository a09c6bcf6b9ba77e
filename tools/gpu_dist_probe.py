#!/usr/bin/env python
"""Development probe: wall time of every host-side piece of the sharded build under torch.distributed (nccl).
Launch with torchrun (any world size the box has GPUs for):  python -m torch.distributed.run --nproc-per-node 1
--master-addr 127.0.0.1 --master-port 29533 tools/gpu_dist_probe.py"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _hip  # noqa: E402
from graphtools_amd import dist as gdist  # noqa: E402
from tools.gpu_perf import make_mix  # noqa: E402


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(lr)
    device = torch.device("cuda", lr)
    dist.init_process_group("nccl")
    n, d = 1000000, 64
    X = make_mix(n, d, 1)
    ctx = _hip.Context(lr)
    params, keep = ctx.make_params(15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    splits = gdist.even_row_splits(n, world)
    x_local = torch.from_numpy(X[splits[rank]:splits[rank + 1]]).to(device)
    torch.cuda.synchronize(device)
    recs = []
    for rep in range(3):
        t = {}

        def lap(name, t0):
            torch.cuda.synchronize(device)
            ctx.sync()
            t[name] = round((time.perf_counter() - t0) * 1e3, 2)

        t0 = time.perf_counter()
        full = gdist.allgather_rows(x_local, splits)
        lap("allgather", t0)
        t0 = time.perf_counter()
        ctx.set_points_device(full.data_ptr(), n, d, np.float32)
        lap("set_points", t0)
        t0 = time.perf_counter()
        send_counts = ctx.graph_begin(params, world, rank, splits)
        lap("graph_begin", t0)
        total = int(send_counts.sum())
        t0 = time.perf_counter()
        send = torch.empty(max(total, 1) * 2, dtype=torch.int64, device=device)
        lap("alloc_send", t0)
        t0 = time.perf_counter()
        ctx.graph_emit(send.data_ptr())
        lap("emit", t0)
        t0 = time.perf_counter()
        recv_counts = gdist.exchange_counts(send_counts, device)
        lap("exchange_counts", t0)
        t0 = time.perf_counter()
        recv = torch.empty(int(recv_counts.sum()) * 2, dtype=torch.int64, device=device)
        lap("alloc_recv", t0)
        t0 = time.perf_counter()
        dist.all_to_all_single(recv, send[: total * 2],
                               output_split_sizes=[int(c) * 2 for c in recv_counts],
                               input_split_sizes=[int(c) * 2 for c in send_counts])
        lap("all_to_all", t0)
        torch.cuda.synchronize(device)
        same = bool(torch.equal(recv, send[: total * 2]))
        w0 = send[: total * 2].view(-1, 2)[:, 0]
        r0 = recv.view(-1, 2)[:, 0]
        t["send_zero_words"] = int((w0 == 0).sum().item())
        t["recv_zero_words"] = int((r0 == 0).sum().item())
        t["recv_equals_send"] = same
        t0 = time.perf_counter()
        nnz, flags = ctx.graph_finish(recv.data_ptr(), int(recv_counts.sum()))
        lap("finish", t0)
        recs.append(t)
        if rank == 0:
            print(json.dumps(t), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
