#!/usr/bin/env python
"""Development probe: does torch.distributed.all_to_all_single (nccl backend = RCCL) deliver large buffers intact?"""
import os
import sys

import torch
import torch.distributed as dist


def main():
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(lr)
    device = torch.device("cuda", lr)
    dist.init_process_group("nccl")
    for mb in (64, 256, 511, 513, 600, 1024, 1150, 2047, 2049):
        n = mb * (1 << 20) // 8
        n -= n % world
        for rep in range(2):
            send = torch.arange(1, n + 1, dtype=torch.int64, device=device)
            recv = torch.zeros(n, dtype=torch.int64, device=device)
            per = n // world
            dist.all_to_all_single(recv, send, output_split_sizes=[per] * world, input_split_sizes=[per] * world)
            torch.cuda.synchronize(device)
            bad = int((recv == 0).sum().item())
            print("rank0 size %5d MB rep %d: zero words after all_to_all = %d (%.1f%%)" % (mb, rep, bad, 100.0 * bad / n), flush=True)
            del send, recv
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
