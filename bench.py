#!/usr/bin/env python
"""Headline benchmark: graphs/sec for the kNN kernel + diffusion operator build (BASELINE.json).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], SURVEY.md 8d "C3"): synthetic `mix` data, N = 1 000 000 points, d = 64,
float32, seed 1; graphtools.Graph(X, knn=15, decay=40, thresh=1e-4) -> kernel K and diff_op P.
One step = one complete build of K and P (row norms / padded copy, kNN candidate pass, fp64 re-rank, radius
pass, affinities, symmetrisation, row normalisation) with the points already resident in HBM; results stay on
the device ("device-complete").  On one GPU the candidate pass is the symmetric one (graphtools_amd/csrc/gt_sym.hip):
a threshold-seeding launch over every row's neighbourhood, then one launch that scores every unordered pair of rows
once and tests the result for both rows.  With N > 1 GPUs the rows are sharded over the ranks: every step additionally
contains the RCCL all-gather of the point slices and the all-to-all of the transposed triplets (strong
scaling: the graph is the same size on any number of GPUs).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel =
knn_select, MFMA-bound, timed with HIP events on the library's stream; `achieved` = the algorithmic 2 N^2 d flop of the
pairwise-distance problem over the time of BOTH candidate launches, the matrix work actually executed is reported next
to it) and, at N = 1, `cpu_baseline`
(the numpy/scipy/scikit-learn oracle port timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md dense matrix peaks (256 CUs @ 2.4 GHz)
MFMA_PEAK_TFLOPS = {"f16x1": 2500.0,  # v_mfma_f32_32x32x16_f16, one chain per product (high float16 plane)
                    "f16": 2500.0,    # same instruction, split float16: 3 chains per product
                    "f32": 157.3}     # v_mfma_f32_32x32x2_f32
MFMA_CHAINS = {"f16x1": 1.0, "f16": 3.0, "f32": 1.0}
SELECT_KERNEL = {"f16x1": "knn_select_kernel<64, 8, 0, 2>", "f16": "knn_select_kernel<64, 8, 0, 1>",
                 "f32": "knn_select_kernel<64, 8, 0, 0>"}


def make_mix(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    c = max(n // 2000, 1)
    centres = rng.uniform(-10, 10, (c, d))
    labels = rng.integers(c, size=n)
    out = np.empty((n, d), dtype=dtype)
    step = 100000
    for s in range(0, n, step):   # chunked to keep the float64 temporary small
        e = min(n, s + step)
        out[s:e] = centres[labels[s:e]] + rng.standard_normal((e - s, d))
    return out


SYM_KERNEL = "knn_select_kernel<64, 8, 2, 2>"   # symmetric collect (single float16 chain), one-stage
SYM2_KERNEL = "knn_select_kernel<16, 8, 3, 2>"  # symmetric collect, two-stage: 16 features in the unit loop ...
SYM_COLD_KERNEL = "sym_cold_kernel<64>"         # ... survivors scored in full by the cold launch
SYM_SEED_KERNEL = "knn_select_kernel<64, 8, 0, 2>"   # threshold-seeding launch of the symmetric pass
BOUND_KERNELS = ("cell_ball_kernel", "cell_mask_kernel<64>", "bound_queue_kernel")   # bound pass (replaces the collect launch)


def measured_traffic(n, d, precision, world, symmetric):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per
    MI355X_MICROARCH.md + WRITE_SIZE, profiles/r*_pmc_fetch_write_per_kernel*.json).  PMC counters cannot be read
    from inside this process, so the number is only reported for the exact workload that was profiled."""
    if not (n == 1000000 and d == 64 and world == 1):
        return None
    try:
        if symmetric:
            # every launch of the candidate pass that the profiled run made: seeding, bound pass or collect, cold launch
            with open(os.path.join(ROOT, "profiles", "r2_pmc_fetch_write_per_kernel.json")) as f:
                ks = json.load(f)["kernels"]
            tot = 0.0
            for name, k in ks.items():
                if name in (SYM_SEED_KERNEL, SYM2_KERNEL, SYM_KERNEL, SYM_COLD_KERNEL) or any(name.endswith(b) or b in name for b in BOUND_KERNELS):
                    tot += (k["hbm_read_GB_per_launch_corrected_x2"] + k["hbm_write_GB_per_launch"]) * k.get("launches", 1)
            return tot * 1e9
        else:
            name = {"f16x1": "f16x1", "f16": "f16split"}[precision]
            with open(os.path.join(ROOT, "profiles", "r1_pmc_fetch_write_per_kernel_%s.json" % name)) as f:
                k = json.load(f)["kernels"][SELECT_KERNEL[precision]]
        return (k["hbm_read_GB_per_launch_corrected_x2"] + k["hbm_write_GB_per_launch"]) * 1e9
    except Exception:
        return None


def cpu_baseline(X, knn, decay, thresh, ctx, params_factory, budget_s=25.0):
    """Oracle port (numpy/scipy + the reference's scikit-learn call sites) on this host's cores.

    kNN search + radius fallback + CSR rows are timed on a block of query rows against the full database and
    scaled by N / block (row-separable work); symmetrisation + diff_op are timed at full size on the real
    unsymmetrised kernel."""
    import oracle
    from scipy import sparse

    n = X.shape[0]
    try:
        import sklearn  # noqa: F401
        engine = "sklearn"
    except Exception:
        engine = "numpy"
    m = min(1024, n)
    t_rows = None
    while True:
        t0 = time.perf_counter()
        oracle.knn_kernel(X, knn=knn + 1, decay=decay, thresh=thresh, Y=X[:m], engine=engine)
        t_rows = time.perf_counter() - t0
        if t_rows >= 0.4 * budget_s or m >= n or m >= 65536:
            break
        m = min(n, m * (4 if t_rows < 0.1 * budget_s else 2))
    t_knn_full = t_rows * (n / m)
    # sparse tail at full size on the unsymmetrised kernel produced by the device (data only, not timed)
    p, keep = params_factory(None)
    ctx.graph_build(p)
    from graphtools_amd import _hip

    d_, i_, p_ = ctx.graph_fetch_csr(_hip.CSR_K)
    K0 = sparse.csr_matrix((d_, i_, p_), shape=(n, n))
    t0 = time.perf_counter()
    K = oracle.symmetrize_kernel(K0, "+")
    oracle.kernel.diff_op_fast(sparse.csr_matrix(K))
    t_tail = time.perf_counter() - t0
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info

        threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        pass
    total = t_knn_full + t_tail
    return {
        "value": 1.0 / total, "unit": "graphs/s", "cores": int(threads), "kind": "port",
        "sample": "oracle port (%s kNN engine): kNN+affinity rows timed on %d of %d query rows against the full database "
                  "(%.2f s, scaled x%.1f), symmetrise+diff_op timed at full size (%.2f s); estimated full build %.1f s"
                  % (engine, m, n, t_rows, n / m, t_tail, total),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--npoints", type=int, default=1000000)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--knn", type=int, default=15)
    ap.add_argument("--decay", type=float, default=40.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--knn-precision", choices=["auto", "f16x1", "f16", "f32"],
                    default=os.environ.get("GT_KNN_PRECISION", "auto"),
                    help="arithmetic of the candidate pass (results are identical; see DESIGN.md); auto = the library "
                         "default: single float16 chain when the data tolerate it, else split float16")
    args = ap.parse_args()

    import torch

    from graphtools_amd import _hip
    from graphtools_amd import dist as gdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GT_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (development aid)
    distributed = world > 1 or os.environ.get("GT_BENCH_FORCE_DIST") == "1"
    if args.gpus != world:
        if rank == 0 and world > 1:
            print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        dist.init_process_group("nccl")
    n, d = args.npoints, args.dim
    thresh = 1e-4
    X = make_mix(n, d, 1)
    ctx = _hip.Context(local_rank)
    ctx.set_option("knn_precision", args.knn_precision)

    def params_factory(symm="+"):
        return ctx.make_params(args.knn, args.decay, thresh, None, 1.0, None, symm, None, 0)

    params, keep = params_factory("+")
    splits = gdist.even_row_splits(n, world)
    x_local = torch.from_numpy(X[splits[rank]:splits[rank + 1]]).to(device)   # inputs resident in HBM
    sharded = gdist.ShardedKnnGraph(ctx, n) if distributed else None
    torch.cuda.synchronize(device)

    def step():
        if distributed:
            sharded.gather_points(x_local)
            return sharded.build(params)
        ctx.set_points_device(x_local.data_ptr(), n, d, np.float32)
        return ctx.graph_build(params)

    def fence():
        torch.cuda.synchronize(device)
        ctx.sync()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    select_ms, seed_ms, cold_ms, bound_ms = [], [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nnz, flags = step()
        # stage timers were recorded with hipEvents on the library's own stream during the step
        select_ms.append(max(ctx.stage_ms("knn_select"), 0.0))   # (-1: the bound pass listed the units, no collect launch)
        bound_ms.append(max(ctx.stage_ms("sym_bound"), 0.0))
        seed_ms.append(max(ctx.stage_ms("sym_seed"), 0.0))
        cold_ms.append(max(ctx.stage_ms("sym_cold"), 0.0))
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nz = torch.tensor([float(nnz)], dtype=torch.float64, device=device)
        dist.all_reduce(nz)
        nnz_total = int(nz.item())
    else:
        nnz_total = int(nnz)
    out = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        nloc = int(splits[1] - splits[0])
        flops = 2.0 * nloc * n * d                      # algorithmic: 2*d flop per (query, database row) pair
        main = ctx.last_knn_precision()                 # what the main candidate pass actually ran on
        peak = MFMA_PEAK_TFLOPS[main]
        kst = ctx.knn_stats()
        symmetric = bool(kst["symmetric"])
        main_ms, seeding_ms, cold_launch_ms = float(np.mean(select_ms)), float(np.mean(seed_ms)), float(np.mean(cold_ms))
        two_stage = symmetric and bool(kst.get("sym_two_stage", False))
        bound_pass = two_stage and bool(kst.get("sym_bound_pass", False))
        bound_launch_ms = float(np.mean(bound_ms)) if two_stage else 0.0
        # the candidate pass = all of its launches (seeding + [bound pass] + [collect] [+ cold launch of the two-stage collect])
        avg_ms = main_ms + (seeding_ms if symmetric else 0.0) + (cold_launch_ms if two_stage else 0.0) + bound_launch_ms
        achieved = flops / (avg_ms * 1e-3) / 1e12
        if symmetric:
            # executed matrix work: every unordered pair of (padded) rows once, plus the tiles of the seeding launch
            # (256-row blocks x 128-row tiles, all d features).  One-stage collect: all d features of every pair;
            # two-stage: 16 features of every pair (1024-row query blocks) + all d of the pairs the cold launch scores
            # (64 queries x 32 rows each).  Row-sharded: every rank walks 1/world of the pieces, seeds 1/world of the blocks.
            seed_flop = 2.0 * d * 256 * 128 * kst.get("sym_seed_tiles", 0)
            if two_stage:
                n_pad = -(-n // 1024) * 1024
                nb = n_pad // 1024
                walk_tiles = 8 * (1 + (nb - 1) // 2) + (0 if nb % 2 else 8)
                cold_flop = 2.0 * d * 64 * 32 * kst.get("sym_cold_pairs", 0) * world
                if bound_pass:
                    # the cell bounds decide the units: no unit loop over the pairs, only the units left are scored
                    # (the bound kernels themselves: L^2 cell pairs x d flop, not counted)
                    collect_flop = cold_flop
                    kname = "bound pass (cell_ball / cell_mask / bound_queue kernels, no collect launch) + " + SYM_COLD_KERNEL
                else:
                    collect_flop = 2.0 * 16 * 1024 * 128 * nb * walk_tiles + cold_flop
                    kname = SYM2_KERNEL + " + " + SYM_COLD_KERNEL
            else:
                n_pad = -(-n // 256) * 256
                nb = n_pad // 256
                walk_tiles = 2 * (1 + (nb - 1) // 2) + (0 if nb % 2 else 2)
                collect_flop = 2.0 * d * 256 * 128 * nb * walk_tiles
                kname = SYM_KERNEL
            executed = (collect_flop + seed_flop) / world
            kernel_name = "%s + the threshold-seeding launch knn_select_kernel<64, 8, 0, 2> (symmetric f16x1 MFMA candidate " \
                          "pass, knn_precision=%s): every unordered pair of rows %s" % (
                              kname, args.knn_precision,
                              "decided once - by its cells' bound or by its score" if bound_pass else "scored once")
        else:
            executed = flops * MFMA_CHAINS[main]
            kernel_name = "%s (%s MFMA candidate pass, knn_precision=%s)" % (SELECT_KERNEL[main], main, args.knn_precision)
        stats = ctx.graph_stats()
        out = {
            "metric": "graphs/sec (kernel+diff_op) at N=1e6 d=64 k=15",
            "value": args.steps / elapsed,
            "unit": "graphs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"f16x1": "f16", "f16": "f16x2-split", "f32": "f32"}[main] + " MFMA candidates + f64 re-rank/affinities",
            "data": "synthetic",
            "config": {"workload": "C3: mix N=%d d=%d float32 seed=1, kNNGraph knn=%d decay=%g thresh=1e-4, "
                                   "kernel_symm='+', device-complete K and P" % (n, d, args.knn, args.decay),
                       "row_sharding": "%d rank(s) x %d rows" % (world, nloc), "nnz_K": nnz_total,
                       "radius_rows_rank0": stats["radius_rows"], "fallback_rows_rank0": stats["fallback_rows"],
                       "symmetric_candidate_pass": symmetric, "two_stage_collect": two_stage, "bound_pass": bound_pass},
            "roofline": {"kernel": kernel_name,
                         "bound": "mfma",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": measured_traffic(n, d, main, world, symmetric), "traffic_unit": "bytes/launch",
                         "avg_launch_ms": avg_ms, "main_launch_ms": main_ms, "seeding_launch_ms": seeding_ms if symmetric else 0.0,
                         "cold_launch_ms": cold_launch_ms if two_stage else 0.0, "bound_pass_ms": bound_launch_ms,
                         "algorithmic_flop_per_launch": flops,
                         "executed_mfma_flop_per_launch": executed,
                         "executed_mfma_frac": executed / (avg_ms * 1e-3) / 1e12 / peak},
            "stage_ms_last_step": {s: round(ctx.stage_ms(s), 3) for s in
                                   ("prep", "query_order", "sym_prepare", "sym_seed", "sym_bound", "knn_select", "sym_cold", "rerank", "fallback",
                                    "radius", "affinity", "symmetrize", "normalize")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X, args.knn, args.decay, thresh, ctx, params_factory)
        print(json.dumps(out))
        sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
