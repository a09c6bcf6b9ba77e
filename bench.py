#!/usr/bin/env python
"""Headline benchmark: graphs/sec for the kNN kernel + diffusion operator build (BASELINE.json).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], SURVEY.md 8d "C3"): synthetic `mix` data, N = 1 000 000 points, d = 64,
float32, seed 1; graphtools.Graph(X, knn=15, decay=40, thresh=1e-4) -> kernel K and diff_op P.
One step = one complete build of K and P (row norms / padded copy, kNN candidate pass, fp64 re-rank, radius
pass, affinities, symmetrisation, row normalisation) with the points already resident in HBM; results stay on
the device ("device-complete").  With N > 1 GPUs the rows are sharded over the ranks: every step additionally
contains the RCCL all-gather of the point slices and the exchanges of the sharded build (strong scaling: the graph
is the same size on any number of GPUs).

Prints ONE JSON line on rank 0 (contract in the task description).  Besides the contract's keys:

  roofline        the kernel with the largest average launch time of the step, priced on the work it EXECUTES
                  (MFMA instructions issued x 32768 flop, or the algorithmic HBM bytes of SURVEY 8d for the streaming
                  kernels) over its launch time (HIP events on the library's stream) against the chip peak: frac <= 1
  kernels         the same figures for every timed launch group of the step
  sparse_tail     affinity + symmetrise + P as one HBM-bound block: algorithmic bytes, counter bytes from
                  the committed rocprofv3 PMC pass of this command (profiles/), waste ratio
  pruning         2 N^2 d (the flops a dense distance contraction would need) against the flops executed - a ratio that
                  says how much the cell bounds prune, NOT a roofline figure
  incl_h2d        the same step with the 256 MB upload of X from pinned host memory inside the timed region
  host_complete   SURVEY 8d's apples-to-apples wall time: host float32 X in -> scipy CSR K and P out
  secondary       the other workloads of SURVEY 8d that fit one GPU (manifold / gauss inputs, decay=None, C2, C4, C5)
  cpu_baseline    the oracle port on this host's cores, bounded sample (and cpu_baseline_full: the whole of C3 once,
                  live with --cpu-full, else the recorded run under profiles/ when it was made on an identical host)
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: dense matrix peaks (256 CUs @ 2.4 GHz) and HBM3E
MFMA_PEAK_TFLOPS = {"f16x1": 2500.0,  # v_mfma_f32_32x32x16_f16, one chain per product (high float16 plane)
                    "f16": 2500.0,    # same instruction, split float16: 3 chains per product
                    "f32": 157.3}     # v_mfma_f32_32x32x2_f32
MFMA_CHAINS = {"f16x1": 1.0, "f16": 3.0, "f32": 1.0}
HBM_PEAK_GBS = 8000.0
PROFILE_TRAFFIC = os.path.join(ROOT, "profiles", "r6_pmc_fetch_write_per_kernel.json")
PROFILE_CPU_FULL = os.path.join(ROOT, "profiles", "r3_cpu_baseline_full_c3.json")

STAGES = ("prep", "query_order", "sym_prepare", "sym_seed", "sym_bound", "knn_select", "sym_cold", "rerank", "fallback",
          "radius", "affinity", "symmetrize", "normalize", "symm_bins", "symm_merge", "symm_compact")


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


def make_manifold(n, d, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((5, d))
    out = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, 100000):
        e = min(n, s + 100000)
        out[s:e] = rng.standard_normal((e - s, 5)) @ a + 0.01 * rng.standard_normal((e - s, d))
    return out


def make_gauss(n, d, seed):
    rng = np.random.default_rng(seed)
    out = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, 100000):
        e = min(n, s + 100000)
        out[s:e] = rng.standard_normal((e - s, d))
    return out


def host_info():
    info = {"logical_cpus": os.cpu_count(), "machine": platform.machine()}
    try:
        import psutil

        info["physical_cores"] = psutil.cpu_count(logical=False)
        info["ram_GB"] = round(psutil.virtual_memory().total / 2**30)
    except Exception:
        info["physical_cores"] = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    info["cpu_model"] = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    try:
        from threadpoolctl import threadpool_info

        info["threadpools"] = [{"api": i.get("user_api"), "lib": i.get("internal_api"), "threads": i.get("num_threads")}
                               for i in threadpool_info()]
    except Exception:
        pass
    vers = {"numpy": np.__version__}
    for mod in ("scipy", "sklearn"):
        try:
            vers[mod] = __import__(mod).__version__
        except Exception:
            vers[mod] = None
    info["versions"] = vers
    return info


def worker_threads(info):
    th = [p.get("threads") or 1 for p in info.get("threadpools", [])]
    return int(max(th + [1]))


def profiled_traffic(kernel_names, per="launch"):
    """HBM-side bytes of the named kernels from the committed rocprofv3 PMC passes of this command (FETCH_SIZE with the gfx950
    correction of MI355X_MICROARCH.md + WRITE_SIZE, tools/pmc_traffic_summary.py): per launch of the (first) matching kernel, or
    per graph (all launches of all matching kernels, divided by the number of builds the profiled command ran).  PMC counters
    cannot be read from inside the benchmarked process; None when the profile does not hold the kernel."""
    try:
        with open(PROFILE_TRAFFIC) as f:
            prof = json.load(f)
        ks = prof["kernels"]
    except Exception:
        return None
    builds = float(prof.get("builds_in_run", 2))   # (bench.py --steps 1 --warmup 0: the timed step and the H2D-inclusive step)
    tot, found = 0.0, False
    for name, k in ks.items():
        if any(name.split("<")[0] == kn.split("<")[0] or (kn.endswith("_") and name.startswith(kn)) for kn in kernel_names):
            one = k["hbm_read_GB_per_launch_corrected_x2"] + k["hbm_write_GB_per_launch"]
            if per == "launch":
                return one * 1e9
            tot += one * k.get("launches", 1) / builds
            found = True
    return tot * 1e9 if found else None


def cpu_baseline_sample(X, knn, decay, thresh, K0, info, budget_s=25.0):
    """Oracle port (numpy/scipy + the reference's scikit-learn call sites) on this host's cores, bounded sample.

    kNN search + radius fallback + CSR rows are timed on a block of query rows against the full database and
    scaled by N / block (row-separable work); symmetrisation + diff_op are timed at full size on the unsymmetrised
    kernel K0 (produced by the device, data only - not timed)."""
    import oracle

    n = X.shape[0]
    engine = "sklearn" if info["versions"].get("sklearn") else "numpy"
    m = min(1024, n)
    while True:
        t0 = time.perf_counter()
        oracle.knn_kernel(X, knn=knn + 1, decay=decay, thresh=thresh, Y=X[:m], engine=engine)
        t_rows = time.perf_counter() - t0
        if t_rows >= 0.4 * budget_s or m >= n or m >= 65536:
            break
        m = min(n, m * (4 if t_rows < 0.1 * budget_s else 2))
    t_knn_full = t_rows * (n / m)
    t0 = time.perf_counter()
    K = oracle.symmetrize_kernel(K0, "+")
    from scipy import sparse

    oracle.kernel.diff_op_fast(sparse.csr_matrix(K))
    t_tail = time.perf_counter() - t0
    total = t_knn_full + t_tail
    return {
        "value": 1.0 / total, "unit": "graphs/s", "cores": worker_threads(info), "physical_cores": info.get("physical_cores"),
        "kind": "port",
        "sample": "oracle port (%s kNN engine): kNN+affinity rows timed on %d of %d query rows against the full database "
                  "(%.2f s, scaled x%.1f), symmetrise+diff_op timed at full size (%.2f s); estimated full build %.1f s"
                  % (engine, m, n, t_rows, n / m, t_tail, total),
        "versions": info["versions"],
    }


def cpu_baseline_full(X, knn, decay, thresh, info):
    """The whole of C3 through the oracle port, once (SURVEY 8d "C3 timed in full"): minutes of host time."""
    import oracle

    engine = "sklearn" if info["versions"].get("sklearn") else "numpy"
    t0 = time.perf_counter()
    K0 = oracle.knn_kernel(X, knn=knn, decay=decay, thresh=thresh, engine=engine)
    t1 = time.perf_counter()
    K = oracle.symmetrize_kernel(K0, "+")
    from scipy import sparse

    K = sparse.csr_matrix(K)
    P = oracle.kernel.diff_op_fast(K)
    t2 = time.perf_counter()
    return {"value": 1.0 / (t2 - t0), "unit": "graphs/s", "seconds": t2 - t0, "kernel_rows_s": t1 - t0, "symmetrise_P_s": t2 - t1,
            "nnz_K": int(K.nnz), "nnz_P": int(P.nnz), "cores": worker_threads(info), "physical_cores": info.get("physical_cores"),
            "kind": "port", "sample": "all %d rows, measured live (--cpu-full)" % X.shape[0], "host": info}


def cpu_baseline_c2(info, n=100000, d=50):
    """BASELINE config 2 through the oracle port, IN FULL (N = 1e5: seconds on the box's cores) - SURVEY 8d asks for the CPU
    reference beside every config."""
    import oracle
    from scipy import sparse

    engine = "sklearn" if info["versions"].get("sklearn") else "numpy"
    X = make_mix(n, d, 0)
    t0 = time.perf_counter()
    K0 = oracle.knn_kernel(X, knn=15, decay=40.0, thresh=1e-4, engine=engine)
    t1 = time.perf_counter()
    K = sparse.csr_matrix(oracle.symmetrize_kernel(K0, "+"))
    P = oracle.kernel.diff_op_fast(K)
    t2 = time.perf_counter()
    return {"value": 1.0 / (t2 - t0), "unit": "graphs/s", "seconds": t2 - t0, "kernel_rows_s": t1 - t0, "symmetrise_P_s": t2 - t1,
            "nnz_K": int(K.nnz), "nnz_P": int(P.nnz), "cores": worker_threads(info), "physical_cores": info.get("physical_cores"),
            "kind": "port", "sample": "oracle port (%s kNN engine), all %d rows, measured live" % (engine, n),
            "versions": info["versions"]}


def cpu_baseline_c4(info, sizes=(5000, 10000, 20000), n_full=200000, d=100):
    """BASELINE config 4 through the oracle port (oracle.exact_graph on a float32 distance matrix: bandwidths by partition,
    exp(-(D / bw)^decay), threshold, (K + K^T) / 2, rows over their sums - graphs.py:1583-1609, base.py:557-561, 645) at three
    sizes the host can hold, fitted t = a N^2 (every step is a pass over the N x N matrix) and EXTRAPOLATED to N = 2e5:
    the full size needs ~1 TB of host arrays in numpy's formulation and is not run."""
    import oracle

    pts = []
    for n in sizes:
        X = make_mix(n, d, 2).astype(np.float64)
        sq = (X * X).sum(axis=1)
        D = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (X @ X.T), 0.0)).astype(np.float32)   # (the config's INPUT: not timed)
        np.fill_diagonal(D, 0.0)
        t0 = time.perf_counter()
        K, P = oracle.exact_graph(D, knn=15, decay=40, thresh=1e-4, precomputed="distance")
        pts.append((n, time.perf_counter() - t0))
        del D, K, P
    a = sum(t * n * n for n, t in pts) / sum(float(n) ** 4 for n, _ in pts)     # least squares through the origin in N^2
    est = a * float(n_full) ** 2
    return {"value": 1.0 / est, "unit": "graphs/s", "seconds": est, "extrapolated": True,
            "measured": [{"n": int(n), "seconds": round(t, 3)} for n, t in pts], "fit": "t = %.3e x N^2 s" % a,
            "cores": worker_threads(info), "physical_cores": info.get("physical_cores"), "kind": "port",
            "sample": "oracle.exact_graph timed at N = %s, fitted to N^2 and extrapolated to N = %d (flag `extrapolated`)"
                      % ("/".join(str(n) for n, _ in pts), n_full),
            "versions": info["versions"]}


def cpu_baseline_c5(info, X, K0, K, n_landmark=2000, budget_s=12.0):
    """BASELINE config 5 on the host's cores: the kernel part as cpu_baseline_sample times it (kNN + affinity rows on a block of
    query rows scaled to N, symmetrise + diff_op at full size), then - in full - the random landmark assignment
    (graphs.py:1200-1213) and the landmark operator (graphs.py:1169-1246) on the finished kernel K (host CSR from the device
    build: data for the timed algebra, its production is not timed)."""
    import oracle

    ker = cpu_baseline_sample(X, 15, 40.0, 1e-4, K0, info, budget_s=budget_s)
    t0 = time.perf_counter()
    clusters, _ = oracle.random_landmark_clusters(X, n_landmark, 42)
    t1 = time.perf_counter()
    op, _ = oracle.landmark_operator(K, clusters)
    t2 = time.perf_counter()
    total = 1.0 / ker["value"] + (t2 - t0)
    return {"value": 1.0 / total, "unit": "graphs/s", "seconds_estimated": total, "kernel": ker["sample"],
            "landmark_assignment_s": t1 - t0, "landmark_operator_s": t2 - t1, "cores": worker_threads(info),
            "physical_cores": info.get("physical_cores"), "kind": "port",
            "sample": "kernel: sampled query rows scaled to N (see `kernel`); random landmarking + landmark_op (L = %d): all rows, live"
                      % n_landmark,
            "versions": info["versions"]}


def recorded_cpu_full(info):
    """The committed full-size run, reported only when it was made on the same kind of host (cpu model, core count,
    library versions) - otherwise it says nothing about this box."""
    try:
        with open(PROFILE_CPU_FULL) as f:
            rec = json.load(f)
    except Exception:
        return None
    h = rec.get("host", {})
    same = (h.get("cpu_model") == info.get("cpu_model") and h.get("logical_cpus") == info.get("logical_cpus")
            and h.get("versions") == info.get("versions"))
    rec = dict(rec)
    rec["sample"] = "all rows, recorded run (profiles/%s), host %s" % (os.path.basename(PROFILE_CPU_FULL),
                                                                       "identical to this one" if same else "DIFFERENT from this one")
    rec["host_matches"] = bool(same)
    rec.pop("host", None)
    return rec


class StageTimes:
    """per-step stage times of a context (HIP events on the library's stream), averaged over the timed steps"""

    def __init__(self):
        self.acc = {s: [] for s in STAGES}

    def record(self, ctx):
        for s in STAGES:
            self.acc[s].append(max(ctx.stage_ms(s), 0.0))   # (-1: the stage did not run)

    def mean(self, s):
        return float(np.mean(self.acc[s])) if self.acc[s] else 0.0


def candidate_counts(ctx, nloc):
    """(candidates in the lists of the symmetric pass, table entries the re-rank kept) of the most recent build, read back
    through the library's development fetch (gt_dbg_fetch_sym: list lengths, table lengths); (None, None) if unavailable"""
    import ctypes
    try:
        fn = ctx.lib.gt_dbg_fetch_sym
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
        lens = np.zeros(nloc, dtype=np.uint32)
        tabs = np.zeros(nloc, dtype=np.uint32)
        if fn(ctx.h, 3, nloc, lens.ctypes.data) != 0 or fn(ctx.h, 13, nloc, tabs.ctypes.data) != 0:
            return None, None
        return float(np.minimum(lens, 512).sum(dtype=np.int64)), float(tabs.sum(dtype=np.int64))
    except Exception:
        return None, None


def kernel_table(ctx, st, n, nloc, d, world, nnz, nnz0, main):
    """One entry per timed launch group: what it executes (flop / algorithmic bytes), its time, its roofline fraction."""
    kst = ctx.knn_stats()
    symmetric = bool(kst["symmetric"])
    two_stage = symmetric and bool(kst.get("sym_two_stage", False))
    bound_pass = two_stage and bool(kst.get("sym_bound_pass", False))
    peak = MFMA_PEAK_TFLOPS[main]
    rows = []

    def mfma(name, kernel, ms, flop, note=""):
        if ms <= 0:
            return
        ach = flop / (ms * 1e-3) / 1e12
        rows.append({"stage": name, "kernel": kernel, "bound": "mfma", "avg_launch_ms": ms, "executed_flop": flop,
                     "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "note": note})

    def hbm(name, kernel, ms, nbytes, note=""):
        if ms <= 0:
            return
        ach = nbytes / (ms * 1e-3) / 1e9
        rows.append({"stage": name, "kernel": kernel, "bound": "hbm", "avg_launch_ms": ms, "algorithmic_bytes": nbytes,
                     "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "note": note})

    if symmetric:
        dp = 16 if d <= 16 else 32 if d <= 32 else 64 if d <= 64 else 128
        bn = 128 if dp <= 64 else 64
        seed_flop = 2.0 * dp * 256 * bn * kst.get("sym_seed_tiles", 0)   # (256-row query block) x (bn-row tile), padded depth
        mfma("sym_seed", ("sym_seed_dense_kernel<%d>" if kst.get("sym_seed_dense") else "knn_select_kernel<%d, 8, 0, 2>") % dp,
             st.mean("sym_seed"), seed_flop,
             "threshold seeding: every row against its neighbourhood cells")
        cold_flop = 2.0 * d * 64 * 32 * kst.get("sym_cold_pairs", 0)
        if two_stage:
            mfma("sym_cold", ("sym_cold_local_kernel<%d>" if kst.get("sym_cold_local") else "sym_cold_kernel<%d>") % dp, st.mean("sym_cold"), cold_flop,
                 "(64 queries x 32 rows) units left by the cell bounds / stage one, scored in full" +
                 (" in the frame of their queries (float16 of (x - o) sc formed on the fly); the stage also forms the group centres"
                  if kst.get("sym_cold_local") else ""))
        if not bound_pass and st.mean("knn_select") > 0:
            if kst.get("sym_listed"):
                mfma("knn_select", "knn_select_kernel<%d, 8, 2, 2>" % dp, st.mean("knn_select"), cold_flop,
                     "one-stage symmetric collect over listed walks: the (256 x 128) tiles the cell bounds leave, scored and filed in one launch")
            elif two_stage:
                n_pad = -(-n // 1024) * 1024
                nb = n_pad // 1024
                walk = 8 * (1 + (nb - 1) // 2) + (0 if nb % 2 else 8)
                flop = 2.0 * 16 * 1024 * 128 * nb * walk / world
                scored = kst.get("sym_stage_one_units", 0)
                if scored > 0:   # (round 6: units whose balls in the stage-one space are too far apart issue no MFMA)
                    flop = 2.0 * 16 * 32 * 32 * scored
                mfma("knn_select", "knn_select_kernel<16, 8, 3, 2>", st.mean("knn_select"), flop,
                     "stage one of the two-stage collect" + ("" if scored <= 0 else
                      ": %d of %d (32 x 32) units scored, the others skipped by their balls" % (scored, int(nb * walk * 256 / world))))
            else:
                n_pad = -(-n // 256) * 256
                nb = n_pad // 256
                walk = 2 * (1 + (nb - 1) // 2) + (0 if nb % 2 else 2)
                flop = 2.0 * d * 256 * 128 * nb * walk / world
                mfma("knn_select", "knn_select_kernel<%d, 8, 2, 2>" % d, st.mean("knn_select"), flop, "one-stage symmetric collect")
    else:
        mfma("knn_select", "knn_select_kernel<%d, 8, 0, %d>" % (d, {"f32": 0, "f16": 1, "f16x1": 2}[main]), st.mean("knn_select"),
             2.0 * nloc * n * d * MFMA_CHAINS[main], "classic candidate pass: every query row against every point")
    # streaming kernels, algorithmic bytes per SURVEY 8d
    tab = 128   # table entries the affinity pass reads per row (first batch)
    # the re-rank is priced on what it actually sees (round 6; the 128-slot capacity flattered it by 10-15 %): the candidates
    # the lists hold (8-byte keys in) and the table entries it keeps (8 + 4 B out), counted on the device after the timed region
    cand_in, tab_out = candidate_counts(ctx, nloc) if symmetric else (None, None)
    if cand_in is None:
        cand_in = tab_out = float(nloc) * tab
    hbm("rerank", ("rerank_sym4_kernel<1, WT>" if (d % 4 == 0 and d <= 64) else "rerank_sym_kernel") if symmetric else "rerank_kernel",
        st.mean("rerank"),
        cand_in * 8.0 + n * d * 4.0 + tab_out * 12.0,
        "candidate lists in (8 B x %d candidates = %.1f per row), X once, exact tables out ((8 + 4) B x %d entries kept); "
        "the row gathers come from L2/MALL" % (int(cand_in), cand_in / max(nloc, 1), int(tab_out)))
    # (tables by sorted position + no row of the radius pass: the pipelined slot kernels of round 5 - st.mean("symm_bins") of such
    #  a build holds no bin_count_kernel; which ones ran shows in the kernel names of the committed rocprofv3 summary)
    hbm("affinity", "bandwidth_kernel + affinity_slots_kernel (+ posj_hist_kernel) | affinity_kernel", st.mean("affinity"),
        nloc * tab * 12.0 + nnz0 * 8.0, "tables in (8 + 4 B per entry), kept values written in place")
    if st.mean("symm_merge") > 0 and st.mean("symm_compact") > 0:
        # the three launch groups of the symmetrisation, each priced on what IT has to move (the stage as a whole is priced
        # on SURVEY 8d's bytes in `sparse_tail`)
        hbm("symm_bins", "bin_count_kernel + bin_emit_kernel + bin_fill_kernel", st.mean("symm_bins"), 2.0 * nnz0 * 12.0,
            "transpose through destination bins: K0 entries in, K0^T entries out (12 B each)")
        hbm("symm_merge", "sort_merge_kernel (+ sort_merge_long_kernel)", st.mean("symm_merge"), 2.0 * nnz0 * 12.0 + nnz * 12.0,
            "both halves of every union row in, merged row out")
        hbm("symm_compact", "compact_kernel", st.mean("symm_compact"), nnz * 12.0 + nnz * 20.0,
            "merged rows in, CSR K (4 + 8 B) and P (8 B) out")
    elif st.mean("symm_merge") > 0:
        # pair-resolved tail: mutual pairs were settled by the affinity pass, only one-sided entries (nnz - nnz0 of them)
        # are transposed, the merge writes K and P at their final place
        one_sided = float(nnz - nnz0)
        hbm("symm_bins", "bin_emit_slots_kernel + bin_fill_kernel | bin_count_kernel + bin_emit_kernel + bin_fill_kernel",
            st.mean("symm_bins"), nnz0 * 12.0 + one_sided * 12.0, "kept entries in, one-sided entries out (12 B each)")
        hbm("symm_merge", "merge_pairs_slots_kernel + merge_final_kernel (+ merge_long_final_kernel)", st.mean("symm_merge"),
            nnz0 * 12.0 + one_sided * 12.0 + nnz * 20.0, "own and received entries in, CSR K (4 + 8 B) and P (8 B) out")
    else:
        hbm("symmetrize", "symmetrise + compact (K, P)", st.mean("symmetrize"),
            2.0 * nnz0 * 12.0 + nnz * 12.0 + nnz * 12.0 + nnz * 8.0,
            "SURVEY 8d: read K0 and K0^T entries, write K; read K, write P")
    return rows, {"symmetric": symmetric, "two_stage": two_stage, "bound_pass": bound_pass, "knn_stats": kst}


def timed_builds(ctx, params, x_dev_ptr, n, d, steps, warmup, sync):
    for _ in range(warmup):
        ctx.set_points_device(x_dev_ptr, n, d, np.float32)
        ctx.graph_build(params)
    sync()
    st = StageTimes()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.set_points_device(x_dev_ptr, n, d, np.float32)
        nnz, _ = ctx.graph_build(params)
        st.record(ctx)
    sync()
    return (time.perf_counter() - t0) / steps * 1e3, nnz, st


def secondary_knn(_hip, torch, device, name, X, knn=15, decay=40.0, steps=3, opts=()):
    """device-complete build of a secondary workload: {ms_per_graph, graphs_per_s, path, stage_ms}"""
    ctx = _hip.Context(device.index or 0)
    try:
        for k, v in opts:
            ctx.set_option(k, v)
        n, d = X.shape
        x_dev = torch.from_numpy(X).to(device)
        params, keep = ctx.make_params(knn, decay, 1e-4, None, 1.0, None, "+", None, 0)

        def sync():
            torch.cuda.synchronize(device)
            ctx.sync()

        ms, nnz, st = timed_builds(ctx, params, x_dev.data_ptr(), n, d, steps, 2, sync)
        kst = ctx.knn_stats()
        path = "classic candidate pass (%s)" % ctx.last_knn_precision()
        if kst["symmetric"]:
            path = "symmetric pass: " + ("cell bounds + cold launch" if kst.get("sym_bound_pass") else
                                         "two-stage collect" if kst.get("sym_two_stage") else "one-stage collect")
        gs = ctx.graph_stats()
        return {"workload": name, "ms_per_graph": ms, "graphs_per_s": 1e3 / ms, "nnz_K": int(nnz), "path": path,
                "radius_rows": gs["radius_rows"], "repaired_rows": gs["fallback_rows"],
                "stage_ms": {s: round(st.mean(s), 3) for s in STAGES if st.mean(s) > 0}}
    finally:
        ctx.close()


def secondary_c4(_hip, torch, device, n=200000, d=100):
    """BASELINE config 4: exact dense graph + diff_op from a device-resident float32 distance matrix (160 GB), in place."""
    import ctypes

    X = torch.from_numpy(make_mix(n, d, 2)).to(device)
    D = torch.empty((n, n), dtype=torch.float32, device=device)
    for r in range(0, n, 8192):
        D[r: r + 8192] = torch.cdist(X[r: r + 8192], X)
    D.fill_diagonal_(0.0)
    torch.cuda.synchronize(device)
    ctx = _hip.Context(device.index or 0)
    try:
        for o in [o for o in os.environ.get("GT_C4_OPTS", "").split(",") if o]:   # development: library options for this leg
            ctx.set_option(*o.split("="))
        flags = ctypes.c_uint32(0)
        # warm-up (the headline has its --warmup steps; this build consumes its input and runs once): the same call on a
        # 4096 x 4096 corner of the matrix, so that the code objects of its kernels are loaded when the clock starts
        wctx = _hip.Context(device.index or 0)
        try:
            for o in [o for o in os.environ.get("GT_C4_OPTS", "").split(",") if o]:
                wctx.set_option(*o.split("="))
            if "dense_rows=" not in os.environ.get("GT_C4_OPTS", ""):
                wctx.set_option("dense_rows", "1")      # (the row-streaming form is the default from 16384 rows)
            Dw = D[:4096, :4096].contiguous()
            rc = wctx.lib.gt_dense_graph_build(wctx.h, ctypes.c_void_p(Dw.data_ptr()), 4096, 0, 0, 1, 1, 15, 40.0, 1e-4, None, 0, 1.0,
                                               _hip.SYMM["+"], 1.0, 0.0, 1, None, ctypes.c_void_p(Dw.data_ptr()), 1, ctypes.byref(flags))
            wctx._check(rc, "gt_dense_graph_build (warm-up)")
            wctx.sync()
            del Dw
        finally:
            wctx.close()
        t0 = time.perf_counter()
        # in place: D -> K -> P = diff_op in the same 160 GB buffer (K and P together would be 320 GB: more than the HBM);
        # what stays is P, the degrees (K = P * degree row by row) and the bandwidths
        rc = ctx.lib.gt_dense_graph_build(ctx.h, ctypes.c_void_p(D.data_ptr()), n, 0, 0, 1, 1, 15, 40.0, 1e-4, None, 0, 1.0,
                                          _hip.SYMM["+"], 1.0, 0.0, 1, None, ctypes.c_void_p(D.data_ptr()), 1, ctypes.byref(flags))
        ctx._check(rc, "gt_dense_graph_build")
        ctx.sync()
        wall = time.perf_counter() - t0
        st = {s: round(ctx.stage_ms(s), 2) for s in ("dense_bandwidth", "dense_rows_scan", "dense_kernel", "dense_rows_transpose",
                                                     "dense_normalize")}
        row_sums = float(D[:4096].double().sum(dim=1).sub(1.0).abs().max().item())   # diff_op rows sum to 1 (float32 entries)
        # bytes this run needs (round-3 verdict: price what runs).  Row-streaming form (round 4, what runs by default): D read once
        # for the bandwidths (4 N^2), which also gives the row sums and the list of kept affinities (else: a scan of its own,
        # 4 N^2); P written over it (4 N^2: zeros streamed, the listed entries placed - or 8 N^2 when the rows are read again).
        # Tile-pair form: bandwidths 4 N^2, K with fused row sums 8 N^2, normalisation 8 N^2.
        rows_form = st["dense_rows_scan"] > 0
        listed = ctx.stage_launches("dense_rows_listed") > 0    # the bandwidth pass listed the kept affinities: no scan of its own
        placed = ctx.stage_launches("dense_rows_placed") > 0    # the write pass read no row
        nbytes = (((8.0 if placed else 12.0) if listed else (12.0 if placed else 16.0)) if rows_form else 20.0) * n * n
        note = ("8 N^2 bytes as run (row-streaming form): 4 N^2 read by the bandwidth pass, which also lists the rows' kept affinities, "
                "4 N^2 written by the write pass (zeros streamed over the distances, the listed entries and their transposed partners "
                "placed: the matrix is read ONCE)") if rows_form and listed and placed else (
                "12 N^2 bytes as run (row-streaming form): 4 N^2 bandwidth pass, which also lists the rows' kept affinities, 8 N^2 write "
                "pass (P over the distances; the transposed half arrives as the list)") if rows_form and listed else (
                "12 N^2 bytes as run (row-streaming form): 4 N^2 bandwidth pass, 4 N^2 scan (row sums + the kept affinities as a "
                "list), 4 N^2 written by the write pass (zeros streamed, the listed entries placed)") if rows_form and placed else (
                "16 N^2 bytes as run (row-streaming form): 4 N^2 bandwidth pass, 4 N^2 scan (row sums + the kept affinities as a "
                "list), 8 N^2 write pass (P over the distances; the transposed half arrives as the list)") if rows_form else (
                "20 N^2 bytes as run: 4 N^2 bandwidth pass (one read), 8 N^2 tile-pair kernel (row sums fused), 8 N^2 normalisation")
        return {"workload": "C4: mix N=%d d=%d seed=2, TraditionalGraph knn=15 decay=40 from a resident float32 distance matrix "
                            "(precomputed='distance'): bandwidths, K, diff_op materialised IN PLACE (the buffer ends as P; K = P x "
                            "degree), degrees (one build: the input is consumed; one 4096-row warm-up build before it)" % (n, d),
                "ms_per_graph": wall * 1e3, "graphs_per_s": 1.0 / wall, "stage_ms": st,
                "diff_op_row_sum_max_dev_first_4096_rows": row_sums,
                "roofline": {"bound": "hbm", "algorithmic_bytes": nbytes, "achieved": nbytes / wall / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": nbytes / wall / 1e9 / HBM_PEAK_GBS,
                             "note": note}}
    finally:
        ctx.close()
        del D, X
        torch.cuda.empty_cache()
        _hip.release_cached_memory()


def secondary_c4_points(_hip, torch, device, n=200000, d=100, steps=2):
    """Config 4's graph built FROM THE POINTS (graphs.py:1546-1609 without forming all pairs): radius search of the kNN path with
    float64 distances, then K and P written out densely (float32, device-resident, one N x N buffer reused for both)."""
    X = torch.from_numpy(make_mix(n, d, 2)).to(device)
    out = torch.empty((n, n), dtype=torch.float32, device=device)
    ctx = _hip.Context(device.index or 0)
    try:
        ctx.set_option("distance_dtype", "float64")
        params, keep = ctx.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
        best = None
        for it in range(steps + 1):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            ctx.set_points_device(X.data_ptr(), n, d, np.float32)
            nnz, _ = ctx.graph_build(params)
            ctx.sync()
            t1 = time.perf_counter()
            ctx.graph_to_dense(_hip.CSR_K, n, np.float32, out_device=out)
            ctx.graph_to_dense(_hip.CSR_P, n, np.float32, out_device=out)
            ctx.sync()
            t2 = time.perf_counter()
            cur = {"build_ms": (t1 - t0) * 1e3, "dense_out_ms": (t2 - t1) * 1e3, "ms_per_graph": (t2 - t0) * 1e3}
            if it > 0 and (best is None or cur["ms_per_graph"] < best["ms_per_graph"]):
                best = cur
        nbytes = 8.0 * n * n   # two float32 N x N matrices written once
        best.update({"workload": "C4 from points: mix N=%d d=%d seed=2 float32, TraditionalGraph knn=15 decay=40 thresh=1e-4, "
                                 "radius search + dense float32 K and P on the device" % (n, d),
                     "graphs_per_s": 1e3 / best["ms_per_graph"], "nnz_K": int(nnz),
                     "dense_out_GBs": nbytes / best["dense_out_ms"] / 1e6})
        return best
    finally:
        ctx.close()
        del out, X
        torch.cuda.empty_cache()
        _hip.release_cached_memory()


def secondary_c5(n=1000000, d=50, cpu_info=None):
    """BASELINE config 5 on one GPU: kNN kernel + landmark operator (random landmarking), host arrays in and out."""
    import warnings

    import graphtools_amd

    X = make_mix(n, d, 3)
    best = None
    for _ in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=2000, random_landmarking=True,
                                     random_state=42, verbose=0)
            t1 = time.perf_counter()
            op = G.landmark_op
            t2 = time.perf_counter()
        cur = {"kernel_s": t1 - t0, "landmark_op_s": t2 - t1, "total_s": t2 - t0, "L": int(op.shape[0])}
        if best is None or cur["total_s"] < best["total_s"]:
            best = cur
        del G, op
    best["workload"] = "C5 on one GPU: mix N=%d d=%d seed=3, knn=15 decay=40, n_landmark=2000 random landmarking -> landmark_op; " \
                       "host-complete (host X in, host operator out)" % (n, d)
    best["graphs_per_s"] = 1.0 / best["total_s"]
    if cpu_info is not None:
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, verbose=0)
                K0 = G.build_kernel()      # the unsymmetrised kernel (graphs.py:771-982): input of the timed symmetrisation
                K = G.K
            best["cpu_baseline"] = cpu_baseline_c5(cpu_info, X, K0, K)
            best["vs_cpu_baseline"] = best["graphs_per_s"] / best["cpu_baseline"]["value"]
            del G, K0, K
        except Exception as e:   # pragma: no cover
            best["cpu_baseline"] = {"error": repr(e)}
    return best


def launch_ranks(n_gpus, visible):
    """`bench.py --gpus N` without WORLD_SIZE: N ranks through torch.distributed.run, started as a child process.

    Returns the exit status to leave with.  The caller has not initialised the GPU (on this pool a process that has must not
    be replaced or followed by another on the same device set); rank 0's JSON line reaches stdout through the child.
    """
    import socket
    import subprocess

    if visible < n_gpus:
        print("bench.py: --gpus %d but only %d GPU(s) are visible on this node - refusing to run fewer ranks than asked for"
              % (n_gpus, visible), file=sys.stderr)
        return 2
    with socket.socket() as s:   # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--npoints", type=int, default=1000000)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--knn", type=int, default=15)
    ap.add_argument("--decay", type=float, default=40.0)
    ap.add_argument("--workload", choices=["c3", "gauss", "c5"], default="c3",
                    help="c3 (default): the headline.  gauss: the same build on isotropic points (nothing to prune: the dense "
                         "contraction runs, sharded by row blocks).  c5: BASELINE config 5, mix N=1e6 d=50 + landmark operator "
                         "(n_landmark=2000, random landmarking).  Both exist for the N > 1 legs: --gpus N --workload gauss|c5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="time the oracle port on ALL rows of the workload (minutes)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads / host-complete legs")
    ap.add_argument("--secondary-only", default="", help="comma list out of manifold,gauss,binary,cosine,c2,c4,c4points,c5")
    ap.add_argument("--knn-precision", choices=["auto", "f16x1", "f16", "f32"],
                    default=os.environ.get("GT_KNN_PRECISION", "auto"),
                    help="arithmetic of the candidate pass (results are identical; see DESIGN.md); auto = the library "
                         "default: single float16 chain when the data tolerate it, else split float16")
    args = ap.parse_args()

    import torch

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves - as a CHILD process, before this one
        # has touched the GPU (device_count() does not initialise it) - relay their output and leave with their status.
        # Never a silent one-rank run: fewer visible devices than asked for is an error.
        sys.exit(launch_ranks(args.gpus, torch.cuda.device_count()))

    from graphtools_amd import _hip
    from graphtools_amd import dist as gdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GT_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (development aid)
    distributed = world > 1 or os.environ.get("GT_BENCH_FORCE_DIST") == "1"
    if args.gpus != world:
        # the line's n_gpus is the number of ranks that ran: a launcher that disagrees with --gpus is a mistake, not a warning
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        dist.init_process_group("nccl")
    n, d = args.npoints, args.dim
    thresh = 1e-4
    if args.workload == "c5":
        d = 50
        X = make_mix(n, d, 3)
    elif args.workload == "gauss":
        X = make_gauss(n, d, 1)
    else:
        X = make_mix(n, d, 1)
    n_landmark = 2000
    ctx = _hip.Context(local_rank)
    ctx.set_option("knn_precision", args.knn_precision)

    def params_factory(symm="+"):
        return ctx.make_params(args.knn, args.decay, thresh, None, 1.0, None, symm, None, 0)

    params, keep = params_factory("+")
    splits = gdist.even_row_splits(n, world)
    x_local = torch.from_numpy(X[splits[rank]:splits[rank + 1]]).to(device)   # inputs resident in HBM
    sharded = gdist.ShardedKnnGraph(ctx, n) if distributed else None
    torch.cuda.synchronize(device)

    landmark_out = {}

    def step():
        if distributed:
            sharded.gather_points(x_local)
            res = sharded.build(params)
            if args.workload == "c5":
                # random landmarking (graphs.py:1200-1213) + the landmark operator (graphs.py:1169-1246) over the ranks:
                # an all-gather of the cluster labels, an all-reduce of the L x L partial products
                clusters = sharded.random_landmark_clusters(n_landmark, 42)
                landmark_out["op"], landmark_out["tnnz"] = sharded.landmark_operator(clusters, n_landmark)
            return res
        ctx.set_points_device(x_local.data_ptr(), n, d, np.float32)
        res = ctx.graph_build(params)
        if args.workload == "c5":
            # random landmarking as graphtools_amd.graphs.LandmarkGraph does it on one GPU: 1-NN of every row against the
            # L landmark rows on the MFMA path (sklearn's euclidean_distances arithmetic, graphs.py:1210)
            landmarks = np.random.default_rng(42).choice(n, n_landmark, replace=False)
            lm_rows = x_local[torch.as_tensor(landmarks, device=device)].contiguous()
            torch.cuda.synchronize(device)
            lm_ctx = _hip.Context(local_rank)
            try:
                lm_ctx.set_points_device(lm_rows.data_ptr(), n_landmark, d, np.float32)
                d_, idx, _ = lm_ctx.knn_search_device(4, x_local.data_ptr(), n)
            finally:
                lm_ctx.close()
            clusters = np.where(d_ == d_[:, :1], idx, np.iinfo(np.int64).max).min(axis=1).astype(np.int32)
            M, R, landmark_out["tnnz"] = ctx.landmark_build(clusters, n_landmark)
            landmark_out["op"] = ctx.landmark_scale(M, R)
        return res

    def fence():
        torch.cuda.synchronize(device)
        ctx.sync()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    st = StageTimes()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nnz, flags = step()
        st.record(ctx)   # stage timers were recorded with hipEvents on the library's own stream during the step
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
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        nloc = int(splits[1] - splits[0])
        main_prec = ctx.last_knn_precision()                 # what the main candidate pass actually ran on
        stats = ctx.graph_stats()
        nnz0 = int(stats["nnz_unsymmetrised"])
        rows, flags_ = kernel_table(ctx, st, n, nloc, d, world, int(nnz), nnz0, main_prec)
        # the dominant KERNEL: rows that time a group of launches ("a + b") do not compete
        singles = [r for r in rows if " + " not in r["kernel"]] or rows
        dominant = max(singles, key=lambda r: r["avg_launch_ms"])
        roof = {k: dominant[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")}
        roof["stage"] = dominant["stage"]
        roof["share_of_step"] = dominant["avg_launch_ms"] / ms_per_step
        roof["traffic"] = profiled_traffic([dominant["kernel"].split(" ")[0]]) if (n == 1000000 and d == 64 and world == 1 and args.workload == "c3") else None
        roof["traffic_unit"] = "bytes/launch"
        roof["traffic_source"] = ("committed rocprofv3 PMC passes of this command (profiles/%s: FETCH_SIZE x 2 per MI355X_MICROARCH.md + "
                                  "WRITE_SIZE), an earlier process - counters cannot be read from inside the benchmarked one"
                                  % os.path.basename(PROFILE_TRAFFIC))
        roof["work_per_launch"] = dominant.get("executed_flop", dominant.get("algorithmic_bytes"))
        executed = sum(r.get("executed_flop", 0.0) for r in rows)
        tail_ms = st.mean("affinity") + st.mean("symmetrize") + st.mean("normalize")
        # tables in / kept values out (affinity); read K0 and K0^T entries, write K; read K, write P
        tail_bytes = (sum(r["algorithmic_bytes"] for r in rows if r["stage"] == "affinity")
                      + 2.0 * nnz0 * 12.0 + nnz * 12.0 + nnz * 12.0 + nnz * 8.0)
        tail_counter = profiled_traffic(["bandwidth_kernel", "affinity_kernel", "affinity_slots_kernel", "posj_hist_kernel", "bin_",
                                         "sort_merge_kernel", "sort_merge_long_kernel", "compact_kernel", "merge_final_kernel",
                                         "merge_pairs_slots_kernel", "merge_long_final_kernel", "pairs_len_kernel", "scan_",
                                         "gather_counts_kernel", "scatter_", "invperm_kernel"], per="graph") if (n == 1000000 and d == 64 and world == 1 and args.workload == "c3") else None
        wl_name = {"c3": "C3: mix N=%d d=%d float32 seed=1, kNNGraph knn=%d decay=%g thresh=1e-4, kernel_symm='+', points resident "
                         "in HBM, device-complete K and P" % (n, d, args.knn, args.decay),
                   "gauss": "gauss: isotropic N=%d d=%d float32 seed=1 (no structure to prune: the classic candidate pass scores "
                            "every query row of a rank against every point), kNNGraph knn=%d decay=%g thresh=1e-4, device-complete "
                            "K and P" % (n, d, args.knn, args.decay),
                   "c5": "C5: mix N=%d d=%d float32 seed=3, kNNGraph knn=%d decay=%g + LandmarkGraph n_landmark=%d random "
                         "landmarking (random_state=42): K, P on the device, landmark_op (L x L) on the host"
                         % (n, d, args.knn, args.decay, n_landmark)}[args.workload]
        out = {
            "metric": {"c3": "graphs/sec (kernel+diff_op) at N=1e6 d=64 k=15",
                       "gauss": "graphs/sec (kernel+diff_op) at N=1e6 d=64 k=15, isotropic input",
                       "c5": "graphs/sec (kernel+diff_op+landmark_op) at N=1e6 d=50 k=15 n_landmark=2000"}[args.workload],
            "value": args.steps / elapsed,
            "unit": "graphs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,   # the reference publishes no number for this metric
            "dtype": {"f16x1": "f16", "f16": "f16x2-split", "f32": "f32"}[main_prec] + " MFMA candidate filter (proven error bound) + f64 re-rank/affinities: every emitted value is float64 as in the reference",
            "data": "synthetic",
            "config": {"workload": wl_name,
                       "row_sharding": ("%d rank(s) x ~%d rows" % (world, nloc)) + (
                           "; rows renumbered by landmark cell (coherent cell numbers: a rank owns whole clusters), candidate lists "
                           "collected by the owner (no candidate exchange); collectives per graph: points all-gather, cell numbers "
                           "all-gather (4 B per row)" + (", bandwidths all-gather (8 B per row: every rank settles its mutual pairs itself, "
                           "only one-sided entries for other ranks' rows travel)" if getattr(sharded, "pairs_used", False) else "") +
                           ", triplet counts, triplet all-to-all"
                           if (distributed and sharded.renumbered) else ""),
                       "pair_resolved_tail_on_ranks": bool(getattr(sharded, "pairs_used", False)) if distributed else None,
                       "nnz_K": nnz_total, "nnz_K0_rank0": nnz0,
                       "radius_rows_rank0": stats["radius_rows"], "fallback_rows_rank0": stats["fallback_rows"],
                       "symmetric_candidate_pass": flags_["symmetric"], "two_stage_collect": flags_["two_stage"],
                       "bound_pass": flags_["bound_pass"],
                       "tail_tables_by_slot": bool(flags_["knn_stats"].get("tables_by_slot", False)),
                       "tail_destinations_fused": bool(flags_["knn_stats"].get("destinations_fused", False))},
            "roofline": roof,
            "kernels": [{k: (round(v, 4) if isinstance(v, float) and k not in ("executed_flop", "algorithmic_bytes") else v)
                         for k, v in r.items()} for r in rows],
            "sparse_tail": {"stages": "affinity + symmetrise + P", "ms": tail_ms, "algorithmic_bytes": tail_bytes,
                            "achieved_GBs": tail_bytes / (tail_ms * 1e-3) / 1e9 if tail_ms > 0 else None,
                            "frac_of_hbm_peak": tail_bytes / (tail_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if tail_ms > 0 else None,
                            "counter_bytes": tail_counter,
                            "waste_ratio": (tail_counter / tail_bytes) if tail_counter else None},
            "pruning": {"dense_contraction_flop_2N2d": 2.0 * nloc * n * d, "executed_mfma_flop": executed,
                        "ratio": (2.0 * nloc * n * d / executed) if executed else None,
                        "note": "how much of the N x d . d x N contraction the cell bounds and fixed thresholds prove "
                                "unnecessary; a pruning ratio, not a roofline figure"},
            "stage_ms": {s: round(st.mean(s), 3) for s in STAGES if st.mean(s) > 0},
        }
        if args.workload == "c5" and "op" in landmark_out:
            op = np.asarray(landmark_out["op"])
            out["config"]["landmark_op"] = {"shape": list(op.shape), "row_sums_max_abs_dev_from_1": float(np.abs(op.sum(axis=1) - 1.0).max()),
                                            "transitions_nnz_rank0": int(landmark_out.get("tnnz", 0))}
        info = host_info()
        single = world == 1 and not distributed and args.workload == "c3"
        if single:
            # ---- the same step with the upload of X inside the timed region (pinned host memory) ----
            try:
                x_pin = torch.from_numpy(X).pin_memory()
                x_dev = torch.empty_like(x_local)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    x_dev.copy_(x_pin, non_blocking=True)
                    torch.cuda.synchronize(device)
                    ctx.set_points_device(x_dev.data_ptr(), n, d, np.float32)
                    ctx.graph_build(params)
                fence()
                ms = (time.perf_counter() - t0) / args.steps * 1e3
                out["incl_h2d"] = {"ms_per_graph": ms, "graphs_per_s": 1e3 / ms,
                                   "note": "X (%.0f MB) copied from pinned host memory inside the timed region" % (X.nbytes / 1e6)}
                del x_pin, x_dev
            except Exception as e:   # pragma: no cover
                out["incl_h2d"] = {"error": repr(e)}
        K0 = None
        if single and not args.no_cpu_baseline:
            from scipy import sparse

            p0, keep0 = params_factory(None)
            ctx.graph_build(p0)
            d_, i_, p_ = ctx.graph_fetch_csr(_hip.CSR_K)
            K0 = sparse.csr_matrix((d_, i_, p_), shape=(n, n))
            ctx.graph_build(params)
        ctx.close()
        ctx = None
        del x_local
        torch.cuda.empty_cache()
        want = [w for w in args.secondary_only.split(",") if w] or ["manifold", "gauss", "binary", "cosine", "c2", "c4", "c4points", "c5"]
        if single and not args.no_secondary:
            # ---- host-complete: host X in -> scipy CSR K, P out (SURVEY 8d headline definition) ----
            try:
                import graphtools_amd

                ts = []
                for _ in range(7):   # (seven: the first call of a process allocates the result arrays, the second is the first to recycle them)
                    t0 = time.perf_counter()
                    G = graphtools_amd.Graph(X, knn=args.knn, decay=args.decay, n_pca=None, verbose=0)
                    Kh, Ph = G.K, G.P
                    ts.append(time.perf_counter() - t0)
                    nnz_h = int(Kh.nnz)
                    del G, Kh, Ph
                out["host_complete"] = {"ms_per_graph": float(np.median(ts)) * 1e3, "graphs_per_s": 1.0 / float(np.median(ts)),
                                        "runs_s": [round(t, 4) for t in ts], "nnz_K": nnz_h,
                                        "note": "graphtools_amd.Graph(X, knn=15, decay=40).K/.P: pageable host X in, scipy CSR out "
                                                "(H2D 256 MB, build, D2H of K values + indices + indptr; the P values are derived "
                                                "from K and the degrees by host threads while K arrives - bit-identical to "
                                                "the device's P, which stays on the device for device consumers); median of seven "
                                                "calls in one process: the result arrays of a dropped graph are recycled "
                                                "(graphtools_amd._hip._HostPool) and a copy into resident memory runs at the "
                                                "link's rate - the first call of a process, into fresh arrays, is the slowest of "
                                                "runs_s"}
                # SURVEY 8d defines the headline as host-complete: the same metric under that definition, next to `value`
                out["value_host_complete"] = out["host_complete"]["graphs_per_s"]
            except Exception as e:   # pragma: no cover
                out["host_complete"] = {"error": repr(e)}
            sec = {}
            jobs = {
                "manifold": lambda: secondary_knn(_hip, torch, device, "manifold N=1e6 d=64 seed=1 (5 dimensions embedded in 64), knn=15 decay=40",
                                                  make_manifold(1000000, 64, 1)),
                "gauss": lambda: secondary_knn(_hip, torch, device, "gauss N=1e6 d=64 seed=1 (isotropic: no structure to prune), knn=15 decay=40",
                                               make_gauss(1000000, 64, 1), steps=2),
                "binary": lambda: secondary_knn(_hip, torch, device, "C3 with decay=None (connectivity kernel): mix N=1e6 d=64 seed=1, knn=15",
                                                X, decay=None),
                "cosine": lambda: secondary_knn(_hip, torch, device, "C3 with distance='cosine': mix N=1e6 d=64 seed=1, knn=15 decay=40",
                                                X, opts=(("metric", "cosine"),)),
                "c2": lambda: secondary_knn(_hip, torch, device, "C2: mix N=1e5 d=50 seed=0, knn=15 decay=40", make_mix(100000, 50, 0), steps=5),
                "c4": lambda: secondary_c4(_hip, torch, device),
                "c4points": lambda: secondary_c4_points(_hip, torch, device),
                "c5": lambda: secondary_c5(cpu_info=None if args.no_cpu_baseline else info),
            }
            for name in want:
                t0 = time.perf_counter()
                try:
                    sec[name] = jobs[name]()
                except Exception as e:   # pragma: no cover
                    sec[name] = {"error": repr(e)}
                sec[name]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
                _hip.release_cached_memory()
                # the CPU reference beside the config, on the box's cores: C2 in full, C4 fitted and extrapolated
                if not args.no_cpu_baseline and name in ("c2", "c4") and "error" not in sec[name]:
                    try:
                        cb = cpu_baseline_c2(info) if name == "c2" else cpu_baseline_c4(info)
                        sec[name]["cpu_baseline"] = cb
                        gps = sec[name].get("graphs_per_s")
                        if gps:
                            sec[name]["vs_cpu_baseline"] = gps / cb["value"]
                            sec[name]["vs_cpu_baseline_note"] = "device-resident GPU build against the host-to-host CPU port" + (
                                "; the CPU figure is extrapolated" if cb.get("extrapolated") else "")
                    except Exception as e:   # pragma: no cover
                        sec[name]["cpu_baseline"] = {"error": repr(e)}
            out["secondary"] = sec
        if single and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_sample(X, args.knn, args.decay, thresh, K0, info)
            full = cpu_baseline_full(X, args.knn, args.decay, thresh, info) if args.cpu_full else recorded_cpu_full(info)
            if full and args.cpu_full:
                # (copied to profiles/ by hand: the recorded run later invocations on an identical host report)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", os.path.basename(PROFILE_CPU_FULL)), "w") as f:
                    json.dump(full, f, indent=1)
            if full:
                full = {k: v for k, v in full.items() if k != "host"}
                out["cpu_baseline_full"] = full
                # like for like: the CPU port builds host arrays from host arrays - against the host-complete GPU build; the
                # device-resident figure is given separately and says so
                if "seconds" in full and "host_complete" in out and "ms_per_graph" in out["host_complete"]:
                    out["vs_cpu_baseline_full"] = full["seconds"] / (out["host_complete"]["ms_per_graph"] * 1e-3)
                    out["vs_cpu_baseline_full_device_resident"] = out["value"] * full["seconds"]
                else:
                    out["vs_cpu_baseline_full"] = None
            if "host_complete" in out and "graphs_per_s" in out["host_complete"]:
                out["vs_cpu_baseline_sampled"] = out["host_complete"]["graphs_per_s"] / out["cpu_baseline"]["value"]
                out["vs_cpu_baseline_sampled_device_resident"] = out["value"] / out["cpu_baseline"]["value"]
            else:
                out["vs_cpu_baseline_sampled"] = out["value"] / out["cpu_baseline"]["value"]
        out["host"] = {k: info[k] for k in ("cpu_model", "logical_cpus", "physical_cores") if k in info}
        print(json.dumps(out))
        sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
