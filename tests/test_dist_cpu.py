"""CPU: the N>1 host path (graphtools_amd/dist.py) on a world_size-2 gloo group."""
import os
import socket
import subprocess
import sys

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_sharded_build_world2_gloo():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.join(ROOT, "tests") + os.pathsep + ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + "\n" + res.stderr[-3000:]
    assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout


def test_bench_gpus_n_without_launcher_never_runs_fewer_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own ranks; on a box with fewer GPUs it refuses
    (non-zero exit, a message) instead of printing a one-rank line labelled n_gpus 1 (missing 2)."""
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("two GPUs visible: the launcher would run the benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "--gpus 2 but only" in res.stderr and "refusing" in res.stderr
    assert '"metric"' not in res.stdout


def test_bench_rejects_a_launcher_that_disagrees_with_gpus():
    env = dict(os.environ)
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr
