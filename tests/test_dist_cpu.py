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
