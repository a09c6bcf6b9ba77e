"""GPU: randomised parity sweep (tools/gpu_fuzz.py) - random shapes, dtypes, metrics, kernels and symmetrisations of
graphtools_amd.Graph against the oracle; CSR structure identical, values within 1e-5 relative."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_configurations_match_the_oracle():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "30", "11"], capture_output=True,
                         text=True, timeout=1500)
    tail = "\n".join(res.stdout.strip().splitlines()[-6:])
    assert res.returncode == 0, tail + "\n" + res.stderr[-2000:]


def test_random_exact_extension_and_mnn_graphs_match_the_oracle():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz_more.py"), "30", "5"], capture_output=True,
                         text=True, timeout=1500)
    tail = "\n".join(res.stdout.strip().splitlines()[-6:])
    assert res.returncode == 0, tail + "\n" + res.stderr[-2000:]


def test_random_configurations_with_grouped_queries():
    """the same sweep with the query ordering and the landmark starting thresholds forced on for small problems
    (GT_QUERY_ORDER_MIN_ROWS=1; by default they only engage from 32768 rows)"""
    env = dict(os.environ, GT_QUERY_ORDER_MIN_ROWS="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "30", "23"], capture_output=True,
                         text=True, timeout=1500, env=env)
    tail = "\n".join(res.stdout.strip().splitlines()[-6:])
    assert res.returncode == 0, tail + "\n" + res.stderr[-2000:]


def test_random_configurations_with_the_symmetric_pass():
    """the same sweep with the symmetric candidate pass (gt_sym.hip) forced on wherever it applies (euclidean self
    queries of at least 2048 rows; by default it only engages from 65536 rows)"""
    env = dict(os.environ, GT_QUERY_ORDER_MIN_ROWS="1", GT_SYMMETRIC="1", GT_SYM_STRIDE="4")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "30", "37"], capture_output=True,
                         text=True, timeout=1500, env=env)
    tail = "\n".join(res.stdout.strip().splitlines()[-6:])
    assert res.returncode == 0, tail + "\n" + res.stderr[-2000:]


def test_random_configurations_with_the_two_stage_symmetric_collect():
    """the symmetric sweep with the two-stage collect forced on wherever it is built (32 ... 64 padded features): partial
    distances first, deferred cold pass, orphans - whatever the data look like (a queue that overflows starts over)"""
    env = dict(os.environ, GT_QUERY_ORDER_MIN_ROWS="1", GT_SYMMETRIC="1", GT_SYM_STRIDE="4", GT_SYM_TWO_STAGE="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "40", "53"], capture_output=True,
                         text=True, timeout=1500, env=env)
    tail = "\n".join(res.stdout.strip().splitlines()[-6:])
    assert res.returncode == 0, tail + "\n" + res.stderr[-2000:]
