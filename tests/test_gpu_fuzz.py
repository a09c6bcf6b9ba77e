"""GPU: randomised parity sweep (tools/gpu_fuzz.py) - random shapes, dtypes, metrics, kernels and symmetrisations of
graphtools_amd.Graph against the oracle; CSR structure identical, values within 1e-5 relative.

The five sweeps are separate processes whose time is the ORACLE's (numpy on the host, ~2 s per case): they are started
together by the first test that needs one and every test waits for its own (round 6: 238 s of a 556 s suite one after the
other, the longest alone 86 s)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWEEPS = {
    "plain": ("gpu_fuzz.py", "30", "11", {}),
    "more": ("gpu_fuzz_more.py", "30", "5", {}),
    # query ordering and the landmark starting thresholds forced on for small problems (by default from 32768 rows)
    "grouped": ("gpu_fuzz.py", "30", "23", {"GT_QUERY_ORDER_MIN_ROWS": "1"}),
    # symmetric candidate pass (gt_sym.hip) forced on wherever it applies (by default from 65536 rows)
    "symmetric": ("gpu_fuzz.py", "30", "37", {"GT_QUERY_ORDER_MIN_ROWS": "1", "GT_SYMMETRIC": "1", "GT_SYM_STRIDE": "4"}),
    # ... with the two-stage collect forced on wherever it is built (32 ... 64 padded features)
    "two_stage": ("gpu_fuzz.py", "40", "53", {"GT_QUERY_ORDER_MIN_ROWS": "1", "GT_SYMMETRIC": "1", "GT_SYM_STRIDE": "4",
                                               "GT_SYM_TWO_STAGE": "1"}),
}


@pytest.fixture(scope="module")
def sweeps(tmp_path_factory):
    out = tmp_path_factory.mktemp("fuzz")
    cores = os.cpu_count() or 8
    procs = {}
    for name, (script, cases, seed, extra) in SWEEPS.items():
        env = dict(os.environ, **extra)
        env["GT_FUZZ_FAILURES"] = str(out / (name + "_failures.json"))
        # (the oracle's BLAS would otherwise start one thread per core in each of the five processes)
        env.setdefault("OMP_NUM_THREADS", str(max(2, cores // len(SWEEPS))))
        env.setdefault("OPENBLAS_NUM_THREADS", env["OMP_NUM_THREADS"])
        so, se = open(out / (name + ".out"), "w"), open(out / (name + ".err"), "w")
        procs[name] = (subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", script), cases, seed], stdout=so, stderr=se,
                                        env=env), so, se, out)
    yield procs
    for p, so, se, _ in procs.values():
        if p.poll() is None:
            p.kill()
        so.close()
        se.close()


def _check(sweeps, name):
    p, so, se, out = sweeps[name]
    rc = p.wait(timeout=1500)
    so.flush()
    se.flush()
    tail = "\n".join(open(out / (name + ".out")).read().strip().splitlines()[-6:])
    assert rc == 0, tail + "\n" + open(out / (name + ".err")).read()[-2000:]


def test_random_configurations_match_the_oracle(sweeps):
    _check(sweeps, "plain")


def test_random_exact_extension_and_mnn_graphs_match_the_oracle(sweeps):
    _check(sweeps, "more")


def test_random_configurations_with_grouped_queries(sweeps):
    """the same sweep with the query ordering and the landmark starting thresholds forced on for small problems
    (GT_QUERY_ORDER_MIN_ROWS=1; by default they only engage from 32768 rows)"""
    _check(sweeps, "grouped")


def test_random_configurations_with_the_symmetric_pass(sweeps):
    """the same sweep with the symmetric candidate pass (gt_sym.hip) forced on wherever it applies (euclidean self
    queries of at least 2048 rows; by default it only engages from 65536 rows)"""
    _check(sweeps, "symmetric")


def test_random_configurations_with_the_two_stage_symmetric_collect(sweeps):
    """the symmetric sweep with the two-stage collect forced on wherever it is built (32 ... 64 padded features): partial
    distances first, deferred cold pass, orphans - whatever the data look like (a queue that overflows starts over)"""
    _check(sweeps, "two_stage")
