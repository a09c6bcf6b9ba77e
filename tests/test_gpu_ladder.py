"""GPU: the decision ladder picks between candidate passes that are 2-10 x apart on a given data family.
Whatever it picks is exact (the other tests); this one bounds how badly `auto` can LOSE: on nine data families at
N = 2e5 the default build must not take more than 1.5 x the time of the better of the two forced routes (symmetric pass forced
on / classic pass forced) - plus 1 ms, the granularity of the fixed costs at this size.  Reference semantics are not involved:
all three builds produce the same graph (asserted).

What is compared is DEVICE time: the HIP-event spans of the build's top-level stages (gt_stage_ms: events recorded on the
library's stream around each stage), best of three builds - and, more loosely (2 x + 2 ms), the WALL clock of the same builds,
which also sees what the spans do not: the synchronisations and read-backs of a refuted attempt, the host gaps between stages.
The majority of up to three attempts decides: a busy box cannot turn the parity suite red with one slow attempt, and a ladder
that really picks the slow route cannot pass with one lucky one."""

import time

import numpy as np
import pytest

from conftest import make_gauss, make_manifold, make_mix

pytestmark = pytest.mark.gpu

N = 200000


def _unequal_scales(seed):
    rng = np.random.default_rng(seed)
    c = 100
    centres = rng.uniform(-10, 10, (c, 32))
    scale = rng.choice([0.2, 1.0, 3.0], size=c)
    lab = rng.integers(c, size=N)
    return (centres[lab] + scale[lab, None] * rng.standard_normal((N, 32))).astype(np.float32)


def _one_big_cell(seed):
    rng = np.random.default_rng(seed)
    X = make_mix(N, 48, seed)
    big = rng.random(N) < 0.5
    X[big] = (rng.standard_normal((int(big.sum()), 48)) * 0.5 + 3.0).astype(np.float32)   # half of the points in one tight blob
    return X


def _hubs(seed):
    rng = np.random.default_rng(seed)
    X = make_mix(N, 32, seed)
    X[:200] = (0.05 * rng.standard_normal((200, 32))).astype(np.float32) + X[200:400].mean(axis=0)   # points many rows are close to
    return X


def _isolated(seed):
    rng = np.random.default_rng(seed)
    X = make_mix(N, 48, seed)
    idx = rng.choice(N, 15, replace=False)
    X[idx] = rng.uniform(-12, 12, (15, 48)).astype(np.float32)   # rows that belong to no cluster (round 4: 4.7 x the build time then)
    return X


FAMILIES = {
    "mix d=64": lambda: make_mix(N, 64, 1),
    "mix, clusters of unequal scale d=32": lambda: _unequal_scales(2),
    "half of the points in one blob d=48": lambda: _one_big_cell(3),
    "manifold (5 dims in 64)": lambda: make_manifold(N, 64, 4),
    "isotropic gauss d=24": lambda: make_gauss(N, 24, 5),
    "isotropic gauss d=64": lambda: make_gauss(N, 64, 6),
    "mix with hubs d=32": lambda: _hubs(7),
    "mix shifted far from the origin d=64": lambda: (make_mix(N, 64, 8) + np.float32(300.0)),
    "mix with 15 isolated points d=48": lambda: _isolated(9),
}


# the top-level stage spans of gt_graph_build (nested ones - symm_bins, symm_merge ... - are parts of these)
TOP_STAGES = ("query_order", "sym_prepare", "sym_seed", "sym_bound", "knn_select", "sym_cold", "rerank", "fallback", "radius",
              "affinity", "symmetrize", "normalize")


def _build_ms(X, opts, reps=3):
    from graphtools_amd import _hip

    c = _hip.Context(0)
    try:
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_points(X)
        p, keep = c.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
        best, wall, nnz = None, None, None
        for _ in range(reps):
            c.set_points(X)
            c.sync()
            t0 = time.perf_counter()
            nnz, _ = c.graph_build(p)
            c.sync()
            w = (time.perf_counter() - t0) * 1e3
            ms = sum(max(c.stage_ms(s), 0.0) for s in TOP_STAGES)
            best = ms if best is None else min(best, ms)
            wall = w if wall is None else min(wall, w)
            if w > 1000.0:   # a route that takes seconds where the others take a fraction (the symmetric pass FORCED on points
                break        # shifted 300 units from the origin: 8.8 s per build) is not timed three times
        sym = bool(c.knn_stats()["symmetric"])
        deg = c.graph_fetch_vec(1)
        return best, int(nnz), sym, float(deg.sum()), wall
    finally:
        c.close()


@pytest.mark.parametrize("family", list(FAMILIES))
def test_auto_is_never_far_behind_the_better_forced_route(family):
    X = FAMILIES[family]()
    passed, failed, last = 0, 0, None
    for attempt in range(3):
        auto, nnz_a, sym_a, s_a, wall_a = _build_ms(X, {})
        forced_sym, nnz_s, sym_s, s_s, wall_s = _build_ms(X, {"select_symmetric": "1"})
        classic, nnz_c, sym_c, s_c, wall_c = _build_ms(X, {"select_symmetric": "0"})
        assert nnz_a == nnz_s == nnz_c and s_a == s_s == s_c, "the routes built different graphs"
        assert not sym_c
        best, best_wall = min(forced_sym, classic), min(wall_s, wall_c)
        print("%-42s auto %.2f ms (%s)  symmetric forced %.2f  classic %.2f  (device time; wall %.2f / %.2f / %.2f; attempt %d)" % (
            family, auto, "symmetric" if sym_a else "classic", forced_sym, classic, wall_a, wall_s, wall_c, attempt + 1))
        last = (auto, best, wall_a, best_wall)
        # device time: the tight bound; wall clock (best of three builds, host gaps, read-backs and refuted attempts of a wrong
        # rung included): a looser one - round-5 advisor: a ladder that loses on the HOST side must not pass on device spans
        ok = auto <= 1.5 * best + 1.0 and wall_a <= 2.0 * best_wall + 2.0
        passed += ok
        failed += not ok
        if passed >= 2 or failed >= 2:   # the majority of three attempts decides (a single lucky attempt does not)
            break
    assert passed >= 2, "auto %.2f ms vs the better forced route %.2f ms (device), wall %.2f vs %.2f, on '%s' (%d of %d attempts passed)" % (
        last[0], last[1], last[2], last[3], family, passed, passed + failed)
