"""GPU: single-rank symmetrisation through destination bins (gt_sparse.hip bin_count / bin_emit / bin_fill) against the
oracle and against the triplet-exchange path it replaces.  The bin path engages by itself from 65536 rows when the
cell-sorted order of the points exists; here it is forced on small point sets (`symmetrize_bins=1`,
`query_order_min_rows=1`).  Bar: K and P identical to the exchange path bit for bit, CSR structure identical to the
oracle's, values within 1e-5 relative."""
import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import make_gauss, make_manifold, make_mix

pytestmark = pytest.mark.gpu


def _build(X, bins, knn=15, decay=40.0, thresh=1e-4, bandwidth=None, symm="+", theta=None, aniso=0, opts=()):
    from graphtools_amd import _hip

    c = _hip.Context(0)
    c.set_option("query_order_min_rows", "1")
    c.set_option("symmetrize_bins", str(bins))
    for k, v in opts:
        c.set_option(k, str(v))
    c.set_points(X)
    p, keep = c.make_params(knn, decay, thresh, bandwidth, 1.0, None, symm, theta, aniso)
    c.graph_build(p)
    ran = c.stage_ms("symm_bins") >= 0
    K = c.graph_fetch_csr(_hip.CSR_K)
    P = c.graph_fetch_csr(_hip.CSR_P)[0]
    st = c.graph_stats()
    c.close()
    return K, P, ran, st


def _same(a, b):
    (Kd, Ki, Kp), P = a[0], a[1]
    (Ld, Li, Lp), Q = b[0], b[1]
    assert np.array_equal(Kp, Lp) and np.array_equal(Ki, Li)
    assert np.array_equal(Kd, Ld) and np.array_equal(P, Q)


@pytest.mark.parametrize("n,d,maker,seed,kw", [
    (5003, 64, make_mix, 0, {}),                                        # ragged last bin
    (300, 10, make_gauss, 1, {"knn": 7}),                               # fewer rows than one bin
    (512, 16, make_mix, 2, {"knn": 5, "decay": 10.0}),                  # exactly one bin
    (4100, 33, make_mix, 3, {"symm": "*"}),
    (4100, 33, make_mix, 3, {"symm": "mnn", "theta": 0.3}),
    (3000, 20, make_manifold, 4, {"aniso": 1.0}),
    (2600, 50, make_mix, 5, {"decay": None, "knn": 9}),                 # binary kernel
])
def test_bin_path_equals_the_exchange_path_and_the_oracle(n, d, maker, seed, kw):
    X = maker(n, d, seed)
    b = _build(X, 1, **kw)
    e = _build(X, 0, **kw)
    assert b[2] and not e[2]
    _same(b, e)
    Ko, Po = oracle.knn_graph(X, knn=kw.get("knn", 15), decay=kw.get("decay", 40.0), kernel_symm=kw.get("symm", "+"),
                              theta=kw.get("theta"), anisotropy=kw.get("aniso", 0))
    Ko = sparse.csr_matrix(Ko)
    Ko.sort_indices()
    if kw.get("symm") == "*":
        Ko.eliminate_zeros()
    (Kd, Ki, Kp) = b[0]
    assert np.array_equal(Kp, Ko.indptr) and np.array_equal(Ki, Ko.indices)
    np.testing.assert_allclose(Kd, Ko.data, rtol=2e-5 if kw.get("symm") == "*" else 1e-5, atol=0)


def test_bin_path_with_rows_from_the_radius_pass_and_long_union_rows():
    """wide kernel: every row comes from the radius lists, union rows of > 512 and > 2048 entries (register sorts of 16 /
    32 keys per lane, global-memory sort)"""
    X = make_gauss(3000, 12, 13)
    kw = dict(knn=5, decay=2, bandwidth=2.5)
    b = _build(X, 1, **kw)
    e = _build(X, 0, **kw)
    assert b[2] and not e[2]
    assert b[3]["radius_rows"] == 3000
    assert len(b[0][0]) / 3000 > 600
    _same(b, e)


def test_bin_path_with_a_hub_row():
    """one point in the middle of everything: its union row receives a triplet from most rows (one bin holds a long run)"""
    rng = np.random.default_rng(5)
    X = rng.standard_normal((6000, 24)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)     # points on a sphere, the centre is everybody's neighbour
    X[1234] = 0
    b = _build(X, 1, knn=10, decay=20.0)
    e = _build(X, 0, knn=10, decay=20.0)
    assert b[2]
    deg = np.diff(b[0][2])
    assert deg[1234] > 3000
    _same(b, e)


@pytest.mark.parametrize("shift", [8, 11])
def test_bin_path_with_other_bin_sizes(shift):
    """256-row bins (one row per thread in the scan) and 2048-row bins (what point sets beyond 4 M rows get)"""
    X = make_mix(5003, 40, 7)
    b = _build(X, 1, opts=(("symmetrize_bin_shift", shift),))
    e = _build(X, 0)
    assert b[2]
    _same(b, e)


def test_bin_path_engages_by_itself_and_is_deterministic():
    X = make_mix(150000, 64, 12)
    a = _build(X, -1)
    b = _build(X, -1)
    e = _build(X, 0)
    assert a[2] and b[2] and not e[2]
    _same(a, b)
    _same(a, e)


@pytest.mark.parametrize("bins", [0, 1])
def test_row_sorts_on_32_bit_keys_equal_the_64_bit_ones(bins):
    """the per-row sort packs (column, tag, position) into 32 bits where they fit (default); 64-bit keys otherwise"""
    X = make_mix(6000, 30, 9)
    a = _build(X, bins, knn=40, decay=15.0)                                   # union rows of 128 ... 512 entries
    b = _build(X, bins, knn=40, decay=15.0, opts=(("symmetrize_key32", 0),))
    assert np.diff(a[0][2]).max() > 128
    _same(a, b)


def _build_pairs(X, pairs, knn=15, decay=40.0, bandwidth=None, opts=()):
    """single-rank build with the symmetric candidate pass and the bin transpose forced on; returns the CSR parts, P and which
    tail ran ("pairs": K and P written by the merge itself, no compaction pass)"""
    from graphtools_amd import _hip

    c = _hip.Context(0)
    for k, v in (("query_order_min_rows", 1), ("symmetrize_bins", 1), ("select_symmetric", 1), ("select_sym_stride", 4),
                 ("symmetrize_pairs", pairs)) + tuple(opts):
        c.set_option(k, str(v))
    c.set_points(X)
    p, keep = c.make_params(knn, decay, 1e-4, bandwidth, 1.0, None, "+", None, 0)
    out = []
    for _ in range(2):      # (twice: a refuted pair path is remembered for the point set)
        c.graph_build(p)
        tail = "pairs" if (c.stage_ms("symm_merge") >= 0 and c.stage_ms("symm_compact") < 0) else "general"
        out.append((c.graph_fetch_csr(_hip.CSR_K), c.graph_fetch_csr(_hip.CSR_P)[0], tail, c.graph_stats(), c.knn_stats(),
                    c.stage_launches("affinity")))
    c.close()
    return out


@pytest.mark.parametrize("pairs", [1, 2])
@pytest.mark.parametrize("n,d,maker,seed,kw", [
    (20000, 32, make_mix, 3, {}),
    (30000, 24, make_manifold, 4, {"knn": 10, "decay": 20.0}),
    (16000, 16, make_mix, 5, {"knn": 5, "decay": 8.0}),                  # wide kernel: radius rows, long union rows
    (9000, 40, make_mix, 6, {"bandwidth": 6.0}),                          # caller's bandwidth
])
def test_pair_resolved_tail_equals_the_general_tail(n, d, maker, seed, kw, pairs):
    """'+' rule, single rank: every row settles its mutual pairs itself from the transposed keys the re-rank left next to its
    table (or from the dot products where the table came from a repair / radius pass), only one-sided entries travel, the merge
    writes K and P at their final place.  Bar: K (structure and values) and P bit for bit those of the general tail.
    pairs = 2 (the default since round 5): the tables lie by sorted position and - where no row took the radius pass - the
    affinity pass looks the destinations up itself, the emit and the merge walk the slots (affinity_slots_kernel,
    bin_emit_slots_kernel, merge_pairs_slots_kernel); with rows of the radius pass the kernels of pairs = 1 read those tables."""
    X = maker(n, d, seed)
    a = _build_pairs(X, pairs, **kw)
    b = _build_pairs(X, 0, **kw)
    assert b[0][2] == "general" and b[1][2] == "general"
    assert a[0][4]["symmetric"], "the symmetric candidate pass did not run: nothing was tested"
    for x, y in zip(a, b):
        _same(x, y)
    # the pair path ran, or was refuted by a union row beyond the register sorts - then it stays off for the point set
    assert a[0][2] == "pairs" or a[1][2] == "general"
    if a[0][2] == "pairs":
        assert a[1][2] == "pairs"


@pytest.mark.parametrize("pairs", [1, 2])
def test_pair_resolved_tail_gives_way_to_hub_rows(pairs):
    """a point set with a hub (many rows keep one row that keeps few of them): its union row is longer than the register sorts
    hold - that row is sorted by rocPRIM's segmented sort behind the others (same bits as the general tail)"""
    rng = np.random.default_rng(8)
    X = make_mix(24000, 12, 8)
    X[:4000] = X[4000] + 0.35 * rng.standard_normal((4000, 12)).astype(np.float32)    # a dense knot around one point
    a = _build_pairs(X, pairs, knn=4, decay=3.0)
    b = _build_pairs(X, 0, knn=4, decay=3.0)
    for x, y in zip(a, b):
        _same(x, y)
    lens = np.diff(a[0][0][2])
    assert lens.max() > 2048, "the case holds no row beyond the register sorts"
    # round 4: such rows are finished by a segmented sort behind the others - the pair-resolved tail stays, one attempt
    assert a[0][2] == "pairs" and a[1][2] == "pairs" and a[0][5] == b[0][5]
    # ... and where the option says so they refute the path as they used to: the build is redone the general way, the refuted
    # first attempt stays in the stage timers (its launches are counted next to the second attempt's); once the verdict is in,
    # a build is one attempt
    c = _build_pairs(X, pairs, knn=4, decay=3.0, opts=(("symmetrize_pairs_huge", 0),))
    for x, y in zip(c, b):
        _same(x, y)
    assert c[0][2] == "general" and c[1][2] == "general"
    assert c[0][5] == 2 * c[1][5] and c[1][5] == b[0][5]


def test_default_build_of_a_large_point_set_takes_the_slot_kernels():
    """a whole single-rank '+' build of 65536+ rows with default options: the tables lie by sorted position and the affinity pass
    looks the destinations up itself (no bin_count launch) - a silent fall-back to the round-4 kernels would cost 1.2 ms per C3 graph
    without any test noticing; same K and P as the general tail"""
    from graphtools_amd import _hip

    X = make_mix(70000, 24, 11)
    out = []
    for opts in ((), (("symmetrize_pairs", "0"),)):
        c = _hip.Context(0)
        for k, v in opts:
            c.set_option(k, v)
        c.set_points(X)
        p, keep = c.make_params(10, 20.0, 1e-4, None, 1.0, None, "+", None, 0)
        c.graph_build(p)
        out.append((c.graph_fetch_csr(_hip.CSR_K), c.graph_fetch_csr(_hip.CSR_P)[0], c.knn_stats(), c.graph_stats()))
        c.close()
    st = out[0][2]
    assert st["symmetric"] and st["tables_by_slot"], st
    assert st["destinations_fused"] == (out[0][3]["radius_rows"] == 0), (st, out[0][3])
    assert not out[1][2]["tables_by_slot"]
    for a, b in zip(out[0][0], out[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(out[0][1], out[1][1])
