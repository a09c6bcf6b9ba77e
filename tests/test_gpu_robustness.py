"""GPU: error paths leave the library usable."""
import ctypes

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def test_a_failed_allocation_does_not_poison_the_next_call():
    """a call that cannot get its device memory reports 'out of memory' and leaves nothing behind: the next call on the thread
    (whose launch checks read HIP's last-error slot) succeeds.  The failure is provoked without filling the device: the dense
    copy of a 300 000-row graph needs a 720 GB staging buffer, refused before anything is written."""
    from graphtools_amd import _hip

    rng = np.random.default_rng(0)
    X = rng.standard_normal((300000, 2)).astype(np.float32)
    ctx = _hip.Context(0)
    try:
        ctx.set_points(X)
        params, keep = _hip.Context.make_params(3, 40.0, 1e-3, None, 1.0, None, "+", None, 0)
        ctx.graph_build(params)
        dummy = np.zeros(8, dtype=np.float64)
        rc = ctx.lib.gt_graph_to_dense(ctx.h, _hip.CSR_K, dummy.ctypes.data_as(ctypes.c_void_p), _hip.GT_F64, 0)
        assert rc != 0 and b"out of memory" in ctx.lib.gt_last_error(ctx.h)
        assert not dummy.any()
        # same context, next call
        d, i, _ = ctx.knn_search(4)
        assert i.shape == (300000, 4)
    finally:
        ctx.close()
    ctx = _hip.Context(0)
    try:
        ctx.set_points(X[:5000])
        d, i, _ = ctx.knn_search(5)
        d0, i0 = oracle.kneighbors(X[:5000], None, 5)
        assert np.array_equal(i, i0)
    finally:
        ctx.close()


def test_a_kernel_that_reaches_too_far_is_a_limit_with_a_message_not_an_allocator_failure():
    """round-3 fuzz leftovers (n ~ 1e5, decay 2, thresh 1e-3, bandwidth_scale > 1): every point lies inside every row's kernel
    radius, the radius lists would need n x n entries.  The build says so (GT_E_LIMIT: what it needs, what the GPU has, what to
    change) instead of dying in hipMalloc; the context stays usable."""
    import numpy as np
    from graphtools_amd import _hip

    rng = np.random.default_rng(0)
    X = rng.standard_normal((140000, 8)).astype(np.float32)
    c = _hip.Context(0)
    c.set_points(X)
    p, keep = c.make_params(5, 2.0, 1e-3, None, 50.0, None, "+", None, 0)
    with pytest.raises(_hip.HipError, match="radius pass: .* GB of lists|does not fit the GPU"):
        c.graph_build(p)
    p2, keep2 = c.make_params(5, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
    nnz, flags = c.graph_build(p2)          # the same context builds an ordinary graph afterwards
    assert nnz > 140000
    c.close()
