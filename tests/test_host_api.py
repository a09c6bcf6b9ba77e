"""CPU: host-side mirror of the reference interface - class selection, argument validation, warnings and
parameter guards (reference behaviours listed in SURVEY.md Appendix D).  No kernel is built here
(initialize=False), so no GPU is needed."""
import warnings

import numpy as np
import pytest

import graphtools_amd
from graphtools_amd import graphs

X = np.random.default_rng(0).standard_normal((60, 10)).astype(np.float32)


def test_class_selection_rules():
    assert type(graphtools_amd.Graph(X, knn=3, decay=10, initialize=False)).__name__ == "kNNGraph"
    assert type(graphtools_amd.Graph(X, knn=3, decay=None, initialize=False)).__name__ == "kNNGraph"
    assert type(graphtools_amd.Graph(X, knn=3, decay=10, thresh=0, initialize=False)).__name__ == "TraditionalGraph"
    D = np.abs(np.random.default_rng(1).standard_normal((20, 20)))
    g = graphtools_amd.Graph(D, precomputed="distance", knn=3, decay=10, initialize=False)
    assert type(g).__name__ == "TraditionalGraph"
    g = graphtools_amd.Graph(X, knn=3, decay=10, n_landmark=10, initialize=False)
    assert type(g).__name__ == "kNNLandmarkGraph"


def test_initialize_false_is_lazy():
    g = graphtools_amd.Graph(X, knn=3, decay=10, initialize=False)
    assert not hasattr(g, "_kernel")


def test_reference_error_messages():
    with pytest.raises(ValueError, match="kNNGraph does not support precomputed"):
        graphtools_amd.Graph(X, graphtype="knn", precomputed="distance", initialize=False)
    with pytest.raises(ValueError, match="graphtype 'hello' not recognized"):
        graphtools_amd.Graph(X, graphtype="hello", initialize=False)
    with pytest.raises(ValueError, match="Cannot instantiate a kNNGraph with `decay=None`, `thresh=0`"):
        graphs.kNNGraph(X, knn=3, decay=10, thresh=0, initialize=False)
    with pytest.raises(ValueError, match="Expected 0 <= anisotropy <= 1"):
        graphtools_amd.Graph(X, knn=3, decay=10, anisotropy=2, initialize=False)
    with pytest.raises(ValueError, match="kernel_symm 'x' not recognized"):
        graphtools_amd.Graph(X, knn=3, decay=10, kernel_symm="x", initialize=False)
    with pytest.raises(TypeError, match="unexpected keyword argument"):
        graphtools_amd.Graph(X, knn=3, decay=10, hello=1, initialize=False)
    with pytest.raises(NotImplementedError):
        graphs.kNNGraph(X, knn=3, decay=10, bandwidth=lambda d: d, initialize=False)


def test_reference_warnings():
    with pytest.warns(UserWarning, match=r"Cannot set knn \(100\) to be greater than n_samples - 2 \(58\)"):
        g = graphs.kNNGraph(X, knn=100, decay=10, initialize=False)
    assert g.knn == 58
    with pytest.warns(UserWarning, match=r"Cannot set knn_max \(2\) to be less than knn \(5\)"):
        g = graphs.kNNGraph(X, knn=5, knn_max=2, decay=10, initialize=False)
    assert g.knn_max == 5
    with pytest.warns(UserWarning, match="`bandwidth` is not used when `decay=None`"):
        graphs.kNNGraph(X, knn=5, decay=None, bandwidth=2.0, initialize=False)
    with pytest.warns(UserWarning, match="kernel_symm='\\+' but theta is not None"):
        g = graphtools_amd.Graph(X, knn=3, decay=10, theta=0.5, initialize=False)
    assert g.kernel_symm == "mnn"
    with pytest.warns(UserWarning, match="kernel_symm='mnn' but theta not given"):
        g = graphtools_amd.Graph(X, knn=3, decay=10, kernel_symm="mnn", initialize=False)
    assert g.theta == 1


def test_thresh_clamped_to_eps():
    g = graphs.kNNGraph(X, knn=3, decay=10, thresh=1e-30, initialize=False)
    assert g.thresh == np.finfo(float).eps


def test_get_set_params_guards():
    g = graphtools_amd.Graph(X, knn=3, decay=10, initialize=False, random_state=4)
    p = g.get_params()
    assert set(p) == {"n_pca", "random_state", "kernel_symm", "theta", "anisotropy", "knn", "decay", "bandwidth",
                      "bandwidth_scale", "knn_max", "distance", "thresh", "n_jobs", "verbose"}
    for key, val in [("knn", 4), ("decay", 11), ("thresh", 1e-3), ("kernel_symm", "*"), ("anisotropy", 0.5),
                     ("bandwidth", 3), ("bandwidth_scale", 2), ("distance", "cosine"), ("theta", 0.3), ("n_pca", 3)]:
        with pytest.raises(ValueError, match="Cannot update {}".format(key)):
            g.set_params(**{key: val})
    g.set_params(n_jobs=4, random_state=13, verbose=2)
    assert (g.n_jobs, g.random_state, g.verbose) == (4, 13, 2)


def test_pygsp_is_out_of_scope_and_mnn_landmark_selection():
    with pytest.raises(NotImplementedError):
        graphtools_amd.Graph(X, use_pygsp=True, initialize=False)
    g = graphtools_amd.Graph(X, sample_idx=np.arange(60) % 2, n_landmark=10, initialize=False)
    assert type(g).__name__ == "MNNLandmarkGraph" and g.n_landmark == 10 and not hasattr(g, "_kernel")
    assert [c.__name__ for c in type(g).__mro__[:3]] == ["MNNLandmarkGraph", "MNNGraph", "LandmarkGraph"]


def test_mnn_graph_selection_and_validation():
    # reference: api.py:196-236, graphs.py:1743-1790, test/test_mnn.py
    idx = np.arange(60) % 3
    g = graphtools_amd.Graph(X, sample_idx=idx, knn=4, initialize=False)
    assert type(g).__name__ == "MNNGraph" and not hasattr(g, "_kernel")
    assert list(g.samples) == [0, 1, 2] and list(g.n_cells) == [20, 20, 20]
    assert g.get_params()["beta"] == 1 and g.get_params()["knn"] == 4
    with pytest.raises(ValueError, match="Cannot update beta"):
        g.set_params(beta=0.5)
    with pytest.raises(ValueError, match="Cannot update knn"):
        g.set_params(knn=7)
    with pytest.raises(ValueError, match="must be the same length as data"):
        graphtools_amd.graphs.MNNGraph(X, sample_idx=idx[:-1], initialize=False)
    with pytest.raises(ValueError, match="more than one unique value"):
        graphtools_amd.graphs.MNNGraph(X, sample_idx=np.zeros(60), initialize=False)
    with pytest.raises(TypeError, match="Expected `theta` as a float"):
        graphtools_amd.graphs.MNNGraph(X, sample_idx=idx, kernel_symm="mnn", theta="a", initialize=False)
    with pytest.warns(DeprecationWarning, match="adaptive_k"):
        graphtools_amd.graphs.MNNGraph(X, sample_idx=idx, adaptive_k="sqrt", initialize=False)
    with pytest.warns(UserWarning, match="Only one unique sample"):
        g1 = graphtools_amd.Graph(X, sample_idx=np.zeros(60), initialize=False)
    assert type(g1).__name__ == "kNNGraph"
    with pytest.raises(ValueError, match="MNNGraph does not support precomputed"):
        graphtools_amd.Graph(X, sample_idx=idx, precomputed="distance", graphtype="mnn", initialize=False)
    with pytest.raises(NotImplementedError):
        g.build_kernel_to_data(X)


def test_landmark_graphs_use_the_landmark_out_of_sample_methods():
    """reference MRO: only DataGraph and LandmarkGraph define extend_to_data / interpolate, so on every landmark graph the
    landmark versions (cluster-aggregated transitions, default to self.transitions) apply (graphs.py:1247-1317)"""
    from graphtools_amd import graphs

    for cls in (graphs.kNNLandmarkGraph, graphs.MNNLandmarkGraph, graphs.TraditionalLandmarkGraph):
        assert cls.extend_to_data is graphs.LandmarkGraph.extend_to_data, cls
        assert cls.interpolate is graphs.LandmarkGraph.interpolate, cls
    assert graphs.kNNGraph.extend_to_data is not graphs.LandmarkGraph.extend_to_data


def test_pca_backend_selection_is_host_logic():
    """Data._reduce_data: the device solver applies to dense float32 inputs with n >= d and at most 118 components, from
    2^24 elements on in auto mode; everything else is scikit-learn's randomized PCA as in the reference"""
    import numpy as np

    from graphtools_amd import _pca, base

    X32 = np.zeros((300, 40), dtype=np.float32)
    assert _pca.device_pca_applies(X32, 10)
    assert not _pca.device_pca_applies(X32.astype(np.float64), 10)      # float64 keeps sklearn's float64 solver
    assert not _pca.device_pca_applies(X32.T.copy(), 10)                # wide data (n < d): sklearn transposes
    assert not _pca.device_pca_applies(np.zeros((5000, 400), np.float32), 119)   # 119 + 10 oversamples > 128 columns
    assert not _pca.device_pca_applies(X32, 40)                         # no reduction at all
    assert base.PCA_BACKEND in ("auto", "device", "sklearn")
    # small inputs stay on the host in auto mode: no GPU is touched here
    rng = np.random.default_rng(0)
    D = base.Data(rng.standard_normal((200, 30)).astype(np.float32), n_pca=5, random_state=0)
    assert type(D.data_pca).__name__ == "PCA" and D.data_nu.shape == (200, 5)


def test_n_pca_rank_estimate():
    """n_pca=True / "auto" (reference base.py:137-283): all but one direction, gated by the singular values"""
    from graphtools_amd.base import Data

    rng = np.random.default_rng(0)
    A = rng.standard_normal((300, 6)) @ rng.standard_normal((6, 20))     # rank 6
    for flag in (True, "auto", "AUTO"):
        d = Data(A, n_pca=flag)
        assert d.n_pca == 6 and d.data_nu.shape == (300, 6)
        assert d.data_pca.components_.shape == (6, 20) and d.data_pca.singular_values_.shape == (6,)
        # default threshold: largest singular value x eps of the dtype x the larger dimension
        s0 = d.data_pca.singular_values_.max()
        assert d.rank_threshold == pytest.approx(s0 * np.finfo(np.float64).eps * 300)
    sv = Data(A, n_pca=True).data_pca.singular_values_
    d = Data(A, n_pca="auto", rank_threshold=float(sv[2]) * 0.999)
    assert d.n_pca == 3
    with pytest.raises(ValueError, match="greater than maximum singular value"):
        Data(A, n_pca=True, rank_threshold=float(sv[0]) * 10)
    with pytest.raises(ValueError, match="rank_threshold must be positive float or 'auto'"):
        Data(A, n_pca=True, rank_threshold=-1.0)
    with pytest.raises(ValueError, match="rank_threshold must be positive float or 'auto'"):
        Data(A, n_pca=True, rank_threshold="big")
    with pytest.raises(ValueError, match="or in \\[None, False, True, 'auto'\\]"):
        Data(A, n_pca="all")
    with pytest.warns(RuntimeWarning, match="Rounding to 5"):
        assert Data(A, n_pca=4.6).n_pca == 5
    with pytest.warns(RuntimeWarning, match="rank_threshold of 0.5 will not be used"):
        Data(A, n_pca=3, rank_threshold=0.5)
    with pytest.raises(ValueError, match="n_pca cannot be negative"):
        Data(A, n_pca=-2)
    from scipy import sparse

    S = sparse.random(200, 30, density=0.2, random_state=1, format="coo")
    d = Data(S, n_pca=True)
    assert d.data_nu.shape == (200, d.n_pca) and 1 <= d.n_pca <= 29


def test_result_arrays_are_recycled_only_when_nobody_refers_to_them():
    """graphtools_amd._hip._HostPool: the big result arrays (CSR values / indices of a graph) come from host blocks that are
    handed out again once every array of an earlier result is gone - a view, a slice or a scipy matrix built on one keeps its
    block out of circulation (scipy shares the memory: the blocks are of the requested dtype), small arrays are plain
    np.empty, release_cached_memory() empties the pool."""
    import gc

    from scipy import sparse

    from graphtools_amd import _hip

    pool = _hip._HostPool()
    n = pool.MIN_BYTES // 8 + 1000

    def addr(a):
        return a.__array_interface__["data"][0]

    a = pool.empty(n, np.float64)
    assert a.dtype == np.float64 and a.shape == (n,) and a.base is not None
    pa = addr(a)
    b = pool.empty(n, np.float64)
    assert addr(b) != pa, "a live array's block was handed out again"
    v = a[5:100]
    del a
    c = pool.empty(n, np.float64)
    assert addr(c) != pa, "a view still refers to the block"
    del v
    d = pool.empty(n - 10, np.float64)
    assert addr(d) == pa, "a released block of the right size should be re-used"
    assert addr(pool.empty(n, np.int32)) not in (pa, addr(b), addr(c)), "blocks are per dtype"
    # scipy keeps the arrays it is given (no copy): the matrix holds the block
    x = pool.empty(n, np.float64)
    x[:] = 1.0
    idx = np.zeros(n, dtype=np.int32)
    ptr = np.array([0, n], dtype=np.int32)
    M = sparse.csr_matrix((x, idx, ptr), shape=(1, 1))
    assert np.shares_memory(M.data, x)
    px = addr(x)
    del x
    assert addr(pool.empty(n, np.float64)) != px
    del M
    gc.collect()
    assert addr(pool.empty(n, np.float64)) == px
    assert pool.empty(100, np.float64).base is None
    pool.cap = 0
    assert pool.empty(n, np.float64).base is None
    pool.clear()
    assert pool.blocks == []
