"""Host-side mirror of the reference's abstract graph layer (graphtools/base.py).

Only what the hot path needs: parameter handling and validation with the reference's
messages, the lazy cached properties ``K``/``kernel``, ``P``/``diff_op``, ``kernel_degree``,
``diff_aff`` and the post-build sanity warnings.  All numerics run in the HIP library
(:mod:`graphtools_amd._hip`); there is no CPU implementation behind these classes.
"""
import numbers
import os
import warnings

import numpy as np
from scipy import sparse

from . import _hip


def log_task(verbose, name, seconds):
    """The reference reports its phases through ``tasklogger`` ("Calculating KNN search..." / "Calculated KNN search in
    1.23 seconds.", graphs.py:873-885, 1197, base.py:242).  Here the phases run on the device back to back and their times
    come from the library's stage timers afterwards: one completed-task line per phase, in tasklogger's wording, when the
    graph was made with a truthy ``verbose``."""
    if verbose:
        print("Calculated {} in {:.2f} seconds.".format(name, seconds), flush=True)


# attributes that hold device state (a ctypes handle, what the context currently holds): never pickled, rebuilt lazily
# (a row-sharded graph also holds its ShardedKnnGraph - a ctypes context and CUDA tensors - and the process group)
_DEVICE_ATTRS = ("_hip_ctx", "_points_bound", "_device_state", "_knn_tree", "_sharded", "group")


class BaseGraph(object):
    """Parent graph class (reference: graphtools/base.py:427-724).

    Parameters
    ----------
    kernel_symm : {'+', '*', 'mnn', None}
    theta : float, mnn symmetrisation constant
    anisotropy : float in [0, 1]
    initialize : bool, build the kernel on construction
    """

    def __init__(self, kernel_symm="+", theta=None, anisotropy=0, gamma=None, initialize=True, **kwargs):
        if gamma is not None:
            warnings.warn("gamma is deprecated. Setting theta={}".format(gamma), FutureWarning)
            theta = gamma
        if kernel_symm in ("gamma", "theta"):
            warnings.warn(
                "kernel_symm='{}' is deprecated. Setting kernel_symm='mnn'".format(kernel_symm), FutureWarning
            )
            kernel_symm = "mnn"
        self.kernel_symm = kernel_symm
        self.theta = theta
        self._check_symmetrization(kernel_symm, theta)
        if not (isinstance(anisotropy, numbers.Real) and 0 <= anisotropy <= 1):
            raise ValueError("Expected 0 <= anisotropy <= 1. Got {}".format(anisotropy))
        self.anisotropy = anisotropy
        if kwargs:
            # reference: Base.__init__ is object.__init__ -> unexpected keyword TypeError (test_api.py:161-165)
            raise TypeError("__init__() got an unexpected keyword argument '{}'".format(sorted(kwargs)[0]))
        if initialize:
            self._initialize()

    def _ensure_built(self):
        """The kernel exists where the device operations read it (default: the host copy is made too)."""
        self.K

    def _initialize(self):
        """``initialize=True`` (reference: base.py:77-83 builds ``self.K``).  A graph whose kernel lives on the device may build it there
        and leave the host copy to the first access of ``K`` / ``P`` (kNNGraph)."""
        self.K

    def _check_symmetrization(self, kernel_symm, theta):
        # reference: base.py:508-532
        if kernel_symm not in ["+", "*", "mnn", None]:
            raise ValueError(
                "kernel_symm '{}' not recognized. Choose from '+', '*', 'mnn', or 'none'.".format(kernel_symm)
            )
        elif kernel_symm != "mnn" and theta is not None:
            warnings.warn("kernel_symm='{}' but theta is not None. Setting kernel_symm='mnn'.".format(kernel_symm))
            self.kernel_symm = kernel_symm = "mnn"
        if kernel_symm == "mnn":
            if theta is None:
                self.theta = theta = 1
                warnings.warn("kernel_symm='mnn' but theta not given. Defaulting to theta={}.".format(self.theta))
            elif not isinstance(theta, numbers.Number) or theta < 0 or theta > 1:
                raise ValueError("theta {} not recognized. Expected a float between 0 and 1".format(theta))

    # ---- parameters -------------------------------------------------------------------------
    def get_params(self):
        return {"kernel_symm": self.kernel_symm, "theta": self.theta, "anisotropy": self.anisotropy}

    def set_params(self, **params):
        # reference: base.py:602-627
        if "theta" in params and params["theta"] != self.theta:
            raise ValueError("Cannot update theta. Please create a new graph")
        if "anisotropy" in params and params["anisotropy"] != self.anisotropy:
            raise ValueError("Cannot update anisotropy. Please create a new graph")
        if "kernel_symm" in params and params["kernel_symm"] != self.kernel_symm:
            raise ValueError("Cannot update kernel_symm. Please create a new graph")
        return self

    # ---- pickling (reference: base.py:887-902 to_pickle; graphs are plain picklable objects there) ----------
    def __getstate__(self):
        """Everything but the device context: the host-side results that were already fetched (K, P, ...) travel, what lives
        on the GPU is rebuilt on first use after loading (``_bind_points`` / ``_ensure_device_graph``)."""
        if not hasattr(self, "_kernel") and getattr(self, "_device_state", None) is not None:
            self.K   # (built on the device, never fetched: the pickle holds what the reference's holds)
        state = dict(self.__dict__)
        for key in _DEVICE_ATTRS:
            state.pop(key, None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    def to_pickle(self, path):
        """Save the current Graph to a pickle (reference: base.py:887-902)."""
        import pickle

        with open(path, "wb") as f:
            pickle.dump(self, f, protocol=pickle.HIGHEST_PROTOCOL)

    # ---- device plumbing --------------------------------------------------------------------
    @property
    def hip(self):
        """The :class:`graphtools_amd._hip.Context` that owns this graph's device state."""
        try:
            return self._hip_ctx
        except AttributeError:
            self._hip_ctx = _hip.Context(getattr(self, "device", 0) or 0)
            return self._hip_ctx

    def _emit_build_warnings(self, flags, K=None):
        """The sanity warnings of BaseGraph._build_kernel (reference: base.py:551-554).  Every symmetrisation of the
        merge kernel is symmetric by construction; without one (kernel_symm=None) the reference's own test
        ``(K - K.T).max() > 1e-5`` is evaluated on the fetched matrix."""
        if K is not None and getattr(self, "kernel_symm", "+") is None:
            diff = K - K.T
            asym = diff.max() if diff.shape[0] else 0.0
            if asym > 1e-5:
                warnings.warn("K should be symmetric", RuntimeWarning)
        if flags & _hip.FLAG_ZERO_DIAGONAL:
            warnings.warn("K should have a non-zero diagonal", RuntimeWarning)

    # ---- abstract ---------------------------------------------------------------------------
    def _build_kernel(self):
        """Build kernel + diffusion operator on the device; returns K (scipy CSR or ndarray)."""
        raise NotImplementedError

    def _fetch_diff_op(self):
        raise NotImplementedError

    def _fetch_degree(self):
        raise NotImplementedError

    # ---- cached properties (reference: base.py:629-724) ---------------------------------------
    @property
    def K(self):
        try:
            return self._kernel
        except AttributeError:
            self._kernel = self._build_kernel()
            return self._kernel

    @property
    def kernel(self):
        return self.K

    @property
    def P(self):
        try:
            return self._diff_op
        except AttributeError:
            self.K
            self._diff_op = self._fetch_diff_op()
            return self._diff_op

    @property
    def diff_op(self):
        return self.P

    @property
    def kernel_degree(self):
        try:
            return self._kernel_degree
        except AttributeError:
            self.K
            self._kernel_degree = np.asarray(self._fetch_degree(), dtype=np.float64).reshape(-1, 1)
            return self._kernel_degree

    @property
    def diff_aff(self):
        """Symmetric diffusion affinity D^-1/2 K D^-1/2 (reference: base.py:668-698).

        Sparse kernels that live on the device (kNN and MNN graphs): one pass over K's entries (``gt_graph_diff_aff``),
        the result shares K's CSR structure.  Dense kernels (exact graphs, whose K was handed to the host when it was
        built): the reference's two numpy divisions."""
        deg = self.kernel_degree
        if sparse.issparse(self.kernel):
            fetch = getattr(self, "_fetch_diff_aff", None)
            if fetch is not None:
                return fetch()
            n = len(deg)
            dm = sparse.csr_matrix((1 / np.sqrt(deg.flatten()), np.arange(n), np.arange(n + 1)))
            return dm @ self.kernel @ dm
        return (self.kernel / np.sqrt(deg)) / np.sqrt(deg.T)

    # convenience used by downstream packages
    @property
    def N(self):
        return self.K.shape[0]


# PCA pre-reduction backend: "auto" (device for large dense float32 inputs the device solver applies to), "device"
# (whenever it applies), "sklearn" (the reference's call, always).  GRAPHTOOLS_AMD_PCA sets the default.
PCA_BACKEND = os.environ.get("GRAPHTOOLS_AMD_PCA", "auto")
_PCA_DEVICE_MIN_ELEMENTS = 1 << 24
# spectral landmark front end (LandmarkGraph, random_landmarking=False): "auto" = device from 20 000 samples on
SPECTRAL_BACKEND = os.environ.get("GRAPHTOOLS_AMD_SPECTRAL", "auto")
_SPECTRAL_DEVICE_MIN_ROWS = 20000


class Data(object):
    """Input coercion and optional PCA pre-reduction (reference: graphtools/base.py:72-424).

    Dense float32 inputs are reduced by ``graphtools_amd._pca.DevicePCA`` - sklearn's randomized solver with its tall
    matrix products on the device (statistical parity, see that module); sparse, float64, wide or small inputs go
    through scikit-learn exactly as in the reference.  ``data_nu`` is what the graph kernels see.
    """

    def __init__(self, data, n_pca=None, rank_threshold=None, random_state=None):
        if hasattr(data, "sparse") and hasattr(data, "columns"):  # pandas sparse frame
            data = data.sparse.to_coo()
        elif hasattr(data, "columns") and hasattr(data, "values"):  # pandas DataFrame
            data = np.array(data)
        elif hasattr(data, "X") and hasattr(data, "obs"):  # anndata
            data = data.X
        if not sparse.issparse(data):
            data = np.asarray(data)
            if data.ndim != 2:
                raise ValueError("Expected 2D array, got {}D array instead".format(data.ndim))
        if min(data.shape) == 0:
            raise ValueError("Found array with 0 sample(s) or feature(s): {}".format(data.shape))
        n_pca, rank_threshold = self._resolve_n_pca(data.shape, n_pca, rank_threshold)
        self.data = data
        self.n_pca = n_pca
        self.rank_threshold = rank_threshold
        self.random_state = random_state
        self.data_nu = self._reduce_data()

    @staticmethod
    def _resolve_n_pca(shape, n_pca, rank_threshold):
        """The accepted forms of ``n_pca`` / ``rank_threshold`` and their warnings (reference: base.py:137-213): an integer
        (fractions are rounded, too large a value switches the reduction off), None / 0 / False (no reduction), True /
        "auto" (rank estimate from the singular values, see ``_reduce_data``)."""
        bad = ValueError("n_pca was not an instance of numbers.Number, could not be cast to False, and not None. "
                         "Please supply an integer 0 <= n_pca < min(n_samples,n_features) or None")
        if isinstance(n_pca, str):
            if n_pca.lower() != "auto":
                raise ValueError("n_pca must be an integer 0 <= n_pca < min(n_samples,n_features), "
                                 "or in [None, False, True, 'auto'].")
            n_pca = "auto"
        elif n_pca is True:
            n_pca = "auto"
        elif n_pca is None or n_pca is False:
            n_pca = None
        elif isinstance(n_pca, numbers.Number):
            if not float(n_pca).is_integer():
                rounded = int(np.round(n_pca))
                warnings.warn("Cannot perform PCA to fractional {} dimensions. Rounding to {}".format(n_pca, rounded),
                              RuntimeWarning)
                n_pca = rounded
            n_pca = int(n_pca)
            if n_pca < 0:
                raise ValueError("n_pca cannot be negative. Please supply an integer "
                                 "0 <= n_pca < min(n_samples,n_features) or None")
            if n_pca >= min(shape):
                warnings.warn("Cannot perform PCA to {} dimensions on data with min(n_samples, n_features) = {}".format(
                    n_pca, min(shape)), RuntimeWarning)
                n_pca = None
            elif n_pca == 0:
                n_pca = None
        else:
            raise bad
        if n_pca != "auto":
            if rank_threshold is not None:
                warnings.warn("n_pca = {}, therefore rank_threshold of {} will not be used. To use rank thresholding, "
                              "set n_pca = True".format(n_pca, rank_threshold), RuntimeWarning)
            return n_pca, rank_threshold
        if rank_threshold is None:
            rank_threshold = "auto"
        elif isinstance(rank_threshold, str):
            rank_threshold = rank_threshold.lower()
        ok = rank_threshold == "auto" or (isinstance(rank_threshold, numbers.Number) and rank_threshold > 0)
        if not ok:
            raise ValueError("rank_threshold must be positive float or 'auto'. ")
        return n_pca, rank_threshold

    def _reduce_auto(self):
        """n_pca = "auto": all but one principal direction, then only those whose singular value reaches the threshold
        (default: largest singular value x machine epsilon of the data's dtype x the larger dimension) are kept
        (reference: base.py:243-283).  Solved by scikit-learn on the host, as in the reference - the rank estimate needs the
        whole spectrum, which is not what the device solver (a few leading components of a tall matrix) is for."""
        from sklearn.decomposition import PCA, TruncatedSVD

        k = self.data.shape[1] - 1
        if sparse.issparse(self.data):
            if not isinstance(self.data, (sparse.csr_matrix, sparse.csc_matrix)):
                self.data = self.data.tocsr()
            op = TruncatedSVD(k, random_state=self.random_state)
        else:
            op = PCA(k, svd_solver="randomized", random_state=self.random_state)
        op.fit(self.data)
        sv = op.singular_values_
        if self.rank_threshold == "auto":
            self.rank_threshold = sv.max() * np.finfo(self.data.dtype).eps * max(self.data.shape)
        keep = np.flatnonzero(sv >= self.rank_threshold)
        if keep.size == 0:
            raise ValueError("Supplied threshold {} was greater than maximum singular value {} for the data matrix".format(
                self.rank_threshold, sv.max()))
        self.n_pca = int(keep.size)
        for name in ("components_", "explained_variance_", "explained_variance_ratio_", "singular_values_"):
            setattr(op, name, getattr(op, name)[keep])
        self.data_pca = op
        return op.transform(self.data)

    def _reduce_data(self):
        if self.n_pca == "auto":
            return self._reduce_auto()
        if self.n_pca is None:
            d = self.data
            if sparse.issparse(d):
                d = d.toarray()
            return d
        from sklearn.decomposition import PCA, TruncatedSVD

        if sparse.issparse(self.data):
            self.data_pca = TruncatedSVD(self.n_pca, random_state=self.random_state)
        else:
            from ._pca import DevicePCA, device_pca_applies

            backend = PCA_BACKEND
            if backend != "sklearn" and device_pca_applies(self.data, self.n_pca) and (
                    backend == "device" or self.data.size >= _PCA_DEVICE_MIN_ELEMENTS):
                from ._pca import DevicePCAUnsuitable

                try:
                    self.data_pca = DevicePCA(self.n_pca, random_state=self.random_state,
                                              device=getattr(self, "device", 0) or 0)
                    return self.data_pca.fit_transform(self.data)
                except DevicePCAUnsuitable:
                    pass   # (large offsets: scikit-learn centres before its products)
            self.data_pca = PCA(self.n_pca, svd_solver="randomized", random_state=self.random_state)
        self.data_pca.fit(self.data)
        return self.data_pca.transform(self.data)

    def get_params(self):
        return {"n_pca": self.n_pca, "random_state": self.random_state}

    def set_params(self, **params):
        if "n_pca" in params and params["n_pca"] != self.n_pca:
            raise ValueError("Cannot update n_pca. Please create a new graph")
        if "random_state" in params:
            self.random_state = params["random_state"]
        return self
