"""Concrete graph classes backed by the HIP library (reference: graphtools/graphs.py).

``kNNGraph``          graphtools/graphs.py:562-982   -> gt_set_points + gt_graph_build
``TraditionalGraph``  graphtools/graphs.py:1320-1704 -> gt_dense_graph_build
``LandmarkGraph``     graphtools/graphs.py:985-1317  -> gt_nearest_landmark / gt_landmark_partial
``MNNGraph``          graphtools/graphs.py:1707-1966 -> per-batch gt_graph_build + gt_graph_extend blocks + gt_csr_graph_build

Same constructor arguments, attribute names, warnings and errors as the reference for this
path; the numerics run on the MI355X.  No CPU implementation exists behind these classes.
"""
import numbers
import warnings

import numpy as np
from scipy import sparse

from . import _hip
from . import base as _base
from .base import BaseGraph, Data


class DataGraph(Data, BaseGraph):
    """Graphs built from a data matrix (reference: graphtools/base.py:1046-1254)."""

    def __init__(self, data, n_pca=None, rank_threshold=None, random_state=None, verbose=True, n_jobs=1,
                 device=None, distributed=False, group=None, **kwargs):
        self.n_jobs = n_jobs
        self.verbose = verbose
        # Where the reference spends all cores of one process (``n_jobs=-1``, api.py:35 / graphs.py:763-768) this package spends
        # all GPUs of one ``torch.distributed`` job: with ``distributed=True`` (or "auto": when a process group with more than
        # one rank is initialised) EVERY rank makes the same call with the same data, the build is row-sharded over the ranks
        # (graphtools_amd.dist, one process per GPU, RCCL) and every rank ends with the full ``K`` / ``P`` on its host, like a
        # single-process build; ``K_local`` / ``P_local`` / ``local_rows`` are the rank's own rows without the gather.
        if distributed not in (False, True, "auto", None):
            raise ValueError("distributed must be True, False or 'auto'. Got {}".format(distributed))
        self.distributed = distributed or False
        self.group = group
        if device is None and self.distributed:
            import os

            device = int(os.environ.get("LOCAL_RANK", "0"))   # (torchrun's convention: one process per GPU of the node)
        self.device = device
        if random_state is None and self._dist_ranks() is not None:
            # Every rank runs this constructor on the same data and contributes ITS rows of data_nu to the all-gather: whatever
            # is random in here (the randomized SVD behind n_pca, the landmark draw) has to be the same draw on every rank.
            # random_state=None means "seed from the OS" - per process; one seed for the job instead: rank 0's, broadcast.
            random_state = self._job_seed()
        Data.__init__(self, data, n_pca=n_pca, rank_threshold=rank_threshold, random_state=random_state)
        BaseGraph.__init__(self, **kwargs)

    def get_params(self):
        params = Data.get_params(self)
        params.update(BaseGraph.get_params(self))
        return params

    def set_params(self, **params):
        if "n_jobs" in params:
            self.n_jobs = params["n_jobs"]
        if "verbose" in params:
            self.verbose = params["verbose"]
        Data.set_params(self, **params)
        BaseGraph.set_params(self, **params)
        return self

    # ---- row-sharded builds over torch.distributed ------------------------------------------------
    def _dist_ranks(self):
        """(group, world, rank) when this graph is built row-sharded over ``torch.distributed``, else None."""
        mode = getattr(self, "distributed", False)
        if not mode:
            return None
        import torch.distributed as tdist

        if not (tdist.is_available() and tdist.is_initialized()):
            if mode == "auto":
                return None
            raise RuntimeError("distributed=True needs an initialised torch.distributed process group (one process per GPU)")
        group = getattr(self, "group", None)
        world = tdist.get_world_size(group)
        if world == 1 and mode == "auto":
            return None
        return group, world, tdist.get_rank(group)

    def _dist_device(self):
        """the torch device the collectives run on: this rank's GPU under RCCL, the host under gloo (CPU tests)"""
        import torch
        import torch.distributed as tdist

        if "nccl" in str(tdist.get_backend(getattr(self, "group", None))):
            return torch.device("cuda", int(getattr(self, "device", 0) or 0))
        return torch.device("cpu")

    def _job_seed(self):
        """one integer seed for all ranks of the job: drawn from the OS on rank 0, broadcast"""
        import torch
        import torch.distributed as tdist

        group, world, rank = self._dist_ranks()
        seed = torch.zeros(1, dtype=torch.int64, device=self._dist_device())
        if rank == 0:
            seed[0] = int(np.random.SeedSequence().generate_state(1, dtype=np.uint32)[0])
        tdist.broadcast(seed, src=tdist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return int(seed.item())

    def _no_sharded(self, what):
        if self._dist_ranks() is not None:
            raise NotImplementedError("graphtools_amd: {} is not available on a row-sharded (distributed=True) graph - build "
                                      "it with distributed=False on one GPU".format(what))

    def _check_extension_shape(self, Y):
        Y = np.asarray(Y)
        if Y.ndim != 2:
            raise ValueError("Expected a 2D matrix. Y has shape {}".format(Y.shape))
        if Y.shape[1] != self.data_nu.shape[1]:
            if Y.shape[1] == self.data.shape[1] and self.n_pca is not None:
                Y = self.data_pca.transform(Y)
            else:
                raise ValueError(
                    "Y must be of shape either (n, {}) or (n, {})".format(self.data.shape[1], self.data_nu.shape[1])
                )
        return Y


    def extend_to_data(self, Y):
        """Row-stochastic transitions from new points to the graph's data (reference: base.py:1166-1193): the kernel of
        ``build_kernel_to_data`` with its rows l1-normalised.  (kNNGraph overrides this with the device's own P.)"""
        Y = self._check_extension_shape(Y)
        return _l1_rows(self.build_kernel_to_data(Y))

    def interpolate(self, transform, transitions=None, Y=None):
        """reference: base.py:1195-1229"""
        if transitions is None:
            if Y is None:
                raise ValueError("Either `transitions` or `Y` must be provided.")
            transitions = self.extend_to_data(Y)
        return transitions.dot(transform)


def _l1_rows(M):
    # sklearn.preprocessing.normalize(M, "l1", axis=1): rows divided by the sum of their absolute values, zero rows kept
    if sparse.issparse(M):
        M = sparse.csr_matrix(M, dtype=np.float64, copy=True)
        sums = np.asarray(abs(M).sum(axis=1)).ravel()
        sums[sums == 0.0] = 1.0
        M.data /= np.repeat(sums, np.diff(M.indptr))
        return M
    M = np.asarray(M, dtype=np.float64)
    sums = np.abs(M).sum(axis=1)
    sums[sums == 0.0] = 1.0
    return M / sums[:, None]


class _KnnTree(object):
    """Stand-in for the reference's ``knn_tree`` attribute (a fitted sklearn NearestNeighbors,
    graphs.py:748-769): ``kneighbors`` runs the brute-force search on the device."""

    def __init__(self, graph):
        self._graph = graph
        self._fit_method = "brute"

    def kneighbors(self, X=None, n_neighbors=None, return_distance=True):
        g = self._graph
        if n_neighbors is None:
            n_neighbors = g.knn + 1
        g._bind_points()
        if X is None or X is g.data_nu:
            dist, idx, _ = g.hip.knn_search(int(n_neighbors))
        else:
            dist, idx, _ = g.hip.knn_search(int(n_neighbors), Y=np.asarray(X))
        return (dist, idx) if return_distance else idx


class kNNGraph(DataGraph):
    """K nearest neighbours graph with optional alpha-decay kernel (reference: graphs.py:562-982)."""

    def __init__(self, data, knn=5, decay=None, knn_max=None, search_multiplier=6, bandwidth=None,
                 bandwidth_scale=1.0, distance="euclidean", thresh=1e-4, n_pca=None, **kwargs):
        # reference: graphs.py:621-661
        if decay is not None:
            if thresh <= 0 and knn_max is None:
                raise ValueError(
                    "Cannot instantiate a kNNGraph with `decay=None`, "
                    "`thresh=0` and `knn_max=None`. Use a TraditionalGraph instead."
                )
            elif thresh < np.finfo(float).eps:
                thresh = np.finfo(float).eps
        if callable(bandwidth):
            raise NotImplementedError(
                "Callable bandwidth is only supported by graphtools.graphs.TraditionalGraph."
            )
        if knn is None and bandwidth is None:
            raise ValueError("Either `knn` or `bandwidth` must be provided.")
        elif knn is None and bandwidth is not None:
            knn = 5
        if decay is None and bandwidth is not None:
            warnings.warn("`bandwidth` is not used when `decay=None`.", UserWarning)
        n_samples = data.shape[0]
        if knn > n_samples - 2:
            warnings.warn(
                "Cannot set knn ({k}) to be greater than "
                "n_samples - 2 ({n}). Setting knn={n}".format(k=knn, n=n_samples - 2)
            )
            knn = n_samples - 2
        if knn_max is not None and knn_max < knn:
            warnings.warn(
                "Cannot set knn_max ({knn_max}) to be less than "
                "knn ({knn}). Setting knn_max={knn}".format(knn=knn, knn_max=knn_max)
            )
            knn_max = knn
        if n_pca in [None, 0, False] and data.shape[1] > 500:
            warnings.warn(
                "Building a kNNGraph on data of shape {} is expensive. Consider setting n_pca.".format(data.shape),
                UserWarning,
            )
        if distance not in ("euclidean", "cosine"):
            raise NotImplementedError(
                "graphtools_amd.kNNGraph: distance='{}' is not available on the HIP path "
                "(euclidean and cosine are)".format(distance)
            )
        self.knn = knn
        self.knn_max = knn_max
        self.search_multiplier = search_multiplier
        self.decay = decay
        self.bandwidth = bandwidth
        self.bandwidth_scale = bandwidth_scale
        self.distance = distance
        self.thresh = thresh
        super().__init__(data, n_pca=n_pca, **kwargs)

    def get_params(self):
        params = super().get_params()
        params.update({
            "knn": self.knn, "decay": self.decay, "bandwidth": self.bandwidth,
            "bandwidth_scale": self.bandwidth_scale, "knn_max": self.knn_max, "distance": self.distance,
            "thresh": self.thresh, "n_jobs": self.n_jobs, "random_state": self.random_state, "verbose": self.verbose,
        })
        return params

    def set_params(self, **params):
        # reference: graphs.py:693-746
        if "knn" in params and params["knn"] != self.knn:
            raise ValueError("Cannot update knn. Please create a new graph")
        if "knn_max" in params and params["knn_max"] != self.knn:
            raise ValueError("Cannot update knn_max. Please create a new graph")
        if "decay" in params and params["decay"] != self.decay:
            raise ValueError("Cannot update decay. Please create a new graph")
        if "bandwidth" in params and params["bandwidth"] != self.bandwidth:
            raise ValueError("Cannot update bandwidth. Please create a new graph")
        if "bandwidth_scale" in params and params["bandwidth_scale"] != self.bandwidth_scale:
            raise ValueError("Cannot update bandwidth_scale. Please create a new graph")
        if "distance" in params and params["distance"] != self.distance:
            raise ValueError("Cannot update distance. Please create a new graph")
        if "thresh" in params and params["thresh"] != self.thresh and self.decay != 0:
            raise ValueError("Cannot update thresh. Please create a new graph")
        super().set_params(**params)
        return self

    # ---- device -------------------------------------------------------------------------------
    def _bind_points(self):
        if getattr(self, "_points_bound", False):
            return
        X = np.ascontiguousarray(self.data_nu)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        # non-finite input: gt_set_points reports it with sklearn's messages (ValueError, as NearestNeighbors.fit in
        # the reference, graphs.py:763-768)
        self.hip.set_option("metric", self.distance)
        self.hip.set_points(X)
        self._points_bound = True

    @property
    def knn_tree(self):
        try:
            return self._knn_tree
        except AttributeError:
            self._knn_tree = _KnnTree(self)
            return self._knn_tree

    def _check_duplicates(self):
        # reference: graphs.py:787-817; only reached when the device flagged a zero distance
        search_knn = min((self.knn + 1) * self.search_multiplier, self.data_nu.shape[0])
        if self.knn_max:
            search_knn = min(search_knn, self.knn_max + 1)
        distances, indices = self.knn_tree.kneighbors(None, n_neighbors=min(search_knn, 96))
        if np.any(distances[:, 1] == 0):
            has_duplicates = distances[:, 1] == 0
            if np.sum(distances[:, 1:] == 0) < 20:
                idx = np.argwhere((distances == 0) & has_duplicates[:, None])
                duplicate_ids = np.array(
                    [[indices[i[0], i[1]], i[0]] for i in idx if indices[i[0], i[1]] < i[0]]
                )
                duplicate_ids = duplicate_ids[np.argsort(duplicate_ids[:, 0])]
                duplicate_names = ", ".join(["{} and {}".format(i[0], i[1]) for i in duplicate_ids])
                warnings.warn(
                    "Detected zero distance between samples {}. Consider removing duplicates to avoid errors in "
                    "downstream processing.".format(duplicate_names),
                    RuntimeWarning,
                )
            else:
                warnings.warn(
                    "Detected zero distance between {} pairs of samples. Consider removing duplicates to avoid "
                    "errors in downstream processing.".format(np.sum(np.sum(distances[:, 1:] == 0)) // 2),
                    RuntimeWarning,
                )

    def _params_struct(self):
        bw = self.bandwidth
        if bw is not None and not isinstance(bw, numbers.Number):
            bw = np.asarray(bw, dtype=np.float64)
            if bw.shape != (self.data_nu.shape[0],):
                raise ValueError("bandwidth must be a scalar or have one entry per sample")
        return _hip.Context.make_params(self.knn, self.decay, self.thresh, bw, self.bandwidth_scale, self.knn_max,
                                        self.kernel_symm, self.theta, self.anisotropy)

    def _device_build(self, kernel_symm, theta, anisotropy):
        self._bind_points()
        params, keep = self._params_struct()
        params.kernel_symm = _hip.SYMM[kernel_symm]
        params.theta = 1.0 if theta is None else float(theta)
        params.anisotropy = float(anisotropy)
        nnz, flags = self.hip.graph_build(params)
        del keep
        self._device_state = (kernel_symm, theta, anisotropy)
        return nnz, flags

    def _ensure_device_graph(self):
        """(Re)build on the device if another build (e.g. build_kernel()) replaced the cached state."""
        want = (self.kernel_symm, self.theta, self.anisotropy)
        if getattr(self, "_device_state", None) != want:
            self._device_build(*want)

    # stage timers of the library behind the reference's two phases (graphs.py:873-885)
    _KNN_STAGES = ("prep", "query_order", "sym_prepare", "sym_seed", "sym_bound", "sym_cold", "knn_select", "rerank", "fallback",
                   "radius")
    _AFFINITY_STAGES = ("affinity", "symmetrize", "normalize")

    def _log_phases(self):
        if self.verbose:
            ms = lambda names: sum(max(self.hip.stage_ms(s), 0.0) for s in names)   # noqa: E731
            _base.log_task(self.verbose, "KNN search", ms(self._KNN_STAGES) * 1e-3)
            _base.log_task(self.verbose, "affinities", ms(self._AFFINITY_STAGES) * 1e-3)

    def _build_kernel_sharded(self):
        """The build of ``_build_kernel`` row-sharded over the ranks of a torch.distributed job (graphtools_amd.dist: points
        all-gather, renumbering by cell, local candidate lists, triplet all-to-all, local merge), then the ranks' row blocks
        of K and P all-gathered so that every rank returns what a single-process build returns."""
        import torch

        from . import dist as gdist

        group, world, rank = self._dist_ranks()
        X = np.ascontiguousarray(self.data_nu)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        n = X.shape[0]
        device = self._dist_device()
        self.hip.set_option("metric", self.distance)
        sk = gdist.ShardedKnnGraph(self.hip, n, group=group)
        r0, r1 = int(sk.input_splits[rank]), int(sk.input_splits[rank + 1])
        sk.gather_points(torch.from_numpy(X[r0:r1]).to(device))   # (only the rank's slice crosses its PCIe link)
        params, keep = self._params_struct()
        nnz_local, flags = sk.build(params)
        del keep
        self._points_bound = False    # (the context holds the renumbered points of the sharded build)
        self._device_state = None
        self._sharded = sk
        self._log_phases()
        data, indices, indptr = self.hip.graph_fetch_csr(_hip.CSR_K)
        pdata, _, _ = self.hip.graph_fetch_csr(_hip.CSR_P, structure=False)
        rows = sk.row_ids()
        self.local_rows = rows
        self.K_local = sparse.csr_matrix((data, indices, indptr), shape=(len(rows), n))
        self.P_local = sparse.csr_matrix((pdata, indices, indptr), shape=(len(rows), n))
        deg_local = np.asarray(self.hip.graph_fetch_vec(_hip.VEC_DEGREE), dtype=np.float64)
        # the full matrices on every rank (what `G.K`, `G.P` mean in the reference); flags: OR over the ranks
        fl = torch.as_tensor(np.array([(flags >> b) & 1 for b in range(8)], dtype=np.int64), device=device)
        torch.distributed.all_reduce(fl, op=torch.distributed.ReduceOp.MAX, group=group)
        flags = int(sum(int(v) << b for b, v in enumerate(fl.cpu().tolist())))
        self._build_flags = flags
        ptr, cols, (kv, pv) = gdist.allgather_csr_blocks(indptr, indices, [data, pdata], rows, (n, n), device, group)
        deg = np.empty(n, dtype=np.float64)
        ids_t = gdist.allgather_vector(torch.as_tensor(rows, device=device), sk.splits, group).cpu().numpy()
        deg[ids_t] = gdist.allgather_vector(torch.as_tensor(deg_local, device=device), sk.splits, group).cpu().numpy()
        self._kernel_degree = deg.reshape(-1, 1)
        if int(ptr[-1]) < 2**31:
            ptr = ptr.astype(np.int32)
        K = sparse.csr_matrix((kv, cols, ptr), shape=(n, n))
        self._diff_op = sparse.csr_matrix((pv, K.indices, K.indptr), shape=(n, n))
        if flags & _hip.FLAG_DUPLICATES:
            warnings.warn("Detected zero distance between samples. Consider removing duplicates to avoid errors in "
                          "downstream processing.", RuntimeWarning)
        self._emit_build_warnings(flags, K)
        return K

    def _sharded_graph(self):
        """this rank's share of the row-sharded build (rebuilt - a collective call, every rank alike - when the graph came out
        of a pickle: the device side does not travel)"""
        self.K
        if getattr(self, "_sharded", None) is None:
            self._build_kernel_sharded()
        return self._sharded

    def _build_on_device(self):
        """Kernel and diffusion operator built and left on the device, with the warnings of the build (reference: base.py:551-554,
        graphs.py:887-914); nothing crosses PCIe.  Idempotent while the device holds this graph's kernel."""
        want = (self.kernel_symm, self.theta, self.anisotropy)
        if getattr(self, "_device_state", None) == want and hasattr(self, "_build_nnz"):
            return self._build_nnz, self._build_flags
        nnz, flags = self._device_build(*want)
        self._log_phases()
        self._build_nnz, self._build_flags = nnz, flags
        if flags & _hip.FLAG_DUPLICATES:
            self._check_duplicates()
        if self.kernel_symm is not None:
            self._emit_build_warnings(flags)
        return nnz, flags

    def _ensure_built(self):
        if self._dist_ranks() is not None or self.kernel_symm is None:
            self.K
        else:
            self._build_on_device()

    def _initialize(self):
        # the landmark operator, the spectral clustering, diffuse() and the torch hand-offs read the kernel where it is: the host
        # copy (1.4 GB over PCIe at N = 1e6: 30 ms, twice the build) is made when K or P are first asked for.  Without a
        # symmetrisation the reference's symmetry warning needs the matrix itself: fetched right away.
        self._ensure_built()

    def _build_kernel(self):
        if self._dist_ranks() is not None:
            return self._build_kernel_sharded()
        nnz, flags = self._build_on_device()
        # K and P in one pass over the link (SURVEY 8d host-complete: scipy CSR K and P out): the P values are derived on the
        # host from K and the degrees by the copy threads while K is still arriving - the division the device made for its
        # own P, same bits - so they never cross PCIe (gt_graph_fetch_kp)
        data, indices, indptr, pdata = self.hip.graph_fetch_kp()
        n = self.data_nu.shape[0]
        if nnz < 2**31:
            indptr = indptr.astype(np.int32)
        K = sparse.csr_matrix((data, indices, indptr), shape=(n, n))
        self._diff_op = sparse.csr_matrix((pdata, K.indices, K.indptr), shape=(n, n))
        if self.kernel_symm is None:
            self._emit_build_warnings(flags, K)
        return K

    def build_kernel(self):
        """The unsymmetrised kernel K0 (reference: graphs.py:771-785), as a scipy CSR matrix."""
        self._no_sharded("build_kernel() (the unsymmetrised kernel)")
        nnz, flags = self._device_build(None, None, 0)
        data, indices, indptr = self.hip.graph_fetch_csr(_hip.CSR_K)
        n = self.data_nu.shape[0]
        if nnz < 2**31:
            indptr = indptr.astype(np.int32)
        return sparse.csr_matrix((data, indices, indptr), shape=(n, n))

    def _fetch_diff_op(self):
        self._ensure_device_graph()
        data, _, _ = self.hip.graph_fetch_csr(_hip.CSR_P, structure=False)
        K = self._kernel
        return sparse.csr_matrix((data, K.indices, K.indptr), shape=K.shape)

    def _fetch_degree(self):
        self._ensure_device_graph()
        return self.hip.graph_fetch_vec(_hip.VEC_DEGREE)

    def _fetch_diff_aff(self):
        self.K
        if self._dist_ranks() is not None:
            # (a sharded graph's device holds the rank's rows only: base.py:683-698 on the gathered host copy)
            d = np.asarray(self.kernel_degree, dtype=np.float64).ravel()
            dm = sparse.csr_matrix((1 / np.sqrt(d), np.arange(len(d)), np.arange(len(d) + 1)))
            return dm @ self._kernel @ dm
        self._ensure_device_graph()
        vals = self.hip.graph_diff_aff()
        K = self._kernel
        return sparse.csr_matrix((vals, K.indices, K.indptr), shape=K.shape)

    def diff_op_torch(self):
        """The diffusion operator as a CUDA ``torch.sparse_csr_tensor`` (no host round trip): the hand-off to
        consumers that continue on the device, e.g. ``P @ X`` diffusion steps."""
        self._no_sharded("diff_op_torch()")
        self._ensure_built()
        self._ensure_device_graph()
        return self.hip.graph_csr_torch(_hip.CSR_P)

    def diffuse(self, X, t=1):
        """``P^t X`` with the diffusion operator left on the device (``gt_graph_spmm``; the reference's consumers do
        ``graph.diff_op.dot(X)`` on the host CSR).  X: [n_samples] or [n_samples, c]."""
        self._no_sharded("diffuse()")
        self._ensure_built()
        self._ensure_device_graph()
        X = np.asarray(X)
        if X.shape[0] != self.data_nu.shape[0]:
            raise ValueError("X must have one row per sample ({}), got {}".format(self.data_nu.shape[0], X.shape[0]))
        for _ in range(int(t)):
            X = self.hip.graph_spmm(_hip.CSR_P, X)
        return X

    def kernel_torch(self):
        """The kernel matrix K as a CUDA ``torch.sparse_csr_tensor``."""
        self._no_sharded("kernel_torch()")
        self._ensure_built()
        self._ensure_device_graph()
        return self.hip.graph_csr_torch(_hip.CSR_K)

    @property
    def build_stats(self):
        """Device-side statistics of the last build (fallback / radius rows, stage timings in ms)."""
        self._ensure_built()
        st = self.hip.graph_stats()
        st["stage_ms"] = {s: self.hip.stage_ms(s) for s in
                          ("knn_select", "rerank", "fallback", "radius", "affinity", "symmetrize", "normalize")}
        return st

    def _extend_on_device(self, Y, knn=None, knn_max=None, bandwidth=None, bandwidth_scale=None):
        self._no_sharded("the out-of-sample extension")
        if knn is None:
            knn = self.knn
        if bandwidth is None:
            bandwidth = self.bandwidth
        if bandwidth_scale is None:
            bandwidth_scale = self.bandwidth_scale
        if knn > self.data.shape[0]:
            warnings.warn(
                "Cannot set knn ({k}) to be greater than "
                "n_samples ({n}). Setting knn={n}".format(k=knn, n=self.data_nu.shape[0])
            )
            knn = self.data_nu.shape[0]
        Y = self._check_extension_shape(Y)
        if bandwidth is not None and not isinstance(bandwidth, numbers.Number):
            # one bandwidth per row of Y (graphs.py:893-897 broadcasts `bandwidth * bandwidth_scale` over the rows)
            bandwidth = np.asarray(bandwidth, dtype=np.float64).ravel()
            if bandwidth.shape[0] not in (1, np.asarray(Y).shape[0]):
                raise ValueError("bandwidth must be a scalar or have one entry per row of Y")
        self._bind_points()
        params, keep = _hip.Context.make_params(knn, self.decay, self.thresh, bandwidth, bandwidth_scale, knn_max,
                                                None, None, 0)
        nnz, flags = self.hip.graph_extend(np.asarray(Y), params)
        del keep
        self._device_state = None   # the device now holds the rectangular kernel, not this graph's K
        return nnz

    def _fetch_rect(self, which, m, nnz):
        data, indices, indptr = self.hip.graph_fetch_csr(which)
        if nnz < 2**31:
            indptr = indptr.astype(np.int32)
        return sparse.csr_matrix((data, indices, indptr), shape=(m, self.data_nu.shape[0]))

    def build_kernel_to_data(self, Y, knn=None, knn_max=None, bandwidth=None, bandwidth_scale=None):
        """Kernel from new points ``Y`` to the graph's data (reference: graphs.py:819-982)."""
        nnz = self._extend_on_device(Y, knn, knn_max, bandwidth, bandwidth_scale)
        return self._fetch_rect(_hip.CSR_K, np.asarray(Y).shape[0], nnz)

    def extend_to_data(self, Y):
        """Row-stochastic transitions from new points to the graph's data (reference: base.py:1166-1193)."""
        nnz = self._extend_on_device(Y)
        return self._fetch_rect(_hip.CSR_P, np.asarray(Y).shape[0], nnz)

    def interpolate(self, transform, transitions=None, Y=None):
        """reference: base.py:1195-1229"""
        if transitions is None:
            if Y is None:
                raise ValueError("Either `transitions` or `Y` must be provided.")
            transitions = self.extend_to_data(Y)
        return transitions.dot(transform)


class LandmarkGraph(DataGraph):
    """Landmark graph mixin (reference: graphtools/graphs.py:985-1317).

    The cluster assignment for ``random_landmarking=True`` and the operator algebra
    (``pmn``/``pnm`` aggregation, both normalisations, the L x L product) run on the device.  The spectral
    front end of the default mode (randomized SVD of ``diff_aff`` + MiniBatchKMeans, graphs.py:1215-1230)
    is RNG/iteration-order dependent pre-processing and stays on host scikit-learn exactly as in the
    reference; its labels are an input to the device step.
    """

    def __init__(self, data, n_landmark=2000, n_svd=100, random_landmarking=False, **kwargs):
        if n_landmark >= data.shape[0]:
            raise ValueError(
                "n_landmark ({}) >= n_samples ({}). Use kNNGraph instead".format(n_landmark, data.shape[0])
            )
        if (n_svd >= data.shape[0]) and (not random_landmarking):
            warnings.warn(
                "n_svd ({}) >= n_samples ({}) Consider using kNNGraph or lower n_svd".format(n_svd, data.shape[0]),
                RuntimeWarning,
            )
        self.random_landmarking = random_landmarking
        self.n_landmark = n_landmark
        self.n_svd = n_svd
        super().__init__(data, **kwargs)

    def get_params(self):
        params = super().get_params()
        params.update({"n_landmark": self.n_landmark, "n_pca": self.n_pca,
                       "random_landmarking": self.random_landmarking})
        return params

    def set_params(self, **params):
        reset_landmarks = False
        if "n_landmark" in params and params["n_landmark"] != self.n_landmark:
            self.n_landmark = params["n_landmark"]
            reset_landmarks = True
        if "n_svd" in params and params["n_svd"] != self.n_svd:
            self.n_svd = params["n_svd"]
            reset_landmarks = True
        if "random_landmarking" in params and params["random_landmarking"] != self.random_landmarking:
            self.random_landmarking = params["random_landmarking"]
            reset_landmarks = True
        super().set_params(**params)
        if reset_landmarks:
            self._reset_landmarks()
        return self

    def _reset_landmarks(self):
        for attr in ("_landmark_op", "_transitions", "_clusters"):
            if hasattr(self, attr):
                delattr(self, attr)

    @property
    def landmark_op(self):
        try:
            return self._landmark_op
        except AttributeError:
            self.build_landmark_op()
            return self._landmark_op

    @property
    def clusters(self):
        try:
            return self._clusters
        except AttributeError:
            self.build_landmark_op()
            return self._clusters

    @property
    def transitions(self):
        try:
            return self._transitions
        except AttributeError:
            self.build_landmark_op()
            return self._transitions

    def extend_to_data(self, data, **kwargs):
        """Transitions from new points to the LANDMARKS (reference: graphs.py:1247-1289): the kernel to the data
        (``build_kernel_to_data`` - the device path for kNN graphs), its columns summed per cluster in the order of
        ``np.unique(clusters)``, rows l1-normalised.  Returns [n_samples_y, n_landmark]."""
        kernel = self.build_kernel_to_data(data, **kwargs)
        landmarks, inverse = np.unique(np.asarray(self.clusters), return_inverse=True)
        n = inverse.shape[0]
        if sparse.issparse(kernel):
            # one-hot cluster membership [n, L]; csr @ csr adds the entries of a row in ascending column order, which is
            # the order the reference's boolean column mask + row sum sees them in
            S = sparse.csr_matrix((np.ones(n), (np.arange(n), inverse)), shape=(n, len(landmarks)))
            pnm = sparse.csr_matrix(sparse.csr_matrix(kernel).dot(S))
            pnm.eliminate_zeros()
            pnm.sort_indices()
        else:
            kernel = np.asarray(kernel)
            pnm = np.array([np.sum(kernel[:, inverse == i], axis=1).T for i in range(len(landmarks))]).transpose()
        return self._l1_rows(pnm)

    _l1_rows = staticmethod(_l1_rows)

    def interpolate(self, transform, transitions=None, Y=None):
        """reference: graphs.py:1291-1317 - without ``transitions`` and ``Y`` the landmark transitions of the graph's own
        data are used (a landmark embedding is mapped back to every sample)."""
        if transitions is None and Y is None:
            transitions = self.transitions
        if transitions is None:
            transitions = self.extend_to_data(Y)
        return transitions.dot(transform)

    def _assign_clusters(self):
        n_samples = self.data.shape[0]
        if self.random_landmarking:
            # reference: graphs.py:1200-1213
            if self.distance != "euclidean":
                raise NotImplementedError("graphtools_amd: random landmarking supports the euclidean metric only")
            if self._dist_ranks() is not None:
                # every rank assigns its own rows, one all-gather of the labels (graphtools_amd.dist)
                return np.asarray(self._sharded_graph().random_landmark_clusters(self.n_landmark, self.random_state), dtype=np.int64)
            rng = np.random.default_rng(self.random_state)
            landmark_indices = rng.choice(n_samples, self.n_landmark, replace=False)
            if n_samples > 5000:
                # sklearn euclidean_distances arithmetic: float64 GEMM form rounded to the input dtype, then sqrt -
                # the same rounding the device kNN search emits.  Nearest landmark = 1-NN against the L landmark
                # rows on the MFMA path; argmin's first-index rule is applied among the (few) nearest that tie
                # after the float32 rounding.
                X = np.ascontiguousarray(self.data_nu)
                if X.dtype not in (np.float32, np.float64):
                    X = X.astype(np.float64)
                lm_ctx = _hip.Context(getattr(self, "device", 0) or 0)
                try:
                    lm_ctx.set_points(X[landmark_indices])
                    k = int(min(4, self.n_landmark))
                    if getattr(self, "_points_bound", False) and self.hip.n == X.shape[0] and self.hip.dtype == X.dtype:
                        # (the graph's own context holds these rows on the device already: queried where they are)
                        self.hip.sync()
                        return lm_ctx.knn_first_nearest(k, y_dev_ptr=self.hip.points_device(0), m=X.shape[0])
                    return lm_ctx.knn_first_nearest(k, Y=X)
                finally:
                    lm_ctx.close()
            self._bind_points()
            return self.hip.nearest_landmark(landmark_indices, 0).astype(np.int64)
        # spectral front end (graphs.py:1215-1230).  Sparse kNN kernels that live on the device: randomized SVD of
        # diff_aff, the embedding product and the labelling pass on the device (graphtools_amd/_spectral.py; statistical
        # parity - the k-means in the middle is scikit-learn's own and RNG-dependent); otherwise host scikit-learn as in
        # the reference.
        self._no_sharded("spectral landmarking (use random_landmarking=True)")
        from . import base as _base
        if (_base.SPECTRAL_BACKEND != "sklearn" and isinstance(self, kNNGraph) and self.n_svd + 10 <= 128
                and (_base.SPECTRAL_BACKEND == "device" or n_samples >= _base._SPECTRAL_DEVICE_MIN_ROWS)):
            from ._spectral import spectral_clusters

            self._ensure_built()
            self._ensure_device_graph()
            return spectral_clusters(self, self.hip)
        from sklearn.cluster import MiniBatchKMeans
        from sklearn.utils.extmath import randomized_svd

        _, _, VT = randomized_svd(self.diff_aff, n_components=self.n_svd, random_state=self.random_state)
        kmeans = MiniBatchKMeans(self.n_landmark, init_size=3 * self.n_landmark, n_init=1, batch_size=10000,
                                 random_state=self.random_state)
        return kmeans.fit_predict(self.diff_op.dot(VT.T))

    def build_landmark_op(self):
        """Landmark operator and sample-to-landmark transitions (reference: graphs.py:1187-1246)."""
        self._ensure_built()
        dense = hasattr(self, "_kernel") and not sparse.issparse(self._kernel)   # (a kNN kernel may exist on the device only)
        if not hasattr(self, "_clusters"):
            self._clusters = self._assign_clusters()
        cl = np.asarray(self._clusters)
        if (cl.ndim == 1 and cl.dtype.kind in "iu" and cl.size and int(cl.min()) == 0 and int(cl.max()) < 4 * self.n_landmark
                and np.all(np.bincount(cl, minlength=int(cl.max()) + 1) > 0)):
            # labels 0 .. L-1, every one in use (random landmarking: a landmark is its own nearest): np.unique's answer without its sort
            landmarks, inverse = np.arange(int(cl.max()) + 1), cl
        else:
            landmarks, inverse = np.unique(cl, return_inverse=True)
        L = len(landmarks)
        if self._dist_ranks() is not None:
            # partial L x L products of the rank's rows, ONE all-reduce (dist.landmark_operator); the rank's rows of the
            # transitions all-gathered like K's
            from . import dist as gdist

            group, world, rank = self._dist_ranks()
            sk = self._sharded_graph()
            self._landmark_op, tnnz = sk.landmark_operator(inverse.astype(np.int32), L)
            data, indices, indptr = self.hip.landmark_fetch_transitions(tnnz)
            n = self.data.shape[0]
            ptr, cols, (tv,) = gdist.allgather_csr_blocks(indptr, indices, [data], sk.row_ids(), (n, L), self._dist_device(), group)
            self._transitions = sparse.csr_matrix((tv, cols, ptr.astype(np.int32)), shape=(n, L))
            return
        if dense:
            # an exact (dense) kernel: its non-zeros are handed to the device as they are (no symmetrisation, no anisotropy -
            # K is final), the landmark products then run as for a kNN kernel; the reference returns dense transitions here
            # (K is symmetric already: the "+" rule returns (a + a) / 2 = a exactly and tells the library so)
            Ks = sparse.csr_matrix(np.asarray(self._kernel, dtype=np.float64))
            self.hip.csr_graph_build(Ks, "+" if self.kernel_symm is not None else None, None, 0, assume_unique=True)
            self._device_state = None   # (the context now holds this copy, not what _ensure_device_graph expects)
        else:
            self._ensure_device_graph()
        M, R, tnnz = self.hip.landmark_build(inverse.astype(np.int32), L)
        self._landmark_op = self.hip.landmark_scale(M, R)
        data, indices, indptr = self.hip.landmark_fetch_transitions(tnnz)
        self._transitions = sparse.csr_matrix((data, indices, indptr), shape=(self.data.shape[0], L))
        if dense:
            self._transitions = self._transitions.toarray()


class TraditionalGraph(DataGraph):
    """Exact dense alpha-decay graph (reference: graphtools/graphs.py:1320-1704)."""

    def __init__(self, data, knn=5, decay=40, bandwidth=None, bandwidth_scale=1.0, distance="euclidean", n_pca=None,
                 thresh=1e-4, precomputed=None, **kwargs):
        # reference: graphs.py:1397-1436
        if decay is None and precomputed not in ["affinity", "adjacency"]:
            raise ValueError("`decay` must be provided for a TraditionalGraph. For kNN kernel, use kNNGraph.")
        if precomputed is not None and n_pca not in [None, 0, False]:
            n_pca = None
            warnings.warn("n_pca cannot be given on a precomputed graph. Setting n_pca=None", RuntimeWarning)
        if knn is None and bandwidth is None:
            raise ValueError("Either `knn` or `bandwidth` must be provided.")
        if knn is not None and knn > data.shape[0] - 2:
            warnings.warn(
                "Cannot set knn ({k}) to be greater than "
                " n_samples - 2 ({n}). Setting knn={n}".format(k=knn, n=data.shape[0] - 2)
            )
            knn = data.shape[0] - 2
        if precomputed is not None:
            if precomputed not in ["distance", "affinity", "adjacency"]:
                raise ValueError(
                    "Precomputed value {} not recognized. "
                    "Choose from ['distance', 'affinity', 'adjacency']".format(precomputed)
                )
            elif data.shape[0] != data.shape[1]:
                raise ValueError("Precomputed {} must be a square matrix. {} was given".format(precomputed, data.shape))
            elif (data < 0).sum() > 0:
                raise ValueError("Precomputed {} should be non-negative".format(precomputed))
        self.knn = knn
        self.decay = decay
        self.bandwidth = bandwidth
        self.bandwidth_scale = bandwidth_scale
        self.distance = distance
        self.thresh = thresh
        self.precomputed = precomputed
        super().__init__(data, n_pca=n_pca, **kwargs)

    def get_params(self):
        params = super().get_params()
        params.update({
            "knn": self.knn, "decay": self.decay, "bandwidth": self.bandwidth,
            "bandwidth_scale": self.bandwidth_scale, "distance": self.distance, "precomputed": self.precomputed,
        })
        return params

    def set_params(self, **params):
        # reference: graphs.py:1462-1512
        if "precomputed" in params and params["precomputed"] != self.precomputed:
            raise ValueError("Cannot update precomputed. Please create a new graph")
        if "distance" in params and params["distance"] != self.distance and self.precomputed is None:
            raise ValueError("Cannot update distance. Please create a new graph")
        if "knn" in params and params["knn"] != self.knn and self.precomputed is None:
            raise ValueError("Cannot update knn. Please create a new graph")
        if "decay" in params and params["decay"] != self.decay and self.precomputed is None:
            raise ValueError("Cannot update decay. Please create a new graph")
        if "bandwidth" in params and params["bandwidth"] != self.bandwidth and self.precomputed is None:
            raise ValueError("Cannot update bandwidth. Please create a new graph")
        if "bandwidth_scale" in params and params["bandwidth_scale"] != self.bandwidth_scale:
            raise ValueError("Cannot update bandwidth_scale. Please create a new graph")
        super().set_params(**params)
        return self

    def _bind_points(self):
        """the points on the device (random landmark assignment, graphs.py:1200-1213)"""
        if self.precomputed is not None:
            raise ValueError("random landmarking needs the points: the graph was built from a precomputed matrix")
        X = self.data_nu.toarray() if sparse.issparse(self.data_nu) else np.ascontiguousarray(self.data_nu)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        self.hip.set_option("metric", "euclidean")
        self.hip.set_points(X)

    # smallest point set sent through the neighbour search: below it the all-pairs tile kernel is as fast
    _NEIGHBOUR_ROUTE_MIN = 4096

    def _build_kernel_through_neighbours(self, data, bandwidth):
        """The exact graph from points without forming all pairs (reference: graphs.py:1546-1609).

        ``K[K < thresh] = 0`` makes every entry beyond ``bandwidth_i * (-log thresh)^(1/decay)`` of row i exactly zero, so
        the kept entries are a radius search: the kNN path's build (candidate distances on the matrix cores, float64
        refinement inside the radius only) with the same ``knn`` (the bandwidth is the distance to the knn-th OTHER point in
        both: graphs.py:1584-1588 here, graphs.py:776 ``knn + 1`` with the point itself there), then written out densely
        on the device.  Distances are float64 (pdist converts to double); the points are centred first, which pdist's
        differences do not see and which keeps the expanded form |x|^2 - 2 x.y + |y|^2 of the search well conditioned.
        Returns None when this route does not apply (small sets, thresh = 0, duplicate points - their warnings and the
        0 / 0 bandwidth rule live in the all-pairs path)."""
        n = data.shape[0]
        # (thresh below float64's eps: the kNN build clamps it to eps - graphs.py:628-629 is a kNNGraph rule, the reference's
        #  TraditionalGraph keeps entries in [thresh, eps) - so such a threshold belongs to the all-pairs path)
        if (n < self._NEIGHBOUR_ROUTE_MIN or not self.thresh or self.thresh < np.finfo(float).eps or self.decay is None
                or (self.knn is not None and self.knn + 1 > _hip.MAX_KNN)):   # (+ 1: the point itself)
            return None
        bw = bandwidth
        if bw is not None and not isinstance(bw, numbers.Number):
            bw = np.asarray(bw, dtype=np.float64)
            if bw.shape != (n,):
                return None
        X = np.asarray(data, dtype=np.float64)
        X = np.ascontiguousarray(X - X.mean(axis=0, keepdims=True))
        try:
            # anything this route cannot hold (more than 2048 features, a radius list or the n x n device copy beyond the
            # memory at hand) is the all-pairs path's to build, as it was before this route existed
            self.hip.set_option("metric", "euclidean")
            self.hip.set_points(X)
            params, keep = _hip.Context.make_params(self.knn if self.knn is not None else 1, self.decay, self.thresh, bw,
                                                    self.bandwidth_scale, None, self.kernel_symm, self.theta, self.anisotropy)
            nnz, flags = self.hip.graph_build(params)
            del keep
            if flags & _hip.FLAG_DUPLICATES:
                return None
            K = self.hip.graph_to_dense(_hip.CSR_K, n)
            self._diff_op = self.hip.graph_to_dense(_hip.CSR_P, n)
            self._kernel_degree = self.hip.graph_fetch_vec(_hip.VEC_DEGREE).reshape(-1, 1)
        except _hip.HipError:
            self._diff_op = None
            if hasattr(self, "_kernel_degree"):
                del self._kernel_degree
            return None
        self._emit_build_warnings(flags, K)
        return K

    def _build_kernel(self):
        self._no_sharded("an exact (TraditionalGraph) build")
        data = self.data_nu
        if self.precomputed in ("affinity", "adjacency"):
            if sparse.issparse(self.data):
                data = self.data   # (a sparse kernel stays sparse, as in the reference; data_nu is the dense copy)
            # reference: graphs.py:1532-1545, 1596-1609 - the caller's matrix IS the unsymmetrised kernel (adjacency: with
            # the diagonal set to 1), entries below thresh are zeroed; then the common tail on the device
            if sparse.issparse(data):
                K0 = sparse.csr_matrix(data, dtype=np.float64, copy=True)
                if self.precomputed == "adjacency":
                    K0 = K0.tolil()
                    K0.setdiag(1)
                    K0 = K0.tocsr()
                K0.data[K0.data < self.thresh] = 0
                K0.eliminate_zeros()
                K0.sum_duplicates()
                self._sparse_k0 = K0
                nnz, flags = self.hip.csr_graph_build(K0, self.kernel_symm, self.theta, self.anisotropy, assume_unique=True)
                self._device_state = (self.kernel_symm, self.theta, self.anisotropy)
                kd, ki, kp = self.hip.graph_fetch_csr(_hip.CSR_K)
                if nnz < 2**31:
                    kp = kp.astype(np.int32)
                K = sparse.csr_matrix((kd, ki, kp), shape=K0.shape)
                pd_, _, _ = self.hip.graph_fetch_csr(_hip.CSR_P, structure=False)
                self._diff_op = sparse.csr_matrix((pd_, K.indices, K.indptr), shape=K.shape)
                self._kernel_degree = self.hip.graph_fetch_vec(_hip.VEC_DEGREE).reshape(-1, 1)
                self._emit_build_warnings(flags, K)
                return K
            data = np.asarray(data)
            K, P, flags = self.hip.dense_graph_build(
                data, self.precomputed, None, None, self.thresh, None, 1.0, self.kernel_symm, self.theta,
                self.anisotropy, want_P=True)
            self._diff_op = P
            self._kernel_degree = self.hip.dense_fetch_vec(_hip.VEC_DEGREE, K.shape[0]).reshape(-1, 1).astype(K.dtype)
            self._emit_build_warnings(flags, K)
            return K
        if sparse.issparse(data):
            data = data.toarray()
        data = np.asarray(data)
        bandwidth = self.bandwidth
        host_pdx = None
        if self.precomputed is None and (self.distance != "euclidean" or callable(bandwidth)):
            # metrics other than the north star's euclidean, and bandwidth callables (which see the whole distance matrix,
            # graphs.py:1588-1589): the distances are formed on the host with scipy exactly as the reference does
            # (graphs.py:1552-1576), everything behind them runs on the device
            from scipy.spatial.distance import pdist, squareform

            host_pdx = squareform(pdist(data, metric=self.distance))
            dup = np.argwhere(np.triu(host_pdx == 0, k=1))
            if len(dup) > 0:
                self._warn_duplicate_pairs([(int(i), int(j)) for i, j in dup] if len(dup) < 20 else None, len(dup))
        if callable(bandwidth):
            bandwidth = np.asarray(bandwidth(host_pdx if host_pdx is not None else data), dtype=np.float64)
        if host_pdx is None and self.precomputed is None:
            K = self._build_kernel_through_neighbours(data, bandwidth)
            if K is not None:
                return K
        K, P, flags = self.hip.dense_graph_build(
            host_pdx if host_pdx is not None else data,
            "distance" if (self.precomputed == "distance" or host_pdx is not None) else None, self.knn, self.decay,
            self.thresh, bandwidth, self.bandwidth_scale, self.kernel_symm, self.theta, self.anisotropy, want_P=True)
        self._diff_op = P
        self._kernel_degree = self.hip.dense_fetch_vec(_hip.VEC_DEGREE, K.shape[0]).reshape(-1, 1).astype(K.dtype)
        if flags & _hip.FLAG_DUPLICATES and self.precomputed is None and host_pdx is None:
            # reference: graphs.py:1553-1574 - the pairs (i < j) at pdist distance 0, i.e. identical rows, named when
            # there are fewer than 20 of them (counted first: a group of g identical rows holds g (g - 1) / 2 pairs)
            _, inverse, counts = np.unique(data, axis=0, return_inverse=True, return_counts=True)
            inverse = np.asarray(inverse).ravel()
            n_pairs = int(np.sum(counts.astype(np.int64) * (counts.astype(np.int64) - 1) // 2))
            if 0 < n_pairs < 20:
                order = np.argsort(inverse, kind="stable")
                groups = np.split(order, np.flatnonzero(np.diff(inverse[order])) + 1)
                pairs = sorted((int(g[a]), int(g[b])) for g in groups if len(g) > 1
                               for a in range(len(g)) for b in range(a + 1, len(g)))
                self._warn_duplicate_pairs(pairs, n_pairs)
            elif n_pairs >= 20:
                self._warn_duplicate_pairs(None, n_pairs)
            # (n_pairs == 0 cannot happen: a float64 difference-form distance is 0 only between identical rows)
        self._emit_build_warnings(flags, K)
        return K

    @staticmethod
    def _warn_duplicate_pairs(pairs, n_pairs):
        if pairs is not None and n_pairs < 20:
            warnings.warn(
                "Detected zero distance between samples {}. Consider removing duplicates to avoid errors in "
                "downstream processing.".format(", ".join("{} and {}".format(i, j) for i, j in pairs)),
                RuntimeWarning)
        else:
            warnings.warn(
                "Detected zero distance between {} pairs of samples. Consider removing duplicates to avoid errors "
                "in downstream processing.".format(n_pairs), RuntimeWarning)

    def _fetch_diff_op(self):
        return self._diff_op

    def _fetch_degree(self):
        return self._kernel_degree

    def build_kernel_to_data(self, Y, knn=None, bandwidth=None, bandwidth_scale=None):
        """Dense kernel from new points ``Y`` to the graph's data (reference: graphs.py:1612-1678) on the device:
        ``cdist`` in float64 difference form, bandwidth = the knn-th smallest distance of each row (or the caller's),
        alpha-decay affinities, entries below ``thresh`` zeroed.  Returns a float64 array [n_samples_y, n_samples]."""
        if knn is None:
            knn = self.knn
        if bandwidth is None:
            bandwidth = self.bandwidth
        if bandwidth_scale is None:
            bandwidth_scale = self.bandwidth_scale
        if self.precomputed is not None:
            raise ValueError("Cannot extend kernel on precomputed graph")
        Y = np.asarray(self._check_extension_shape(Y))
        X = np.ascontiguousarray(self.data_nu.toarray() if sparse.issparse(self.data_nu) else self.data_nu)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        if self.distance != "euclidean":
            raise NotImplementedError(
                "graphtools_amd.TraditionalGraph.build_kernel_to_data: distance='{}' is not on the HIP path "
                "(euclidean only)".format(self.distance))
        if callable(bandwidth):
            # the callable sees the whole distance matrix (graphs.py:1657-1658): formed on the host like the reference's
            from scipy.spatial.distance import cdist

            bandwidth = np.asarray(bandwidth(cdist(Y, X, metric=self.distance)), dtype=np.float64)
        # scipy's cdist(Y, X) promotes both sides to float64 and keeps Y's precision (graphs.py:1653): a float64 Y on a float32
        # graph (e.g. the output of data_pca.transform) must not be rounded to the points' dtype first
        Ya = np.asarray(Y)
        if Ya.dtype == np.float64 and X.dtype == np.float32:
            X = X.astype(np.float64)
        self.hip.set_points(X)
        return self.hip.dense_extend(Y, knn, self.decay, self.thresh, bandwidth, bandwidth_scale)


class MNNGraph(DataGraph):
    """Mutual nearest neighbours graph for batch correction (reference: graphs.py:1707-1966).

    A composition of the accelerated path: every batch gets a ``kNNGraph`` (kernel_symm='+') on the device, every
    ordered pair of batches a ``build_kernel_to_data`` block, each block row is scaled by
    ``min(1, within / between) * beta`` and the assembled kernel goes back to the device for symmetrisation,
    anisotropy and the diffusion operator (``gt_csr_graph_build``).
    """

    def __init__(self, data, sample_idx, knn=5, beta=1, n_pca=None, decay=None, adaptive_k=None, bandwidth=None,
                 distance="euclidean", thresh=1e-4, n_jobs=1, **kwargs):
        # reference: graphs.py:1743-1790
        self.beta = beta
        self.sample_idx = sample_idx
        self.samples, self.n_cells = np.unique(self.sample_idx, return_counts=True)
        self.knn = knn
        self.decay = decay
        self.distance = distance
        self.bandwidth = bandwidth
        self.thresh = thresh
        self.n_jobs = n_jobs
        if sample_idx is None:
            raise ValueError("sample_idx must be given. For a graph without batch correction, use kNNGraph.")
        elif len(sample_idx) != data.shape[0]:
            raise ValueError(
                "sample_idx ({}) must be the same length as data ({})".format(len(sample_idx), data.shape[0])
            )
        elif len(self.samples) == 1:
            raise ValueError("sample_idx must contain more than one unique value")
        if adaptive_k is not None:
            warnings.warn("`adaptive_k` has been deprecated. Using fixed knn.", DeprecationWarning)
        if decay is not None and thresh <= 0:
            raise NotImplementedError(
                "graphtools_amd.MNNGraph: thresh=0 with a decaying kernel makes every block an exact dense graph; "
                "only the sparse (kNN) composition is on the HIP path"
            )
        super().__init__(data, n_pca=n_pca, n_jobs=n_jobs, **kwargs)

    def _check_symmetrization(self, kernel_symm, theta):
        # reference: graphs.py:1792-1803
        if (kernel_symm == "theta" or kernel_symm == "mnn") and theta is not None and \
                not isinstance(theta, numbers.Number):
            raise TypeError("Expected `theta` as a float. Got {}.".format(type(theta)))
        else:
            super()._check_symmetrization(kernel_symm, theta)

    def get_params(self):
        params = super().get_params()
        params.update({"beta": self.beta, "knn": self.knn, "decay": self.decay, "bandwidth": self.bandwidth,
                       "distance": self.distance, "thresh": self.thresh, "n_jobs": self.n_jobs})
        return params

    def set_params(self, **params):
        # reference: graphs.py:1822-1868
        if "beta" in params and params["beta"] != self.beta:
            raise ValueError("Cannot update beta. Please create a new graph")
        for arg in ["knn", "decay", "distance", "thresh", "bandwidth"]:
            if arg in params and params[arg] != getattr(self, arg):
                raise ValueError("Cannot update {}. Please create a new graph".format(arg))
        for arg in ["n_jobs", "random_state", "verbose"]:
            if arg in params:
                self.__setattr__(arg, params[arg])
                for g in getattr(self, "subgraphs", []):
                    g.set_params(**{arg: params[arg]})
        super().set_params(**params)
        return self

    def _assemble_kernel0(self):
        """The unsymmetrised MNN kernel (reference: graphs.py:1870-1946) as CSR arrays with UNSORTED columns inside a
        row: the blocks (one device kNN kernel per batch, one device ``build_kernel_to_data`` block per ordered batch
        pair, rows scaled by ``min(1, within / between) * beta``) are written straight into their rows - O(nnz)
        copies, no COO round trip and no host sort (the device merge sorts every row anyway)."""
        n = self.data_nu.shape[0]
        masks = [np.asarray(self.sample_idx) == s for s in self.samples]
        index = [np.nonzero(m)[0] for m in masks]
        self.subgraphs = []
        for m in masks:
            self.subgraphs.append(kNNGraph(
                self.data_nu[m], n_pca=None, knn=self.knn, decay=self.decay, bandwidth=self.bandwidth,
                distance=self.distance, thresh=self.thresh, verbose=self.verbose, random_state=self.random_state,
                n_jobs=self.n_jobs, kernel_symm="+", initialize=True, device=self.device,
            ))
            self.subgraphs[-1].kernel_degree   # cache the row sums before the context is reused for cross kernels
        blocks = []   # (batch of the rows, batch of the columns, CSR block in batch-local ids, per-row scale or None)
        for i, X in enumerate(self.subgraphs):
            blocks.append((i, i, sparse.csr_matrix(X.K), None))
            within_batch_norm = np.asarray(X.kernel_degree).flatten()
            for j, Y in enumerate(self.subgraphs):
                if i == j:
                    continue
                Kij = sparse.csr_matrix(Y.build_kernel_to_data(X.data_nu, knn=self.knn))
                between_batch_norm = np.array(np.sum(Kij, 1)).flatten()
                scale = np.minimum(1, within_batch_norm / between_batch_norm) * self.beta
                blocks.append((i, j, Kij, scale))
        row_len = np.zeros(n, dtype=np.int64)
        for i, _, M, _ in blocks:
            row_len[index[i]] += np.diff(M.indptr)
        indptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(row_len, out=indptr[1:])
        nnz = int(indptr[-1])
        data = np.empty(nnz, dtype=np.float64)
        indices = np.empty(nnz, dtype=np.int32)
        cursor = indptr[:-1].copy()
        for i, j, M, scale in blocks:
            # entry e of local row r goes to cursor[row] + (e - M.indptr[r]); threaded copy in the library
            _hip.host_place_block(M, index[i], index[j], scale, cursor, indices, data)
        if nnz < 2**31:
            indptr = indptr.astype(np.int32)
        K = sparse.csr_matrix((data, indices, indptr), shape=(n, n))
        K.has_sorted_indices = False
        return K

    def build_kernel(self):
        """The unsymmetrised MNN kernel (reference: graphs.py:1870-1946) as a canonical scipy CSR matrix."""
        K = self._assemble_kernel0().copy()
        K.sort_indices()
        return K

    def _device_build_from_k0(self):
        nnz, flags = self.hip.csr_graph_build(self._kernel0, self.kernel_symm, self.theta, self.anisotropy,
                                              assume_unique=True)
        self._device_state = (self.kernel_symm, self.theta, self.anisotropy)
        return nnz, flags

    def _ensure_device_graph(self):
        """(Re)build K and P on the device from the assembled kernel if the context lost or changed them."""
        if getattr(self, "_device_state", None) != (self.kernel_symm, self.theta, self.anisotropy):
            if not hasattr(self, "_kernel0"):
                self._kernel0 = self._assemble_kernel0()
            self._device_build_from_k0()

    def _bind_points(self):
        # binding the points (random landmark assignment) reuses this context's workspace
        self._device_state = None
        kNNGraph._bind_points(self)

    def _build_kernel(self):
        self._no_sharded("an MNN build")
        K0 = self._kernel0 = self._assemble_kernel0()
        nnz, flags = self._device_build_from_k0()
        data, indices, indptr = self.hip.graph_fetch_csr(_hip.CSR_K)
        n = K0.shape[0]
        if nnz < 2**31:
            indptr = indptr.astype(np.int32)
        K = sparse.csr_matrix((data, indices, indptr), shape=(n, n))
        self._emit_build_warnings(flags, K)
        return K

    def _fetch_diff_op(self):
        self._ensure_device_graph()
        data, _, _ = self.hip.graph_fetch_csr(_hip.CSR_P, structure=False)
        K = self._kernel
        return sparse.csr_matrix((data, K.indices, K.indptr), shape=K.shape)

    def _fetch_degree(self):
        self._ensure_device_graph()
        return self.hip.graph_fetch_vec(_hip.VEC_DEGREE)

    def _fetch_diff_aff(self):
        self.K
        self._ensure_device_graph()
        vals = self.hip.graph_diff_aff()
        K = self._kernel
        return sparse.csr_matrix((vals, K.indices, K.indptr), shape=K.shape)

    def build_kernel_to_data(self, Y, theta=None):
        # reference: graphs.py:1948-1966
        raise NotImplementedError


class kNNLandmarkGraph(kNNGraph, LandmarkGraph):
    # kNNGraph carries its own (device) extend_to_data / interpolate; on a landmark graph the landmark versions apply, as
    # in the reference, where only DataGraph and LandmarkGraph define them (graphs.py:1247-1317)
    extend_to_data = LandmarkGraph.extend_to_data
    interpolate = LandmarkGraph.interpolate


class MNNLandmarkGraph(MNNGraph, LandmarkGraph):
    """reference: graphs.py:1973-1974 - the landmark algebra on the batch-corrected kernel"""
    extend_to_data = LandmarkGraph.extend_to_data
    interpolate = LandmarkGraph.interpolate


class TraditionalLandmarkGraph(TraditionalGraph, LandmarkGraph):
    extend_to_data = LandmarkGraph.extend_to_data
    interpolate = LandmarkGraph.interpolate
