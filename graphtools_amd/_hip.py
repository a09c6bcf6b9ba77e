"""ctypes binding of libgraphtools_amd.so (the C ABI declared in include/graphtools_amd.h).

The product path fails loudly when the HIP library is missing or no MI355X is visible:
there is no CPU fallback (``HipUnavailableError``).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRAPHTOOLS_AMD_LIB: development override (kernel-variant experiments, tools/build_variant.py)
LIB_PATH = os.environ.get("GRAPHTOOLS_AMD_LIB") or os.path.join(_HERE, "libgraphtools_amd.so")

GT_F32, GT_F64 = 0, 1
GT_E_NONFINITE = -6  # include/graphtools_amd.h
SYMM = {None: 0, "none": 0, "+": 1, "*": 2, "mnn": 3}
MAX_KNN = 447   # deepest neighbour count of the candidate tables (gt_knn.cpp)
FLAG_DUPLICATES, FLAG_ZERO_DIAGONAL, FLAG_FALLBACK_ROWS, FLAG_RADIUS_ROWS = 1, 2, 4, 8
CSR_K, CSR_P = 0, 1
VEC_BANDWIDTH, VEC_DEGREE = 0, 1


class HipUnavailableError(RuntimeError):
    """The HIP extension (or a GPU) is not available; graphtools_amd has no CPU path."""


class HipError(RuntimeError):
    pass


class KnnParams(ctypes.Structure):
    """mirror of gt_knn_params"""

    _fields_ = [
        ("knn", ctypes.c_int32),
        ("kernel_symm", ctypes.c_int32),
        ("decay", ctypes.c_double),
        ("thresh", ctypes.c_double),
        ("bandwidth_scale", ctypes.c_double),
        ("theta", ctypes.c_double),
        ("anisotropy", ctypes.c_double),
        ("bandwidth", ctypes.c_void_p),
        ("bandwidth_len", ctypes.c_int64),
        ("knn_max", ctypes.c_int64),
    ]


_lib = None

_c = ctypes
_SIGNATURES = {
    "gt_abi_version": (_c.c_int, []),
    "gt_ctx_create": (_c.c_int, [_c.c_int, _c.POINTER(_c.c_void_p)]),
    "gt_ctx_destroy": (None, [_c.c_void_p]),
    "gt_last_error": (_c.c_char_p, [_c.c_void_p]),
    "gt_device_count": (_c.c_int, []),
    "gt_stage_ms": (_c.c_double, [_c.c_void_p, _c.c_char_p]),
    "gt_stage_launches": (_c.c_int, [_c.c_void_p, _c.c_char_p]),
    "gt_set_option": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_char_p]),
    "gt_set_points": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32]),
    "gt_knn_search": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_int32,
                                 _c.c_void_p, _c.c_void_p, _c.c_int32, _c.POINTER(_c.c_uint32)]),
    "gt_last_knn_precision": (_c.c_int, [_c.c_void_p]),
    "gt_graph_begin": (_c.c_int, [_c.c_void_p, _c.POINTER(KnnParams), _c.c_int32, _c.c_int32, _c.c_void_p, _c.c_void_p]),
    "gt_graph_emit": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "gt_graph_sym_plan": (_c.c_int, [_c.c_void_p, _c.POINTER(KnnParams), _c.c_int32, _c.c_int32, _c.c_void_p,
                                     _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int64), _c.c_void_p]),
    "gt_graph_sym_seed": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_int64), _c.c_void_p]),
    "gt_graph_sym_collect": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.POINTER(_c.c_int32), _c.c_void_p]),
    "gt_graph_sym_emit": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "gt_graph_sym_finish": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64]),
    "gt_points_cell_sort": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int32)]),
    "gt_points_cells_begin": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int64, _c.c_int64,
                                         _c.c_void_p, _c.POINTER(_c.c_int32)]),
    "gt_points_cells_finish": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "gt_points_shard_splits": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_void_p]),
    "gt_points_row_ids": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_int32]),
    "gt_points_device": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_int32),
                                    _c.POINTER(_c.c_int32)]),
    "gt_graph_shard_local": (_c.c_int, [_c.c_void_p, _c.POINTER(KnnParams), _c.c_int32, _c.c_int32, _c.c_void_p,
                                        _c.POINTER(_c.c_int32)]),
    "gt_graph_finish": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_uint32)]),
    "gt_graph_anisotropy": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "gt_graph_stage_counts": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int32, _c.c_int32, _c.c_void_p, _c.c_void_p,
                                         _c.POINTER(_c.c_int32)]),
    "gt_graph_set_stage_totals": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int32]),
    "gt_graph_bandwidth_local": (_c.c_int, [_c.c_void_p, _c.POINTER(KnnParams), _c.c_int32, _c.c_int32, _c.c_void_p, _c.c_void_p,
                                            _c.POINTER(_c.c_int32)]),
    "gt_graph_set_bandwidths": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "gt_graph_build": (_c.c_int, [_c.c_void_p, _c.POINTER(KnnParams), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_uint32)]),
    "gt_graph_extend": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.POINTER(KnnParams),
                                   _c.POINTER(_c.c_int64), _c.POINTER(_c.c_uint32)]),
    "gt_csr_graph_build": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int32,
                                      _c.c_double, _c.c_double, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_uint32)]),
    "gt_graph_rows": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    "gt_graph_fetch_csr": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int32]),
    "gt_graph_fetch_kp": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "gt_graph_to_dense": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_void_p, _c.c_int32, _c.c_int32]),
    "gt_release_cached_memory": (_c.c_int, []),
    "gt_host_place_block": (_c.c_int, [_c.c_int64] + [_c.c_void_p] * 9),
    "gt_graph_spmm": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int32]),
    "gt_graph_fetch_vec": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_void_p, _c.c_int32]),
    "gt_graph_diff_aff": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int32]),
    "gt_graph_stats": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "gt_knn_stats": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "gt_dense_graph_build": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32,
                                        _c.c_int32, _c.c_int32, _c.c_double, _c.c_double, _c.c_void_p, _c.c_int64,
                                        _c.c_double, _c.c_int32, _c.c_double, _c.c_double, _c.c_int32, _c.c_void_p,
                                        _c.c_void_p, _c.c_int32, _c.POINTER(_c.c_uint32)]),
    "gt_dense_extend": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_double, _c.c_double,
                                   _c.c_void_p, _c.c_int64, _c.c_double, _c.c_void_p, _c.c_int32]),
    "gt_dense_fetch_vec": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_void_p]),
    "gt_landmark_build": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int32, _c.c_void_p, _c.c_void_p, _c.c_int32,
                                     _c.POINTER(_c.c_int64)]),
    "gt_landmark_scale": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int32, _c.c_int32]),
    "gt_landmark_fetch_transitions": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int32]),
    "gt_nearest_landmark": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_int32, _c.c_int32, _c.c_void_p]),
    "gt_knn_first_nearest": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_void_p, _c.c_void_p]),
    "gt_pca_begin": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_void_p, _c.c_void_p]),
    "gt_pca_matmul": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_void_p, _c.c_int32, _c.c_int32, _c.c_void_p, _c.c_int32]),
    "gt_pca_tmatmul": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_int32, _c.c_void_p, _c.c_void_p]),
    "gt_pca_gram": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_int32, _c.c_void_p]),
    "gt_pca_fetch": (_c.c_int, [_c.c_void_p, _c.c_int32, _c.c_int32, _c.c_void_p, _c.c_int32]),
    "gt_pca_end": (_c.c_int, [_c.c_void_p]),
    "gt_thin_scale_rows": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_void_p, _c.c_double]),
    "gt_thin_gram": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_void_p]),
    "gt_thin_rmul": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_void_p, _c.c_int32, _c.c_void_p]),
    "gt_dev_alloc": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_void_p)]),
    "gt_dev_free": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "gt_dev_upload": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t]),
    "gt_dev_download": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t]),
    "gt_dev_sync": (_c.c_int, [_c.c_void_p]),
    "gt_stream_order": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int32]),
}


def load_library():
    """dlopen the in-tree library and declare every prototype (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipUnavailableError(
            "%s not found: build it with `python -m graphtools_amd._build` (needs hipcc). "
            "graphtools_amd has no CPU fallback." % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def release_cached_memory():
    """Hand the device memory parked by closed contexts (see gt_release_cached_memory) back to the driver, and drop the host
    blocks kept for result arrays (``_HostPool``)."""
    load_library().gt_release_cached_memory()
    _host_pool.clear()


class _HostPool:
    """Host memory for the big result arrays (the CSR of a graph: 2.3 GB at N = 1e6), RE-USED once nobody refers to an earlier
    result any more.  A fresh ``np.empty`` costs the copy threads a page fault per 4 KB they write - at 2.3 GB that is a third of
    the device -> host time of a graph (31 ms against 25 for resident memory, more with torch's allocator in the process).  A block
    is an array OF THE REQUESTED DTYPE the pool keeps a reference to (scipy's CSR constructor copies a float64 view of a byte
    block, it keeps a slice of a float64 array); callers get the head of it, whose ``base`` is the block: the block's
    reference count tells whether any array of an earlier result is still alive, and only a block nobody else refers to is
    handed out again (``np.empty`` semantics: the content is whatever was there).  At most ``GRAPHTOOLS_AMD_HOST_POOL_GB``
    (default 5 - two graphs of 10^6 rows: the one a caller still holds and the one being copied out -, 0 = off) stay cached
    after the graphs that used them are dropped; ``release_cached_memory()`` drops them.

    What the recycling rests on is CPython's reference count of the block (checked by a self-test at start-up; an interpreter
    where it fails, or a free-threaded build - whose counts are not a safe "nobody else holds this" -, never recycles).  A
    consumer that keeps a RAW POINTER into a result array without holding the array (ctypes, a C extension that does not take
    a buffer) is not seen by it: such a consumer must keep a reference to the array for as long as it uses the memory - as it
    must for any numpy array - or switch the pool off."""

    MIN_BYTES = 32 << 20
    ROUND = 16 << 20

    def __init__(self):
        import threading

        self.blocks = []   # least recently used first
        self.lock = threading.Lock()
        try:
            self.cap = int(float(os.environ.get("GRAPHTOOLS_AMD_HOST_POOL_GB", "5")) * (1 << 30))
        except ValueError:
            self.cap = 5 << 30
        if not self._refcounts_behave() or self._free_threaded():
            self.cap = 0   # (an interpreter whose reference counts do not say "nobody else holds this": never recycle)

    @staticmethod
    def _free_threaded():
        """a build without the GIL: reference counts are biased / deferred there, and two threads may race the check"""
        import sys
        import sysconfig

        try:
            if sysconfig.get_config_var("Py_GIL_DISABLED"):
                return True
            gil = getattr(sys, "_is_gil_enabled", None)
            return gil is not None and not gil()
        except Exception:
            return True

    @staticmethod
    def _refcounts_behave():
        """the pool's one assumption, checked once: a block held by a list alone counts 2 (the list, getrefcount's argument),
        and every view of it - however derived - adds one until it dies"""
        import sys

        try:
            held = [np.empty(64, dtype=np.float64)]
            if sys.getrefcount(held[0]) != 2:
                return False
            v = held[0][:8]
            w = v[2:4].view(np.int64)
            ok = sys.getrefcount(held[0]) == 4 and w.base is held[0]
            del v, w
            return ok and sys.getrefcount(held[0]) == 2
        except Exception:
            return False

    def clear(self):
        with self.lock:
            self.blocks = []

    def empty(self, count, dtype):
        import sys

        dtype = np.dtype(dtype)
        nbytes = int(count) * dtype.itemsize
        if self.cap <= 0 or nbytes < self.MIN_BYTES:
            return np.empty(count, dtype=dtype)
        with self.lock:
            best = -1
            for idx in range(len(self.blocks)):
                # references to a free block: the list's, and getrefcount's own argument
                if (nbytes <= self.blocks[idx].nbytes <= nbytes + nbytes // 4 + 4 * self.ROUND
                        and self.blocks[idx].dtype == dtype and sys.getrefcount(self.blocks[idx]) == 2
                        and (best < 0 or self.blocks[idx].nbytes < self.blocks[best].nbytes)):
                    best = idx
            if best >= 0:
                block = self.blocks.pop(best)
            else:
                size = (nbytes + self.ROUND - 1) // self.ROUND * self.ROUND
                total = sum(b.nbytes for b in self.blocks)
                idx = 0
                while total + size > self.cap and idx < len(self.blocks):   # make room: free blocks go first, oldest first
                    if sys.getrefcount(self.blocks[idx]) == 2:
                        total -= self.blocks[idx].nbytes
                        self.blocks.pop(idx)
                    else:
                        idx += 1
                if total + size > self.cap:
                    return np.empty(count, dtype=dtype)   # (the cap is held by live results: this one is not cached)
                block = np.empty(size // dtype.itemsize, dtype=dtype)
            self.blocks.append(block)
            return block[:int(count)]


_host_pool = _HostPool()


def host_place_block(M, rows_global, cols_global, scale, cursor, out_indices, out_data):
    """rows of the CSR block ``M`` (batch-local ids) -> their rows of the assembled kernel (gt_host_place_block)"""
    indptr = np.ascontiguousarray(M.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(M.indices, dtype=np.int32)
    data = np.ascontiguousarray(M.data, dtype=np.float64)
    rows_global = np.ascontiguousarray(rows_global, dtype=np.int64)
    cols_global = np.ascontiguousarray(cols_global, dtype=np.int64)
    if scale is not None:
        scale = np.ascontiguousarray(scale, dtype=np.float64)
    rc = load_library().gt_host_place_block(M.shape[0], _ptr(indptr), _ptr(indices), _ptr(data), _ptr(rows_global),
                                            _ptr(cols_global), _ptr(scale), _ptr(cursor), _ptr(out_indices), _ptr(out_data))
    if rc != 0:
        raise HipError("gt_host_place_block failed (%d)" % rc)


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class Context:
    """One gt_ctx (one HIP device, one stream, all device workspace)."""

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.gt_device_count() <= 0:
            raise HipUnavailableError("no HIP device visible; graphtools_amd has no CPU fallback")
        h = ctypes.c_void_p()
        rc = self.lib.gt_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise HipUnavailableError("gt_ctx_create failed: %s" % self.lib.gt_last_error(None).decode())
        self.h = h
        self.device = int(device)
        self._points = None
        self.n = 0
        self.d = 0
        self.dtype = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.gt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc == GT_E_NONFINITE:   # the reference raises sklearn's ValueError here (NearestNeighbors.fit / kneighbors)
            raise ValueError(self.lib.gt_last_error(self.h).decode())
        if rc != 0:
            raise HipError("%s failed (%d): %s" % (what, rc, self.lib.gt_last_error(self.h).decode()))

    def stage_ms(self, stage):
        return float(self.lib.gt_stage_ms(self.h, stage.encode()))

    def stage_launches(self, stage):
        return int(self.lib.gt_stage_launches(self.h, stage.encode()))

    def set_option(self, name, value):
        self._check(self.lib.gt_set_option(self.h, name.encode(), str(value).encode()), "gt_set_option")

    # ---- points -----------------------------------------------------------------------------
    def set_points(self, X):
        """Bind host data (numpy float32/float64, C order).  The array is copied to the device."""
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        self._points = X
        self.n, self.d = X.shape
        self.dtype = X.dtype
        self._check(
            self.lib.gt_set_points(self.h, _ptr(X), self.n, self.d, GT_F32 if X.dtype == np.float32 else GT_F64, 0),
            "gt_set_points",
        )

    def set_points_device(self, dev_ptr, n, d, dtype):
        """Bind a device-resident n x d matrix (pointer must stay valid while bound)."""
        self._points = None
        self.n, self.d = int(n), int(d)
        self.dtype = np.dtype(dtype)
        self._check(
            self.lib.gt_set_points(self.h, ctypes.c_void_p(int(dev_ptr)), self.n, self.d,
                                   GT_F32 if self.dtype == np.float32 else GT_F64, 1),
            "gt_set_points",
        )

    # ---- kNN --------------------------------------------------------------------------------
    def points_device(self, row0=0):
        """device address of row `row0` of the bound points as the context holds them"""
        p = ctypes.c_void_p()
        self._check(self.lib.gt_points_device(self.h, int(row0), ctypes.byref(p), None, None), "gt_points_device")
        return p.value

    def knn_search_device(self, k, y_dev_ptr, m):
        """gt_knn_search with a device-resident query matrix (m rows of the bound points' dtype and width)"""
        idx = np.empty((m, k), dtype=np.int64)
        dist = np.empty((m, k), dtype=np.float64)
        flags = ctypes.c_uint32(0)
        self._check(self.lib.gt_knn_search(self.h, 0, 0, ctypes.c_void_p(int(y_dev_ptr)), int(m), 1, int(k), _ptr(idx), _ptr(dist), 0,
                                           ctypes.byref(flags)), "gt_knn_search")
        return dist, idx, flags.value

    def knn_first_nearest(self, k, Y=None, y_dev_ptr=None, m=None):
        """gt_knn_first_nearest: for every query row the lowest index among its nearest bound points that tie (int64 [m]);
        Y: host matrix, or y_dev_ptr / m: a device-resident one of the bound points' dtype and width"""
        if Y is not None:
            Y = np.ascontiguousarray(Y, dtype=self.dtype)
            m, ptr, on_dev = Y.shape[0], _ptr(Y), 0
        else:
            ptr, on_dev = ctypes.c_void_p(int(y_dev_ptr)), 1
        out = np.empty(int(m), dtype=np.int64)
        flags = ctypes.c_uint32(0)
        self._check(self.lib.gt_knn_first_nearest(self.h, ptr, int(m), on_dev, int(k), _ptr(out), ctypes.byref(flags)),
                    "gt_knn_first_nearest")
        return out

    def knn_search(self, k, rows=None, Y=None):
        """(distances float64 [m,k], indices int64 [m,k], flags)"""
        if Y is not None:
            Y = np.ascontiguousarray(Y, dtype=self.dtype)
            m = Y.shape[0]
            r0, r1 = 0, 0
        else:
            r0, r1 = rows if rows is not None else (0, self.n)
            m = r1 - r0
        idx = np.empty((m, k), dtype=np.int64)
        dist = np.empty((m, k), dtype=np.float64)
        flags = ctypes.c_uint32(0)
        self._check(
            self.lib.gt_knn_search(self.h, r0, r1, _ptr(Y), m, 0, int(k), _ptr(idx), _ptr(dist), 0, ctypes.byref(flags)),
            "gt_knn_search",
        )
        return dist, idx, flags.value

    # ---- graph ------------------------------------------------------------------------------
    @staticmethod
    def make_params(knn, decay, thresh, bandwidth, bandwidth_scale, knn_max, kernel_symm, theta, anisotropy):
        p = KnnParams()
        p.knn = int(knn)
        p.kernel_symm = SYMM[kernel_symm]
        p.decay = float("nan") if decay is None else float(decay)
        p.thresh = float(thresh)
        p.bandwidth_scale = float(bandwidth_scale)
        p.theta = 1.0 if theta is None else float(theta)
        p.anisotropy = float(anisotropy)
        keep = None
        if bandwidth is None:
            p.bandwidth = None
            p.bandwidth_len = 0
        else:
            keep = np.ascontiguousarray(np.atleast_1d(np.asarray(bandwidth, dtype=np.float64)))
            p.bandwidth = keep.ctypes.data
            p.bandwidth_len = keep.shape[0]
        p.knn_max = -1 if knn_max is None else int(knn_max)
        return p, keep

    def graph_build(self, params):
        nnz = ctypes.c_int64(0)
        flags = ctypes.c_uint32(0)
        self._check(self.lib.gt_graph_build(self.h, ctypes.byref(params), ctypes.byref(nnz), ctypes.byref(flags)),
                    "gt_graph_build")
        return nnz.value, flags.value

    def graph_extend(self, Y, params):
        """K_yx / transitions for new points Y against the bound points; returns (nnz, flags)"""
        Y = np.ascontiguousarray(Y, dtype=self.dtype)
        nnz = ctypes.c_int64(0)
        flags = ctypes.c_uint32(0)
        self._check(self.lib.gt_graph_extend(self.h, _ptr(Y), Y.shape[0], 0, ctypes.byref(params), ctypes.byref(nnz),
                                             ctypes.byref(flags)), "gt_graph_extend")
        return nnz.value, flags.value

    def last_knn_precision(self):
        """arithmetic of the last main candidate pass: 'f32', 'f16' (split, 3 chains) or 'f16x1' (single chain)"""
        return {0: "f32", 1: "f16", 2: "f16x1"}[self.lib.gt_last_knn_precision(self.h)]

    def csr_graph_build(self, K0, kernel_symm, theta, anisotropy, assume_unique=False):
        """symmetrise + anisotropy + row-normalise a host CSR kernel on the device; returns (nnz, flags).  The device
        merge sorts every row itself; the columns of a row only have to be unique (``assume_unique``: the caller
        vouches for that, else duplicates are summed here first)."""
        K0 = K0.tocsr()
        if not assume_unique:
            K0.sum_duplicates()
        indptr = np.ascontiguousarray(K0.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(K0.indices, dtype=np.int32)
        data = np.ascontiguousarray(K0.data, dtype=np.float64)
        nnz = ctypes.c_int64(0)
        flags = ctypes.c_uint32(0)
        self._check(self.lib.gt_csr_graph_build(self.h, K0.shape[0], _ptr(indptr), _ptr(indices), _ptr(data),
                                                SYMM[kernel_symm], float(theta if theta is not None else 1.0),
                                                float(anisotropy), ctypes.byref(nnz), ctypes.byref(flags)),
                    "gt_csr_graph_build")
        return nnz.value, flags.value

    def graph_stage_counts(self, params, world, rank, row_splits):
        """row-sharded builds with knn_max: this rank's counts for the reference's search-expansion loop (int64 array, possibly
        empty: nothing to exchange)"""
        splits = np.ascontiguousarray(row_splits, dtype=np.int64)
        counts = np.zeros(4, dtype=np.int64)
        n = ctypes.c_int32(0)
        self._check(self.lib.gt_graph_stage_counts(self.h, ctypes.byref(params), world, rank, _ptr(splits), _ptr(counts),
                                                   ctypes.byref(n)), "gt_graph_stage_counts")
        return counts[: n.value].copy()

    def graph_set_stage_totals(self, totals):
        totals = np.ascontiguousarray(totals, dtype=np.int64)
        buf = np.zeros(4, dtype=np.int64)
        buf[: len(totals)] = totals
        self._check(self.lib.gt_graph_set_stage_totals(self.h, _ptr(buf), int(len(totals))), "gt_graph_set_stage_totals")

    def graph_begin(self, params, world, rank, row_splits):
        splits = np.ascontiguousarray(row_splits, dtype=np.int64)
        counts = np.zeros(world, dtype=np.int64)
        self._check(self.lib.gt_graph_begin(self.h, ctypes.byref(params), world, rank, _ptr(splits), _ptr(counts)),
                    "gt_graph_begin")
        return counts

    # ---- row-sharded symmetric candidate pass (gt_knn_shard.cpp); dist.ShardedKnnGraph drives the stages ----
    def graph_sym_plan(self, params, world, rank, row_splits):
        """-> (applies, n_pad_sorted, sorted_splits): whether this context can run the sharded symmetric pass for these
        parameters, and the split of the cell-sorted positions whose thresholds each rank seeds"""
        splits = np.ascontiguousarray(row_splits, dtype=np.int64)
        applies = ctypes.c_int32(0)
        n_pad = ctypes.c_int64(0)
        sorted_splits = np.zeros(world + 1, dtype=np.int64)
        self._check(self.lib.gt_graph_sym_plan(self.h, ctypes.byref(params), world, rank, _ptr(splits), ctypes.byref(applies),
                                               ctypes.byref(n_pad), _ptr(sorted_splits)), "gt_graph_sym_plan")
        return bool(applies.value), n_pad.value, sorted_splits

    def graph_sym_seed(self, thr_local_ptr):
        """launch A for the planned share; float32 {threshold, far-kept seeds} pairs of its positions to thr_local_ptr
        (device, [rows][2]); -> statistics of the
        share the host sums over the ranks: numpy float64 [far-kept rows, then the four sums behind the orphan cut]"""
        far = ctypes.c_int64(0)
        rad = np.zeros(4, dtype=np.float64)
        self._check(self.lib.gt_graph_sym_seed(self.h, ctypes.c_void_p(int(thr_local_ptr)) if thr_local_ptr else None,
                                               ctypes.byref(far), _ptr(rad)), "gt_graph_sym_seed")
        return np.concatenate([[float(far.value)], rad])

    def graph_sym_collect(self, thr_all_ptr, stats_total, world):
        """launch B for this rank's pieces against the gathered pairs of all rows (device, float32 [n_pad_sorted][2]) and the
        summed statistics of graph_sym_seed; -> (applies, send_counts): records per destination rank, or applies False
        when the predictor refuses"""
        applies = ctypes.c_int32(0)
        counts = np.zeros(world, dtype=np.int64)
        st = np.ascontiguousarray(stats_total, dtype=np.float64)
        rad = np.ascontiguousarray(st[1:5])
        self._check(self.lib.gt_graph_sym_collect(self.h, ctypes.c_void_p(int(thr_all_ptr)), int(round(st[0])), _ptr(rad),
                                                  ctypes.byref(applies), _ptr(counts)), "gt_graph_sym_collect")
        return bool(applies.value), counts

    def graph_sym_emit(self, send_ptr):
        self._check(self.lib.gt_graph_sym_emit(self.h, ctypes.c_void_p(int(send_ptr)) if send_ptr else None),
                    "gt_graph_sym_emit")

    def graph_sym_finish(self, recv_ptr, n_recv):
        self._check(self.lib.gt_graph_sym_finish(self.h, ctypes.c_void_p(int(recv_ptr)) if recv_ptr else None, int(n_recv)),
                    "gt_graph_sym_finish")

    # ---- cell-sorted renumbering: the row-sharded build without candidate exchange (gt_knn_shard.cpp) ----
    def points_cell_sort(self):
        """renumber the bound points in cell-sorted order; -> True when applied (False: too few / too wide points)"""
        applied = ctypes.c_int32(0)
        self._check(self.lib.gt_points_cell_sort(self.h, ctypes.byref(applied)), "gt_points_cell_sort")
        return bool(applied.value)

    def points_cells_begin(self, dev_ptr, n, d, dtype, row0, row1, cells_out_ptr):
        """bind device-resident points and assign the rows [row0, row1) to their landmark cells (uint32 per row at
        cells_out_ptr, device); -> True when a cell order applies (then all-gather the cells and call points_cells_finish)"""
        self.n, self.d, self.dtype = int(n), int(d), np.dtype(dtype)
        applied = ctypes.c_int32(0)
        self._check(self.lib.gt_points_cells_begin(self.h, ctypes.c_void_p(int(dev_ptr)), self.n, self.d,
                                                   GT_F32 if self.dtype == np.float32 else GT_F64, int(row0), int(row1),
                                                   ctypes.c_void_p(int(cells_out_ptr)) if cells_out_ptr else None,
                                                   ctypes.byref(applied)), "gt_points_cells_begin")
        return bool(applied.value)

    def points_cells_finish(self, cells_all_ptr):
        self._check(self.lib.gt_points_cells_finish(self.h, ctypes.c_void_p(int(cells_all_ptr))), "gt_points_cells_finish")

    def points_shard_splits(self, world):
        splits = np.zeros(world + 1, dtype=np.int64)
        self._check(self.lib.gt_points_shard_splits(self.h, int(world), _ptr(splits)), "gt_points_shard_splits")
        return splits

    def points_row_ids(self, row0, row1):
        """the caller's row numbers of the context's rows [row0, row1) (numpy int32)"""
        out = np.zeros(max(int(row1 - row0), 0), dtype=np.int32)
        if len(out):
            self._check(self.lib.gt_points_row_ids(self.h, int(row0), int(row1), _ptr(out), 0), "gt_points_row_ids")
        return out

    def graph_shard_local(self, params, world, rank, row_splits):
        splits = np.ascontiguousarray(row_splits, dtype=np.int64)
        applies = ctypes.c_int32(0)
        self._check(self.lib.gt_graph_shard_local(self.h, ctypes.byref(params), int(world), int(rank), _ptr(splits),
                                                  ctypes.byref(applies)), "gt_graph_shard_local")
        return bool(applies.value)

    def graph_bandwidth_local(self, params, world, rank, row_splits, bw_local_ptr):
        """Row-sharded pair-resolved tail: the first half of graph_begin for the rank's rows, its bandwidths (float64, device,
        on the library's stream) into ``bw_local_ptr``.  False: not this tail's parameters - nothing was done, the same answer
        on every rank; ``graph_begin`` runs the whole build."""
        splits = np.ascontiguousarray(row_splits, dtype=np.int64)
        applies = ctypes.c_int32(0)
        self._check(self.lib.gt_graph_bandwidth_local(self.h, ctypes.byref(params), int(world), int(rank), _ptr(splits),
                                                      ctypes.c_void_p(int(bw_local_ptr)) if bw_local_ptr else None,
                                                      ctypes.byref(applies)), "gt_graph_bandwidth_local")
        return bool(applies.value)

    def graph_set_bandwidths(self, bw_all_ptr):
        """... the bandwidths of all rows (float64 [n], device, rank slices in rank order; alive until graph_begin returned)"""
        self._check(self.lib.gt_graph_set_bandwidths(self.h, ctypes.c_void_p(int(bw_all_ptr))), "gt_graph_set_bandwidths")

    def graph_emit(self, send_ptr):
        self._check(self.lib.gt_graph_emit(self.h, ctypes.c_void_p(int(send_ptr)) if send_ptr else None), "gt_graph_emit")

    def graph_finish(self, recv_ptr, n_recv):
        nnz = ctypes.c_int64(0)
        flags = ctypes.c_uint32(0)
        self._check(
            self.lib.gt_graph_finish(self.h, ctypes.c_void_p(int(recv_ptr)) if recv_ptr else None, int(n_recv),
                                     ctypes.byref(nnz), ctypes.byref(flags)),
            "gt_graph_finish",
        )
        return nnz.value, flags.value

    def graph_anisotropy(self, degree_all_ptr):
        self._check(self.lib.gt_graph_anisotropy(self.h, ctypes.c_void_p(int(degree_all_ptr))), "gt_graph_anisotropy")

    def graph_rows(self):
        r0, r1, nnz = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.gt_graph_rows(self.h, ctypes.byref(r0), ctypes.byref(r1), ctypes.byref(nnz)), "gt_graph_rows")
        return r0.value, r1.value, nnz.value

    def graph_fetch_csr(self, which, structure=True):
        """host copies: (data float64, indices int32, indptr int64) for the owned rows; ``structure=False`` copies
        the values only (K and P share indices / indptr) and returns (data, None, None)"""
        r0, r1, nnz = self.graph_rows()
        data = _host_pool.empty(nnz, np.float64)
        indices = _host_pool.empty(nnz, np.int32) if structure else None
        indptr = np.empty(r1 - r0 + 1, dtype=np.int64) if structure else None
        self._check(self.lib.gt_graph_fetch_csr(self.h, which, _ptr(data), _ptr(indices) if structure else None,
                                                _ptr(indptr) if structure else None, 0), "gt_graph_fetch_csr")
        return data, indices, indptr

    def graph_fetch_kp(self):
        """host copies of K and P in one pass over the link: (K data, indices int32, indptr int64, P data); the P values are
        derived on the host from K and the degrees while K is still arriving (bit-identical to the device's P)"""
        r0, r1, nnz = self.graph_rows()
        kd = _host_pool.empty(nnz, np.float64)
        pd = _host_pool.empty(nnz, np.float64)
        indices = _host_pool.empty(nnz, np.int32)
        indptr = np.empty(r1 - r0 + 1, dtype=np.int64)
        self._check(self.lib.gt_graph_fetch_kp(self.h, _ptr(kd), _ptr(indices), _ptr(indptr), _ptr(pd)), "gt_graph_fetch_kp")
        return kd, indices, indptr, pd

    def graph_to_dense(self, which, n, dtype=np.float64, out_device=None):
        """dense copy [owned rows, n] of K or P (n = columns of the graph): a host ndarray, or (``out_device``: a CUDA torch
        tensor of that shape and dtype) written in place on the device"""
        r0, r1, _ = self.graph_rows()
        code = GT_F32 if np.dtype(dtype) == np.float32 else GT_F64
        if out_device is not None:
            self._check(self.lib.gt_graph_to_dense(self.h, which, ctypes.c_void_p(out_device.data_ptr()), code, 1),
                        "gt_graph_to_dense")
            return out_device
        out = np.empty((r1 - r0, n), dtype=dtype)
        self._check(self.lib.gt_graph_to_dense(self.h, which, _ptr(out), code, 0), "gt_graph_to_dense")
        return out

    def graph_csr_torch(self, which):
        """Device-resident hand-off: the owned rows of K or P as a ``torch.sparse_csr_tensor`` on this context's GPU
        (device-to-device copies into torch-owned memory; nothing crosses PCIe).  For consumers that keep working
        on the device - diffusion powers ``P^t X``, spectral front ends.  torch ships its own copy of the HIP
        runtime: initialise it (``torch.cuda.init()``) before the first graphtools_amd call of the process."""
        import torch

        if not torch.cuda.is_initialized():
            try:
                torch.cuda.init()
            except RuntimeError as e:
                raise RuntimeError("torch could not initialise its HIP runtime after libgraphtools_amd.so had been "
                                   "used in this process; call torch.cuda.init() first") from e

        r0, r1, nnz = self.graph_rows()
        dev = torch.device("cuda", self.device)
        data = torch.empty(nnz, dtype=torch.float64, device=dev)
        indices = torch.empty(nnz, dtype=torch.int32, device=dev)
        indptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)   # torch's allocator may hand out memory with work pending on its own streams
        self._check(self.lib.gt_graph_fetch_csr(self.h, which, ctypes.c_void_p(data.data_ptr()),
                                                ctypes.c_void_p(indices.data_ptr()), ctypes.c_void_p(indptr.data_ptr()), 1),
                    "gt_graph_fetch_csr")
        n_cols = self.n
        # torch wants one index dtype for both arrays: narrow the short one (row pointers) when the counts allow it
        if nnz < 2**31:
            indptr = indptr.to(torch.int32)
        else:
            indices = indices.to(torch.int64)
        return torch.sparse_csr_tensor(indptr, indices, data, size=(r1 - r0, n_cols))

    def graph_spmm(self, which, X):
        """(K or P)[owned rows] @ X on the device; X: host array [n_columns, c] (any float dtype, converted to
        float64 like scipy would) -> host float64 array [owned rows, c]"""
        r0, r1, _ = self.graph_rows()
        X = np.ascontiguousarray(X, dtype=np.float64)
        squeeze = X.ndim == 1
        if squeeze:
            X = X[:, None]
        out = np.empty((r1 - r0, X.shape[1]), dtype=np.float64)
        self._check(self.lib.gt_graph_spmm(self.h, which, _ptr(X), X.shape[1], _ptr(out), 0), "gt_graph_spmm")
        return out[:, 0] if squeeze else out

    def graph_diff_aff(self):
        """values of D^-1/2 K D^-1/2 on the CSR structure of K (owned rows; single-rank builds)"""
        _, _, nnz = self.graph_rows()
        out = np.empty(nnz, dtype=np.float64)
        self._check(self.lib.gt_graph_diff_aff(self.h, None, _ptr(out), 0), "gt_graph_diff_aff")
        return out

    def graph_fetch_vec(self, which):
        r0, r1, _ = self.graph_rows()
        out = np.empty(r1 - r0, dtype=np.float64)
        self._check(self.lib.gt_graph_fetch_vec(self.h, which, _ptr(out), 0), "gt_graph_fetch_vec")
        return out

    # ---- tall-matrix products of the PCA pre-reduction (gt_pca.hip; algorithm in graphtools_amd/_pca.py) ----------
    def pca_begin(self, X):
        """bind the float32 matrix; returns (column means, centred sums of squares) as float64"""
        X = np.ascontiguousarray(X, dtype=np.float32)
        n, d = X.shape
        mean = np.empty(d, dtype=np.float64)
        ssq = np.empty(d, dtype=np.float64)
        self._check(self.lib.gt_pca_begin(self.h, _ptr(X), n, d, 0, _ptr(mean), _ptr(ssq)), "gt_pca_begin")
        self._pca_shape = (n, d)
        return mean, ssq

    def pca_matmul(self, src, W, sub, dst):
        W = np.ascontiguousarray(W, dtype=np.float64)
        sub = None if sub is None else np.ascontiguousarray(sub, dtype=np.float64)
        self._check(self.lib.gt_pca_matmul(self.h, int(src), _ptr(W), W.shape[0], W.shape[1], _ptr(sub), int(dst)),
                    "gt_pca_matmul")

    def pca_tmatmul(self, ybuf, k):
        out = np.empty((self._pca_shape[1], k), dtype=np.float64)
        colsum = np.empty(k, dtype=np.float64)
        self._check(self.lib.gt_pca_tmatmul(self.h, int(ybuf), int(k), _ptr(out), _ptr(colsum)), "gt_pca_tmatmul")
        return out, colsum

    def pca_gram(self, ybuf, k):
        out = np.empty((k, k), dtype=np.float64)
        self._check(self.lib.gt_pca_gram(self.h, int(ybuf), int(k), _ptr(out)), "gt_pca_gram")
        return out

    def pca_fetch(self, ybuf, k):
        out = np.empty((self._pca_shape[0], k), dtype=np.float32)
        self._check(self.lib.gt_pca_fetch(self.h, int(ybuf), int(k), _ptr(out), 0), "gt_pca_fetch")
        return out

    def pca_end(self):
        self._check(self.lib.gt_pca_end(self.h), "gt_pca_end")

    def knn_stats(self):
        st = np.zeros(12, dtype=np.int64)
        self._check(self.lib.gt_knn_stats(self.h, _ptr(st)), "gt_knn_stats")
        out = {"symmetric": bool(st[0]), "sym_overflow_rows": int(st[1]), "repaired_rows": int(st[2]),
               "exhaustive_rows": int(st[3])}
        out["sym_far_kept"] = int(st[10])
        out["sym_bound_pass"] = bool(st[0]) and bool(st[4])
        if st[0]:
            out.update(sym_nseg=int(st[6]), sym_seed_tiles=int(st[9]), sym_cold_pairs=int(st[5]), sym_two_stage=bool(st[7] & 1), sym_seed_dense=bool(st[7] & 2), sym_cold_local=bool(st[7] & 4), tables_by_slot=bool(st[7] & 8), destinations_fused=bool(st[7] & 16), sym_listed=bool(st[7] & 32),
                       sym_rows_over_256=int(st[8]),   # (without dbg_select bit 256, after a two-stage collect: the units stage one scored)
                       sym_stage_one_units=int(st[8]) if (st[7] & 1) and not st[4] else 0,
                       sym_rows_over_128=int(st[11]))   # list-length counters: only with dbg_select bit 256
        return out

    def graph_stats(self):
        st = np.zeros(4, dtype=np.int64)
        self._check(self.lib.gt_graph_stats(self.h, _ptr(st)), "gt_graph_stats")
        return {"fallback_rows": int(st[0]), "radius_rows": int(st[1]), "nnz_unsymmetrised": int(st[2]),
                "radius_retries": int(st[3])}

    # ---- exact dense graph -------------------------------------------------------------------
    def dense_graph_build(self, data, precomputed, knn, decay, thresh, bandwidth, bandwidth_scale, kernel_symm, theta,
                          anisotropy, want_P=True):
        """(K ndarray, P ndarray or None, flags).  Result dtype follows numpy's rules in the reference.
        ``precomputed``: None / False (points), "distance" / True, "affinity", "adjacency"."""
        mode = {None: 0, False: 0, True: 1, "distance": 1, "affinity": 2, "adjacency": 3}[precomputed]
        precomputed = mode != 0
        data = np.ascontiguousarray(data)
        if data.dtype not in (np.float32, np.float64):
            data = data.astype(np.float64)
        n = data.shape[0]
        d = 0 if precomputed else data.shape[1]
        bw = None
        bw_len = 0
        if bandwidth is not None:
            bw = np.ascontiguousarray(np.atleast_1d(np.asarray(bandwidth, dtype=np.float64)))
            bw_len = bw.shape[0]
        out_f64 = (not precomputed) or data.dtype == np.float64 or bw_len > 1
        odt = np.float64 if out_f64 else np.float32
        K = np.empty((n, n), dtype=odt)
        P = np.empty((n, n), dtype=odt) if want_P else None
        flags = ctypes.c_uint32(0)
        self._check(
            self.lib.gt_dense_graph_build(
                self.h, _ptr(data), n, d, GT_F32 if data.dtype == np.float32 else GT_F64, 0, mode,
                int(knn) if knn is not None else 0, float("nan") if decay is None else float(decay), float(thresh), _ptr(bw),
                bw_len, float(bandwidth_scale),
                SYMM[kernel_symm], 1.0 if theta is None else float(theta), float(anisotropy), 0, _ptr(K), _ptr(P), 0,
                ctypes.byref(flags)),
            "gt_dense_graph_build",
        )
        if not precomputed:
            self.n, self.d, self.dtype = n, data.shape[1], data.dtype
        return K, P, flags.value

    def dense_extend(self, Y, knn, decay, thresh, bandwidth, bandwidth_scale):
        """dense float64 kernel [m, n] from new points Y to the points of the last from-data dense build"""
        Y = np.ascontiguousarray(Y, dtype=self.dtype)
        m = Y.shape[0]
        bw = None
        bw_len = 0
        if bandwidth is not None:
            bw = np.ascontiguousarray(np.atleast_1d(np.asarray(bandwidth, dtype=np.float64)))
            bw_len = bw.shape[0]
        K = np.empty((m, self.n), dtype=np.float64)
        self._check(self.lib.gt_dense_extend(self.h, _ptr(Y), m, 0, int(knn) if knn is not None else 0, float(decay),
                                             float(thresh), _ptr(bw), bw_len, float(bandwidth_scale), _ptr(K), 0),
                    "gt_dense_extend")
        return K

    def dense_fetch_vec(self, which, n):
        out = np.empty(n, dtype=np.float64)
        self._check(self.lib.gt_dense_fetch_vec(self.h, which, _ptr(out)), "gt_dense_fetch_vec")
        return out

    # ---- landmarks ----------------------------------------------------------------------------
    def nearest_landmark(self, landmarks, mode, rows=None):
        r0, r1 = rows if rows is not None else (0, self.n)
        lm = np.ascontiguousarray(landmarks, dtype=np.int64)
        out = np.empty(r1 - r0, dtype=np.int32)
        self._check(self.lib.gt_nearest_landmark(self.h, r0, r1, _ptr(lm), lm.shape[0], int(mode), _ptr(out)),
                    "gt_nearest_landmark")
        return out

    def landmark_build(self, clusters, n_landmark):
        """(M [L,L], R [L], transitions nnz): unscaled partial operator of the owned rows (host copies)"""
        cl = np.ascontiguousarray(clusters, dtype=np.int32)
        M = np.empty((n_landmark, n_landmark), dtype=np.float64)
        R = np.empty(n_landmark, dtype=np.float64)
        tnnz = ctypes.c_int64(0)
        self._check(self.lib.gt_landmark_build(self.h, _ptr(cl), int(n_landmark), _ptr(M), _ptr(R), 0, ctypes.byref(tnnz)),
                    "gt_landmark_build")
        return M, R, tnnz.value

    def landmark_build_device(self, clusters, n_landmark, buf_ptr):
        """the same partial products left ON THE DEVICE: ``buf_ptr`` addresses L*L + L float64 (M row-major, then R) - the buffer
        a row-sharded build all-reduces over the ranks before :meth:`landmark_scale_device`.  Returns the transitions' nnz."""
        cl = np.ascontiguousarray(clusters, dtype=np.int32)
        L = int(n_landmark)
        tnnz = ctypes.c_int64(0)
        self._check(self.lib.gt_landmark_build(self.h, _ptr(cl), L, ctypes.c_void_p(int(buf_ptr)),
                                               ctypes.c_void_p(int(buf_ptr) + 8 * L * L), 1, ctypes.byref(tnnz)), "gt_landmark_build")
        return tnnz.value

    def landmark_scale_device(self, buf_ptr, n_landmark):
        """landmark_op = M / R in place on the device buffer of :meth:`landmark_build_device` (after the ranks' sum)"""
        L = int(n_landmark)
        self._check(self.lib.gt_landmark_scale(self.h, ctypes.c_void_p(int(buf_ptr)), ctypes.c_void_p(int(buf_ptr) + 8 * L * L), L, 1),
                    "gt_landmark_scale")

    def landmark_scale(self, M, R):
        M = np.ascontiguousarray(M, dtype=np.float64)
        R = np.ascontiguousarray(R, dtype=np.float64)
        self._check(self.lib.gt_landmark_scale(self.h, _ptr(M), _ptr(R), M.shape[0], 0), "gt_landmark_scale")
        return M

    def landmark_fetch_transitions(self, tnnz):
        r0, r1, _ = self.graph_rows()
        data = np.empty(tnnz, dtype=np.float64)
        indices = np.empty(tnnz, dtype=np.int32)
        indptr = np.empty(r1 - r0 + 1, dtype=np.int64)
        self._check(self.lib.gt_landmark_fetch_transitions(self.h, _ptr(data), _ptr(indices), _ptr(indptr), 0),
                    "gt_landmark_fetch_transitions")
        return data, indices, indptr

    # ---- tall thin float64 matrices on the device (gt_thin.hip; driver: graphtools_amd/_spectral.py) -------------
    def thin_scale_rows(self, A, n, k, v, power):
        self._check(self.lib.gt_thin_scale_rows(self.h, ctypes.c_void_p(int(A)), int(n), int(k), ctypes.c_void_p(int(v)),
                                                float(power)), "gt_thin_scale_rows")

    def thin_gram(self, A, n, k):
        out = np.empty((k, k), dtype=np.float64)
        self._check(self.lib.gt_thin_gram(self.h, ctypes.c_void_p(int(A)), int(n), int(k), _ptr(out)), "gt_thin_gram")
        return out

    def thin_rmul(self, A, n, k, R, B):
        R = np.ascontiguousarray(R, dtype=np.float64)
        assert R.shape[0] == k
        self._check(self.lib.gt_thin_rmul(self.h, ctypes.c_void_p(int(A)), int(n), int(k), _ptr(R), R.shape[1],
                                          ctypes.c_void_p(int(B))), "gt_thin_rmul")

    def graph_spmm_device(self, which, X_dev, ncols, out_dev):
        """(K or P) @ X with X and the result in device memory (row-major float64 [n, ncols])"""
        self._check(self.lib.gt_graph_spmm(self.h, which, ctypes.c_void_p(int(X_dev)), int(ncols),
                                           ctypes.c_void_p(int(out_dev)), 1), "gt_graph_spmm")

    def graph_fetch_vec_device(self, which, out_dev):
        self._check(self.lib.gt_graph_fetch_vec(self.h, which, ctypes.c_void_p(int(out_dev)), 1), "gt_graph_fetch_vec")

    # ---- raw device memory --------------------------------------------------------------------
    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self.lib.gt_dev_alloc(self.h, int(nbytes), ctypes.byref(p)), "gt_dev_alloc")
        return p.value

    def dev_free(self, p):
        self._check(self.lib.gt_dev_free(self.h, ctypes.c_void_p(int(p))), "gt_dev_free")

    def dev_upload(self, dst, arr):
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.gt_dev_upload(self.h, ctypes.c_void_p(int(dst)), _ptr(arr), arr.nbytes), "gt_dev_upload")

    def dev_download(self, arr, src):
        self._check(self.lib.gt_dev_download(self.h, _ptr(arr), ctypes.c_void_p(int(src)), arr.nbytes), "gt_dev_download")

    def sync(self):
        self._check(self.lib.gt_dev_sync(self.h), "gt_dev_sync")

    def wait_for_stream(self, stream_handle):
        """the context's later work waits (on the device, not the host) for what ``stream_handle`` holds now"""
        self._check(self.lib.gt_stream_order(self.h, ctypes.c_void_p(int(stream_handle)), 0), "gt_stream_order")

    def stream_waits_for_me(self, stream_handle):
        """``stream_handle`` waits for everything the context has queued so far"""
        self._check(self.lib.gt_stream_order(self.h, ctypes.c_void_p(int(stream_handle)), 1), "gt_stream_order")
