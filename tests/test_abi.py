"""CPU: the C-ABI library loads, exports every symbol declared in include/graphtools_amd.h, its structs
have the declared layout, and the product path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from graphtools_amd import _hip

HEADER = os.path.join(ROOT, "include", "graphtools_amd.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_hip.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def header_options():
    """the option names include/graphtools_amd.h documents (between its OPTIONS-BEGIN / OPTIONS-END markers, one per line)"""
    text = open(HEADER).read()
    block = text[text.index("OPTIONS-BEGIN"):text.index("OPTIONS-END")]
    return re.findall(r"^ \*   ([a-z][a-z0-9_]*) ", block, flags=re.M)


def test_header_documents_exactly_the_options_the_parser_accepts():
    """the public header is the boundary's documentation: every option it names is one gt_set_option parses
    (csrc/gt_api.cpp) and the other way round (the device-side check that each is ACCEPTED: test_gpu_dropin.py)"""
    src = open(os.path.join(ROOT, "graphtools_amd", "csrc", "gt_api.cpp")).read()
    body = src[src.index("int gt_set_option("):src.index('"unknown option"')]
    parsed = re.findall(r'k == "([a-z0-9_]+)"', body)
    named = header_options()
    assert len(named) == len(set(named)) and len(parsed) == len(set(parsed))
    assert sorted(named) == sorted(parsed), (sorted(set(named) - set(parsed)), sorted(set(parsed) - set(named)))
    # ... and no option name appears in the header's prose that is not in the list (stale documentation)
    text = open(HEADER).read()
    prose = text[:text.index("OPTIONS-BEGIN")] + text[text.index("OPTIONS-END"):]
    quoted = set(re.findall(r'"((?:select|symmetrize|query_order|rerank|xcd|row)_[a-z0-9_]+)"', prose))
    assert quoted <= set(named), sorted(quoted - set(named))


def test_python_binding_covers_header():
    assert set(_declared_functions()) == set(_hip._SIGNATURES)


def test_abi_version_and_struct_layout():
    lib = _hip.load_library()
    assert lib.gt_abi_version() == 1
    # gt_knn_params: 2 x int32, 5 x double, pointer, 2 x int64 = 72 bytes on LP64
    assert ctypes.sizeof(_hip.KnnParams) == 72
    assert _hip.KnnParams.decay.offset == 8 and _hip.KnnParams.bandwidth.offset == 48


def test_make_params_roundtrip():
    p, keep = _hip.Context.make_params(15, 40, 1e-4, None, 1.0, None, "+", None, 0)
    assert (p.knn, p.kernel_symm, p.bandwidth_len, p.knn_max) == (15, 1, 0, -1)
    p, keep = _hip.Context.make_params(5, None, 1e-4, np.arange(3.0), 2.0, 7, "mnn", 0.3, 0.5)
    assert np.isnan(p.decay) and p.kernel_symm == 3 and p.bandwidth_len == 3 and p.knn_max == 7
    assert p.theta == 0.3 and p.anisotropy == 0.5


def test_no_gpu_fails_loudly():
    lib = _hip.load_library()
    if lib.gt_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_hip.HipUnavailableError):
        _hip.Context(0)
    import graphtools_amd

    X = np.random.default_rng(0).standard_normal((50, 8)).astype(np.float32)
    with pytest.raises(_hip.HipUnavailableError):
        graphtools_amd.Graph(X, knn=3, decay=10)


def test_product_never_imports_oracle():
    """the shipped package must not reference the oracle (tests/bench/smoke only)"""
    pkg = os.path.join(ROOT, "graphtools_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_host_place_block_matches_scipy_assembly():
    """gt_host_place_block (host-only helper of the MNN composition): blocks in batch-local ids land in their rows
    of the assembled CSR exactly like the reference's COO concatenation (compared after sorting the rows)."""
    import numpy as np
    from scipy import sparse

    from graphtools_amd import _hip

    rng = np.random.default_rng(0)
    n = 500
    owner = rng.integers(0, 3, size=n)
    index = [np.nonzero(owner == b)[0] for b in range(3)]
    blocks, rows, cols, vals = [], [], [], []
    for i in range(3):
        for j in range(3):
            M = sparse.random(len(index[i]), len(index[j]), density=0.05, random_state=10 * i + j, format="csr")
            scale = rng.uniform(0.2, 1.0, size=len(index[i])) if i != j else None
            blocks.append((i, j, M, scale))
            C = (M if scale is None else sparse.csr_matrix(M.multiply(scale[:, None]))).tocoo()
            rows.append(index[i][C.row])
            cols.append(index[j][C.col])
            vals.append(C.data)
    ref = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    row_len = np.zeros(n, dtype=np.int64)
    for i, _, M, _ in blocks:
        row_len[index[i]] += np.diff(M.indptr)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(row_len, out=indptr[1:])
    data = np.empty(indptr[-1])
    indices = np.empty(indptr[-1], dtype=np.int32)
    cursor = indptr[:-1].copy()
    for i, j, M, scale in blocks:
        _hip.host_place_block(M, index[i], index[j], scale, cursor, indices, data)
    assert np.array_equal(cursor, indptr[1:])
    out = sparse.csr_matrix((data, indices, indptr), shape=(n, n))
    out.sort_indices()
    assert np.array_equal(out.indptr, ref.indptr) and np.array_equal(out.indices, ref.indices)
    assert np.array_equal(out.data, ref.data)
