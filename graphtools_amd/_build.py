"""Build libgraphtools_amd.so (HIP, gfx950 only) in-tree with hipcc.

``python -m graphtools_amd._build`` or ``graphtools_amd._build.build()``.  Objects are
compiled in parallel (one hipcc process per translation unit; the MFMA candidate kernel is
compiled once per padded feature count) and linked into ``graphtools_amd/libgraphtools_amd.so``.
Only out-of-date objects are rebuilt.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libgraphtools_amd.so")
ARCH = "gfx950"
# (precision, padded feature count) instantiations; keep in sync with gt_knn_select_dispatch.cpp
SELECT_UNITS = ([(0, dp) for dp in (16, 32, 56, 64, 104, 128)] + [(1, dp) for dp in (16, 32, 48, 64, 80, 96, 112, 128)] +
                [(2, dp) for dp in (16, 32, 48, 64, 80, 96, 112, 128)])

# 128-row-workgroup variants (single-chain float16, DP <= 64) for launches with few query rows (multi-GPU shards)
SELECT_NARROW_UNITS = [(2, dp) for dp in (16, 32, 48, 64)]

# candidate kernels: scores are never NaN (finite data, -inf seeds only on pad rows), so the v_max3 reduction of the
# admission test needs no canonicalising moves
SELECT_FLAGS = ["-fno-honor-nans"]

# GT_BUILD_DEBUG_HOOKS=0 leaves gt_debug.hip (the gt_dbg_* hooks the GPU unit tests of the device primitives call) out of the
# library; the default build - the one the tests run against - has them
DEBUG_HOOKS = os.environ.get("GT_BUILD_DEBUG_HOOKS", "1") != "0"

COMMON_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
                "-ffp-contract=off"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found - the HIP toolchain is required to build graphtools_amd")
    return exe


def _units():
    units = []
    for name in ("gt_api.cpp", "gt_knn.cpp", "gt_knn_shard.cpp", "gt_knn_select_dispatch.cpp", "gt_hostcopy.cpp", "gt_devpool.cpp"):
        units.append((name, name.replace(".cpp", ".o"), ["-x", "hip"]))
    for name in ("gt_prep.hip", "gt_rerank.hip", "gt_sparse.hip", "gt_dense.hip", "gt_landmark.hip", "gt_debug.hip", "gt_order.hip", "gt_sym.hip", "gt_pca.hip", "gt_thin.hip"):
        if name == "gt_debug.hip" and not DEBUG_HOOKS:
            continue   # (the gt_dbg_* development / unit-test entry points: not part of the ABI of include/graphtools_amd.h)
        if os.path.exists(os.path.join(CSRC, name)):
            units.append((name, name.replace(".hip", ".o"), []))
    units.append(("gt_seed.hip", "gt_seed.o", SELECT_FLAGS))
    for prec, dp in SELECT_UNITS:
        units.append(("gt_knn_select.hip", "gt_knn_select_p%d_dp%d.o" % (prec, dp),
                      ["-DGT_SEL_PREC=%d" % prec, "-DGT_SEL_DP=%d" % dp] + SELECT_FLAGS))
    for prec, dp in SELECT_NARROW_UNITS:
        units.append(("gt_knn_select.hip", "gt_knn_select_narrow_p%d_dp%d.o" % (prec, dp),
                      ["-DGT_SEL_PREC=%d" % prec, "-DGT_SEL_DP=%d" % dp, "-DGT_SEL_QT1=1"] + SELECT_FLAGS))
    return units


def _compile(unit):
    src, obj, extra = unit
    obj_path = os.path.join(OBJ, obj)
    stamp = obj_path + ".stamp"
    # every unit depends on all headers; hash headers + its own source + flags
    h = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)):
        if fn.endswith(".h") or fn == src:
            with open(os.path.join(CSRC, fn), "rb") as f:
                h.update(fn.encode())
                h.update(f.read())
    with open(os.path.join(os.path.dirname(HERE), "include", "graphtools_amd.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(COMMON_FLAGS + extra).encode())
    digest = h.hexdigest()
    if os.path.exists(obj_path) and os.path.exists(stamp) and open(stamp).read() == digest:
        return obj_path, False, ""
    cmd = [_hipcc()] + COMMON_FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj_path]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), res.stderr[-8000:]))
    with open(stamp, "w") as f:
        f.write(digest)
    return obj_path, True, res.stderr


def build(verbose=False, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    units = _units()
    jobs = jobs or min(8, os.cpu_count() or 1)
    objs, rebuilt = [], False
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        for obj_path, did, log in ex.map(_compile, units):
            objs.append(obj_path)
            rebuilt |= did
            if verbose and log.strip():
                print(log, file=sys.stderr)
    link_stamp = os.path.join(OBJ, "link.stamp")
    linked = " ".join(sorted(os.path.basename(o) for o in objs))
    if rebuilt or not os.path.exists(LIB) or not os.path.exists(link_stamp) or open(link_stamp).read() != linked:
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), res.stderr[-8000:]))
        with open(link_stamp, "w") as f:
            f.write(linked)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
