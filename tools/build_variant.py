#!/usr/bin/env python
"""Development: link a variant of libgraphtools_amd.so whose (precision, DP) candidate kernel (GT_VARIANT_UNIT="p:dp",
default 1:64) is compiled with extra -D flags, for ablation runs (GRAPHTOOLS_AMD_LIB=<path> python tools/...).
usage: build_variant.py NAME -DGT_SEL_EXP=1 [...]   ->  graphtools_amd/_variants/libgt_NAME.so"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtools_amd import _build  # noqa: E402

if __name__ == "__main__":
    name, flags = sys.argv[1], sys.argv[2:]
    _build.build()
    out_dir = os.path.join(_build.HERE, "_variants")
    os.makedirs(out_dir, exist_ok=True)
    unit = os.environ.get("GT_VARIANT_UNIT", "1:64")
    obj = os.path.join(out_dir, "sel_%s.o" % name)
    if ":" in unit:
        prec, dp = [int(v) for v in unit.split(":")]
        replaced = "gt_knn_select_p%d_dp%d.o" % (prec, dp)
        cmd = [_build._hipcc()] + _build.COMMON_FLAGS + ["-DGT_SEL_PREC=%d" % prec, "-DGT_SEL_DP=%d" % dp] + _build.SELECT_FLAGS + flags + [
            "-c", os.path.join(_build.CSRC, "gt_knn_select.hip"), "-o", obj]
    else:   # GT_VARIANT_UNIT=<source file>, e.g. gt_seed.hip: that translation unit with the extra flags
        replaced = unit.replace(".hip", ".o").replace(".cpp", ".o")
        extra = [e for s_, o_, e in _build._units() if o_ == replaced][0]
        cmd = [_build._hipcc()] + _build.COMMON_FLAGS + extra + flags + ["-c", os.path.join(_build.CSRC, unit), "-o", obj]
    subprocess.run(cmd, check=True)
    objs = []
    for src, o, extra in _build._units():
        objs.append(obj if o == replaced else os.path.join(_build.OBJ, o))
    lib = os.path.join(out_dir, "libgt_%s.so" % name)
    subprocess.run([_build._hipcc(), "--offload-arch=" + _build.ARCH, "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)
