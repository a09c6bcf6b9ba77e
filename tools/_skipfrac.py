import sys, ctypes
sys.path.insert(0, ".")
import numpy as np
from graphtools_amd import _hip
from bench import make_manifold
n, d = 1000000, 64
X = make_manifold(n, d, 1)
c = _hip.Context(0)
c.set_points(X)
p, keep = c.make_params(15, 40.0, 1e-4, None, 1.0, None, "+", None, 0)
c.graph_build(p)
ng = ((n + 1023) // 1024 * 1024) // 32
fn = c.lib.gt_dbg_fetch_sym
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
zc = np.zeros((ng, 16), dtype=np.float32); zrn = np.zeros((ng, 2), dtype=np.float32)
assert fn(c.h, 14, ng * 16, zc.ctypes.data) == 0 and fn(c.h, 15, ng * 2, zrn.ctypes.data) == 0
rng = np.random.default_rng(0)
a = rng.integers(0, ng - 32, 2000000); b = rng.integers(0, ng - 32, 2000000)
D = np.linalg.norm(zc[a] - zc[b], axis=1)
need = np.maximum(zrn[a, 1], zrn[b, 1])
far = D * (1 - 1e-4) - (zrn[a, 0] + zrn[b, 0]) > need
print("groups", ng, "radius median %.4g need median %.4g centre spread %.4g" % (np.median(zrn[:, 0]), np.median(zrn[:, 1][np.isfinite(zrn[:,1])]), np.median(D)))
print("fraction of random unit pairs skipped: %.3f" % far.mean())
# per tile (4 consecutive sub-tiles vs 8 consecutive query tiles): fraction of tiles with ALL 32 units skipped
