"""graphtools_amd - MI355X-native kNN / affinity-kernel / diffusion-operator builder with the
``graphtools.Graph(...)`` API surface.  Numerics run in hand-written HIP kernels for gfx950
(``libgraphtools_amd.so``, C ABI in ``include/graphtools_amd.h``); there is no CPU fallback."""
from .api import Graph  # noqa: F401
from .graphs import MNNGraph, MNNLandmarkGraph, TraditionalGraph, kNNGraph, kNNLandmarkGraph  # noqa: F401
from . import graphs  # noqa: F401
from ._hip import release_cached_memory  # noqa: F401

__version__ = "0.1.0"
