"""CPU oracle for the graphtools kNN -> affinity kernel -> diffusion operator path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy/scipy restatement of the
reference algorithm (KrishnaswamyLab/graphtools v2.1.0, non-numba float64 branch)
and of the scikit-learn 1.7.2 / scipy 1.15.3 arithmetic that the reference
delegates to on this path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it - and there only as the
checker, never as the thing measured or shipped.  The product path
(``graphtools_amd``) never imports this package and fails loudly when the HIP
extension is missing.

Parity pinning: every function here is checked (tests/test_oracle_golden.py)
against golden vectors produced by importing the reference itself in the build
container (tools/make_golden.py -> tests/golden/*.npz), and the kNN
restatement additionally against scikit-learn's own ``NearestNeighbors`` when
that package is importable.

Each function cites the reference ``file:line`` it follows (paths relative to the
reference checkout; ``sklearn:`` = the installed scikit-learn 1.7.2 sources).
"""
from .knn import kneighbors, radius_neighbors, row_norms_sq  # noqa: F401
from .kernel import (  # noqa: F401
    apply_anisotropy,
    build_csr_from_neighbors,
    diff_aff,
    diff_op,
    kernel_degree,
    knn_kernel,
    knn_graph,
    symmetrize_kernel,
)
from .exact import (  # noqa: F401
    cross_distances_exact,
    exact_graph,
    exact_graph_rows,
    exact_kernel,
    exact_kernel_to_data,
    pairwise_distances_exact,
)
from .landmark import landmark_extend, landmark_operator, random_landmark_clusters  # noqa: F401
from .mnn import mnn_graph, mnn_kernel  # noqa: F401
