"""``graphtools_amd.Graph`` - the reference's factory (graphtools/api.py:14-295) over the HIP classes.

Same signature and class-selection rules; combinations that are outside the hot path
(PyGSP inheritance, MNN landmark graphs) raise ``NotImplementedError``.
"""
import warnings

import numpy as np

from . import graphs


def Graph(
    data,
    n_pca=None,
    rank_threshold=None,
    knn=5,
    decay=40,
    bandwidth=None,
    bandwidth_scale=1.0,
    knn_max=None,
    anisotropy=0,
    distance="euclidean",
    thresh=1e-4,
    kernel_symm="+",
    theta=None,
    precomputed=None,
    beta=1,
    sample_idx=None,
    adaptive_k=None,
    n_landmark=None,
    n_svd=100,
    random_landmarking=False,
    n_jobs=-1,
    verbose=False,
    random_state=None,
    graphtype="auto",
    use_pygsp=False,
    initialize=True,
    **kwargs,
):
    """Create a graph built from data (see the reference docstring, graphtools/api.py:43-186)."""
    if sample_idx is not None and len(np.unique(sample_idx)) == 1:
        warnings.warn("Only one unique sample. Not using MNNGraph")
        sample_idx = None
        if graphtype == "mnn":
            graphtype = "auto"
    if graphtype == "auto":
        # reference: api.py:196-212
        if sample_idx is not None:
            graphtype = "mnn"
        elif precomputed is not None:
            graphtype = "exact"
        elif decay is None:
            graphtype = "knn"
        elif (thresh == 0 and knn_max is None) or callable(bandwidth):
            graphtype = "exact"
        else:
            graphtype = "knn"

    if graphtype == "knn":
        if precomputed is not None:
            raise ValueError(
                "kNNGraph does not support precomputed values. Use `graphtype='exact'` or `precomputed=None`"
            )
        if sample_idx is not None:
            raise ValueError(
                "kNNGraph does not support batch correction. Use `graphtype='mnn'` or `sample_idx=None`"
            )
        base = "kNN"
    elif graphtype == "mnn":
        if precomputed is not None:
            raise ValueError(
                "MNNGraph does not support precomputed values. Use `graphtype='exact'` and `sample_idx=None` or "
                "`precomputed=None`"
            )
        base = "MNN"
    elif graphtype == "exact":
        if sample_idx is not None:
            raise ValueError(
                "TraditionalGraph does not support batch correction. Use `graphtype='mnn'` or `sample_idx=None`"
            )
        base = "Traditional"
    else:
        raise ValueError(
            "graphtype '{}' not recognized. Choose from ['knn', 'mnn', 'exact', 'auto']".format(graphtype)
        )
    if use_pygsp:
        raise NotImplementedError("graphtools_amd: PyGSP inheritance is outside the accelerated hot path")

    name = base + ("Landmark" if n_landmark is not None else "") + "Graph"
    try:
        cls = getattr(graphs, name)
    except AttributeError:
        raise RuntimeError("unknown graph classes {}".format(name))

    params = dict(kwargs)
    params.update(
        data=data, n_pca=n_pca, rank_threshold=rank_threshold, knn=knn, decay=decay, bandwidth=bandwidth,
        bandwidth_scale=bandwidth_scale, anisotropy=anisotropy, distance=distance, thresh=thresh,
        kernel_symm=kernel_symm, theta=theta, n_jobs=n_jobs, verbose=verbose, random_state=random_state,
        initialize=initialize,
    )
    if base == "kNN":
        params["knn_max"] = knn_max
    elif base == "MNN":
        # reference: api.py:253-283 passes only what MNNGraph.__init__ and its parents accept
        del params["bandwidth_scale"]
        params.update(sample_idx=sample_idx, beta=beta, adaptive_k=adaptive_k)
    else:
        params["precomputed"] = precomputed
    if n_landmark is not None:
        params.update(n_landmark=n_landmark, n_svd=n_svd, random_landmarking=random_landmarking)
    return cls(**params)
