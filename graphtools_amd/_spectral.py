"""Spectral landmark front end with the sparse products on the device.

Reference (graphtools/graphs.py:1215-1230), default mode of ``LandmarkGraph``::

    _, _, VT = randomized_svd(self.diff_aff, n_components=self.n_svd, random_state=self.random_state)
    kmeans = MiniBatchKMeans(self.n_landmark, init_size=3 * self.n_landmark, n_init=1, batch_size=10000, ...)
    self._clusters = kmeans.fit_predict(self.diff_op.dot(VT.T))

``diff_aff = D^-1/2 K D^-1/2`` is an N x N sparse matrix (1e8 non-zeros at N = 1e6): sklearn's randomized SVD is 16 sparse
products with an N x (n_svd + 10) block plus LU / QR factorisations of that block - minutes on the host.  Here the block
stays on the device in float64:

* ``A @ Q`` = scale rows by d^-1/2, ``gt_graph_spmm`` with the graph's own K, scale rows again (A is symmetric, so
  ``A.T @ Q`` is the same product);
* every normalisation (sklearn: LU in the power iterations, QR at the end) is a Cholesky-QR - ``gt_thin_gram`` (k x k Gram
  matrix to the host), Cholesky and triangular inverse there, ``gt_thin_rmul`` - in exact arithmetic the same subspaces;
* the SVD of the projected matrix ``B = Q.T @ A = (A Q).T`` comes from the Gram matrix of ``Z = A Q``:
  ``Z.T Z = U S^2 U.T``, right singular vectors ``V = Z U S^-1``;
* the embedding ``diff_op @ V`` is one more ``gt_graph_spmm`` (with P).

The test matrix is drawn like sklearn's (same generator, same shape), the iteration count follows its rule.  The k-means
that follows is graphtools_amd/_kmeans.py: scikit-learn's MiniBatchKMeans algorithm with every nearest-centre search (the
batches' and the final labelling) on the device kNN path.  Parity with the reference is statistical: the subspace is the
same up to the accuracy of the randomized method, the k-means trajectory is RNG- and rounding-dependent by design.
"""
import numpy as np

__all__ = ["spectral_embedding", "spectral_clusters"]


def _chol_qr(ctx, n, k, src, dst):
    """dst = orthonormal basis of the columns of src (device pointers, n x k float64); returns the pointer that holds it"""
    G = ctx.thin_gram(src, n, k)
    G = 0.5 * (G + G.T)
    try:
        L = np.linalg.cholesky(G)
    except np.linalg.LinAlgError:
        # nearly dependent columns: lift the spectrum by a few ulps of the largest entry, as a shifted Cholesky-QR does
        L = np.linalg.cholesky(G + np.eye(k) * (1e-13 * np.trace(G)))
    R = np.linalg.inv(L.T)            # src @ R has orthonormal columns
    ctx.thin_rmul(src, n, k, R, dst)
    return dst


def spectral_embedding(ctx, n, n_svd, random_state, n_oversamples=10, n_iter="auto"):
    """``diff_op @ V`` for V = the n_svd leading right singular vectors of ``diff_aff`` of the graph ``ctx`` holds;
    returns a host float64 array [n, n_svd]."""
    from sklearn.utils import check_random_state

    from . import _hip

    k = min(n_svd + n_oversamples, n)
    if k > 128:
        raise ValueError("spectral_embedding: n_svd + oversamples must not exceed 128")
    if n_iter == "auto":
        n_iter = 7 if n_svd < 0.1 * n else 4                    # sklearn.utils.extmath.randomized_svd
    rs = check_random_state(random_state)
    Q0 = rs.normal(size=(n, k))                                 # sklearn: random_state.normal(size=(A.shape[1], size))
    nbytes = n * k * 8
    bufs = [ctx.dev_alloc(nbytes) for _ in range(3)]
    deg = ctx.dev_alloc(n * 8)
    try:
        ctx.graph_fetch_vec_device(_hip.VEC_DEGREE, deg)
        ctx.dev_upload(bufs[0], Q0)
        del Q0
        q, y, t = bufs

        def apply_a(src, dst):
            ctx.thin_scale_rows(src, n, k, deg, -0.5)           # (src is scratch from here on)
            ctx.graph_spmm_device(_hip.CSR_K, src, k, dst)
            ctx.thin_scale_rows(dst, n, k, deg, -0.5)

        for _ in range(int(n_iter)):                            # Q <- lu(A Q); Q <- lu(A.T Q)
            apply_a(q, y)
            _chol_qr(ctx, n, k, y, t)
            apply_a(t, y)
            _chol_qr(ctx, n, k, y, q)
        apply_a(q, y)                                           # Q <- qr(A Q)
        _chol_qr(ctx, n, k, y, t)
        apply_a(t, y)                                           # Z = A Q = (Q.T A).T
        G = ctx.thin_gram(y, n, k)
        w, U = np.linalg.eigh(0.5 * (G + G.T))
        order = np.argsort(w)[::-1][:n_svd]
        S = np.sqrt(np.maximum(w[order], 1e-300))
        ctx.thin_rmul(y, n, k, U[:, order] / S[None, :], q)     # V = Z U S^-1   [n, n_svd]
        ctx.graph_spmm_device(_hip.CSR_P, q, n_svd, t)          # diff_op @ V
        E = np.empty((n, n_svd), dtype=np.float64)
        ctx.dev_download(E, t)
        return E, S
    finally:
        for p in bufs + [deg]:
            ctx.dev_free(p)


def spectral_clusters(graph, ctx):
    """The reference's default landmark assignment with the SVD, the embedding product and the k-means searches on the device"""
    from ._kmeans import DeviceMiniBatchKMeans

    n = graph.data_nu.shape[0]
    E, _ = spectral_embedding(ctx, n, graph.n_svd, graph.random_state)
    kmeans = DeviceMiniBatchKMeans(graph.n_landmark, init_size=3 * graph.n_landmark, batch_size=10000,
                                   random_state=graph.random_state, device=getattr(graph, "device", 0) or 0)
    return kmeans.fit_predict(E).astype(np.int64)
