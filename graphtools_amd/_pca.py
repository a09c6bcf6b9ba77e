"""PCA pre-reduction with the tall products on the device.

Reference: ``Data._reduce_data`` (graphtools/base.py:227-294) fits ``sklearn.decomposition.PCA(n_pca,
svd_solver="randomized", random_state=...)`` on the dense data and feeds ``data_pca.transform(data)`` to the graph.
sklearn's randomized solver (``sklearn.utils.extmath.randomized_svd``) is a range finder with power iterations: a
Gaussian test matrix ``Q`` (n_features x (n_components + 10)), ``n_iter`` rounds of ``Q <- normalise(A @ Q)``,
``Q <- normalise(A.T @ Q)`` on the centred matrix ``A``, a thin QR, and the SVD of the projected matrix.

``DevicePCA`` runs the same iteration with the products ``A @ Q`` (n x d by d x k) and ``A.T @ Y`` (d x n by n x k) on
the device (``gt_pca_matmul`` / ``gt_pca_tmatmul``: float32 MFMA, the data matrix never leaves HBM) and the small
factors on the host in float64:

* the test matrix is drawn from the same generator in the same way, and the iteration count follows sklearn's rule;
* the basis is re-orthonormalised on the SMALL side only (QR of the d x k matrix ``A.T @ Y``) - the tall factor ``A @ Q``
  is never factorised: in exact arithmetic every normalisation spans the same subspace as sklearn's LU steps;
* instead of the tall QR + SVD of ``Q.T @ A`` the last step projects on the right: ``Y = A @ Q`` with ``Q`` spanning the
  row space sklearn's projected matrix has (one more half step), the k x k Gram matrix ``Y.T @ Y`` (``gt_pca_gram``,
  float64 accumulation) is diagonalised, ``components_ = (Q @ V).T``, ``singular_values_ = sqrt(eigenvalues)`` and the
  transformed data are ``Y @ V`` - again without leaving the device.

Both are approximations of the same truncated SVD; they agree with each other (and with the exact SVD) to the accuracy
of the randomized method itself - parity with the reference is statistical here, as SURVEY 8f notes, not bit for bit.
float64 data, wide data (n < d) and more than 118 components keep sklearn's own solver.
"""
import numpy as np

__all__ = ["DevicePCA", "device_pca_applies"]

_MAX_THIN = 128     # columns the device products hold (n_components + oversamples)


def device_pca_applies(data, n_components, n_oversamples=10):
    data = np.asarray(data)
    return (data.dtype == np.float32 and data.ndim == 2 and data.shape[0] >= data.shape[1]
            and n_components + n_oversamples <= _MAX_THIN and n_components < data.shape[1])


class DevicePCAUnsuitable(ValueError):
    """the device solver would lose precision on this matrix (see DevicePCA.fit_transform); use scikit-learn"""


class DevicePCA(object):
    """The attributes and methods of ``sklearn.decomposition.PCA`` the reference uses (``components_``, ``mean_``,
    ``singular_values_``, ``explained_variance_``, ``explained_variance_ratio_``, ``noise_variance_``, ``transform``,
    ``inverse_transform``), fitted as described in the module docstring."""

    def __init__(self, n_components, random_state=None, n_oversamples=10, iterated_power="auto", device=0):
        self.n_components = int(n_components)
        self.random_state = random_state
        self.n_oversamples = int(n_oversamples)
        self.iterated_power = iterated_power
        self.device = device
        self.svd_solver = "randomized"

    def fit(self, X):
        self.fit_transform(X)
        return self

    def fit_transform(self, X):
        from scipy import linalg
        from sklearn.utils import check_random_state

        from . import _hip

        X = np.ascontiguousarray(X)
        if not device_pca_applies(X, self.n_components, self.n_oversamples):
            raise ValueError("DevicePCA: float32 data with n_samples >= n_features and at most %d components + oversamples"
                             % _MAX_THIN)
        n, d = X.shape
        k = self.n_components
        kp = min(k + self.n_oversamples, d)
        n_iter = self.iterated_power
        if n_iter == "auto":
            n_iter = 7 if k < 0.1 * min(n, d) else 4       # sklearn.utils.extmath.randomized_svd
        rs = check_random_state(self.random_state)
        # sklearn draws float64 normals and casts them to the data's float32
        Q = rs.normal(size=(d, kp)).astype(np.float32).astype(np.float64)
        import time

        t0 = time.perf_counter()
        phase = {}

        def mark(name):
            nonlocal t0
            t1 = time.perf_counter()
            phase[name] = round(phase.get(name, 0.0) + t1 - t0, 4)
            t0 = t1

        ctx = _hip.Context(self.device)
        try:
            mark("context")
            mean, ssq = ctx.pca_begin(X)
            mark("upload+moments")
            # The products are formed on the uncentred float32 matrix and centred afterwards (Y = X Q - 1 mean^T Q): for
            # data whose column means dwarf its spread (unnormalised counts with a large offset) the float32 products
            # cancel digits scikit-learn keeps by centring first - about offset / spread x 2^-24 relative.  Beyond a
            # ratio of 512 (3e-5) this solver steps aside and the caller takes scikit-learn's.
            spread = float(np.sqrt(max(np.sum(ssq) / max((n - 1) * d, 1), 1e-300)))
            if float(np.max(np.abs(mean))) > 512.0 * spread:
                ctx.pca_end()
                raise DevicePCAUnsuitable("column means up to %.3g against a spread of %.3g per feature"
                                          % (float(np.max(np.abs(mean))), spread))

            def half_steps(Q):
                ctx.pca_matmul(0, Q, mean @ Q, 1)                 # Y = (X - 1 mean^T) Q
                Z, colsum = ctx.pca_tmatmul(1, Q.shape[1])        # X^T Y, 1^T Y
                mark("products")
                Z -= np.outer(mean, colsum)                       # (X - 1 mean^T)^T Y
                Qn, _ = linalg.qr(Z, mode="economic", check_finite=False)
                mark("host qr")
                return Qn

            for _ in range(int(n_iter)):
                Q = half_steps(Q)
            Q = half_steps(Q)                                     # row space of sklearn's projected matrix Q^T A
            ctx.pca_matmul(0, Q, mean @ Q, 1)                     # Y = A Q
            C = ctx.pca_gram(1, kp)                               # Q^T A^T A Q
            mark("products")
            w, V = np.linalg.eigh(0.5 * (C + C.T))
            order = np.argsort(w)[::-1][:k]
            w = np.maximum(w[order], 0.0)
            V = V[:, order]
            comps = (Q @ V).T                                     # [k, d]
            # sklearn: svd_flip(U, Vt, u_based_decision=False) - the largest |entry| of every component is positive
            signs = np.sign(comps[np.arange(k), np.argmax(np.abs(comps), axis=1)])
            signs[signs == 0] = 1.0
            comps *= signs[:, None]
            ctx.pca_matmul(1, V * signs[None, :], None, 2)        # transformed = Y V  (= A components^T)
            mark("host eigh")
            T = ctx.pca_fetch(2, k)
            mark("fetch")
            ctx.pca_end()
        finally:
            ctx.close()
        mark("release")
        self.phase_s_ = phase
        S = np.sqrt(w)
        self.n_samples_, self.n_features_in_ = n, d
        self.n_components_ = k
        self.mean_ = mean.astype(np.float32)
        self.components_ = comps.astype(np.float32)
        self.singular_values_ = S.astype(np.float32)
        self.explained_variance_ = ((S ** 2) / (n - 1)).astype(np.float32)
        total_var = float(np.sum(ssq) / (n - 1))
        self.explained_variance_ratio_ = (self.explained_variance_.astype(np.float64) / total_var).astype(np.float32)
        if k < min(n, d):
            self.noise_variance_ = (total_var - float(np.sum((S ** 2) / (n - 1)))) / (min(n, d) - k)
        else:
            self.noise_variance_ = 0.0
        return T

    def transform(self, X):
        """sklearn: X @ components_.T - mean_ @ components_.T (host; out-of-sample batches are small)"""
        X = np.asarray(X)
        Xt = X @ self.components_.T
        Xt -= np.reshape(self.mean_, (1, -1)) @ self.components_.T
        return Xt

    def inverse_transform(self, X):
        return np.asarray(X) @ self.components_ + self.mean_
