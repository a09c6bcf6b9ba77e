"""GPU: PCA pre-reduction with the tall products on the device (gt_pca.hip + graphtools_amd/_pca.py).

The three device products are checked against float64 numpy (float32 MFMA arithmetic: 1e-5 relative to the operand
norms); the fitted model against the exact SVD and against sklearn's PCA(svd_solver="randomized") with the same seed.
Parity with the reference is statistical for this step (both sides are randomized approximations of one truncated SVD):
singular values within 1e-3 relative, components and transformed data equal up to that accuracy where the spectrum has
gaps, and the kNN graph built on the reduced data shares > 99 % of its edges with the one built on sklearn's reduction."""
import numpy as np
import pytest

import graphtools_amd
from graphtools_amd import _hip
from graphtools_amd._pca import DevicePCA

pytestmark = pytest.mark.gpu


def _low_rank(n, d, r, seed, noise=0.05):
    rng = np.random.default_rng(seed)
    sv = 10.0 * 0.8 ** np.arange(r)
    U = rng.standard_normal((n, r))
    V = np.linalg.qr(rng.standard_normal((d, r)))[0]
    X = (U * sv) @ V.T + noise * rng.standard_normal((n, d)) + rng.uniform(-3, 3, d)
    return X.astype(np.float32)


@pytest.mark.parametrize("n,d,k", [(5000, 300, 37), (4097, 130, 128), (3000, 67, 5), (2500, 1029, 64), (129, 64, 16)])
def test_device_products_match_numpy(n, d, k):
    rng = np.random.default_rng(n + d + k)
    X = rng.standard_normal((n, d)).astype(np.float32) + 2.0
    W = rng.standard_normal((d, k))
    ctx = _hip.Context(0)
    mean, ssq = ctx.pca_begin(X)
    X64 = X.astype(np.float64)
    np.testing.assert_allclose(mean, X64.mean(axis=0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ssq, ((X64 - X64.mean(axis=0)) ** 2).sum(axis=0), rtol=1e-9)
    sub = mean @ W
    ctx.pca_matmul(0, W, sub, 1)
    Y = ctx.pca_fetch(1, k)
    W32 = W.astype(np.float32).astype(np.float64)
    Yref = X64 @ W32 - sub.astype(np.float32).astype(np.float64)
    scale = np.linalg.norm(X64, axis=1)[:, None] * np.linalg.norm(W32, axis=0)[None, :]
    assert np.max(np.abs(Y - Yref) / scale) < 1e-5
    Z, colsum = ctx.pca_tmatmul(1, k)
    Y64 = Y.astype(np.float64)
    np.testing.assert_allclose(colsum, Y64.sum(axis=0), rtol=1e-9, atol=1e-6)
    Zref = X64.T @ Y64
    zscale = np.linalg.norm(X64, axis=0)[:, None] * np.linalg.norm(Y64, axis=0)[None, :]
    assert np.max(np.abs(Z - Zref) / zscale) < 1e-5
    C = ctx.pca_gram(1, k)
    np.testing.assert_allclose(C, Y64.T @ Y64, rtol=1e-10, atol=1e-8)
    # thin @ small
    R = rng.standard_normal((k, min(k, 9)))
    ctx.pca_matmul(1, R, None, 2)
    T = ctx.pca_fetch(2, R.shape[1])
    Tref = Y64 @ R.astype(np.float32).astype(np.float64)
    tscale = np.linalg.norm(Y64, axis=1)[:, None] * np.linalg.norm(R, axis=0)[None, :] + 1e-30
    assert np.max(np.abs(T - Tref) / tscale) < 1e-5
    ctx.pca_end()
    ctx.close()


def test_device_pca_matches_exact_svd_and_sklearn():
    from sklearn.decomposition import PCA

    X = _low_rank(20000, 200, 30, 0)
    k = 20
    dev = DevicePCA(k, random_state=42)
    T = dev.fit_transform(X)
    Xc = X.astype(np.float64) - X.astype(np.float64).mean(axis=0)
    U, S, Vt = np.linalg.svd(Xc, full_matrices=False)
    np.testing.assert_allclose(dev.singular_values_, S[:k], rtol=1e-3)
    np.testing.assert_allclose(dev.mean_, X.astype(np.float64).mean(axis=0), rtol=1e-5, atol=1e-6)
    # components up to sign (the spectrum 0.8^i has gaps), orthonormal rows
    cos = np.abs(np.sum(dev.components_.astype(np.float64) * Vt[:k], axis=1))
    assert cos.min() > 1 - 1e-4
    G = dev.components_.astype(np.float64) @ dev.components_.astype(np.float64).T
    np.testing.assert_allclose(G, np.eye(k), atol=1e-5)
    # transformed data = centred data times components
    np.testing.assert_allclose(T, Xc @ dev.components_.astype(np.float64).T, rtol=0, atol=2e-3 * S[0] / np.sqrt(len(X)))
    np.testing.assert_allclose(dev.transform(X[:50]), T[:50], rtol=0, atol=1e-3)
    np.testing.assert_allclose(dev.inverse_transform(T[:50]), X[:50], rtol=0, atol=1.0)
    sk = PCA(k, svd_solver="randomized", random_state=42).fit(X)
    np.testing.assert_allclose(dev.singular_values_, sk.singular_values_, rtol=1e-3)
    np.testing.assert_allclose(dev.explained_variance_, sk.explained_variance_, rtol=2e-3)
    np.testing.assert_allclose(dev.explained_variance_ratio_, sk.explained_variance_ratio_, rtol=2e-3)
    np.testing.assert_allclose(dev.noise_variance_, sk.noise_variance_, rtol=2e-2)
    # same sign convention as sklearn (largest |entry| of a component positive)
    assert np.all(np.sum(dev.components_ * sk.components_, axis=1) > 1 - 1e-3)
    np.testing.assert_allclose(T, sk.transform(X), rtol=0, atol=5e-3 * S[0] / np.sqrt(len(X)))


def test_graph_on_the_device_reduction_is_the_graph_on_sklearns_reduction(monkeypatch):
    """Graph(X, n_pca=...) end to end: > 99 % of the kNN kernel's entries are shared between the two reductions"""
    from graphtools_amd import base

    X = _low_rank(12000, 150, 25, 3, noise=0.02)
    out = {}
    for backend in ("device", "sklearn"):
        monkeypatch.setattr(base, "PCA_BACKEND", backend)
        G = graphtools_amd.Graph(X, n_pca=15, knn=10, decay=20, random_state=7, verbose=0)
        assert type(G.data_pca).__name__ == ("DevicePCA" if backend == "device" else "PCA")
        assert G.data_nu.shape == (12000, 15)
        out[backend] = G.K
    A, B = out["device"], out["sklearn"]
    common = A.multiply(B != 0).nnz
    assert common > 0.99 * max(A.nnz, B.nnz)
    both = A.multiply(B != 0), B.multiply(A != 0)
    assert np.max(np.abs(both[0].data - both[1].data)) < 0.05


def test_large_offsets_take_the_sklearn_solver():
    """column means far beyond the spread: the device solver (centring after its float32 products) steps aside"""
    from graphtools_amd import _pca

    rng = np.random.default_rng(5)
    X = (rng.standard_normal((20000, 64)) + 5000.0).astype(np.float32)
    with pytest.raises(_pca.DevicePCAUnsuitable):
        _pca.DevicePCA(10, random_state=0).fit_transform(X)
    import graphtools_amd.base as gb

    old = gb.PCA_BACKEND
    gb.PCA_BACKEND = "device"
    try:
        import graphtools_amd

        G = graphtools_amd.Graph(X, n_pca=10, knn=5, decay=10, verbose=0, random_state=0)
        from sklearn.decomposition import PCA

        assert isinstance(G.data_pca, PCA)
    finally:
        gb.PCA_BACKEND = old
