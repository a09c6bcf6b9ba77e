"""GPU: spectral landmark front end on the device (gt_thin.hip + graphtools_amd/_spectral.py).

The thin float64 helpers against numpy; the randomized SVD of diff_aff against the exact leading eigenpairs (scipy eigsh)
- A is symmetric positive semi-definite here, so its right singular vectors are eigenvectors; the embedding against the
host product; and the whole landmark graph against the reference's pipeline run on the host: parity is statistical (the
k-means in the middle is RNG- and order-dependent), measured as agreement of the two partitions."""
import warnings

import numpy as np
import pytest
from scipy import sparse
from scipy.sparse.linalg import eigsh

import graphtools_amd
from conftest import make_mix
from graphtools_amd import _hip
from graphtools_amd._spectral import spectral_embedding

pytestmark = pytest.mark.gpu


def test_thin_helpers_match_numpy():
    rng = np.random.default_rng(0)
    ctx = _hip.Context(0)
    for n, k, m in ((5000, 37, 20), (1025, 128, 128), (300, 5, 3)):
        A = rng.standard_normal((n, k))
        v = rng.uniform(0.5, 2.0, n)
        R = rng.standard_normal((k, m))
        a = ctx.dev_alloc(A.nbytes)
        b = ctx.dev_alloc(n * m * 8)
        vd = ctx.dev_alloc(v.nbytes)
        ctx.dev_upload(a, A)
        ctx.dev_upload(vd, v)
        np.testing.assert_allclose(ctx.thin_gram(a, n, k), A.T @ A, rtol=1e-12, atol=1e-10)
        ctx.thin_rmul(a, n, k, R, b)
        out = np.empty((n, m))
        ctx.dev_download(out, b)
        np.testing.assert_allclose(out, A @ R, rtol=1e-12, atol=1e-12)
        ctx.thin_scale_rows(a, n, k, vd, -0.5)
        out = np.empty((n, k))
        ctx.dev_download(out, a)
        np.testing.assert_allclose(out, A / np.sqrt(v)[:, None], rtol=1e-14)
        for p in (a, b, vd):
            ctx.dev_free(p)
    ctx.close()


def test_spectral_embedding_finds_the_leading_eigenpairs_of_diff_aff():
    X = make_mix(8000, 30, 5)     # 4 clusters
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = graphtools_amd.Graph(X, knn=10, decay=20, n_pca=None, verbose=0)
        K, P = G.K, G.P
    n_svd = 12
    G._ensure_device_graph()
    E, S = spectral_embedding(G.hip, X.shape[0], n_svd, random_state=3)
    deg = np.asarray(K.sum(axis=1)).ravel()
    A = sparse.diags(deg ** -0.5) @ K @ sparse.diags(deg ** -0.5)
    w, V = eigsh(A, k=n_svd, which="LA")
    w = w[::-1]
    # the cluster part of the spectrum (eigenvalue 1 per connected cluster) is separated by a gap and comes out exactly;
    # the continuum below it converges like any randomized SVD does on a flat spectrum: from below, within a few per cent
    # (sklearn's own result on this matrix is off by the same amount)
    sep = w > 0.97
    np.testing.assert_allclose(S[sep], w[sep], rtol=2e-6)
    assert np.all(S <= w * (1 + 1e-9)) and np.all(S >= 0.95 * w)
    from sklearn.utils.extmath import randomized_svd

    _, S_sk, _ = randomized_svd(sparse.csr_matrix(A), n_components=n_svd, random_state=3)
    np.testing.assert_allclose(S, S_sk, rtol=2e-2)
    # the embedding is diff_op @ V_dev with V_dev an orthonormal basis of (nearly) the same invariant subspace: compare
    # the subspaces through their projectors on the cluster part of the spectrum (eigenvalues ~ 1, separated by a gap)
    top = int(np.sum(w > 0.97))
    assert top >= 2
    Et = P @ V[:, ::-1][:, :top]
    # E spans diff_op @ span(V_dev); regress the exact leading coordinates on E: they must be reproduced
    coef, res, rank, sv = np.linalg.lstsq(E, Et, rcond=None)
    assert np.linalg.norm(E @ coef - Et) <= 1e-3 * np.linalg.norm(Et)


@pytest.mark.parametrize("n_landmark", [40])
def test_landmark_graph_with_the_device_front_end(monkeypatch, n_landmark):
    """default (spectral) landmarking end to end: a valid operator, and a partition that agrees with the one the
    reference's host pipeline produces from the same kernel as far as two k-means runs agree with each other"""
    from sklearn.metrics import adjusted_rand_score

    from graphtools_amd import base

    X = make_mix(12000, 20, 9)     # 6 clusters
    out = {}
    for backend in ("device", "sklearn"):
        monkeypatch.setattr(base, "SPECTRAL_BACKEND", backend)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            G = graphtools_amd.Graph(X, knn=10, decay=20, n_pca=None, n_landmark=n_landmark, n_svd=20, random_state=1,
                                     verbose=0)
            op = G.landmark_op
        cl = np.asarray(G.clusters)
        assert op.shape[0] == len(np.unique(cl)) <= n_landmark
        np.testing.assert_allclose(op.sum(axis=1), 1.0, atol=1e-10)
        assert G.transitions.shape == (12000, op.shape[0])
        out[backend] = cl
    rng = np.random.default_rng(9)     # the labels make_mix(12000, 20, 9) drew
    c = max(12000 // 2000, 1)
    rng.uniform(-10, 10, (c, 20))
    truth = rng.integers(c, size=12000)
    # both partitions refine the true clusters ...
    for cl in out.values():
        pure = [np.bincount(truth[cl == u]).max() / np.sum(cl == u) for u in np.unique(cl)]
        assert np.mean(pure) > 0.98
    # ... and resemble each other like two k-means runs on one embedding do
    assert adjusted_rand_score(out["device"], out["sklearn"]) > 0.3


def test_device_minibatch_kmeans_against_scikit_learn():
    """graphtools_amd/_kmeans.py: scikit-learn's MiniBatchKMeans algorithm (reference graphs.py:1223-1230) with the
    nearest-centre searches on the device.  Statistical parity, the criterion: same data, same parameters, same seed - the
    final inertia within 2 % of scikit-learn's (and not worse than 5 % above the best of three scikit-learn seeds), no
    empty cluster, planted clusters recovered as pure partitions."""
    from sklearn.cluster import MiniBatchKMeans

    from graphtools_amd._kmeans import DeviceMiniBatchKMeans

    rng = np.random.default_rng(3)
    centres = rng.uniform(-6, 6, (25, 12))
    truth = rng.integers(25, size=30000)
    X = centres[truth] + 0.6 * rng.standard_normal((30000, 12))
    k = 60

    def inertia(C):
        d2 = ((X[:, None, :] - C[None, :, :]) ** 2).sum(-1) if X.shape[0] * C.shape[0] < 4e6 else None
        if d2 is None:
            d2 = (X ** 2).sum(1)[:, None] - 2 * X @ C.T + (C ** 2).sum(1)[None, :]
        return float(d2.min(axis=1).sum())

    ref = {}
    for seed in (0, 1, 2):
        km = MiniBatchKMeans(k, init_size=3 * k, n_init=1, batch_size=2000, random_state=seed).fit(X)
        ref[seed] = inertia(km.cluster_centers_)
    dk = DeviceMiniBatchKMeans(k, init_size=3 * k, batch_size=2000, random_state=0).fit(X)
    got = inertia(dk.cluster_centers_)
    assert abs(dk.inertia_ - got) <= 1e-6 * got                      # the device's own labelling pass agrees
    assert abs(got - ref[0]) <= 0.02 * ref[0], (got, ref)
    assert got <= 1.05 * min(ref.values())
    labels = dk.labels_
    assert len(np.unique(labels)) == k                               # no empty cluster
    pure = [np.bincount(truth[labels == u]).max() / np.sum(labels == u) for u in range(k)]
    assert np.mean(pure) > 0.99
    assert 1 < dk.n_steps_ < (100 * 30000) // 2000                   # stopped by the no-improvement rule
