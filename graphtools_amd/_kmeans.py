"""Mini-batch k-means with the assignment step on the device.

Reference (graphtools/graphs.py:1223-1230), default landmark mode::

    kmeans = MiniBatchKMeans(self.n_landmark, init_size=3 * self.n_landmark, n_init=1, batch_size=10000, ...)
    self._clusters = kmeans.fit_predict(self.diff_op.dot(VT.T))

scikit-learn's MiniBatchKMeans spends its time in the nearest-centre search of every batch (batch x n_clusters x n_features
multiply-adds) and of the final labelling; everything else is O(batch x n_features) per step.  Here the searches are 1-NN
queries of the device kNN path against the current centres (exact float64 re-rank, ties to the lower index); the algorithm
around them is scikit-learn's own, restated step by step so that the two behave alike:

* initialisation: k-means++ (``sklearn.cluster.kmeans_plusplus``, as the reference's estimator uses) on ``init_size`` rows drawn
  with ``random_state.randint``; a validation set of ``init_size`` rows scores the initialisation like ``_mini_batch_step`` does;
* a step: draw ``batch_size`` rows with ``randint``, assign, move every touched centre to the running mean of everything ever
  assigned to it (``centre = (centre * count + sum of the batch rows) / (count + rows)``);
* reassignment (``reassignment_ratio`` = 0.01): whenever a centre has never been used, or every ``10 * n_clusters`` samples,
  centres whose count is below 1 % of the largest are re-seeded on random rows of the batch;
* stopping: exponentially weighted average of the batch inertia, ``max_no_improvement`` = 10 steps without a new minimum, at
  most ``max_iter`` = 100 epochs.

Parity with scikit-learn is STATISTICAL: the random draws are the same calls in the same order, but the trajectories
separate at the first tie or rounding difference, as two scikit-learn runs on different BLAS builds do.  The criterion tested
(tests/test_gpu_spectral.py): on the same data and seed the final inertia is within 2 % of scikit-learn's, no cluster is empty
where scikit-learn leaves none, and planted clusters are recovered as pure partitions.
"""
import numpy as np
from scipy import sparse

__all__ = ["DeviceMiniBatchKMeans"]


class DeviceMiniBatchKMeans(object):
    def __init__(self, n_clusters, init_size=None, batch_size=10000, random_state=None, max_iter=100,
                 max_no_improvement=10, reassignment_ratio=0.01, device=0):
        self.n_clusters = int(n_clusters)
        self.init_size = init_size
        self.batch_size = int(batch_size)
        self.random_state = random_state
        self.max_iter = int(max_iter)
        self.max_no_improvement = max_no_improvement
        self.reassignment_ratio = float(reassignment_ratio)
        self.device = device

    # ---- device: nearest centre of every row --------------------------------------------------------
    def _assign(self, ctx, centres, rows):
        """(labels int64 [m], squared distances float64 [m]) of ``rows`` against ``centres``"""
        ctx.set_points(np.ascontiguousarray(centres, dtype=np.float64))
        dist, idx, _ = ctx.knn_search(1, Y=np.ascontiguousarray(rows, dtype=np.float64))
        return idx[:, 0], dist[:, 0] ** 2

    def _step(self, ctx, rows, centres, counts, rng, reassign):
        """one mini-batch update in place (scikit-learn ``_mini_batch_step``); returns the batch inertia"""
        labels, d2 = self._assign(ctx, centres, rows)
        m = rows.shape[0]
        k = self.n_clusters
        onehot = sparse.csr_matrix((np.ones(m), (labels, np.arange(m))), shape=(k, m))
        sums = onehot @ rows                                   # [k, features]
        hit = np.asarray(onehot.sum(axis=1)).ravel()
        touched = hit > 0
        new_counts = counts + hit
        centres[touched] = (centres[touched] * counts[touched, None] + sums[touched]) / new_counts[touched, None]
        counts[:] = new_counts
        if reassign and self.reassignment_ratio > 0:
            low = counts < self.reassignment_ratio * counts.max()
            if low.sum() > 0.5 * m:                            # never more than half a batch of new centres
                keep = np.argsort(counts)[int(0.5 * m):]
                low[keep] = False
            n_low = int(low.sum())
            if n_low:
                picks = rng.choice(m, replace=False, size=n_low)
                centres[low] = rows[picks]
                counts[low] = np.min(counts[~low]) if (~low).any() else 1.0
        return float(d2.sum())

    def fit(self, X):
        from sklearn.cluster import kmeans_plusplus
        from sklearn.utils import check_random_state

        from . import _hip

        X = np.ascontiguousarray(X, dtype=np.float64)
        n = X.shape[0]
        k = self.n_clusters
        if n < k:
            raise ValueError("n_samples={} should be >= n_clusters={}.".format(n, k))
        rng = check_random_state(self.random_state)
        init_size = self.init_size if self.init_size is not None else 3 * self.batch_size
        init_size = int(min(max(init_size, k), n))
        batch = int(min(self.batch_size, n))
        ctx = _hip.Context(self.device or 0)
        try:
            valid = X[rng.randint(0, n, init_size)]
            init_rows = X[rng.randint(0, n, init_size)]
            centres, _ = kmeans_plusplus(init_rows, k, random_state=rng)
            centres = np.ascontiguousarray(centres, dtype=np.float64)
            counts = np.zeros(k, dtype=np.float64)
            self._step(ctx, valid, centres, counts, rng, reassign=False)       # (scores and warms the initialisation)
            n_steps = (self.max_iter * n) // batch
            ewa, ewa_min, stale, since_reassign = None, None, 0, 0
            alpha = min(batch * 2.0 / (n + 1), 1.0)
            self.n_steps_ = 0
            for step in range(n_steps):
                rows = X[rng.randint(0, n, batch)]
                since_reassign += batch
                reassign = bool((counts == 0).any() or since_reassign >= 10 * k)
                if reassign:
                    since_reassign = 0
                inertia = self._step(ctx, rows, centres, counts, rng, reassign) / batch
                self.n_steps_ = step + 1
                if step == 0:
                    continue                                   # (the first batch is not representative: as scikit-learn)
                ewa = inertia if ewa is None else ewa * (1 - alpha) + inertia * alpha
                if ewa_min is None or ewa < ewa_min:
                    ewa_min, stale = ewa, 0
                else:
                    stale += 1
                if self.max_no_improvement is not None and stale >= self.max_no_improvement:
                    break
            self.cluster_centers_ = centres
            self._counts = counts
            self.labels_, d2 = self._assign(ctx, centres, X)
            self.inertia_ = float(d2.sum())
        finally:
            ctx.close()
        return self

    def fit_predict(self, X):
        return self.fit(X).labels_
