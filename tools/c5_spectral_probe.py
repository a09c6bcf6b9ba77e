import sys, time, warnings
sys.path.insert(0, ".")
import numpy as np
from bench import make_mix
import graphtools_amd
X = make_mix(1000000, 50, 3)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    t = time.perf_counter()
    G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=2000, random_state=1, verbose=0)
    G.K
    t1 = time.perf_counter()
    cl = G.clusters
    t2 = time.perf_counter()
    op = G.landmark_op
    t3 = time.perf_counter()
print("kernel %.2f s, spectral clusters %.2f s, landmark op %.2f s; clusters used %d" % (t1 - t, t2 - t1, t3 - t2, len(np.unique(cl))))
