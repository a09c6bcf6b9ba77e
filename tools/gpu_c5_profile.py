import cProfile, pstats, sys, warnings
sys.path.insert(0, '/root/repo')
import numpy as np, graphtools_amd
from tools.gpu_perf import make_mix
X = make_mix(1000000, 50, 3)
def run():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = graphtools_amd.Graph(X, knn=15, decay=40, n_pca=None, n_landmark=2000, random_landmarking=True, random_state=42, verbose=0)
        op = G.landmark_op
    return G
run()
pr = cProfile.Profile(); pr.enable(); G = run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
