import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_ladder as tl
tl.N = 1000000
fams = ["mix, clusters of unequal scale d=32", "half of the points in one blob d=48", "mix with hubs d=32", "mix with 15 isolated points d=48", "isotropic gauss d=24"]
for f in fams:
    X = tl.FAMILIES[f]()
    for coh in ("1", "0"):
        auto, nnz_a, sym_a, s_a, wall_a = tl._build_ms(X, {"query_order_coherent": coh}, reps=2)
        print("%-40s coherent %s auto %.1f ms (%s) wall %.1f" % (f, coh, auto, "symmetric" if sym_a else "classic", wall_a), flush=True)
    fs, nnz_s, sym_s, s_s, wall_s = tl._build_ms(X, {"select_symmetric": "1"}, reps=1)
    cl, nnz_c, sym_c, s_c, wall_c = tl._build_ms(X, {"select_symmetric": "0"}, reps=1)
    print("%-40s forced symmetric %.1f  classic %.1f   same graph %s" % (f, fs, cl, nnz_s == nnz_c == nnz_a and s_s == s_c == s_a), flush=True)
