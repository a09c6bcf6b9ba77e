set -x
mkdir -p gpurun_out/r6a
GT_VARIANTS="select_sym_listed=0;select_sym_listed=auto;select_sym_listed=auto,dbg_select=2048" python tools/gpu_ab_probe.py 1000000 64 manifold > gpurun_out/r6a/manifold.log 2>&1
tail -5 gpurun_out/r6a/manifold.log | cut -c1-1500
GT_VARIANTS="select_sym_listed=0;select_sym_listed=auto" python tools/gpu_ab_probe.py 1000000 64 mix > gpurun_out/r6a/mix.log 2>&1
tail -3 gpurun_out/r6a/mix.log | cut -c1-800
GT_VARIANTS="select_sym_listed=0;select_sym_listed=auto;select_sym_listed=1" python tools/gpu_ab_probe.py 200000 64 manifold > gpurun_out/r6a/manifold2e5.log 2>&1
tail -4 gpurun_out/r6a/manifold2e5.log | cut -c1-800
python tools/cosine_f32_probe.py > gpurun_out/r6a/cosine.log 2>&1; cat gpurun_out/r6a/cosine.log
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_ladder.py tests/test_gpu_dropin.py "tests/test_gpu_dense_landmark.py::test_exact_graph_from_points_hands_what_the_search_cannot_hold_to_the_all_pairs_path" -q -x --durations=12 -s > gpurun_out/r6a/tests.log 2>&1; tail -30 gpurun_out/r6a/tests.log
