set -x
mkdir -p gpurun_out/r6b
python -m pytest tests/test_gpu_shard_full.py -q -x -s --durations=5 > gpurun_out/r6b/shard_full.log 2>&1; tail -15 gpurun_out/r6b/shard_full.log
GT_VARIANTS="select_symmetric=0;;select_sym_pca=0" python tools/gpu_ab_probe.py 1000000 64 manifold > gpurun_out/r6b/manifold.log 2>&1
tail -4 gpurun_out/r6b/manifold.log | cut -c1-700
python -m pytest tests/test_gpu_graph.py -k "cosine" tests/test_gpu_dropin.py tests/test_gpu_symmetric.py -q -x -s --durations=8 > gpurun_out/r6b/tests.log 2>&1; tail -25 gpurun_out/r6b/tests.log
