set -x
mkdir -p gpurun_out/r6d
GT_VARIANTS="select_symmetric=0;" python tools/gpu_ab_probe.py 1000000 64 manifold > gpurun_out/r6d/manifold.log 2>&1
tail -3 gpurun_out/r6d/manifold.log | cut -c1-600
python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_shard_local.py tests/test_gpu_shard_full.py -q -x --durations=5 > gpurun_out/r6d/tests.log 2>&1; tail -9 gpurun_out/r6d/tests.log
python tools/gpu_shard_local_probe.py 1000000 64 8 manifold gpurun_out/r6d/shard_sim_manifold_world8.json > gpurun_out/r6d/shard_manifold.log 2>&1; tail -1 gpurun_out/r6d/shard_manifold.log | cut -c1-1400
