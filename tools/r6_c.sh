set -x
mkdir -p gpurun_out/r6c
python -m pytest tests/test_gpu_shard_local.py tests/test_gpu_shard_full.py tests/test_gpu_shard_sym.py -q -x --durations=6 > gpurun_out/r6c/shard_tests.log 2>&1; tail -12 gpurun_out/r6c/shard_tests.log
python tools/gpu_shard_local_probe.py 1000000 64 8 manifold gpurun_out/r6c/shard_sim_manifold_world8.json > gpurun_out/r6c/shard_manifold.log 2>&1; tail -1 gpurun_out/r6c/shard_manifold.log | cut -c1-1300
