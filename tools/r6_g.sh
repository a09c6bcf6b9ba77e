set -x
mkdir -p gpurun_out/r6g
GT_VARIANTS="select_symmetric=0;;select_sym_two_skip=0" python tools/gpu_ab_probe.py 1000000 64 manifold > gpurun_out/r6g/manifold.log 2>&1
python - <<'P'
import json
for l in open('gpurun_out/r6g/manifold.log'):
    if l.startswith('{'):
        j=json.loads(l); print(j['opts'], j['wall_ms'], {k:j['stage_ms'].get(k) for k in ('sym_prepare','knn_select','sym_cold','rerank')}, j['knn'].get('sym_cold_pairs'), j.get('equal_to_first'))
    else: print(l.strip()[:200])
P
python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_shard_local.py tests/test_gpu_shard_full.py tests/test_gpu_fuzz.py -q -x --durations=5 > gpurun_out/r6g/tests.log 2>&1; tail -9 gpurun_out/r6g/tests.log
