set -x
mkdir -p gpurun_out/r6e
for v in old main w3 old main w3; do
  if [ $v = main ]; then unset GRAPHTOOLS_AMD_LIB; else export GRAPHTOOLS_AMD_LIB=$PWD/graphtools_amd/_variants/libgt_$v.so; fi
  GT_REPS=6 GT_VARIANTS=";" python tools/gpu_ab_probe.py 1000000 64 mix 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', j['wall_ms'], {k:j['stage_ms'][k] for k in ('sym_cold','rerank','sym_bound')}, j.get('equal_to_first'))
    else: print('$v', l.strip())
" | tail -3
done
unset GRAPHTOOLS_AMD_LIB
python -m pytest tests/test_gpu_full_reference.py tests/test_gpu_shard_full.py -q -x 2>&1 | tail -3
