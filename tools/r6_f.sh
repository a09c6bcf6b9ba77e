for v in main nobar main nobar; do
  if [ $v = main ]; then unset GRAPHTOOLS_AMD_LIB; else export GRAPHTOOLS_AMD_LIB=$PWD/graphtools_amd/_variants/libgt_$v.so; fi
  GT_REPS=3 GT_COMPARE=0 GT_VARIANTS=";select_sym_two_skip=0" python tools/gpu_ab_probe.py 1000000 64 manifold 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', j['opts'], j['wall_ms'], {k:j['stage_ms'].get(k) for k in ('knn_select','sym_cold','rerank')}, j['knn'].get('sym_cold_pairs'))
" | tail -2
done
