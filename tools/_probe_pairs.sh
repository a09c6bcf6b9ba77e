cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pairs
python -m pytest tests/test_gpu_shard_local.py tests/test_gpu_shard_full.py tests/test_gpu_symmetric.py -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d.get('secondary',{}).keys() if isinstance(d.get('secondary'),dict) else '')
"
for w in 8; do
  python tools/gpu_shard_local_probe.py 1000000 64 $w mix gpurun_out/pairs/mix_w$w.json > /dev/null 2>gpurun_out/pairs/mix_w$w.err
  GT_OPTS=query_order_coherent=0 python tools/gpu_shard_local_probe.py 1000000 64 $w mix gpurun_out/pairs/mix_w${w}_incoh.json > /dev/null 2>>gpurun_out/pairs/mix_w$w.err
  GT_PAIRS=0 python tools/gpu_shard_local_probe.py 1000000 64 $w mix gpurun_out/pairs/mix_w${w}_general.json > /dev/null 2>>gpurun_out/pairs/mix_w$w.err
done
python - <<'PY'
import json
for f in ("mix_w8","mix_w8_incoh","mix_w8_general"):
    d=json.load(open("gpurun_out/pairs/%s.json"%f))
    print(f, d["pair_resolved_tail"], d["single_rank_ms"], d["per_rank"]["wall_ms"], d["speedup_before_collectives"])
    print("   ", d["per_rank"]["stage_ms"])
    print("   ", d["single_rank_stage_ms"])
    print("   ", d["collectives"]["all_to_all_triplets_bytes_received_by_rank"], d["collectives"]["all_to_all_triplets_bytes_sent_by_rank"])
PY
