#!/bin/bash
# Round-6 artefacts on the GPU box (run from the repo root): the bench line, rocprofv3 kernel statistics and PMC traffic of
# bench.py, per-rank simulations of the row-sharded build (C3 at world 2 / 4 / 8, gauss / manifold / C5 at world 8) and the
# kernel timeline of one rank.  Summaries are copied into profiles/ by hand.
set -u
OUT=$PWD/gpurun_out/r6_final
mkdir -p $OUT
export TMPDIR=/tmp
PART=${1:-all}
if [ "$PART" = all ] || [ "$PART" = bench ]; then
  timeout 1500 python3 bench.py --steps 20 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench_line.err
fi
if [ "$PART" = all ] || [ "$PART" = prof ]; then
  timeout 1500 bash tools/profile_bench.sh r6_final/prof > $OUT/profile_bench.log 2>&1
fi
if [ "$PART" = all ] || [ "$PART" = shard ]; then
  for w in 8 4 2; do
    GT_REPS=4 timeout 300 python3 tools/gpu_shard_local_probe.py 1000000 64 $w mix $OUT/shard_sim_world$w.json > $OUT/shard_sim_world$w.log 2>&1
  done
  GT_REPS=3 timeout 300 python3 tools/gpu_shard_local_probe.py 1000000 64 8 gauss $OUT/shard_sim_gauss_world8.json > $OUT/shard_sim_gauss_world8.log 2>&1
  GT_REPS=3 timeout 300 python3 tools/gpu_shard_local_probe.py 1000000 64 8 manifold $OUT/shard_sim_manifold_world8.json > $OUT/shard_sim_manifold_world8.log 2>&1
  GT_REPS=3 GT_LANDMARKS=2000 timeout 300 python3 tools/gpu_shard_local_probe.py 1000000 50 8 c5 $OUT/shard_sim_c5_world8.json > $OUT/shard_sim_c5_world8.log 2>&1
  GT_REPS=2 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/tl -o probe -- python3 tools/gpu_shard_local_probe.py 1000000 64 8 mix > $OUT/tl.log 2>&1
  python3 tools/rocpd_timeline.py $(ls $OUT/tl/*.db $OUT/tl/*/*.db 2>/dev/null | head -1) max_abs > $OUT/shard_timeline_world8.txt 2>&1
  rm -rf $OUT/tl
fi
ls -la $OUT $OUT/prof 2>/dev/null | head -40
