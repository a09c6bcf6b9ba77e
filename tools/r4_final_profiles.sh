#!/bin/bash
# Round-4 artefacts on the GPU box (run from the repo root): per-rank simulations of the row-sharded build at world 2 / 4 / 8,
# the kernel timeline of one rank, the rocprofv3 kernel statistics and PMC traffic of bench.py, and the bench line itself.
set -u
OUT=$PWD/gpurun_out/r4_final
mkdir -p $OUT
export TMPDIR=/tmp
for w in 8 4 2; do
  GT_REPS=4 timeout 300 python3 tools/gpu_shard_local_probe.py 1000000 64 $w mix $OUT/shard_sim_world$w.json > $OUT/shard_sim_world$w.log 2>&1
done
GT_REPS=3 timeout 300 python3 tools/gpu_shard_local_probe.py 1000000 64 8 gauss $OUT/shard_sim_gauss_world8.json > $OUT/shard_sim_gauss_world8.log 2>&1
GT_REPS=3 timeout 300 python3 tools/gpu_shard_local_probe.py 1000000 64 8 manifold $OUT/shard_sim_manifold_world8.json > $OUT/shard_sim_manifold_world8.log 2>&1
GT_REPS=2 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/tl -o probe -- python3 tools/gpu_shard_local_probe.py 1000000 64 8 mix > $OUT/tl.log 2>&1
python3 tools/rocpd_timeline.py $(ls $OUT/tl/*.db $OUT/tl/*/*.db 2>/dev/null | head -1) max_abs > $OUT/shard_timeline_world8.txt 2>&1
rm -rf $OUT/tl
timeout 1500 bash tools/profile_bench.sh r4_final/prof > $OUT/profile_bench.log 2>&1
timeout 900 python3 bench.py --steps 20 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench_line.err
ls -la $OUT $OUT/prof | head -40
