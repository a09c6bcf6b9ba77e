#!/bin/bash
# Round profile of bench.py on the GPU box (run from the repo root): kernel trace + stats, then separate PMC passes
# for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md: counters in their own runs).  Outputs under gpurun_out/$1.
set -u
TAG=$1
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/trace.log 2>&1
python3 tools/rocpd_summary.py $(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null | head -1) > $OUT/kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $OUT/pmc_$c.log 2>&1
done
python3 tools/pmc_traffic_summary.py $OUT > $OUT/pmc_fetch_write_per_kernel.json
head -12 $OUT/kernel_stats.txt
