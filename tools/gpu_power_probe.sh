#!/bin/bash
# Development probe: shader clock / power of GPU 0 sampled while bench.py loops (is the candidate pass power-capped?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( for i in $(seq 1 60); do rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_probe.txt &
SMI=$!
python bench.py --steps 60 --warmup 2 2>&1 | tail -1 | cut -c1-260
wait $SMI
sort gpurun_out/power_probe.txt | uniq -c | sort -rn | head -12
