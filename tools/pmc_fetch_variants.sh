#!/bin/bash
# Development: FETCH_SIZE of the candidate kernel (393216 query rows at N=1e6) for tile-order / query-order variants.
set -u
OUT=$PWD/gpurun_out/pmc_fetch_variants
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, env assignments...
  name=$1; shift
  for kv in "$@"; do export "$kv"; done
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$name -o sel -- python3 tools/gpu_select_ablate.py 1000000 f16x1 0 > $OUT/$name.log 2>&1
  for kv in "$@"; do unset "${kv%%=*}"; done
  echo "$name: $(python3 tools/pmc_summary.py $OUT/$name | tr -d '\n ' | cut -c1-200)"
}
export GT_NQ=393216
run default
run qorder_off GT_QORDER=off
run stride16_nocut GT_SAMP=16:16 GT_SAMP2=0:64
run stride16_nocut_qoff GT_SAMP=16:16 GT_SAMP2=0:64 GT_QORDER=off
run nosamp GT_SAMP=0:0
