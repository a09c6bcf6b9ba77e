#!/bin/bash
# Development: PMC passes over the dense exact-graph kernels (tools/gpu_dense_max.py N d).
set -u
OUT=$PWD/gpurun_out/$1; shift
ARGS=${*:-60000 100}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
  "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o den -- python3 tools/gpu_dense_max.py $ARGS > $OUT/p$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT dense_kernel_tiles
