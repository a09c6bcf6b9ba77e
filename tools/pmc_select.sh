#!/bin/bash
# Development: PMC passes over the candidate kernel (gpu_select_ablate.py, 131072 query rows at N=1e6).
# usage (on the GPU box, from the repo root): bash tools/pmc_select.sh OUTNAME [extra args of gpu_select_ablate.py]
set -u
OUT=$PWD/gpurun_out/$1; shift
ARGS=${*:-1000000 f16 0}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH" \
  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS SQ_CYCLES GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o sel -- python3 tools/gpu_select_ablate.py $ARGS > $OUT/p$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT
