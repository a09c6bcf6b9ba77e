#!/bin/bash
# Development: PMC passes over the symmetric collect kernel (gpu_sym_ablate.py) for one option set.
# usage (GPU box, repo root): bash tools/pmc_sym.sh OUTNAME "opts" [kernel-name pattern]
set -u
OUT=$PWD/gpurun_out/$1
OPTS=${2:-}
PAT=${3:-"8, 3, 2"}
mkdir -p $OUT
export TMPDIR=/tmp
j=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" \
  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_INSTS SQ_CYCLES GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE"; do
  j=$((j+1))
  GT_OPTS="$OPTS" timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$j -o sym -- python3 tools/gpu_sym_ablate.py > $OUT/p$j.log 2>&1
  python3 tools/pmc_summary.py $OUT/p$j "$PAT" | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('%-32s %14.5g   (kernel %.2f ms)'%(k, v['mean'], v['mean_ns']/1e6))
"
done
