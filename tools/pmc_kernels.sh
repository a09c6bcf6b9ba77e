#!/bin/bash
# SQ counter passes (separate --pmc runs per group, MI355X_MICROARCH.md) for the kernels whose names match PATTERNS.
# usage (GPU box, repo root): bash tools/pmc_kernels.sh OUTNAME "pat1|pat2" [probe args...]     env GT_VARIANTS as gpu_ab_probe.py
set -u
OUT=$PWD/gpurun_out/$1
PATS=$2
shift 2
mkdir -p $OUT
export TMPDIR=/tmp
j=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum"; do
  j=$((j+1))
  GT_REPS=2 GT_COMPARE=0 timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$j -o k -- python3 tools/gpu_ab_probe.py "$@" > $OUT/p$j.log 2>&1
done
python3 - "$OUT" "$PATS" <<'PY'
import csv, glob, json, os, sys
root, pats = sys.argv[1], sys.argv[2].split("|")
out = {}
for fn in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(fn) as f:
        for row in csv.DictReader(f):
            for pat in pats:
                if pat in row["Kernel_Name"]:
                    a = out.setdefault(pat, {}).setdefault(row["Counter_Name"], [0.0, 0, 0.0])
                    a[0] += float(row["Counter_Value"]); a[1] += 1
                    a[2] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
res = {k: {c: {"mean": v[0] / v[1], "dispatches": v[1], "kernel_ms_under_pmc": v[2] / v[1] / 1e6} for c, v in cs.items()} for k, cs in out.items()}
json.dump({"note": "rocprofv3 --pmc passes (five counter groups, separate runs); counter values are sums over all SEs as rocprofv3 reports them", "kernels": res}, open(os.path.join(root, "sq_summary.json"), "w"), indent=1)
for k, cs in res.items():
    print(k, "ms under pmc:", round(list(cs.values())[0]["kernel_ms_under_pmc"], 3))
    for c, v in sorted(cs.items()):
        print("   %-28s %14.5g" % (c, v["mean"]))
PY
