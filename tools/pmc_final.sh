#!/bin/bash
# Round profile: SQ counters (MFMA / VALU / LDS / wait cycles) of the three largest kernels of a graph build at N = 1e6
# (seeding launch, re-rank, cold launch), separate --pmc passes per counter group.  usage (GPU box, repo root): bash tools/pmc_final.sh OUTNAME
set -u
OUT=$PWD/gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
j=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM"; do
  j=$((j+1))
  GT_REPS=2 timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$j -o sym -- python3 tools/gpu_sym_probe.py 1000000 64 mix > $OUT/p$j.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
root = sys.argv[1]
pats = {"seeding knn_select_kernel<64, 8, 0, 2>": "8, 0, 2", "rerank_sym_kernel": "rerank_sym", "sym_cold_kernel<64>": "sym_cold"}
out = {}
for fn in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(fn) as f:
        for row in csv.DictReader(f):
            for name, pat in pats.items():
                if pat in row["Kernel_Name"]:
                    a = out.setdefault(name, {}).setdefault(row["Counter_Name"], [0.0, 0, 0.0])
                    a[0] += float(row["Counter_Value"]); a[1] += 1
                    a[2] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
res = {k: {c: {"mean": v[0] / v[1], "dispatches": v[1], "kernel_ms_under_pmc": v[2] / v[1] / 1e6} for c, v in cs.items()} for k, cs in out.items()}
json.dump({"note": "rocprofv3 --pmc passes (three counter groups, separate runs) of tools/gpu_sym_probe.py 1000000 64 mix; counter values are sums over all SEs as rocprofv3 reports them", "kernels": res}, open(os.path.join(root, "sq_summary.json"), "w"), indent=1)
for k, cs in res.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %14.5g" % (c, v["mean"]))
PY
