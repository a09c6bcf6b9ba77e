#!/bin/bash
# development: where the waves of the tail kernels (and the re-rank) spend their cycles - one SQ pass per option set of GT_PMC_VARIANTS
# (';'-separated), optionally FETCH_SIZE / WRITE_SIZE passes (GT_PMC_BYTES=1); one build of C3 each
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r5_pmc_tail
mkdir -p $O
export TMPDIR=/tmp
IFS=';' read -ra VARS <<< "${GT_PMC_VARIANTS:-symmetrize_pairs=2}"
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD")
if [ "${GT_PMC_BYTES:-0}" = 1 ]; then SETS+=("FETCH_SIZE" "WRITE_SIZE"); fi
for v in "${VARS[@]}"; do
  s=0
  for c in "${SETS[@]}"; do
    d=$O/$(echo "$v" | tr '=,' '__')_$s
    GT_REPS=1 GT_COMPARE=0 GT_VARIANTS="$v" timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o k -- python3 tools/gpu_ab_probe.py ${GT_PROBE_ARGS:-1000000 64 mix} > $d.log 2>&1
    s=$((s+1))
  done
done
python3 - "$O" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
want = ("affinity_kernel<float, false, 1>", "affinity_slots_kernel", "bin_count_kernel", "bin_emit_kernel", "bin_emit_slots_kernel",
        "bin_fill_kernel<256>", "merge_final_kernel", "merge_pairs_slots_kernel<unsigned int>", "rerank_sym4_kernel<1, true, 1>",
        "sym_cold_local_kernel<64>", "sym_thresholds_kernel<float>", "sym_seed_dense_kernel<64>", "assign_cells2_kernel<64>",
        "bound_queue_kernel", "landmark_neighbours_kernel", "_ZN12_GLOBAL__N_126landmark_neighbours_kernelEPKDF16_iiiPiPKjS4_")
for d in sorted(x for x in glob.glob(os.path.join(root, "*")) if os.path.isdir(x)):
    acc = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0]); a[0] += float(row["Counter_Value"]); a[1] += 1
    print(os.path.basename(d))
    for k in want:
        if k not in acc: continue
        c = {n: v[0] / v[1] for n, v in acc[k].items()}
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            print("   %-36s waves %.0f  parked %.0f%%  issue-stall %.0f%%  active %.0f%% (VALU %.0f%%)  VALU insts/wave %.0f  VMEM rd/wave %.1f  wave-cycles/wave %.0f" % (
                k, c["SQ_WAVES"], 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc,
                100 * c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"], 4 * wc / c["SQ_WAVES"]))
        for n in ("FETCH_SIZE", "WRITE_SIZE"):
            if n in c: print("   %-36s %s %.3f GB per launch%s" % (k, n, c[n] * 1024 / 1e9 * (2 if n == "FETCH_SIZE" else 1), " (x2 applied)" if n == "FETCH_SIZE" else ""))
PY
