#!/bin/bash
# round 5: the re-rank's shape - variants of gt_rerank.hip (tools/build_variant.py) x rows per wave, on C3; same bits as the first
cd "$(dirname "$0")/.."
O=gpurun_out/r5_rerank
mkdir -p $O
V="rerank_rows_per_wave=1;rerank_rows_per_wave=2;rerank_rows_per_wave=4;rerank_rows_per_wave=8;rerank_rows_per_wave=16"
for v in main $(ls graphtools_amd/_variants/*.so 2>/dev/null); do
  name=$(basename $v .so)
  if [ "$v" = main ]; then unset GRAPHTOOLS_AMD_LIB; else export GRAPHTOOLS_AMD_LIB=$PWD/$v; fi
  GT_REPS=4 GT_VARIANTS="$V" python tools/gpu_ab_probe.py 1000000 64 mix > $O/$name.txt 2>&1
  python - "$name" "$O/$name.txt" <<'PY'
import json, sys
name, f = sys.argv[1], sys.argv[2]
for l in open(f).read().splitlines():
    if l.startswith("{"):
        d = json.loads(l); st = d["stage_ms"]
        print("%-12s %-26s wall %.2f cold %.2f rerank %.3f aff %.2f symm %.2f equal=%s" % (name, ",".join(d["opts"]), d["wall_ms"], st.get("sym_cold", 0), st.get("rerank", 0), st.get("affinity", 0), st.get("symmetrize", 0), d.get("equal_to_first")))
    elif l in ("ALL_EQUAL", "MISMATCH"): print(name, l)
PY
done
