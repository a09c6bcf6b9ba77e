#!/bin/bash
# round 5: build variants (tools/build_variant.py -> graphtools_amd/_variants/*.so) against the main library on C3: stage times
cd "$(dirname "$0")/.."
O=gpurun_out/r5_variants
mkdir -p $O
for v in main $(ls graphtools_amd/_variants/*.so 2>/dev/null); do
  name=$(basename $v .so)
  if [ "$v" = main ]; then unset GRAPHTOOLS_AMD_LIB; else export GRAPHTOOLS_AMD_LIB=$PWD/$v; fi
  GT_COMPARE=${GT_COMPARE:-1} GT_REPS=5 GT_VARIANTS="${GT_PROBE_OPTS:-select_sym_cold_local=1}" python tools/gpu_ab_probe.py ${GT_PROBE_ARGS:-1000000 64 mix} > $O/$name.txt 2>&1
  python - "$name" "$O/$name.txt" <<'PY'
import json, sys
name, f = sys.argv[1], sys.argv[2]
for l in open(f).read().splitlines():
    if l.startswith("{"):
        d = json.loads(l); st = d["stage_ms"]
        print("%-22s wall %.2f order %.2f prep %.2f seed %.2f bound %.2f cold %.3f rerank %.3f aff %.2f symm %.2f nnz %d" % (name, d["wall_ms"], st.get("query_order", 0), st.get("sym_prepare", 0), st.get("sym_seed", 0), st.get("sym_bound", 0), st.get("sym_cold", 0), st.get("rerank", 0), st.get("affinity", 0), st.get("symmetrize", 0), d["nnz"]))
PY
done
