#!/bin/bash
# round 5: the local-frame cold launch against the global-frame one - same bits, stage times (tools/gpu_ab_probe.py)
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/r5_cold_local
mkdir -p $O
V="select_sym_cold_local=0;select_sym_cold_local=1"
GT_VARIANTS="$V" python tools/gpu_ab_probe.py 1000000 64 mix > $O/mix_1e6_d64.txt 2>&1
GT_VARIANTS="$V" python tools/gpu_ab_probe.py 100000 50 mix > $O/mix_1e5_d50.txt 2>&1
GT_VARIANTS="$V" python tools/gpu_ab_probe.py 1000000 64 manifold > $O/manifold_1e6_d64.txt 2>&1
GT_VARIANTS="metric=cosine,select_sym_cold_local=0;metric=cosine,select_sym_cold_local=1" python tools/gpu_ab_probe.py 1000000 64 mix > $O/cosine_1e6_d64.txt 2>&1
GT_VARIANTS="$V" python tools/gpu_ab_probe.py 300000 36 mix > $O/mix_3e5_d36.txt 2>&1
GT_VARIANTS="$V" GT_KNN=5 python tools/gpu_ab_probe.py 1000000 64 mix > $O/mix_1e6_d64_knn5.txt 2>&1
GT_VARIANTS="$V" GT_KNN=30 GT_DECAY=10 python tools/gpu_ab_probe.py 500000 48 mix > $O/mix_5e5_d48_knn30.txt 2>&1
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5_cold_local/*.txt")):
    lines = [l for l in open(f).read().splitlines() if l.startswith("{") or l in ("ALL_EQUAL", "MISMATCH")]
    print(f.split("/")[-1], lines[-1] if lines else "NO RESULT")
    for l in lines[:-1]:
        d = json.loads(l)
        st = d["stage_ms"]
        print("   %-40s wall %.2f cold %.2f rerank %.2f aff %.2f symm %.2f local=%s" % (",".join(d["opts"]), d["wall_ms"], st.get("sym_cold", 0), st.get("rerank", 0), st.get("affinity", 0), st.get("symmetrize", 0), d["knn"].get("sym_cold_local")))
PY
