#!/bin/bash
# round 5: calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on known access patterns (tools/micro/fetch_calib.hip)
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r5_fetch_calib
mkdir -p $O
export TMPDIR=/tmp
B=$PWD/tools/micro/fetch_calib
$B 8 3 > $O/rates.json 2> $O/rates.err
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$name -o k -- $B 8 1 > $O/pmc_$name.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, json, os, sys
root = sys.argv[1]
acc = {}
for fn in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(fn) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].replace("void ", "").split("(")[0].strip()
            acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
rates = json.load(open(os.path.join(root, "rates.json")))
out = {"rates": rates, "counters_per_launch": {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}}
for k, cs in out["counters_per_launch"].items():
    r = rates["kernels"].get(k)
    if not r: continue
    units = r["distinct_slots_or_units"]
    if "FETCH_SIZE" in cs: cs["FETCH_SIZE_bytes_per_unit"] = cs["FETCH_SIZE"] * 1024 / units
    if "WRITE_SIZE" in cs: cs["WRITE_SIZE_bytes_per_unit"] = cs["WRITE_SIZE"] * 1024 / units
    if "TCC_EA0_RDREQ_sum" in cs: cs["RDREQ_per_unit"] = cs["TCC_EA0_RDREQ_sum"] / units
    if "TCC_EA0_WRREQ_sum" in cs: cs["WRREQ_per_unit"] = cs["TCC_EA0_WRREQ_sum"] / units
json.dump(out, open(os.path.join(root, "fetch_calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
