#!/usr/bin/env python
"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output) of bench.py."""
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
acc = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(os.path.join(root, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                name = re.sub(r"\(.*", "", name).strip()
                a = acc.setdefault(name, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                a[counter][0] += float(row["Counter_Value"])
                a[counter][1] += 1
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 0`; "
               "counter unit KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced "
               "stream).  Calibrated in round 5 on kernels with known traffic (profiles/r5_fetch_calibration.json): the x2 holds for "
               "coalesced streams AND for gathers (one RDREQ = one 128-byte line per missing slot: the random-gather rate sits at "
               "the HBM limit for 128 B, not 64 B) - it applies to every kernel of this table, the gather-dominated ones "
               "(affinity_kernel, bin_count_kernel, rerank_sym4_kernel) included; WRITE_SIZE is exact for coalesced stores "
               "(scattered 16-byte stores are tallied at 32 B); Infinity-Cache hits are counted, so this is L2<->fabric traffic",
       "builds_in_run": 2,   # the timed step and the step with the H2D inside
       "kernels": {}}
for name, a in sorted(acc.items(), key=lambda kv: -kv[1]["FETCH_SIZE"][0]):
    nf, nw = max(a["FETCH_SIZE"][1], 1), max(a["WRITE_SIZE"][1], 1)
    fk, wk = a["FETCH_SIZE"][0] / nf, a["WRITE_SIZE"][0] / nw
    out["kernels"][name] = {"launches": a["FETCH_SIZE"][1], "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                            "hbm_read_GB_per_launch_corrected_x2": 2 * fk * 1024 / 1e9,
                            "hbm_write_GB_per_launch": wk * 1024 / 1e9}
print(json.dumps(out, indent=1))
