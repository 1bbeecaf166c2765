"""Group rocprofv3 --pmc counter CSVs by (kernel, grid size): mean counter value per launch over the later half of the
launches of each group.  usage: pmc_by_kernel.py out.json pass_dir [pass_dir ...] [--match REGEX]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

args = sys.argv[1:]
match = None
if "--match" in args:
    i = args.index("--match")
    match = re.compile(args[i + 1])
    del args[i:i + 2]
out_path, dirs = args[0], args[1:]
vals = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if match and not match.search(name):
                continue
            short = re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", name)[:90]
            key = f"{short} grid={row.get('Grid_Size', '?')}"
            vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, counters in vals.items():
    r = {}
    for c, v in counters.items():
        half = v[len(v) // 2:]
        r[c] = sum(half) / len(half)
        r["launches_counted"] = len(half)
    if r.get("SQ_BUSY_CU_CYCLES"):
        r["mfma_busy_per_simd"] = round(r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / r["SQ_BUSY_CU_CYCLES"] / 4, 4)
    if r.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in r:
                r[c + "_frac_of_wave_cycles"] = round(r[c] / r["SQ_WAVE_CYCLES"], 4)
    res[k] = r
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: {c: (round(v, 4) if isinstance(v, float) else v) for c, v in r.items()} for k, r in res.items()}, indent=1))
