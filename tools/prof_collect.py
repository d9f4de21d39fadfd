#!/usr/bin/env python3
"""Copies the summaries of a tools/prof.sh run (gpurun_out/prof_<tag>_*) into profiles/<round>/<tag>_*."""
import csv, glob, json, re, sys, collections, shutil, os
tag = sys.argv[1]; rnd = sys.argv[2] if len(sys.argv) > 2 else "r1"
out = f"profiles/{rnd}"; os.makedirs(out, exist_ok=True)
st = glob.glob(f"gpurun_out/prof_{tag}_stats/*/*kernel_stats.csv")[0]
shutil.copy(st, f"{out}/{tag}_kernel_stats.csv")
shutil.copy(f"gpurun_out/prof_{tag}_bench.json", f"{out}/{tag}_bench_under_rocprof.json")
res = {}
for c in ("fetch", "write"):
    f = glob.glob(f"gpurun_out/prof_{tag}_{c}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        agg[k].append(float(r["Counter_Value"]))
    res[c.upper() + "_SIZE_KB_per_dispatch"] = {k: sum(v) / len(v) for k, v in agg.items() if k.startswith(("trk_", "acq_"))}
json.dump(res, open(f"{out}/{tag}_pmc_fetch_write.json", "w"), indent=1)
print(json.dumps(res, indent=1))
for r in list(csv.DictReader(open(st)))[:8]:
    m = re.search(r"(\w+_kernel)", r["Name"]); print((m.group(1) if m else r["Name"][:30]), r["Calls"], float(r["AverageNs"]) / 1e3, "us")
