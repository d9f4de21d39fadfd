#!/bin/bash
# One counter per pass (combining TCC counters exceeds the hardware's capacity and aborts).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE TCC_HIT_sum TCC_MISS_sum; do
  timeout -k 10 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_f_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-acq > /dev/null 2> gpurun_out/pmc_f_$c.err || { echo "pass $c failed"; tail -3 gpurun_out/pmc_f_$c.err; }
  echo "pass $c done"
done
python3 - <<'PY'
import csv,glob,collections,re
for f in glob.glob("gpurun_out/pmc_f_*/*/*counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m=re.search(r'(\w+_kernel)',r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:20]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in [x for x in agg if x.startswith("trk_corr")]:
        for c,v in agg[k].items(): print(k,c,"%.5g"%(sum(v)/len(v)))
PY
