#!/bin/bash
# instruction counts of trk_corr_ps per launch with phases switched off (GNSSCORR_TRK_ABLATE: 1 = no phase B, 2 = no phase A)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 0 1 2 3 8; do
  GNSSCORR_TRK_ABLATE=$a timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_abl$a -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-acq --loop-periods 0 > /dev/null 2> gpurun_out/pmc_abl$a.err
done
python3 - <<'PY'
import csv,glob,collections,re
for a in (0, 1, 2, 3, 8):
    for f in glob.glob("gpurun_out/pmc_abl%d/*/*counter_collection.csv" % a):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "trk_corr_ps" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("ablate", a, {c: "%.4g" % (sum(v)/len(v)) for c,v in agg.items()})
PY
