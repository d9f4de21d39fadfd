#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_trk1 -- python3 bench.py --steps 1 --warmup 1 --inner 8 --no-cpu --no-acq --loop-periods 0 > /dev/null 2> gpurun_out/pmc_trk1.err
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_trk2 -- python3 bench.py --steps 1 --warmup 1 --inner 8 --no-cpu --no-acq --loop-periods 0 > /dev/null 2> gpurun_out/pmc_trk2.err
python3 - <<'PY'
import csv,glob,collections,re
for d in ("gpurun_out/pmc_trk1","gpurun_out/pmc_trk2"):
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            m=re.search(r'(\w+_kernel)',r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:20]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in [x for x in agg if x.startswith("trk_corr")]:
            for c,v in agg[k].items(): print(k,c,"%.4g"%(sum(v)/len(v)))
PY
tail -n 3 gpurun_out/pmc_trk1.err
