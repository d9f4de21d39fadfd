#!/bin/bash
# kernel timeline of a few tracking steps (rocprofv3 --kernel-trace), gaps between correlator launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 8 --warmup 2 --no-cpu --no-acq > /dev/null 2> gpurun_out/tl.err
python3 - <<'PY'
import csv,glob,re
f=glob.glob("gpurun_out/tl/*/*kernel_trace.csv")[0]
def short(n):
    m=re.search(r'(\w+_kernel)',n); return m.group(1) if m else n[:30]
ks=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),short(r['Kernel_Name'])) for r in csv.DictReader(open(f)))
idx=[i for i,k in enumerate(ks) if k[2].startswith('trk_corr')]
t0=ks[idx[4]][0]
for k in ks[idx[4]-1:idx[7]+2]: print(f"{(k[0]-t0)/1e3:9.1f} -> {(k[1]-t0)/1e3:9.1f} us ({(k[1]-k[0])/1e3:6.1f})  {k[2]}")
cs=[ks[i] for i in idx]
print("corr->corr gaps us:", [round((cs[i+1][0]-cs[i][1])/1e3,1) for i in range(len(cs)-1)])
PY
