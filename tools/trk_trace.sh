#!/bin/bash
# Phase time stamps of sampled trk_corr workgroups (library built with -DGC_TRK_TRACE).
export GNSSCORR_LIB=$PWD/tools/variants/lib_trace.so GNSSCORR_TRACE_OUT=$PWD/gpurun_out/trk_trace.npy
mkdir -p gpurun_out
python bench.py --steps 1 --warmup 1 --no-cpu --no-acq > gpurun_out/trk_trace.json 2> gpurun_out/trk_trace.err || { tail -5 gpurun_out/trk_trace.err; exit 1; }
python - <<'PY'
import numpy as np, json
t=np.load("gpurun_out/trk_trace.npy").astype(np.int64)
ok=(t[:,7]>0)&(t[:,0]>0)
t=t[ok]; print("sampled workgroups", len(t))
d=np.diff(t[:,:8],axis=1)
names=["unit load","data issue","fill","barrier1","main loop","reduce+barrier2","store"]
tot=t[:,7]-t[:,0]
wall=(t[:,9]-t[:,8])
print("cycle-counter ticks per WG: mean %.0f median %.0f ; wall(100MHz) mean %.1f ticks"%(tot.mean(),np.median(tot),wall.mean()))
for i,n in enumerate(names): print("  %-16s mean %8.0f  median %8.0f  p90 %8.0f"%(n,d[:,i].mean(),np.median(d[:,i]),np.percentile(d[:,i],90)))
span=(t[:,9].max()-t[:,8].min())/100e6*1e3
print("kernel span from stamps: %.3f ms"%span)
# concurrency: average number of sampled WGs alive * 31
ev=np.concatenate([np.stack([t[:,8],np.ones(len(t))],1),np.stack([t[:,9],-np.ones(len(t))],1)]); ev=ev[np.argsort(ev[:,0],kind="stable")]
alive=np.cumsum(ev[:,1]); dtv=np.diff(ev[:,0]); print("mean concurrent WGs (x7): %.0f"%((alive[:-1]*dtv).sum()/dtv.sum()*7))


print(json.load(open("gpurun_out/trk_trace.json"))["kernels_ms_per_step"])
PY
