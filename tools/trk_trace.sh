#!/bin/bash
# Phase time stamps of sampled trk_corr workgroups (library built with -DGC_TRK_TRACE): lane 0 of
# wave 0 and of wave 3, round 1 of the period.
export GNSSCORR_LIB=$PWD/tools/variants/lib_trace.so GNSSCORR_TRACE_OUT=$PWD/gpurun_out/trk_trace.npy
mkdir -p gpurun_out
python bench.py --steps 1 --warmup 1 --no-cpu --no-acq > gpurun_out/trk_trace.json 2> gpurun_out/trk_trace.err || { tail -5 gpurun_out/trk_trace.err; exit 1; }
python - <<'PY'
import numpy as np, json
t=np.load("gpurun_out/trk_trace.npy").astype(np.int64)
for wv,name in ((0,"wave 0"),(1,"wave 3")):
    x=t[wv::2]; x=x[(x[:,7]>0)&(x[:,0]>0)&(x[:,2]>0)]
    print(name,"sampled",len(x),"WG lifetime cycles mean %.0f"%(x[:,7]-x[:,0]).mean(), "wall us %.2f"%((x[:,9]-x[:,8]).mean()/100))
    seq=[("prologue (start->round0)",0,10),("round0",10,2),("A (mix+scan+atomics)",2,3),("barrier1",3,4),("lbase+barrier2",4,5),("B look-ups",5,6),("barrier3",6,1),("rounds 2..",1,11),("reduce+store",11,7)]
    for n,a,b in seq:
        d=x[:,b]-x[:,a]; print("  %-24s mean %7.0f median %7.0f p90 %7.0f"%(n,d.mean(),np.median(d),np.percentile(d,90)))
print(json.load(open("gpurun_out/trk_trace.json"))["kernels_ms_per_step"])
PY
