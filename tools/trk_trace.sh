#!/bin/bash
# Shader-clock stamps of sampled trk_corr_ps workgroups (library built with -DGC_PS_TRACE: gnsscorr_ps.h), lane 0 of
# wave 0: where a (channel, period) unit spends its lifetime.
export GNSSCORR_LIB=$PWD/tools/variants/lib_pstrace.so
mkdir -p gpurun_out
python - <<'PY'
import sys, os, ctypes as C, time
import numpy as np
sys.path.insert(0, os.getcwd())
import gnsscorr_loader
gc = gnsscorr_loader.load()
NS, E = 16368, 1000
rng = np.random.default_rng(3)
data = rng.integers(-60, 61, size=((E + 4) * NS, 2), dtype=np.int8)
eng = gc.Engine(0)
eng.ring_create(1, 2, data.shape[0]); eng.ring_push_raw(1, data, data.shape[0])
chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in range(1, 33)]
eng.set_channels(chans)
st0 = [dict(carrfreq=float(rng.uniform(-5000, 5000)), codefreq=c.crate + float(rng.uniform(-2, 2)), remcode=float(rng.uniform(0.01, 0.99)),
            remcarr=float(rng.uniform(0, 6.2)), buffloc=int(rng.integers(0, NS))) for c in chans]
for rep in range(3):
    eng.trk_set_state(st0)
    eng.trk_run(E)
    eng.sync()
t = np.zeros(2048 * 16, dtype=np.uint64)
gc.lib().gnsscorr_debug_ps_trace(C.c_void_p(t.ctypes.data))
x = t.reshape(2048, 16).astype(np.int64)
x = x[(x[:, 10] > x[:, 0]) & (x[:, 0] > 0)]
# stamps of wavefront 0 (gnsscorr_ps.h): 0 entry, 1 constants, 2 tables staged, 11 first round entered, 12 its loads issued,
# 13 piece scan done, 3 mixing starts, 4 mixed,
# 5 scanned, 7 looked up, 9 all its rounds done, 10 reduced and stored
idx = [0, 1, 2, 11, 12, 13, 3, 4, 5, 7, 9, 10]
names = ["unit + channel constants", "tables staged (+barrier)", "accumulators, tap offsets", "first round: record, loads issued", "first round: piece scan", "first round: lane phase", "first round: mixing",
         "first round: scan", "first round: look-ups", "the wavefront's other rounds", "barrier + reduce + store"]
print("sampled workgroups", len(x), " lifetime clocks mean %.0f median %.0f" % ((x[:, 10] - x[:, 0]).mean(), np.median(x[:, 10] - x[:, 0])))
for i, n in enumerate(names):
    d = x[:, idx[i + 1]] - x[:, idx[i]]
    print("  %-46s mean %7.0f median %7.0f p90 %7.0f" % (n, d.mean(), np.median(d), np.percentile(d, 90)))
PY
