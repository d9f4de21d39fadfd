"""debug: which NCO's brackets miss the exact starts, per batch (library variant lib_missdbg.so: stats[7] = code starts
outside their bracket, stats[6] = carrier starts outside theirs), one channel at a time."""
import ctypes as C, json, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gnsscorr_loader
gc = gnsscorr_loader.load()
nepoch, nbatch = 500, 3
nsamples = 16368 * (nepoch * nbatch + 12)
rng = np.random.default_rng(515)
data = rng.integers(-60, 61, size=(nsamples, 2), dtype=np.int8)
allchans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in range(1, 17)]
allstates = [dict(carrfreq=float(rng.uniform(-9000, 9000)), codefreq=c.crate + float(rng.uniform(-6, 6)),
                  remcode=float(rng.uniform(0.01, 0.99)), remcarr=float(rng.uniform(0, 6.2)), buffloc=5 + 700 * i)
             for i, c in enumerate(allchans)]
allstates[0].update(carrfreq=2200.0, codefreq=allchans[0].crate, remcode=0.0, remcarr=0.0)
for ch in range(16):
    eng = gc.Engine(0)
    eng.ring_create(1, 2, nsamples)
    eng.ring_push_raw(1, data, nsamples)
    eng.set_channels([allchans[ch]])
    eng.trk_set_state([allstates[ch]])
    stats = np.zeros(8, dtype=np.uint64)
    gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
    per = []
    for b in range(nbatch):
        eng.trk_run(nepoch)
        eng.sync()
        gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
        per.append((int(stats[7]), int(stats[6]), int(stats[1] + stats[2]), int(stats[4] + stats[5])))
    print("ch %2d carr %8.1f code %+6.2f: per batch (code miss, carrier miss, code slow, carrier slow) %s" % (
        ch, allstates[ch]["carrfreq"], allstates[ch]["codefreq"] - allchans[ch].crate, per))
    eng.close()
