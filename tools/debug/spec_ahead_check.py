import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, gnsscorr_loader
gc = gnsscorr_loader.load()
eng = gc.Engine(0)
n = 16368 * 64
rng = np.random.default_rng(1)
data = rng.integers(-60, 61, size=(n, 2), dtype=np.int8)
eng.ring_create(1, 2, n)
eng.ring_push_raw(1, data, n)
chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in (1, 2, 3, 4)]
eng.set_channels(chans)
eng.trk_set_state([dict(carrfreq=1200.0 * (i + 1), codefreq=c.crate + 0.5, remcode=0.25 * i, remcarr=0.1 * i, buffloc=17 + i) for i, c in enumerate(chans)])
eng.timing(1); eng.timing_reset()
for k in range(6):
    eng.trk_run(8)
eng.sync()
print("spec launches", eng.timing_read("trk_spec"), "plan launches", eng.timing_read("trk_plan"))
