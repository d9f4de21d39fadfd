"""debug: one closed-loop run per mode (for rocprofv3 --kernel-trace: the gaps between the step kernels)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gnsscorr_loader
gc = gnsscorr_loader.load()
NS, seed, NP = 16368, 20240601, int(os.environ.get("NP", "400"))
rng0 = np.random.default_rng(1)
data = rng0.integers(-60, 61, size=((NP + 4) * NS, 2), dtype=np.int8)
eng = gc.Engine(0)
eng.ring_create(1, 2, data.shape[0]); eng.ring_push_raw(1, data, data.shape[0])
chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in range(1, 33)]
eng.set_channels(chans)
rng = np.random.default_rng(seed)
st0 = [dict(carrfreq=float(rng.uniform(-5000, 5000)), codefreq=c.crate + float(rng.uniform(-2, 2)),
            remcode=float(rng.uniform(0.01, 0.99)), remcarr=float(rng.uniform(0, 6.2)),
            buffloc=int(rng.integers(0, NS))) for c in chans]
for mode, flagsync, cnt0 in (("loop1", 0, 0), ("loop10", 1, 2001)):
    for rep in range(2):
        eng.trk_set_state(st0)
        eng.loop_set([eng.loop_state(i, 200.0 * round(st0[i]["carrfreq"] / 200.0), flagsync=flagsync, synci=(7 * i) % 20, cnt=cnt0) for i in range(32)])
        eng.sync()
        t0 = time.perf_counter()
        eng.trk_run_loop(NP); eng.sync()
        dt = time.perf_counter() - t0
    print(mode, "us per period %.2f  x real time %.1f" % (dt / NP * 1e6, NP * 1e-3 / dt))
