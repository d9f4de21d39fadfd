"""debug: the planner chain alone on the chip (no correlator beside it): per-launch time of trk_plan from HIP events."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gnsscorr_loader
gc = gnsscorr_loader.load()
NS, E = 16368, 1000
rng = np.random.default_rng(20240601)
DTYPE, F_IF = int(os.environ.get("DTYPE", "2")), float(os.environ.get("F_IF", "0"))
data = np.random.default_rng(3).integers(-60, 61, size=((E + 4) * NS, 2) if DTYPE == 2 else ((E + 4) * NS,), dtype=np.int8)
eng = gc.Engine(0)
eng.ring_create(1, DTYPE, data.shape[0]); eng.ring_push_raw(1, data, data.shape[0])
NCH = int(os.environ.get('NCH', '32'))
chans = [gc.Channel(p, dtype=DTYPE, f_if=F_IF, corrn=2, corrd=3, corrp=3) for p in range(1, NCH + 1)]
eng.set_channels(chans)
st0 = [dict(carrfreq=F_IF + float(rng.uniform(-5000, 5000)), codefreq=c.crate + float(rng.uniform(-2, 2)), remcode=float(rng.uniform(0.01, 0.99)),
            remcarr=float(rng.uniform(0, 6.2)), buffloc=int(rng.integers(0, NS))) for c in chans]
eng.timing(1)
for rep in range(4):
    eng.trk_set_state(st0)          # (a touched state: no look-ahead, the plan runs before the correlator, alone)
    eng.timing_reset()
    eng.trk_run(E)
    eng.sync()
    print(NCH, {k: round(eng.timing_read(k)[0] / max(eng.timing_read(k)[1], 1), 4) for k in ("trk_spec", "trk_plan", "trk_expand", "trk_edges", "trk_corr")})
if hasattr(gc.lib(), "gnsscorr_debug_plan_prof"):
    import ctypes
    pp = np.zeros(64 * 16, dtype=np.uint64)
    gc.lib().gnsscorr_debug_plan_prof(ctypes.c_void_p(pp.ctypes.data))
    pp = pp.reshape(64, 2, 8)[:NCH].astype(np.float64) / 4
    for ch in range(min(NCH, 4)):
        print("  ch %d carrfreq %.1f code loop %.0f slow %.0f | carrier loop %.0f slow %.0f (%.1f periods) waiting for n %.0f; per period: top %.0f step %.0f" % (
            ch, st0[ch]["carrfreq"], pp[ch, 0, 0], pp[ch, 0, 1], pp[ch, 1, 0], pp[ch, 1, 1], pp[ch, 1, 4], pp[ch, 1, 3], pp[ch, 1, 5] / 1000, pp[ch, 1, 6] / 1000))
