"""debug: the planner's claims rows for the long-batch test configuration: how many periods got a bracket, per channel
and NCO, and why not (the exact starts come from the plan the batch produced)."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gnsscorr_loader
gc = gnsscorr_loader.load()
nepoch = int(os.environ.get("NEPOCH", "400"))
NS = 16368
nsamples = NS * (nepoch + 12)
rng = np.random.default_rng(909)
data = rng.integers(-60, 61, size=(nsamples, 2), dtype=np.int8)
eng = gc.Engine(0)
eng.ring_create(1, 2, nsamples); eng.ring_push_raw(1, data, nsamples)
if os.environ.get("CONFIG") == "wide":      # test_planner_chain_wide_correlator_spacing
    prns = [4, 17, 25, 32]
    chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=6, corrd=3, corrp=6) for p in prns]
    eng.set_channels(chans)
    rng = np.random.default_rng(1213)
    rng.integers(-60, 61, size=(nsamples, 2), dtype=np.int8)
    states = []
    for i, c in enumerate(chans):
        states.append(dict(carrfreq=0.0 + (rng.uniform(-5000, 5000) if i else 2200.0), codefreq=c.crate + (rng.uniform(-3, 3) if i else 0.0),
                           remcode=(rng.uniform(0.01, 0.99) if i > 1 else 0.0), remcarr=rng.uniform(0, 6.2) if i else 0.0,
                           buffloc=40 + 1000 * i + (i % 3)))
else:
    prns = [1, 6, 14, 23, 31, 9, 18, 27]
    chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in prns]
    eng.set_channels(chans)
    rng = np.random.default_rng(910)
    states = []
    for i, c in enumerate(chans):
        states.append(dict(carrfreq=float(rng.uniform(-9000, 9000)) * (-1 if i % 2 else 1), codefreq=c.crate + float(rng.uniform(-8, 8)),
                           remcode=float(rng.uniform(0.01, 0.99)), remcarr=float(rng.uniform(0, 6.2)), buffloc=3 + 1000 * i))
eng.trk_set_state(states)
nb = int(os.environ.get("NBATCH", "2"))
for b in range(nb):
    eng.trk_run(nepoch)
    eng.sync()
    L = gc.lib()
    L.gnsscorr_debug_spec_rows.restype = C.c_int
    ROW = 24
    buf = np.zeros(2 * len(chans) * nepoch * ROW, dtype=np.int32)
    n = L.gnsscorr_debug_spec_rows(C.c_void_p(eng.h.value if hasattr(eng.h, "value") else eng.h), C.c_void_p(buf.ctypes.data), C.c_int(buf.size))
    assert n == len(chans) * nepoch, n
    rows = buf.reshape(2, len(chans), nepoch, ROW)
    lohi = rows[:, :, :, 20:24].copy().view(np.float64)          # (2, nch, nepoch, 2)
    print("batch", b)
    for ch in range(len(chans)):
        tc, tk = rows[0, ch, :, 0], rows[1, ch, :, 0]
        print("  ch %d carrfreq %9.1f: code tag1 %4d tag0 %4d | carrier tag1 %4d tag2 %4d tag0 %4d  first tag0 periods %s  lo[0..2] %s" % (
            ch, states[ch]["carrfreq"], (tc == 1).sum(), (tc == 0).sum(), (tk == 1).sum(), (tk == 2).sum(), (tk == 0).sum(),
            np.nonzero(tk == 0)[0][:6].tolist(), lohi[1, ch, :3, 0].tolist()))
        print("      code lo[0..2] %s  n[0..2] %s  codefreq %.6f remcode %.6f" % (lohi[0, ch, :3, 0].tolist(), rows[0, ch, :3, 18].tolist(), states[ch]["codefreq"], states[ch]["remcode"]))
    stats = np.zeros(8, dtype=np.uint64)
    L.gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
    print("  stats", stats.tolist())
