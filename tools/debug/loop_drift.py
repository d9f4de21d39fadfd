"""debug: closed loop, bench configuration -- first period at which any logged quantity differs bitwise from the
oracle's, per channel, and the values around it (hex)."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gnsscorr_loader
gc = gnsscorr_loader.load()
import importlib, oracle as orc
synth = importlib.import_module("erlangnetwork_gnsslib_sdr_amd.synth")
NSAMP, seed = 16368, 20240601
nper = int(sys.argv[1]) if len(sys.argv) > 1 else 350
prns = list(range(1, 33))
codes = {p: gc.gencode(p, gc.CTYPE_L1CA) for p in prns}
sats = synth.default_sats(prns, seed=seed)
data = synth.make_if(codes, (nper + 4) * NSAMP, f_sf=16.368e6, f_if=0.0, dtype=2, sats=sats, seed=seed)
ns_ = data.shape[0]
eng = gc.Engine(0)
eng.ring_create(1, 2, ns_); eng.ring_push_raw(1, data, ns_)
chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in prns]
eng.set_channels(chans)
rng = np.random.default_rng(seed)
st0 = [dict(carrfreq=float(rng.uniform(-5000, 5000)), codefreq=c.crate + float(rng.uniform(-2, 2)),
            remcode=float(rng.uniform(0.01, 0.99)), remcarr=float(rng.uniform(0, 6.2)),
            buffloc=int(rng.integers(0, NSAMP))) for c in chans]
eng.trk_set_state(st0)
ring = orc.make_ring(data, ns_, ns_)
ochs, bl = [], []
loops = []
for i, (c, st) in enumerate(zip(chans, st0)):
    acqfreq = 200.0 * round(st["carrfreq"] / 200.0)
    o = orc.make_chan(c.prn, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
    o.acq.acqfreq = acqfreq
    o.carrfreq, o.codefreq, o.remcode, o.remcarr = st["carrfreq"], st["codefreq"], st["remcode"], st["remcarr"]
    o.flagsync, o.synci, o.cnt = 0, (7 * i) % 20, 2001
    ochs.append(o); bl.append(C.c_uint64(st["buffloc"]))
    loops.append(eng.loop_state(i, acqfreq, flagsync=0, synci=o.synci, cnt=o.cnt))
eng.loop_set(loops)
eng.trk_run_loop(nper)
II, QQ, ns = eng.trk_fetch()
log, ndone = eng.trk_fetch_log()
print("ndone", ndone.min(), ndone.max())
hx = lambda v: float(v).hex()
L = orc.lib()
names = ["carrfreq", "codefreq", "carrNco", "codeNco", "carrErr", "codeErr", "freqErr", "remcode", "remcarr"]
summary = []
for i, o in enumerate(ochs):
    first = {}
    firstsum = None
    for e in range(nper):
        L.orc_sdrthread_step(C.byref(o), C.byref(ring), C.byref(bl[i]))
        r = log[i, e]
        if firstsum is None and not (np.array_equal(II[i, e], np.ctypeslib.as_array(o.II)[:5]) and np.array_equal(QQ[i, e], np.ctypeslib.as_array(o.QQ)[:5]) and ns[i, e] == o.currnsamp):
            firstsum = e
        for nm in names:
            if nm not in first and float(r[nm]) != float(getattr(o, nm)):
                first[nm] = (e, hx(r[nm]), hx(getattr(o, nm)))
    summary.append((i, firstsum, first))
    print("ch", i, "first sum mismatch", firstsum, "| first bit differences:", {k: v[0] for k, v in first.items()})
    for k, v in first.items():
        if k in ("carrErr", "freqErr", "codeErr", "carrNco", "carrfreq", "remcarr"):
            print("     ", k, v)
