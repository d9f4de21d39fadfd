#!/usr/bin/env python3
"""Generates tests/golden/full_size.json: the BASELINE configurations at their full channel counts, expected
outputs from the CPU oracle (oracle/gnss_oracle.c, the literal restatement: sequential fp64 NCOs).

  configs[1]  32-SV GPS L1CA cold acquisition (71 bins x <= 10 x 1 ms) on the bench's seeded 16.368 Msps
              int8 IQ stream: per PRN flagacq, iterations, code phase, Doppler bin, peak ratio, C/N0, buffloc
  configs[2]  32 channels x 50 code periods of 5-tap E/P/L correlator sums from seeded states on that
              stream: SHA-256 of the II / QQ arrays and of the samples-per-period table, first and last rows
  configs[3]  32 GPS L1CA on stream 1 + 14 GLONASS G1 (frequency numbers -7..+6) on stream 2: acquisition
              decisions of all 46 channels, then 20 periods of tracking sums (hashes) from the acquisition result

The inputs are regenerated from seeds by the same generator (erlangnetwork-gnsslib-sdr_amd/synth.py); nothing
of the reference is involved ("parity unpinned"): these pin the HIP path to the oracle
at full size.  Takes a few minutes on 8 cores.  Re-run:  python tests/golden/make_golden_full.py"""
import ctypes as C
import hashlib
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc          # noqa: E402
import gnsscorr_loader        # noqa: E402
import full_size_inputs as fs  # noqa: E402

gc = gnsscorr_loader.load()
synth = __import__("importlib").import_module("erlangnetwork_gnsslib_sdr_amd.synth")
NS = 16368


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def acquire(o, data, nsamples, wrpos):
    ring = orc.make_ring(data, nsamples, wrpos)
    xc = orc.codespectrum(o)
    o.xcode = xc.ctypes.data
    P = np.zeros(o.nfreq * o.nsamp)
    it = C.c_int()
    b = orc.lib().orc_sdracquisition(C.byref(o), C.byref(ring), P.ctypes.data, C.byref(it))
    return dict(flagacq=o.flagacq, iters=it.value, acqcodei=o.acq.acqcodei, freqi=o.acq.freqi, acqfreq=o.acq.acqfreq,
                peakr=o.acq.peakr, cn0=o.acq.cn0, buffloc=int(b))


def track(o, data, nsamples, st, nep, ntap):
    ring = orc.make_ring(data, nsamples, nsamples)
    o.carrfreq, o.codefreq, o.remcode, o.remcarr = st["carrfreq"], st["codefreq"], st["remcode"], st["remcarr"]
    b = st["buffloc"]
    II, QQ, ns = np.zeros((nep, ntap)), np.zeros((nep, ntap)), np.zeros(nep, np.int32)
    for e in range(nep):
        orc.lib().orc_sdrtracking(C.byref(o), C.byref(ring), b)
        assert o.flagtrk == 1
        II[e], QQ[e], ns[e] = np.ctypeslib.as_array(o.II)[:ntap], np.ctypeslib.as_array(o.QQ)[:ntap], o.currnsamp
        b += o.currnsamp
    return II, QQ, ns, dict(remcode=o.remcode, remcarr=o.remcarr, buffloc=b)


def main():
    out = {"note": "oracle-generated (literal restatement); inputs from tests/full_size_inputs.py seeds"}
    pool = ThreadPoolExecutor(os.cpu_count() or 4)

    # ---- configs[1]
    data, sats = fs.gps_stream(gc, synth, fs.ACQ_MS)
    nsamples = data.shape[0]
    res = list(pool.map(lambda p: acquire(orc.make_chan(p, dtype=2, f_if=0.0), data, nsamples, fs.ACQ_WRPOS), range(1, 33)))
    out["config1"] = dict(present=[dict(prn=s["prn"], cn0=s["cn0"], doppler=s["doppler"]) for s in sats], wrpos=fs.ACQ_WRPOS,
                          nsamples=nsamples, results=res)
    print("config1 acquired", [i + 1 for i, r in enumerate(res) if r["flagacq"]], flush=True)

    # ---- configs[2]
    data, _ = fs.gps_stream(gc, synth, fs.TRK_MS)
    nsamples = data.shape[0]
    states = fs.trk_states(gc)
    r2 = list(pool.map(lambda i: track(orc.make_chan(i + 1, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3), data, nsamples,
                                       states[i], fs.TRK_EPOCHS, 5), range(32)))
    II = np.stack([r[0] for r in r2])
    QQ = np.stack([r[1] for r in r2])
    ns = np.stack([r[2] for r in r2])
    out["config2"] = dict(nsamples=nsamples, epochs=fs.TRK_EPOCHS, II_sha256=sha(II), QQ_sha256=sha(QQ), ns_sha256=sha(ns),
                          II_first=II[:, 0].tolist(), II_last=II[:, -1].tolist(), QQ_first=QQ[:, 0].tolist(),
                          QQ_last=QQ[:, -1].tolist(), final=[r[3] for r in r2])
    print("config2 done", flush=True)

    # ---- configs[3]
    d1, d2, _, sat2 = fs.two_streams(gc, synth)
    nsamples = d1.shape[0]
    chans = fs.config3_channels(gc)

    def both(c):
        o = orc.make_chan(c.prn, ctype=c.ctype, dtype=2, f_if=0.0)
        d = d2 if c.ftype == 2 else d1
        a = acquire(o, d, nsamples, fs.C3_WRPOS)
        t = None
        if a["flagacq"]:
            st = dict(carrfreq=a["acqfreq"], codefreq=c.crate, remcode=0.0, remcarr=0.0, buffloc=a["buffloc"])
            II, QQ, ns, fin = track(o, d, nsamples, st, fs.C3_EPOCHS, 5)
            t = dict(II_sha256=sha(II), QQ_sha256=sha(QQ), ns=ns.tolist(), II_last=II[-1].tolist(), QQ_last=QQ[-1].tolist(),
                     final=fin)
        return dict(prn=c.prn, ctype=c.ctype, ftype=c.ftype, acq=a, trk=t)

    out["config3"] = dict(nsamples=nsamples, wrpos=fs.C3_WRPOS, epochs=fs.C3_EPOCHS, channels=list(pool.map(both, chans)),
                          glonass_present=[s["prn"] for s in sat2])
    print("config3 acquired", [(c["prn"], c["ctype"]) for c in out["config3"]["channels"] if c["acq"]["flagacq"]], flush=True)
    json.dump(out, open(os.path.join(HERE, "full_size.json"), "w"), indent=1)
    print("wrote full_size.json")


if __name__ == "__main__":
    main()
