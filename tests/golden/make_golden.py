#!/usr/bin/env python3
"""Generates tests/golden/vectors.json: seeded inputs (regenerated from the seed, not stored) and the
expected outputs of the correlation path, produced by the CPU oracle (oracle/gnss_oracle.c).

The reference itself cannot be run in this image and ships no vectors, so these are
ORACLE-generated regression anchors ("parity unpinned"): they pin the oracle and the HIP path to each
other and to this commit, not to a run of the reference binary.  Re-run:  python tests/golden/make_golden.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc          # noqa: E402

TI = 1 / 16.368e6


def data_for(seed, n, dtype, amp=90):
    rng = np.random.default_rng(seed)
    d = rng.integers(-amp, amp + 1, size=n * dtype, dtype=np.int8)
    d[:4] = [-128, 127, -128, 127]
    return d


def main():
    out = {"note": "oracle-generated (literal restatement: sequential fp64 NCOs); inputs = numpy default_rng(seed).integers(-amp, amp+1)"
                   " with the first four bytes set to -128,127,-128,127", "correlator": [], "acquisition": []}
    cases = [  # seed, dtype, n, freq, phi0, dcode, coff, taps, prn
        (101, 2, 16368, 2345.6, 0.0, 0.0, 0.0, [3, 6], 1),
        (102, 2, 16369, -4321.0, 1.25, -1.7, 511.25, [3, 6], 17),
        (103, 1, 16367, 4.092e6 + 777.0, 5.9, 2.9, 1022.6, [3, 6, 9, 12, 15, 18], 32),
        (104, 2, 16368, -3.94e6, 0.4, 0.3, 100.5, [8], 5),
        (105, 1, 4000, 4.092e6 - 5000.0, 0.0, 0.0, 0.0, [4, 8, 12, 16, 20, 24], 120),
    ]
    for seed, dtype, n, freq, phi0, dcode, coff, taps, prn in cases:
        code, crate = orc.gencode(prn, 1)
        d = data_for(seed, n, dtype)
        II, QQ, remc, remp = orc.correlator(d, dtype, TI, n, freq, phi0, crate + dcode, coff, taps, code)
        out["correlator"].append(dict(seed=seed, dtype=dtype, n=n, freq=freq, phi0=phi0, codefreq=crate + dcode,
                                      coff=coff, taps=taps, prn=prn, amp=90, II=list(II), QQ=list(QQ),
                                      remc=remc, remp=remp))
    # acquisition decision on a noise-free-ish synthetic signal built from the PRN itself
    for seed, prn, doppler, delay in ((201, 8, 1200.0, 5000), (202, 23, -3400.0, 12345)):
        o = orc.make_chan(prn, dtype=2, f_if=0.0)
        n = o.nsamp
        code = np.ctypeslib.as_array(o.code).astype(np.float64)
        nsamples = 16 * 16384
        k = np.arange(nsamples)
        chips = code[(((k - delay) * 0.0625) % 1023).astype(np.int64)]
        ph = 2 * np.pi * doppler * k * TI
        rng = np.random.default_rng(seed)
        xi = np.rint(20 * chips * np.cos(ph) + rng.normal(0, 12, nsamples))
        xq = np.rint(-20 * chips * np.sin(ph) + rng.normal(0, 12, nsamples))
        data = np.stack([np.clip(xi, -127, 127), np.clip(xq, -127, 127)], axis=1).astype(np.int8)
        wrpos = 13 * n + 100
        ring = orc.make_ring(data, nsamples, wrpos)
        xc = orc.codespectrum(o)
        o.xcode = xc.ctypes.data
        P = np.zeros(o.nfreq * n)
        it = C.c_int()
        b = orc.lib().orc_sdracquisition(C.byref(o), C.byref(ring), P.ctypes.data, C.byref(it))
        out["acquisition"].append(dict(seed=seed, prn=prn, doppler=doppler, delay=delay, wrpos=wrpos,
                                       nsamples=nsamples, flagacq=o.flagacq, iters=it.value,
                                       acqcodei=o.acq.acqcodei, freqi=o.acq.freqi, acqfreq=o.acq.acqfreq,
                                       peakr=o.acq.peakr, cn0=o.acq.cn0, buffloc=int(b),
                                       P_peak=float(P.max()), P_sum=float(P.sum())))
    json.dump(out, open(os.path.join(HERE, "vectors.json"), "w"), indent=1)
    print("wrote", len(out["correlator"]), "correlator and", len(out["acquisition"]), "acquisition vectors")


if __name__ == "__main__":
    main()
