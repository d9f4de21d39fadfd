"""Committed fixtures (tests/golden/vectors.json, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU (-m gpu): the HIP path reproduces them without the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "vectors.json")))
TI = 1 / 16.368e6


def _data(v):
    rng = np.random.default_rng(v["seed"])
    d = rng.integers(-v["amp"], v["amp"] + 1, size=v["n"] * v["dtype"], dtype=np.int8)
    d[:4] = [-128, 127, -128, 127]
    return d


def _acq_signal(v, code):
    nsamples, n = v["nsamples"], 16368
    k = np.arange(nsamples)
    chips = code.astype(np.float64)[(((k - v["delay"]) * 0.0625) % 1023).astype(np.int64)]
    ph = 2 * np.pi * v["doppler"] * k * TI
    rng = np.random.default_rng(v["seed"])
    xi = np.rint(20 * chips * np.cos(ph) + rng.normal(0, 12, nsamples))
    xq = np.rint(-20 * chips * np.sin(ph) + rng.normal(0, 12, nsamples))
    return np.stack([np.clip(xi, -127, 127), np.clip(xq, -127, 127)], axis=1).astype(np.int8)


@pytest.mark.parametrize("v", VEC["correlator"], ids=lambda v: f"seed{v['seed']}")
def test_oracle_reproduces_correlator_vectors(orc, v):
    code, _ = orc.gencode(v["prn"], 1)
    II, QQ, remc, remp = orc.correlator(_data(v), v["dtype"], TI, v["n"], v["freq"], v["phi0"], v["codefreq"],
                                        v["coff"], v["taps"], code)
    assert list(II) == v["II"] and list(QQ) == v["QQ"] and remc == v["remc"] and remp == v["remp"]


@pytest.mark.gpu
@pytest.mark.parametrize("v", VEC["correlator"], ids=lambda v: f"seed{v['seed']}")
def test_hip_reproduces_correlator_vectors(gc, v):
    code, _ = gc.gencode(v["prn"], gc.CTYPE_L1CA if v["prn"] < 120 else gc.CTYPE_L1SBAS)
    d = _data(v)
    s = np.array(v["taps"], np.int32)
    nt = 1 + 2 * len(s)
    II, QQ = np.zeros(nt), np.zeros(nt)
    remc, remp = C.c_double(), C.c_double()
    c16 = code.astype(np.int16)
    gc.lib().correlator(d.ctypes.data, v["dtype"], TI, v["n"], v["freq"], v["phi0"], v["codefreq"], v["coff"],
                        s.ctypes.data, len(s), II.ctypes.data, QQ.ctypes.data, C.byref(remc), C.byref(remp),
                        c16.ctypes.data, len(c16))
    assert list(II) == v["II"] and list(QQ) == v["QQ"]
    assert remc.value == v["remc"] and remp.value == v["remp"]


@pytest.mark.gpu
@pytest.mark.parametrize("v", VEC["acquisition"], ids=lambda v: f"seed{v['seed']}")
def test_hip_reproduces_acquisition_vectors(gc, engine, v):
    code, _ = gc.gencode(v["prn"], gc.CTYPE_L1CA)
    data = _acq_signal(v, code)
    engine.ring_create(1, 2, v["nsamples"])
    engine.ring_push_raw(1, data, v["nsamples"])
    engine.set_channels([gc.Channel(v["prn"], dtype=2, f_if=0.0)])
    engine.acq_run(v["wrpos"])
    r = engine.acq_fetch()[0]
    assert (r["flagacq"], r["iters"], r["acqcodei"], r["freqi"], r["acqfreq"], r["buffloc"]) == \
           (v["flagacq"], v["iters"], v["acqcodei"], v["freqi"], v["acqfreq"], v["buffloc"])
    assert abs(r["peakr"] - v["peakr"]) <= 1e-4 * v["peakr"] and abs(r["cn0"] - v["cn0"]) <= 1e-4 * abs(v["cn0"])
    P = engine.acq_power(0)
    assert abs(P.max() - v["P_peak"]) <= 1e-4 * v["P_peak"] and abs(P.sum() - v["P_sum"]) <= 1e-4 * v["P_sum"]
    # the generator's truth: Doppler bin and code delay
    assert abs(r["acqfreq"] - v["doppler"]) <= 100.0
    start = v["wrpos"] - 11 * 16368
    assert (r["acqcodei"] - (v["delay"] - start)) % 16368 in (0, 1, 16367)
