"""The BASELINE configurations at their full channel counts (-m gpu) against committed fixtures
(tests/golden/full_size.json, made by tests/golden/make_golden_full.py from the literal oracle):

  configs[1]  32-SV GPS L1CA cold acquisition: every decision output identical, peak ratio / C/N0 to 1e-4
  configs[2]  32 channels x 50 periods, 5 taps: SHA-256 of the E/P/L sums and of the samples per period,
              i.e. every one of the 16 000 sums bit for bit
  configs[3]  32 GPS + 14 GLONASS G1 channels on two streams: acquisition decisions of all 46, then 20
              periods of tracking sums from the acquisition result, bit for bit

The oracle is not involved at test time: the inputs are regenerated from seeds (tests/full_size_inputs.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

import full_size_inputs as fs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "full_size.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _check_acq(r, v, who):
    for k in ("flagacq", "iters", "acqcodei", "freqi", "acqfreq", "buffloc"):
        assert r[k] == v[k], (who, k, r[k], v[k])
    assert abs(r["peakr"] - v["peakr"]) <= 1e-4 * v["peakr"], (who, r["peakr"], v["peakr"])
    assert abs(r["cn0"] - v["cn0"]) <= 1e-4 * abs(v["cn0"]), (who, r["cn0"], v["cn0"])


def test_config1_32sv_acquisition(gc, synth, engine):
    f = FIX["config1"]
    data, sats = fs.gps_stream(gc, synth, fs.ACQ_MS)
    assert data.shape[0] == f["nsamples"]
    engine.ring_create(1, 2, data.shape[0])
    engine.ring_push_raw(1, data, data.shape[0])
    engine.set_channels([gc.Channel(p, dtype=2, f_if=0.0) for p in range(1, 33)])
    engine.acq_run(f["wrpos"])
    res = engine.acq_fetch()
    for p, (r, v) in enumerate(zip(res, f["results"]), start=1):
        _check_acq(r, v, p)
    # what the search found makes sense: nothing that is not there, everything above 41 dB-Hz
    found = {p for p, r in enumerate(res, start=1) if r["flagacq"]}
    assert found <= {s["prn"] for s in f["present"]}
    assert {s["prn"] for s in f["present"] if s["cn0"] >= 41.0} <= found


def test_config2_32ch_tracking_50_periods(gc, synth, engine):
    f = FIX["config2"]
    data, _ = fs.gps_stream(gc, synth, fs.TRK_MS)
    assert data.shape[0] == f["nsamples"]
    engine.ring_create(1, 2, data.shape[0])
    engine.ring_push_raw(1, data, data.shape[0])
    engine.set_channels([gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in range(1, 33)])
    engine.trk_set_state(fs.trk_states(gc))
    engine.trk_run(f["epochs"])
    II, QQ, ns = engine.trk_fetch()
    assert np.array_equal(II[:, 0], np.array(f["II_first"])) and np.array_equal(QQ[:, 0], np.array(f["QQ_first"]))
    assert np.array_equal(II[:, -1], np.array(f["II_last"])) and np.array_equal(QQ[:, -1], np.array(f["QQ_last"]))
    assert sha(ns) == f["ns_sha256"]
    assert sha(II) == f["II_sha256"] and sha(QQ) == f["QQ_sha256"]
    for st, v in zip(engine.trk_get_state(), f["final"]):
        assert st["remcode"] == v["remcode"] and st["remcarr"] == v["remcarr"] and st["buffloc"] == v["buffloc"]


def test_config3_gps_plus_glonass_46_channels(gc, synth, engine):
    f = FIX["config3"]
    d1, d2, _, _ = fs.two_streams(gc, synth)
    n = d1.shape[0]
    assert n == f["nsamples"]
    engine.ring_create(1, 2, n)
    engine.ring_create(2, 2, n)
    engine.ring_push_raw(1, d1, n)
    engine.ring_push_raw(2, d2, n)
    chans = fs.config3_channels(gc)
    engine.set_channels(chans)
    engine.acq_run(f["wrpos"])
    res = engine.acq_fetch()
    for c, r, v in zip(chans, res, f["channels"]):
        assert (c.prn, c.ctype, c.ftype) == (v["prn"], v["ctype"], v["ftype"])
        _check_acq(r, v["acq"], (c.prn, c.ctype))
    assert {v["prn"] for v in f["channels"] if v["ctype"] == gc.CTYPE_G1 and v["acq"]["flagacq"]} == set(f["glonass_present"])
    # tracking from the acquisition result, handed over on the device
    parked = [dict(carrfreq=0.0, codefreq=c.crate, remcode=0.5, remcarr=0.0, buffloc=100) for c in chans]
    engine.trk_set_state(parked)
    engine.trk_start_from_acq()
    engine.trk_run(f["epochs"])
    II, QQ, ns = engine.trk_fetch()
    fin = engine.trk_get_state()
    for i, v in enumerate(f["channels"]):
        if not v["trk"]:
            continue
        t = v["trk"]
        assert ns[i].tolist() == t["ns"], v["prn"]
        assert sha(II[i]) == t["II_sha256"] and sha(QQ[i]) == t["QQ_sha256"], v["prn"]
        assert II[i, -1].tolist() == t["II_last"] and QQ[i, -1].tolist() == t["QQ_last"]
        assert fin[i]["remcode"] == t["final"]["remcode"] and fin[i]["remcarr"] == t["final"]["remcarr"]
        assert fin[i]["buffloc"] == t["final"]["buffloc"]
