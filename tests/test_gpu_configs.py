"""Parity cases for BASELINE configs[3] and configs[4] on one GPU (-m gpu): GPS L1CA + GLONASS G1
channels on two IF streams (ref frontend/stereo_L1G1.ini: two front ends, FTYPE1/FTYPE2), a fine
acquisition Doppler grid, and 10 ms coherent sums (cumsumcorr over LOOP_MS epochs, ref
src/sdrtrk.c:64-76).  Everything is checked against the CPU oracle on the same seeded streams."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

F_SF = 16.368e6
NS = 16368


def _streams(gc, synth, seed):
    """Stream 1: GPS PRN 5 and 17; stream 2: GLONASS frequency numbers -3 and +2 (ref
    src/sdrinit.c:600-606: carrier 1602 MHz + 562.5 kHz * k, seen at f_if + 562.5 kHz * k)."""
    rng = np.random.default_rng(seed)
    nsamp = 16 * 16384
    gps = {p: gc.gencode(p, gc.CTYPE_L1CA) for p in (5, 17, 22)}
    sat1 = [dict(prn=p, doppler=float(rng.uniform(-3000, 3000)), codephase=float(rng.uniform(0, 1023)),
                 cn0=47.0, phase=float(rng.uniform(0, 6.28))) for p in (5, 17)]
    d1 = synth.make_if(gps, nsamp, f_sf=F_SF, f_if=0.0, dtype=2, sats=sat1, seed=seed)
    g1code = gc.gencode(1, gc.CTYPE_G1)
    # make_if places a signal at f_if + doppler: the FDMA offset is folded into "doppler" here and
    # the code Doppler that would come with it is far below what the test resolves
    sat2 = [dict(prn=k, doppler=562.5e3 * k + float(rng.uniform(-800, 800)), codephase=float(rng.uniform(0, 511)),
                 cn0=48.0, phase=float(rng.uniform(0, 6.28))) for k in (-3, 2)]
    d2 = synth.make_if({k: g1code for k in (-3, 2)}, nsamp, f_sf=F_SF, f_if=0.0, dtype=2, sats=sat2, seed=seed + 1,
                       f_cf=1.602e9 * 1e6)     # huge f_cf: no code Doppler from the folded offset
    return nsamp, d1, d2, sat1, sat2


def _oracle_acq(orc, o, data, nsamples, wrpos):
    ring = orc.make_ring(data, nsamples, wrpos)
    xc = orc.codespectrum(o)
    o.xcode = xc.ctypes.data
    power = np.zeros(o.nfreq * o.nsamp)
    iters = C.c_int()
    buffloc = orc.lib().orc_sdracquisition(C.byref(o), C.byref(ring), power.ctypes.data, C.byref(iters))
    return buffloc, iters.value


def test_gps_plus_glonass_on_two_streams(gc, orc, synth, engine):
    """configs[3]: L1CA channels on front end 1, G1 channels on front end 2, one engine: acquisition of
    every channel on its own stream and grid, then a tracking batch started from the acquisition result."""
    nsamp, d1, d2, sat1, sat2 = _streams(gc, synth, 31)
    engine.ring_create(1, 2, nsamp)
    engine.ring_create(2, 2, nsamp)
    engine.ring_push_raw(1, d1, nsamp)
    engine.ring_push_raw(2, d2, nsamp)
    chans = [gc.Channel(5), gc.Channel(22), gc.Channel(17),
             gc.Channel(-3, ctype=gc.CTYPE_G1, ftype=2), gc.Channel(2, ctype=gc.CTYPE_G1, ftype=2)]
    assert chans[3].clen == 511 and chans[3].nsamp == NS
    engine.set_channels(chans)
    wrpos = 13 * NS + 4321
    engine.acq_run(wrpos)
    res = engine.acq_fetch()
    ochs = []
    for c, r in zip(chans, res):
        o = orc.make_chan(c.prn, ctype=c.ctype, dtype=2, f_if=0.0)
        assert o.nfreq == c.nfreq and np.array_equal(np.ctypeslib.as_array(o.freq)[:c.nfreq], c.freq)
        buffloc, iters = _oracle_acq(orc, o, d2 if c.ftype == 2 else d1, nsamp, wrpos)
        assert r["flagacq"] == o.flagacq == (0 if c.prn == 22 else 1), (c.prn, r)
        assert r["iters"] == iters and r["buffloc"] == buffloc
        assert r["acqcodei"] == o.acq.acqcodei and r["freqi"] == o.acq.freqi and r["acqfreq"] == o.acq.acqfreq
        assert abs(r["peakr"] - o.acq.peakr) <= 1e-4 * o.acq.peakr
        ochs.append(o)
    # the G1 channels were found at their FDMA offsets
    for c, r, s in zip(chans[3:], res[3:], sat2):
        assert abs(r["acqfreq"] - s["doppler"]) <= 100.0 + 1e-6

    # device-side hand-over: acquired channels take over the acquisition result, the others keep their state
    parked = [dict(carrfreq=1.0 + i, codefreq=chans[i].crate, remcode=0.25, remcarr=0.5, buffloc=1000 + i)
              for i in range(len(chans))]
    engine.trk_set_state(parked)
    engine.trk_start_from_acq()
    for i, (st, r) in enumerate(zip(engine.trk_get_state(), res)):
        if r["flagacq"]:
            assert st == dict(carrfreq=r["acqfreq"], codefreq=chans[i].crate, remcode=0.0, remcarr=0.0, buffloc=r["buffloc"])
        else:
            assert st == parked[i]

    # tracking batch of every acquired channel from its acquisition result (ref src/sdracq.c:54-55)
    live = [i for i, r in enumerate(res) if r["flagacq"]]
    states = [dict(carrfreq=res[i]["acqfreq"], codefreq=chans[i].crate, remcode=0.0, remcarr=0.0,
                   buffloc=res[i]["buffloc"]) if i in live else
              dict(carrfreq=0.0, codefreq=chans[i].crate, remcode=0.5, remcarr=0.0, buffloc=100) for i in range(len(chans))]
    engine.trk_set_state(states)
    nep = 4
    engine.trk_run(nep)
    II, QQ, ns = engine.trk_fetch()
    L = orc.lib()
    for i in live:
        o, st = ochs[i], states[i]
        ring = orc.make_ring(d2 if chans[i].ftype == 2 else d1, nsamp, nsamp)
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = st["carrfreq"], st["codefreq"], st["remcode"], st["remcarr"]
        buffloc = st["buffloc"]
        for e in range(nep):
            L.orc_sdrtracking(C.byref(o), C.byref(ring), buffloc)
            assert o.flagtrk == 1 and o.currnsamp == ns[i, e]
            assert np.array_equal(np.ctypeslib.as_array(o.II)[:5], II[i, e])
            assert np.array_equal(np.ctypeslib.as_array(o.QQ)[:5], QQ[i, e])
            buffloc += o.currnsamp
        # a channel that sits on its signal: prompt power well above the early/late skirts' noise floor
        p = np.hypot(II[i, :, 0], QQ[i, :, 0]).mean()
        assert p > 2000.0, (chans[i].prn, p)


def test_fine_doppler_grid_and_10ms_sums(gc, orc, synth, engine):
    """configs[4]: two IF streams, acquisition on a 50 Hz grid (+-1 kHz around a coarse estimate), and
    tracking read out as 10-epoch coherent sums (sumI/sumQ of cumsumcorr)."""
    nsamp, d1, d2, sat1, sat2 = _streams(gc, synth, 57)
    engine.ring_create(1, 2, nsamp)
    engine.ring_create(2, 2, nsamp)
    engine.ring_push_raw(1, d1, nsamp)
    engine.ring_push_raw(2, d2, nsamp)
    chans = [gc.Channel(5, hband=1000, step=50), gc.Channel(2, ctype=gc.CTYPE_G1, ftype=2, hband=1000, step=50)]
    # centre the fine grids on the coarse 200 Hz bin of the true frequency
    for c, s in zip(chans, (sat1[0], sat2[1])):
        centre = 200.0 * round((s["doppler"] - c.foffset) / 200.0)
        c.freq = c.freq + centre
    assert chans[0].nfreq == 41
    engine.set_channels(chans)
    wrpos = 12 * NS + 99
    engine.acq_run(wrpos)
    res = engine.acq_fetch()
    ochs = []
    for c, r, s, data in zip(chans, res, (sat1[0], sat2[1]), (d1, d2)):
        o = orc.make_chan(c.prn, ctype=c.ctype, dtype=2, f_if=0.0)
        o.nfreq = c.nfreq
        for i, f in enumerate(c.freq):
            o.freq[i] = f
        buffloc, iters = _oracle_acq(orc, o, data, nsamp, wrpos)
        assert r["flagacq"] == o.flagacq == 1
        assert r["acqcodei"] == o.acq.acqcodei and r["freqi"] == o.acq.freqi and r["acqfreq"] == o.acq.acqfreq
        assert r["buffloc"] == buffloc and r["iters"] == iters
        # 1 ms coherent: the main lobe is ~1 kHz wide, neighbouring 50 Hz bins differ by noise only
        assert abs(r["acqfreq"] - s["doppler"]) <= 100.0
        ochs.append(o)

    states = [dict(carrfreq=r["acqfreq"], codefreq=c.crate, remcode=0.0, remcarr=0.0, buffloc=r["buffloc"])
              for c, r in zip(chans, res)]
    engine.trk_set_state(states)
    engine.trk_run(10)                                   # LOOP_MS = 10 epochs, ref src/sdr.h:152
    II, QQ, ns = engine.trk_fetch()
    sI, sQ = engine.trk_fetch_sums()
    L = orc.lib()
    for i, (o, st, data) in enumerate(zip(ochs, states, (d1, d2))):
        ring = orc.make_ring(data, nsamp, nsamp)
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = st["carrfreq"], st["codefreq"], st["remcode"], st["remcarr"]
        L.orc_clearcumsumcorr(C.byref(o))
        buffloc = st["buffloc"]
        for e in range(10):
            L.orc_sdrtracking(C.byref(o), C.byref(ring), buffloc)
            L.orc_cumsumcorr(C.byref(o), 1)
            buffloc += o.currnsamp
        assert np.array_equal(np.ctypeslib.as_array(o.sumI)[:5], sI[i])
        assert np.array_equal(np.ctypeslib.as_array(o.sumQ)[:5], sQ[i])
        assert np.array_equal(sI[i], II[i].sum(axis=0)) and np.array_equal(sQ[i], QQ[i].sum(axis=0))


def test_mixed_sample_formats_in_one_engine(gc, orc, engine):
    """IQ channels on front end 1 and real-IF channels on front end 2 in one batch: one correlator launch
    per sample format, each serving only its own channels."""
    from test_gpu_tracking import _oracle_run
    rng = np.random.default_rng(5150)
    nsamp = 16 * 8192
    d_iq = rng.integers(-70, 71, size=(nsamp, 2), dtype=np.int8)
    d_re = rng.integers(-90, 91, size=nsamp, dtype=np.int8)
    engine.ring_create(1, 2, nsamp)
    engine.ring_create(2, 1, nsamp)
    engine.ring_push_raw(1, d_iq, nsamp)
    engine.ring_push_raw(2, d_re, nsamp)
    spec = [(4, 2, 1, 0.0), (9, 1, 2, 4.092e6), (21, 2, 1, 0.0), (30, 1, 2, 4.092e6)]     # prn, dtype, ftype, f_if
    chans = [gc.Channel(p, dtype=dt, ftype=ft, f_if=fi) for p, dt, ft, fi in spec]
    engine.set_channels(chans)
    states = [dict(carrfreq=fi + float(rng.uniform(-4000, 4000)), codefreq=c.crate + float(rng.uniform(-2, 2)),
                   remcode=float(rng.uniform(0.05, 0.95)), remcarr=float(rng.uniform(0, 6)), buffloc=77 + 500 * i)
              for i, (c, (_, _, _, fi)) in enumerate(zip(chans, spec))]
    engine.trk_set_state(states)
    engine.trk_run(5)
    II, QQ, ns = engine.trk_fetch()
    for i, (p, dt, ft, fi) in enumerate(spec):
        o = orc.make_chan(p, dtype=dt, f_if=fi)
        oII, oQQ, ons, _ = _oracle_run(orc, [o], [states[i]], d_iq if dt == 2 else d_re, nsamp, nsamp, 5)
        assert np.array_equal(ns[i], ons[0])
        assert np.array_equal(II[i], oII[0]) and np.array_equal(QQ[i], oQQ[0])
