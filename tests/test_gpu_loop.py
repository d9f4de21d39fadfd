"""Closed loop on the device (gnsscorr_trk_run_loop) vs the oracle's restatement of sdrthread()'s tracking
branch (ref src/sdrmain.c:264-312: sdrtracking, cumsumcorr, pll, dll, clearcumsumcorr in the
flagsync / swloop cadence of src/sdrnav.c:241-262).

Bar: correlator sums, samples per period, the filter-update flags and the NCO remainders bit for bit over
hundreds of periods.  The loop filters call atan2 / atan (ref src/sdrtrk.c:106-113), whose last bit belongs to
the maths library: the device's and glibc's differ in about one call in three, and a frequency that differs by
one ulp can change a binade's rounded NCO step, i.e. the remainders after thousands of samples (measured: 1.4e-11
rad after 208 periods, tools/debug/loop_drift.py).  So every period's filter outputs are compared with the
oracle's from IDENTICAL inputs (1e-14: a few ulps) and then handed to the oracle (`_adopt`): what is held bit
for bit is everything the library's rounding does not touch -- which is everything else."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F_SF = 16.368e6


def _signal(gc, synth, prns, dop, cph, nper, dtype=2, f_if=0.0, cn0=47.0, seed=41):
    codes = {p: gc.gencode(p, gc.CTYPE_L1CA) for p in prns}
    rng = np.random.default_rng(seed)
    sats = [dict(prn=p, doppler=d, codephase=c, cn0=cn0, phase=0.4 * i, bits=rng.choice([-1.0, 1.0], size=64))
            for i, (p, d, c) in enumerate(zip(prns, dop, cph))]
    return synth.make_if(codes, 16368 * nper, f_sf=F_SF, f_if=f_if, dtype=dtype, sats=sats, seed=seed)


def _close(a, b, tol=1e-14):
    return abs(a - b) <= tol * max(1.0, abs(a), abs(b))


FILTER_FIELDS = ("carrfreq", "codefreq", "carrNco", "codeNco", "carrErr", "codeErr", "freqErr")


def _adopt(o, r, where, tol=1e-14):
    """This period's filter outputs: equal to the oracle's to a few ulps (same inputs), then the oracle continues
    from the device's values, so that the next period starts from identical frequencies on both sides.  The nav
    bit synchronisation (flagsync, decided bits) involves no library call: exact."""
    assert r["flagsync"] == o.flagsync, (where, "flagsync")
    assert r["navbit"] == (o.bit if (o.flagsync and o.swsync) else 0), (where, "navbit")
    for f in FILTER_FIELDS:
        assert _close(float(r[f]), getattr(o, f), tol), (where, f, float(r[f]), getattr(o, f))
        setattr(o, f, float(r[f]))


def _run_case(gc, orc, synth, engine, dtype, f_if, corrn, corrd, corrp, nper, flagsync, chunks):
    prns = [5, 12, 25, 30]
    dop = [1517.0, -3222.0, 4630.0, -120.0]
    cph = [311.3, 12.8, 870.1, 555.5]
    sig = _signal(gc, synth, prns, dop, cph, nper + 3, dtype=dtype, f_if=f_if)
    nsamples = sig.shape[0]
    engine.ring_create(1, dtype, nsamples)
    engine.ring_push_raw(1, sig, nsamples)
    chans = [gc.Channel(p, dtype=dtype, f_if=f_if, corrn=corrn, corrd=corrd, corrp=corrp) for p in prns]
    engine.set_channels(chans)
    L = orc.lib()
    ring = orc.make_ring(sig, nsamples, nsamples)
    ochs, bufflocs, states, loops = [], [], [], []
    for i, c in enumerate(chans):
        acqfreq = f_if + 200.0 * round(dop[i] / 200.0)          # the 200 Hz grid of sdracquisition()
        o = orc.make_chan(c.prn, dtype=dtype, f_if=f_if, corrn=corrn, corrd=corrd, corrp=corrp)
        o.acq.acqfreq = acqfreq
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = acqfreq, c.crate, 0.0, 0.0
        o.flagsync, o.synci, o.cnt = flagsync, (3 + 5 * i) % 20, 2001 + 7 * i
        b = int(round((1023 - cph[i]) * 16)) % 16368
        ochs.append(o)
        bufflocs.append(C.c_uint64(b))
        states.append(dict(carrfreq=acqfreq, codefreq=c.crate, remcode=0.0, remcarr=0.0, buffloc=b))
        loops.append(engine.loop_state(i, acqfreq, flagsync=flagsync, synci=o.synci, cnt=o.cnt))
    engine.trk_set_state(states)
    engine.loop_set(loops)
    # the observables (setobsdata, ref src/sdrtrk.c:160-209): replayed over the device's log, against the oracle's calls
    # inside its loop
    obs = []
    for o in ochs:
        st = gc.ObsState()
        st.f_sf, st.f_if, st.foffset, st.ctime, st.loopms = o.f_sf, o.f_if, o.foffset, o.ctime, o.loopms
        st.oldremcode = o.remcode
        obs.append(st)
    done = 0
    for nrun in chunks:                                         # several runs: the state must carry over
        engine.trk_run_loop(nrun)
        II, QQ, ns = engine.trk_fetch()
        log, ndone = engine.trk_fetch_log()
        assert np.all(ndone == nrun)
        for i, o in enumerate(ochs):
            cnt0, want = o.cnt, []
            for e in range(nrun):
                nobs = o.obs_n
                assert L.orc_sdrthread_step(C.byref(o), C.byref(ring), C.byref(bufflocs[i])) == 1
                if o.obs_n != nobs:
                    want.append((o.obs_tow, o.obs_remcout, o.obs_L, o.obs_D, o.obs_S, o.obs_Isum == 0.0, o.obs_codei, o.obs_cntout))
                ntap = 1 + 2 * corrn
                where = (i, done + e)
                assert ns[i, e] == o.currnsamp, where
                # after the step the oracle's II/QQ hold this period's correlator outputs
                assert np.array_equal(II[i, e], np.ctypeslib.as_array(o.II)[:ntap]), where
                assert np.array_equal(QQ[i, e], np.ctypeslib.as_array(o.QQ)[:ntap]), where
                r = log[i, e]
                assert r["flagloopfilter"] == o.flagloopfilter, where
                assert r["remcode"] == o.remcode and r["remcarr"] == o.remcarr, where
                _adopt(o, r, where)
            rows = gc.obs_replay(obs[i], log[i], II[i, :, 0], cnt0=cnt0)
            assert len(rows) == len(want), (i, len(rows), len(want))
            for r, w in zip(rows, want):
                # (Doppler and what follows from it inherit the ulp of the filters' atan: _adopt)
                assert r["tow"] == w[0] and int(r["codei"]) == w[6] and int(r["cntout"]) == w[7], (i, w)
                assert _close(r["remcout"], w[1]) and _close(r["L"], w[2], 1e-12) and _close(r["D"], w[3]), (i, w)
                assert bool(r["snr"]) == w[5] and (not w[5] or _close(r["S"], w[4], 1e-12)), (i, w)
        done += nrun
    fin = engine.trk_get_state()
    lst = engine.loop_get()
    for i, o in enumerate(ochs):
        assert fin[i]["buffloc"] == bufflocs[i].value and fin[i]["remcode"] == o.remcode and fin[i]["remcarr"] == o.remcarr
        assert lst[i].cnt == o.cnt and lst[i].navcnt == o.navcnt
        ntap = 1 + 2 * corrn
        for name in ("sumI", "sumQ", "oldsumI", "oldsumQ", "II", "QQ"):
            assert np.array_equal(np.ctypeslib.as_array(getattr(lst[i], name))[:ntap],
                                  np.ctypeslib.as_array(getattr(o, name))[:ntap]), (i, name)
        # the loops pulled in
        assert abs(o.carrfreq - (f_if + dop[i])) < 150.0
    return ochs


def test_closed_loop_before_bit_sync(gc, orc, synth, engine):
    """prm1 every period (ref src/sdrmain.c:272-276): 220 periods from the acquisition hand-over state,
    BASELINE tap layout (5 taps, int8 IQ, zero IF)."""
    _run_case(gc, orc, synth, engine, 2, 0.0, 2, 3, 3, nper=220, flagsync=0, chunks=(1, 100, 119))


def test_closed_loop_after_bit_sync(gc, orc, synth, engine):
    """prm2 whenever checkbit() raises swloop (every 10 periods counted from each channel's own bit edge,
    ref src/sdrnav.c:241-262): 240 periods, the shipped 13-tap layout on the real 4.092 MHz IF."""
    _run_case(gc, orc, synth, engine, 1, 4.092e6, 6, 3, 6, nper=240, flagsync=1, chunks=(37, 203))


def test_closed_loop_stops_where_data_ends(gc, orc, synth, engine):
    """sdrtracking() does nothing until bufflocnow > buffloc (ref src/sdrtrk.c:26-30): a run that asks for more
    periods than the ring holds stops there, per channel, and resumes after more samples arrive."""
    prns = [7, 19]
    sig = _signal(gc, synth, prns, [900.0, -2100.0], [100.0, 600.0], 40)
    nsamples = sig.shape[0]
    engine.ring_create(1, 2, nsamples)
    half = 16368 * 20
    engine.ring_push_raw(1, sig[:half], half)
    chans = [gc.Channel(p, dtype=2, f_if=0.0) for p in prns]
    engine.set_channels(chans)
    engine.trk_set_state([dict(carrfreq=800.0, codefreq=c.crate, remcode=0.0, remcarr=0.0, buffloc=5000 * (i + 1))
                          for i, c in enumerate(chans)])
    engine.loop_set([engine.loop_state(i, 800.0) for i in range(2)])
    engine.trk_run_loop(30)
    _, _, ns = engine.trk_fetch()
    log, ndone = engine.trk_fetch_log()
    # period e of channel i starts at buffloc_e; it runs iff wrpos - nsamp > buffloc_e
    for i in range(2):
        b, cnt = 5000 * (i + 1), 0
        while half - 16368 > b and cnt < 30:
            b += int(ns[i, cnt])
            cnt += 1
        assert ndone[i] == cnt and 15 <= cnt <= 19
        assert np.all(ns[i, cnt:] == 0)
    engine.ring_push_raw(1, sig[half:], nsamples - half)
    engine.trk_run_loop(10)
    _, ndone2 = engine.trk_fetch_log()
    assert np.all(ndone2 == 10)


def _check_against_oracle(orc, engine, ochs, ring, bufflocs, nrun, ntap, done=0, tol=1e-14):
    """One trk_run_loop(nrun) against nrun oracle steps per channel: sums, samples, flags and remainders bit for
    bit, filter states to 1e-12."""
    L = orc.lib()
    engine.trk_run_loop(nrun)
    II, QQ, ns = engine.trk_fetch()
    log, ndone = engine.trk_fetch_log()
    assert np.all(ndone == nrun), ndone
    for i, o in enumerate(ochs):
        for e in range(nrun):
            assert L.orc_sdrthread_step(C.byref(o), C.byref(ring), C.byref(bufflocs[i])) == 1
            where = (i, done + e)
            assert ns[i, e] == o.currnsamp, where
            assert np.array_equal(II[i, e], np.ctypeslib.as_array(o.II)[:ntap]), where
            assert np.array_equal(QQ[i, e], np.ctypeslib.as_array(o.QQ)[:ntap]), where
            r = log[i, e]
            assert r["flagloopfilter"] == o.flagloopfilter, where
            assert r["remcode"] == o.remcode and r["remcarr"] == o.remcarr, where
            _adopt(o, r, where, tol)


def test_closed_loop_bench_configuration_32_channels(gc, orc, synth, engine):
    """The closed-loop leg of bench.py as a parity case: the 32 channels of BASELINE configs[2] from the bench's
    own start states (seed 20240601: frequencies anywhere in +-5 kHz, most channels on no satellite, so the loops
    wander), filter update every period (flagsync = 0), 320 periods in one launch and 30 more in a second.  (A
    round-2 build of the closed-loop kernel stalled on the device in this configuration after 257 periods: the
    scans over the period's carrier pieces relied on a closing sentinel alone)"""
    NSAMP, seed, nper = 16368, 20240601, 350
    prns = list(range(1, 33))
    codes = {p: gc.gencode(p, gc.CTYPE_L1CA) for p in prns}
    sats = synth.default_sats(prns, seed=seed)
    data = synth.make_if(codes, (nper + 4) * NSAMP, f_sf=F_SF, f_if=0.0, dtype=2, sats=sats, seed=seed)
    nsamples = data.shape[0]
    engine.ring_create(1, 2, nsamples)
    engine.ring_push_raw(1, data, nsamples)
    chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in prns]
    engine.set_channels(chans)
    rng = np.random.default_rng(seed)
    states0 = [dict(carrfreq=float(rng.uniform(-5000, 5000)), codefreq=c.crate + float(rng.uniform(-2, 2)),
                    remcode=float(rng.uniform(0.01, 0.99)), remcarr=float(rng.uniform(0, 6.2)),
                    buffloc=int(rng.integers(0, NSAMP))) for c in chans]
    engine.trk_set_state(states0)
    ring = orc.make_ring(data, nsamples, nsamples)
    ochs, bufflocs, loops = [], [], []
    for i, (c, st) in enumerate(zip(chans, states0)):
        acqfreq = 200.0 * round(st["carrfreq"] / 200.0)
        o = orc.make_chan(c.prn, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
        o.acq.acqfreq = acqfreq
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = st["carrfreq"], st["codefreq"], st["remcode"], st["remcarr"]
        o.flagsync, o.synci, o.cnt = 0, (7 * i) % 20, 2001
        ochs.append(o)
        bufflocs.append(C.c_uint64(st["buffloc"]))
        loops.append(engine.loop_state(i, acqfreq, flagsync=0, synci=o.synci, cnt=o.cnt))
    engine.loop_set(loops)
    _check_against_oracle(orc, engine, ochs, ring, bufflocs, 320, 5)
    _check_against_oracle(orc, engine, ochs, ring, bufflocs, 30, 5, done=320)


@pytest.mark.parametrize("seed,dtype,f_if,f_sf,taps,flagsync", [
    (101, 2, 0.0, 16.368e6, (2, 3, 3), 1),
    (102, 2, 0.0, 16.368e6, (6, 3, 6), 0),
    (103, 1, 4.092e6, 16.368e6, (2, 3, 3), 1),
    (104, 1, 4.092e6, 16.368e6, (1, 8, 8), 0),
    (105, 2, 0.0, 4.092e6, (2, 1, 1), 1),
    (106, 2, 0.0, 20.0e6, (2, 3, 3), 1),
    (107, 2, 0.0, 20.0e6, (6, 3, 6), 0),
    (108, 2, 0.0, 16.368e6, (2, 3, 3), -1),         # GLONASS G1 channels (511 chips, 10-ms bits), before ...
    (109, 2, 0.0, 16.368e6, (6, 3, 6), -2),         # ... and after nav bit synchronisation
])
def test_closed_loop_random_states_across_front_ends(gc, orc, engine, seed, dtype, f_if, f_sf, taps, flagsync):
    """Closed loop from random start states on noise (no satellite: the filters wander), per front end of the
    shipped configurations -- int8 IQ at 16.368, 4.092 and 20 Msps, real samples at a 4.092 MHz IF -- with 3-, 5- and
    13-tap sets, before (filter update every period) and after nav bit synchronisation (every 10 periods: intervals
    planned from claims discovered for brackets around the closed-form period starts, gnsscorr_loop.hip).  Starts
    include a remainder of exactly 0, remainders within 1e-6 of 0 and of 1 chip, a carrier phase of 0 and Doppler up
    to +-9 kHz.  12 channels x 130 periods in two runs, everything bit for bit against orc_sdrthread_step."""
    corrn, corrd, corrp = taps
    ctype = gc.CTYPE_L1CA
    if flagsync < 0:                                    # (negative: the G1 cases)
        ctype, flagsync = gc.CTYPE_G1, -flagsync - 1
    nper, nch = 130, 12
    nsamp = int(f_sf * 1e-3)
    rng = np.random.default_rng(seed)
    nsamples = nsamp * (nper + 4)
    data = rng.integers(-60, 61, size=(nsamples, 2) if dtype == 2 else (nsamples,), dtype=np.int8)
    engine.ring_create(1, dtype, nsamples)
    engine.ring_push_raw(1, data, nsamples)
    prns = [1 + (3 * i) % 32 for i in range(nch)] if ctype == gc.CTYPE_L1CA else [-7 + i for i in range(nch)]   # G1: frequency numbers
    chans = [gc.Channel(p, ctype=ctype, dtype=dtype, f_if=f_if, f_sf=f_sf, corrn=corrn, corrd=corrd, corrp=corrp) for p in prns]
    engine.set_channels(chans)
    ring = orc.make_ring(data, nsamples, nsamples)
    states0, ochs, bufflocs, loops = [], [], [], []
    for i, c in enumerate(chans):
        edge = i % 4
        remcode = (0.0 if edge == 0 else float(rng.uniform(0.0, 1e-6)) if edge == 1 else
                   float(1.0 - rng.uniform(0.0, 1e-6)) if edge == 2 else float(rng.uniform(0.01, 0.99)))
        st = dict(carrfreq=f_if + c.foffset + float(rng.uniform(-9000, 9000)), codefreq=c.crate + float(rng.uniform(-6, 6)),
                  remcode=remcode, remcarr=float(rng.uniform(0, 6.2)) if i % 5 else 0.0, buffloc=int(rng.integers(0, nsamp)))
        states0.append(st)
        acqfreq = f_if + c.foffset + 200.0 * round((st["carrfreq"] - f_if - c.foffset) / 200.0)
        o = orc.make_chan(c.prn, ctype=ctype, dtype=dtype, f_sf=f_sf, f_if=f_if, corrn=corrn, corrd=corrd, corrp=corrp)
        o.acq.acqfreq = acqfreq
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = st["carrfreq"], st["codefreq"], st["remcode"], st["remcarr"]
        o.flagsync, o.synci, o.cnt = flagsync, (3 + 7 * i) % 20, 2001 + 3 * i
        ochs.append(o)
        bufflocs.append(C.c_uint64(st["buffloc"]))
        loops.append(engine.loop_state(i, acqfreq, flagsync=flagsync, synci=o.synci, cnt=o.cnt))
    engine.trk_set_state(states0)
    engine.loop_set(loops)
    ntap = 1 + 2 * corrn
    # (filter outputs: the device's atan / atan2 and glibc's differ by an ulp now and then, times the loop gains -- 1e-13
    # relative here, north_star's bound is 1e-4; everything the correlators return stays exact)
    _check_against_oracle(orc, engine, ochs, ring, bufflocs, 97, ntap, tol=1e-12)
    _check_against_oracle(orc, engine, ochs, ring, bufflocs, 33, ntap, done=97, tol=1e-12)
    fin = engine.trk_get_state()
    for i, o in enumerate(ochs):
        assert fin[i]["buffloc"] == bufflocs[i].value and fin[i]["remcode"] == o.remcode and fin[i]["remcarr"] == o.remcarr


def test_closed_loop_tie_in_the_top_binade_state(gc, orc, engine):
    """The single state on which the round-2 closed-loop kernel stalled (tools/debug/loop_hang3.py): a chip step
    that is a rounding tie in the code's top binade, so the period step falls back to the general walkers, and a
    carrier phase of -5794 rad (one piece per period).  Four periods against the oracle."""
    rng = np.random.default_rng(1)
    n = 16368 * 8
    data = rng.integers(-60, 61, size=(n, 2), dtype=np.int8)
    engine.ring_create(1, 2, n)
    engine.ring_push_raw(1, data, n)
    engine.set_channels([gc.Channel(32, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)])
    st = dict(carrfreq=-3560.0113868445246, codefreq=1022997.1414221136, remcode=-0.045902522978210625,
              remcarr=-5793.900519752811, buffloc=5000)
    engine.trk_set_state([st])
    engine.loop_set([engine.loop_state(0, -3600.0)])
    o = orc.make_chan(32, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
    o.acq.acqfreq = -3600.0
    o.carrfreq, o.codefreq, o.remcode, o.remcarr = st["carrfreq"], st["codefreq"], st["remcode"], st["remcarr"]
    o.flagsync, o.synci, o.cnt = 0, 0, 0
    ring = orc.make_ring(data, n, n)
    _check_against_oracle(orc, engine, [o], ring, [C.c_uint64(5000)], 4, 5)


def test_closed_loop_bit_sync_on_the_device(gc, orc, synth, engine):
    """f3: sdrnavigation()'s bit synchronisation and bit decision on the device (ref src/sdrnav.c:18-36,198-282):
    three satellites with 50 bps data, loops closed every period until checksync() reports the bit edge, then every
    10 periods counted from it -- acquisition hand-over state to loop-10 without the host.  PRN 3 goes through the
    vote histogram (more than NAVSYNCTH = 50 sign changes at one position: about 2000 periods), PRN 12 and 25
    through the sign shift register the fork uses for every PRN above 5.  5200 periods per channel: flagsync
    period, decided bits, swloop cadence, sums and remainders bit for bit."""
    prns, dop, cph = [3, 12, 25], [1517.0, -3222.0, 2630.0], [311.3, 12.8, 870.1]
    nper, NS = 5200, 16368
    codes = {p: gc.gencode(p, gc.CTYPE_L1CA) for p in prns}
    rng = np.random.default_rng(5)
    sats = [dict(prn=p, doppler=d, codephase=c, cn0=50.0, phase=0.3 * i, bits=rng.choice([-1.0, 1.0], size=257))
            for i, (p, d, c) in enumerate(zip(prns, dop, cph))]
    sig = synth.make_if(codes, NS * (nper + 3), f_sf=F_SF, f_if=0.0, dtype=2, sats=sats, seed=77)
    nsamples = sig.shape[0]
    engine.ring_create(1, 2, nsamples)
    engine.ring_push_raw(1, sig, nsamples)
    chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in prns]
    engine.set_channels(chans)
    ring = orc.make_ring(sig, nsamples, nsamples)
    ochs, bufflocs, states, loops = [], [], [], []
    for i, c in enumerate(chans):
        acqfreq = 200.0 * round(dop[i] / 200.0)
        o = orc.make_chan(c.prn, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
        o.acq.acqfreq = acqfreq
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = acqfreq, c.crate, 0.0, 0.0
        o.flagsync, o.synci, o.cnt = 0, 0, 1990 + i          # checksync() starts once cnt > 2000 (ref src/sdrnav.c:30)
        b = int(round((1023 - cph[i]) * 16)) % NS
        ochs.append(o)
        bufflocs.append(C.c_uint64(b))
        states.append(dict(carrfreq=acqfreq, codefreq=c.crate, remcode=0.0, remcarr=0.0, buffloc=b))
        loops.append(engine.loop_state(i, acqfreq, flagsync=0, synci=0, cnt=o.cnt))
    engine.trk_set_state(states)
    engine.loop_set(loops)
    done = 0
    for nrun in (1500, 1700, 2000):
        _check_against_oracle(orc, engine, ochs, ring, bufflocs, nrun, 5, done=done)
        done += nrun
    lst = engine.loop_get()
    for i, o in enumerate(ochs):
        assert o.flagsync == 1 and lst[i].flagsync == 1, i             # every channel found its bit edge ...
        for f in ("synci", "biti", "navcnt", "swloop", "bit", "swsync", "swreset", "bitIP", "cnt"):
            assert getattr(lst[i], f) == getattr(o, f), (i, f)
        assert list(lst[i].bitsync) == list(o.bitsync), i
        assert abs(o.carrfreq - dop[i]) < 30.0, (i, o.carrfreq)       # ... and the loops stayed locked on it


def test_acquisition_state_to_frame_and_observables(gc, orc, synth, engine):
    """The whole thread loop of a channel on a signal that carries a navigation subframe (ref src/sdrmain.c:264-312,
    src/sdrnav.c:15-84): from the state sdracquisition() leaves, the device runs the one-period loop, finds the bit
    edge (checksync), switches to the ten-period loop and decides the bits (checkbit); the host then finds the frame
    in the log's bits (gnsscorr_frame_replay: preamble, parity, hand-over word) and replays setobsdata()
    (gnsscorr_obs_replay).  4.092 Msps, 8.6 s of signal, one satellite at +1234 Hz."""
    from lnav_frames import l1ca_subframe
    f_sf, nsamp, prn, dop, cph = 4.092e6, 4092, 7, 1234.0, 200.25
    rng = np.random.default_rng(99)
    lead = [int(x) for x in rng.choice([-1, 1], size=115)]
    tow_count = 70001
    sf, _ = l1ca_subframe(rng, [lead[-2], lead[-1]], tow_count, 2, 1)
    bits = np.array(lead + sf + [int(x) for x in rng.choice([-1, 1], size=15)], np.float64)
    nper = 8580
    codes = {prn: gc.gencode(prn, gc.CTYPE_L1CA)}
    sats = [dict(prn=prn, doppler=dop, codephase=cph, cn0=50.0, phase=0.7, bits=bits)]
    sig = synth.make_if(codes, nsamp * (nper + 4), f_sf=f_sf, f_if=0.0, dtype=2, sats=sats, seed=7)
    engine.ring_create(1, 2, sig.shape[0])
    engine.ring_push_raw(1, sig, sig.shape[0])
    ch = gc.Channel(prn, dtype=2, f_sf=f_sf, f_if=0.0, corrn=2, corrd=1, corrp=1)
    assert ch.nsamp == nsamp
    engine.set_channels([ch])
    acqfreq = 200.0 * round(dop / 200.0)
    b0 = int(round((1023 - cph) * 4)) % nsamp
    engine.trk_set_state([dict(carrfreq=acqfreq, codefreq=ch.crate, remcode=0.0, remcarr=0.0, buffloc=b0)])
    engine.loop_set([engine.loop_state(0, acqfreq, flagsync=0, synci=0, cnt=0)])
    engine.trk_run_loop(nper)
    II, QQ, ns = engine.trk_fetch()
    log, ndone = engine.trk_fetch_log()
    assert ndone[0] == nper
    rows = log[0]
    # bit synchronisation on the device: after period 2000, then the ten-period loop
    first_sync = int(np.argmax(rows["flagsync"] != 0))
    assert 2000 < first_sync < 2400 and np.all(rows["flagsync"][first_sync:] == 1)
    assert np.all(rows["flagloopfilter"][:first_sync] == 1) and set(np.unique(rows["flagloopfilter"][first_sync + 20:])) == {0, 2}
    decided = rows["navbit"][rows["navbit"] != 0]
    assert len(decided) >= 300 + 5
    # the frame in the decided bits
    fr = gc.FrameState()
    gc.frame_replay(fr, rows, cnt0=0)
    end = (115 + 300) * 20                                      # the period in which the subframe's last bit ends
    assert fr.flagtow == 1 and fr.flagsyncf == 1 and fr.flagdec == 1 and fr.polarity in (1, -1)
    assert abs(int(fr.firstsfcnt) - end) <= 25 and fr.sfid == 2 and fr.firstsftow == tow_count * 6.0
    assert fr.firstsf == int(rows["buffloc"][int(fr.firstsfcnt)])
    # the observables, with what the frame decoder found
    st = gc.ObsState()
    st.f_sf, st.f_if, st.foffset, st.ctime, st.loopms = f_sf, 0.0, 0.0, 1e-3, 10
    st.flagsyncf, st.polarity, st.firstsftow, st.firstsfcnt = fr.flagsyncf, fr.polarity, fr.firstsftow, fr.firstsfcnt
    obs = gc.obs_replay(st, rows, II[0, :, 0], cnt0=0)
    nupd = int((rows["flagloopfilter"] == 2).sum())
    assert len(obs) == nupd and nupd > 550
    assert np.all(np.abs(obs["D"][-200:] + dop) < 25.0)          # Doppler = -(carrfreq - f_if)
    tail = obs[obs["cntout"] >= fr.firstsfcnt]                  # (before the frame's end the reference's tow is meaningless too)
    assert len(tail) >= 20
    assert np.allclose(np.diff(tail["tow"]), 0.01, atol=1e-9) and np.allclose(np.diff(tail["L"]), tail["D"][1:] * 0.01, rtol=1e-3)
    k = int(np.argmax(obs["cntout"] >= fr.firstsfcnt))
    assert abs(obs["tow"][k] - (tow_count * 6.0 + (int(obs["cntout"][k]) - int(fr.firstsfcnt)) * 1e-3)) < 1e-9
    assert int(obs["snr"].sum()) == (nupd + 9) // 10 and np.all(obs["S"][obs["snr"] == 1][5:] > 30.0)
