"""Host C side of the boundary (no GPU needed): INI surface, code generation,
channel set-up, loop filters -- product (libgnsscorr.so, host C) against the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _write_inis(tmp_path, rcv_extra=""):
    fe = tmp_path / "fe.ini"
    dat = tmp_path / "if.dat"
    dat.write_bytes(b"\0" * 64)
    fe.write_text(open(os.path.join(HERE, "golden", "frontend_file.ini")).read().replace("@FILE1@", str(dat)))
    rcv = tmp_path / "gnss-sdrcli.ini"
    rcv.write_text(open(os.path.join(HERE, "golden", "gnss-sdrcli.ini")).read().replace("@FENDCONF@", str(fe)) + rcv_extra)
    return rcv, fe, dat


def test_readinifile_keys_and_quirks(gc, tmp_path):
    """Same keys / parsing rules as ref src/sdrinit.c:17-46,106-211."""
    rcv, fe, dat = _write_inis(tmp_path)
    ini = gc.SdrIni()
    assert gc.lib().readinifile_at(C.byref(ini), str(rcv).encode()) == 0
    assert ini.fend == 10 and ini.useif1 == 1 and ini.useif2 == 0
    assert ini.file1.decode() == str(dat)
    assert (ini.f_cf[0], ini.f_sf[0], ini.f_if[0], ini.dtype[0]) == (1575.42e6, 16.368e6, 4.092e6, 1)
    assert (ini.trkcorrn, ini.trkcorrd, ini.trkcorrp) == (6, 3, 6)
    assert list(ini.trkdllb) == [5.0, 1.0] and list(ini.trkpllb) == [30.0, 10.0] and list(ini.trkfllb) == [200.0, 50.0]
    assert ini.nch == 4 and list(ini.prn)[:4] == [1, 2, 3, 120] and list(ini.ctype)[:4] == [1, 1, 1, 27]
    assert ini.nchL1 == 3 and ini.log == 1 and ini.outms == 400 and ini.rtcmport == 9999
    assert gc.lib().chk_initvalue(C.byref(ini)) == 0
    # ';' starts a comment; the first match in the section wins; keys are right-trimmed
    fe.write_text(fe.read_text().replace("CORRN    =6", "CORRN\t =4 ;was 6\nCORRN    =9"))
    ini2 = gc.SdrIni()
    assert gc.lib().readinifile_at(C.byref(ini2), str(rcv).encode()) == 0
    assert ini2.trkcorrn == 4


def test_readinifile_errors(gc, tmp_path, capfd):
    ini = gc.SdrIni()
    assert gc.lib().readinifile_at(C.byref(ini), str(tmp_path / "nope.ini").encode()) == -1
    rcv, fe, dat = _write_inis(tmp_path)
    fe.write_text(fe.read_text().replace("TYPE     =FILE", "TYPE     =FOO"))
    assert gc.lib().readinifile_at(C.byref(ini), str(rcv).encode()) == -1
    rcv.write_text(rcv.read_text().replace("NCH      =  4", "NCH      =  0"))
    fe.write_text(fe.read_text().replace("TYPE     =FOO", "TYPE     =FILE"))
    assert gc.lib().readinifile_at(C.byref(ini), str(rcv).encode()) == -1
    C.CDLL(None).fflush(None)          # the library prints through C stdio
    out = capfd.readouterr().out
    assert "doesn't exist" in out and "wrong frontend type: FOO" in out and "wrong inifile value NCH=0" in out


def test_gencode_product_vs_oracle_and_known_answers(gc, orc):
    kat = json.load(open(os.path.join(HERE, "golden", "ca_first10_octal.json")))["first10_octal"]
    for prn in list(range(1, 38)) + [120, 138, 193, 210]:
        a, cra = gc.gencode(prn, gc.CTYPE_L1SBAS if prn >= 120 else gc.CTYPE_L1CA)
        b, crb = orc.gencode(prn, 1)
        assert np.array_equal(a, b) and cra == crb
        if str(prn) in kat:
            bits = "".join("1" if c == 1 else "0" for c in a[:10])
            assert format(int(bits, 2), "o") == kat[str(prn)]
    a, cr = gc.gencode(3, gc.CTYPE_G1)
    b, _ = orc.gencode(3, 20)
    assert np.array_equal(a, b) and cr == 0.511e6


def test_initsdrch_matches_reference_shapes(gc, orc):
    """n = 16368, nsampchip = 16, nfft = 32736, nfreq = 71, +-7 kHz / 200 Hz grid."""
    L = gc.lib()
    ini = gc.sdrini()
    ini.trkcorrn, ini.trkcorrd, ini.trkcorrp = 6, 3, 6
    ini.trkdllb[0], ini.trkdllb[1] = 5.0, 1.0
    ini.trkpllb[0], ini.trkpllb[1] = 30.0, 10.0
    ini.trkfllb[0], ini.trkfllb[1] = 200.0, 50.0
    sdr = gc.SdrCh()
    assert L.initsdrch(1, gc.SYS_GPS, 7, gc.CTYPE_L1CA, 1, 1, 1575.42e6, 16.368e6, 4.092e6, C.byref(sdr)) == 0
    assert (sdr.nsamp, sdr.nsampchip, sdr.clen, sdr.acq.nfft, sdr.acq.nfreq, sdr.acq.intg) == (16368, 16, 1023, 32736, 71, 10)
    assert sdr.satstr == b"G07" and sdr.sat == 7 and sdr.ci == 0.0625 and sdr.ctime == 1e-3
    freq = np.ctypeslib.as_array(sdr.acq.freq, shape=(71,))
    assert freq[0] == 4.092e6 - 7000 and freq[35] == 4.092e6 and freq[70] == 4.092e6 + 7000
    assert (sdr.trk.corrn, sdr.trk.ne, sdr.trk.nl, sdr.trk.loopms) == (6, 3, 4, 10)
    assert list(np.ctypeslib.as_array(sdr.trk.corrp, shape=(6,))) == [3, 6, 9, 12, 15, 18]
    o = orc.make_chan(7, dtype=1, f_if=4.092e6, corrn=6, corrd=3, corrp=6)
    assert sdr.trk.prm1.pllw2 == o.pllw2[0] and sdr.trk.prm2.dllaw == o.dllaw[1] and sdr.trk.prm1.fllw == o.fllw[0]
    assert np.array_equal(np.ctypeslib.as_array(sdr.code, shape=(1023,)), np.ctypeslib.as_array(o.code))
    # python mirror agrees with the C initsdrch
    ch = gc.Channel(7, dtype=1, f_if=4.092e6, corrn=6, corrd=3, corrp=6)
    assert (ch.nsamp, ch.nsampchip, ch.nfreq, ch.nfft, ch.ne, ch.nl) == (16368, 16, 71, 32736, 3, 4)
    assert np.array_equal(ch.freq, freq)
    L.freesdrch(C.byref(sdr))
    # GLONASS FDMA offsets (ref src/sdrinit.c:612-615); SBAS satellite numbering
    g = gc.SdrCh()
    assert L.initsdrch(2, gc.SYS_GLO, -3, gc.CTYPE_G1, 2, 2, 1602e6, 20e6, 0.0, C.byref(g)) == 0
    assert g.satstr == b"R-3" and g.foffset == -3 * 0.5625e6 and g.clen == 511 and g.nsamp == 20000
    L.freesdrch(C.byref(g))
    s = gc.SdrCh()
    assert L.initsdrch(3, gc.SYS_SBS, 120, gc.CTYPE_L1SBAS, 1, 1, 1575.42e6, 16.368e6, 4.092e6, C.byref(s)) == 0
    assert s.satstr == b"120" and s.sat == 33 and s.trk.loopms == 2
    L.freesdrch(C.byref(s))


def test_loop_filters_product_vs_oracle(gc, orc):
    L = gc.lib()
    ini = gc.sdrini()
    ini.trkcorrn, ini.trkcorrd, ini.trkcorrp = 2, 3, 3
    sdr = gc.SdrCh()
    assert L.initsdrch(1, gc.SYS_GPS, 1, gc.CTYPE_L1CA, 2, 1, 1575.42e6, 16.368e6, 0.0, C.byref(sdr)) == 0
    o = orc.make_chan(1, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
    rng = np.random.default_rng(0)
    sdr.acq.acqfreq = o.acq.acqfreq = 1400.0
    sdr.trk.carrfreq = o.carrfreq = 1400.0
    sdr.trk.codefreq = o.codefreq = o.crate
    for step in range(20):
        vals = rng.uniform(-4000, 4000, size=(4, 5))
        for i in range(5):
            sdr.trk.II[i] = o.II[i] = vals[0, i]
            sdr.trk.QQ[i] = o.QQ[i] = vals[1, i]
            sdr.trk.oldI[i] = o.oldI[i] = vals[2, i]
            sdr.trk.oldQ[i] = o.oldQ[i] = vals[3, i]
        L.cumsumcorr(C.byref(sdr.trk), 1)
        orc.lib().orc_cumsumcorr(C.byref(o), 1)
        prm = sdr.trk.prm1 if step < 10 else sdr.trk.prm2
        dt = 1e-3 if step < 10 else 1e-2
        L.pll(C.byref(sdr), C.byref(prm), dt)
        L.dll(C.byref(sdr), C.byref(prm), dt)
        orc.lib().orc_pll(C.byref(o), 0 if step < 10 else 1, dt)
        orc.lib().orc_dll(C.byref(o), 0 if step < 10 else 1, dt)
        assert sdr.trk.carrfreq == o.carrfreq and sdr.trk.codefreq == o.codefreq
        assert sdr.trk.carrNco == o.carrNco and sdr.trk.codeErr == o.codeErr and sdr.trk.freqErr == o.freqErr
        if step % 3 == 2:
            L.clearcumsumcorr(C.byref(sdr.trk))
            orc.lib().orc_clearcumsumcorr(C.byref(o))
    L.freesdrch(C.byref(sdr))


def test_setobsdata_product_vs_oracle(gc, orc):
    """setobsdata() (ref src/sdrtrk.c:160-209) on the reference's sdrch_t against the oracle's restatement, 250 calls
    with random loop outputs: Doppler, accumulated carrier phase, remaining code phase, the SNR of every 10th call, and
    the histories moving down by one."""
    L = gc.lib()
    sdr = gc.SdrCh()
    assert L.initsdrch(1, gc.SYS_GPS, 1, gc.CTYPE_L1CA, 2, 1, 1575.42e6, 16.368e6, 0.0, C.byref(sdr)) == 0
    o = orc.make_chan(1, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
    L.setobsdata.argtypes = [C.POINTER(gc.SdrCh), C.c_uint64, C.c_uint64, C.POINTER(gc.SdrTrk), C.c_int]
    L.setobsdata.restype = None
    rng = np.random.default_rng(77)
    buffloc, prevL = 1000, 0.0
    for k in range(250):
        cf, df = 1400.0 + rng.uniform(-30, 30), o.crate + rng.uniform(-2, 2)
        rc, rp, n, si = rng.uniform(0, 1), rng.uniform(-50, 6.2), int(rng.integers(16366, 16371)), rng.uniform(-5e4, 5e4)
        sdr.trk.carrfreq = o.carrfreq = cf
        sdr.trk.codefreq = o.codefreq = df
        sdr.trk.oldremcode = o.oldremcode = rc
        sdr.trk.remcarr = o.remcarr = rp
        sdr.currnsamp = o.currnsamp = n
        sdr.trk.sumI[0] = o.sumI[0] = si
        snr = 1 if k % 10 == 0 else 0
        L.setobsdata(C.byref(sdr), buffloc, 2001 + 10 * k, C.byref(sdr.trk), snr)
        orc.lib().orc_setobsdata(C.byref(o), buffloc, 2001 + 10 * k, snr)
        assert sdr.trk.L[0] == o.obs_L and sdr.trk.D[0] == o.obs_D and sdr.trk.remcout[0] == o.obs_remcout
        assert sdr.trk.tow[0] == o.obs_tow and sdr.trk.codei[0] == o.obs_codei and sdr.trk.cntout[0] == o.obs_cntout
        assert sdr.trk.Isum == o.obs_Isum and sdr.trk.S[0] == o.obs_S
        if k:
            assert sdr.trk.L[1] == prevL and sdr.trk.cntout[1] == 2001 + 10 * (k - 1)
        prevL = sdr.trk.L[0]
        buffloc += 10 * n
    assert sdr.trk.flagremcarradd == 1 and o.obs_nsnr == 25
    L.freesdrch(C.byref(sdr))


def test_observables_replayed_over_a_loop_log(gc, orc, synth):
    """gnsscorr_obs_replay: setobsdata() over the per-period log of a closed-loop run equals setobsdata() called inside
    the loop where the reference calls it (ref src/sdrmain.c:277-288; here the oracle's thread loop on a synthetic
    signal, nav bit synchronised, an inverted frame reported half way) -- bit for bit, in one piece and in two."""
    NS, NP = 16368, 420
    rng = np.random.default_rng(5)
    data = rng.integers(-40, 41, size=((NP + 4) * NS, 2), dtype=np.int8)
    ring = orc.make_ring(data, data.shape[0], data.shape[0])
    o = orc.make_chan(3, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
    o.acq.acqfreq = o.carrfreq = 1200.0
    o.codefreq = o.crate
    o.remcode, o.remcarr = 0.25, 1.5
    o.flagacq = 1
    o.flagsync, o.synci, o.cnt, o.prn = 1, 7, 2001, 3
    st = gc.ObsState()
    st.f_sf, st.f_if, st.foffset, st.ctime, st.loopms = o.f_sf, o.f_if, o.foffset, o.ctime, o.loopms
    st.oldremcode = o.remcode
    st.firstsftow, st.firstsfcnt = 345600.0, 1500
    o.firstsftow, o.firstsfcnt = 345600.0, 1500
    log = np.zeros(NP, dtype=np.dtype(gc.TrkLog))
    II0 = np.zeros(NP)
    want = []
    buffloc = C.c_uint64(11)
    for p in range(NP):
        if p == NP // 2:            # the frame decoder has found an inverted preamble
            o.flagsyncf, o.polarity = 1, 1
        n_before = o.obs_n
        b0 = buffloc.value
        assert orc.lib().orc_sdrthread_step(C.byref(o), C.byref(ring), C.byref(buffloc)) == 1
        log[p]["carrfreq"], log[p]["codefreq"] = o.carrfreq, o.codefreq
        log[p]["remcode"], log[p]["remcarr"] = o.remcode, o.remcarr
        log[p]["buffloc"], log[p]["currnsamp"], log[p]["flagloopfilter"] = b0, o.currnsamp, o.flagloopfilter
        II0[p] = o.II[0]
        if o.obs_n != n_before:
            want.append((o.obs_tow, o.obs_remcout, o.obs_L, o.obs_D, o.obs_S if o.obs_Isum == 0.0 else None, o.obs_codei,
                         o.obs_cntout))
    assert len(want) >= 40
    # in two pieces, the frame decoder's report handed over between them
    half = NP // 2
    rows = list(gc.obs_replay(st, log[:half], II0[:half], cnt0=2001))
    st.flagsyncf, st.polarity = 1, 1
    rows += list(gc.obs_replay(st, log[half:], II0[half:], cnt0=2001 + half))
    assert len(rows) == len(want)
    for r, w in zip(rows, want):
        assert (r["tow"], r["remcout"], r["L"], r["D"], int(r["codei"]), int(r["cntout"])) == (w[0], w[1], w[2], w[3], w[5], w[6])
        if w[4] is not None:
            assert r["snr"] == 1 and r["S"] == w[4]
        else:
            assert r["snr"] == 0
    assert any(r["snr"] for r in rows) and st.flagpolarityadd == 1


from lnav_frames import l1ca_subframe as _l1ca_subframe


def test_frame_sync_on_batched_nav_bits(gc, orc):
    """gnsscorr_frame_replay (preamble search, parity over the ten words, subframe number and time of week of the
    hand-over word: ref src/sdrnav.c:41-82) against the oracle's restatement, bit by bit: random bits (no false lock
    that the oracle does not share), then three consecutive subframes for either polarity of the bit stream."""
    rng = np.random.default_rng(2024)
    for polarity in (1, -1):
        bits = [int(x) for x in rng.choice([-1, 1], size=777)]
        prev2 = [polarity * bits[-2], polarity * bits[-1]]
        tow0 = 34567
        for k in range(3):
            sf, prev2 = _l1ca_subframe(rng, prev2, tow0 + k, 1 + k, polarity)
            bits += sf
        bits += [int(x) for x in rng.choice([-1, 1], size=40)]
        # one decided bit every 20 periods (rate 20), the first at period 7
        nper = 7 + 20 * len(bits)
        log = np.zeros(nper, dtype=np.dtype(gc.TrkLog))
        for i, bv in enumerate(bits):
            log[7 + 20 * i]["navbit"] = bv
            log[7 + 20 * i]["buffloc"] = 1000 + 16368 * (7 + 20 * i)
        of = orc.Frame()
        st = gc.FrameState()
        cnt0, done = 5000, 0
        sync_at = None
        for chunk in (1, 4000, 9000, nper - 13001):                 # in pieces: the state carries over
            gc.frame_replay(st, log[done:done + chunk], cnt0=cnt0 + done)
            for p in range(done, done + chunk):
                orc.lib().orc_navframe_l1ca(C.byref(of), int(log[p]["navbit"]), int(log[p]["buffloc"]), cnt0 + p)
                if of.flagtow and sync_at is None:
                    sync_at = p
            done += chunk
            assert list(st.fbits) == list(of.fbits)
            for f in ("polarity", "flagsyncf", "flagtow", "flagdec", "sfid", "firstsf", "firstsfcnt", "firstsftow", "tow_gpst"):
                assert getattr(st, f) == getattr(of, f), (polarity, done, f)
        # locked at the last bit of the first constructed subframe, and never lost
        first_end = 7 + 20 * (777 + 300 - 1)
        assert sync_at == first_end and st.flagtow == 1 and st.flagsyncf == 1 and st.firstsfcnt == cnt0 + first_end
        assert st.firstsf == 1000 + 16368 * first_end
        assert st.flagdec == 1 and st.firstsftow == tow0 * 6.0 and st.sfid == 3 and st.tow_gpst == (tow0 + 2) * 6.0
