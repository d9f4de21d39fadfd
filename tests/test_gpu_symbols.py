"""The reference-named per-call symbols of libgnsscorr.so (include/sdr_compat.h) on the GPU,
against the oracle.  These are the drop-in replacements a maintainer links instead of
src/sdracq.c, src/sdrtrk.c and the DSP helpers of src/sdrcmn.c."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TI = 1 / 16.368e6


def test_mixcarr_rescode_cpxcpx(gc, orc):
    L, O = gc.lib(), orc.lib()
    rng = np.random.default_rng(0)
    # incl. the acquisition hand-over state (phase 0, bin-centre frequencies: ref src/sdracq.c:51-55) and
    # a negative, never-wrapped phase (ref src/sdrcmn.c:667)
    for dtype, freq, phi0 in ((2, 1234.5, 0.4), (1, 4.092e6 - 700.0, 0.0), (2, -3.9e6, 2.0), (2, 2200.0, 0.0),
                              (2, -1400.0, 0.0), (1, 4.0932e6, 0.0), (2, -3400.0, -12345.678), (2, 137.77, 6.1)):
        n = 16369
        data = rng.integers(-128, 128, size=n * dtype, dtype=np.int8)
        I, Q, oI, oQ = (np.zeros(n, np.int16) for _ in range(4))
        r = L.mixcarr(data.ctypes.data, dtype, TI, n, freq, phi0, I.ctypes.data, Q.ctypes.data)
        ro = O.orc_mixcarr_seq(data.ctypes.data, dtype, TI, n, freq, phi0, oI.ctypes.data, oQ.ctypes.data)
        assert np.array_equal(I, oI) and np.array_equal(Q, oQ) and r == ro
    code, crate = orc.gencode(3, 1)
    # incl. an integer code phase with a non-dyadic chip step: the chip choice hinges on the rounding of
    # the reference's running sum (ref src/sdrcmn.c:616-619)
    for coff, smax, dc in ((0.0, 0, 0.0), (100.25, 6, 1.5), (1022.9, 18, -2.0), (0.0, 6, 1.7), (0.0, 18, -2.3),
                           (512.0, 6, 0.4), (1022.9999999, 3, 2.9)):
        n = 16368
        a, b = np.zeros(n + 2 * smax, np.int16), np.zeros(n + 2 * smax, np.int16)
        ci = TI * (crate + dc)
        r = L.rescode(code.ctypes.data, 1023, coff, smax, ci, n, a.ctypes.data)
        ro = O.orc_rescode_seq(code.ctypes.data, 1023, coff, smax, ci, n, b.ctypes.data)
        assert np.array_equal(a, b) and r == ro
    I = rng.integers(-8000, 8000, size=1000).astype(np.int16)
    Q = rng.integers(-8000, 8000, size=1000).astype(np.int16)
    x, xo = np.zeros(2000, np.float32), np.zeros(2000, np.float32)
    L.cpxcpx(I.ctypes.data, Q.ctypes.data, (1 / 32) / 32736, 1000, x.ctypes.data)
    O.orc_cpxcpx(I.ctypes.data, Q.ctypes.data, (1 / 32) / 32736, 1000, xo.ctypes.data)
    assert np.array_equal(x, xo)
    L.cpxcpx(I.ctypes.data, None, 1.0, 1000, x.ctypes.data)
    assert np.all(x[1::2] == 0) and np.array_equal(x[0::2], I.astype(np.float32))


@pytest.mark.parametrize("n", [1000, 4092, 16384, 32736, 32768])
def test_cpxfft_cpxpspec_any_length(gc, orc, n):
    L = gc.lib()
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    y = x.copy()
    L.cpxfft(None, y.ctypes.data, n)
    assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-6
    z = y.copy()
    L.cpxifft(None, z.ctypes.data, n)
    assert np.abs(z / n - x).max() < 2e-5
    ps = np.full(n, 1.5)
    xx = x.copy()
    L.cpxpspec(None, xx.ctypes.data, n, 1, ps.ctypes.data)           # flagsum: accumulates
    assert rel_err(ps - 1.5, np.abs(ref) ** 2) < 1e-5
    xx = x.copy()
    L.cpxpspec(None, xx.ctypes.data, n, 0, ps.ctypes.data)
    assert rel_err(ps, np.abs(ref) ** 2) < 1e-5


def test_cpxconv_and_pcorrelator_reference_length(gc, orc):
    """m = 2*nsamp = 32736 exactly as the reference calls them (ref src/sdrcmn.c:738-773)."""
    L, O = gc.lib(), orc.lib()
    rng = np.random.default_rng(1)
    o = orc.make_chan(6, dtype=2, f_if=0.0)
    n, m = o.nsamp, o.nfft
    xc = orc.codespectrum(o)
    data = rng.integers(-60, 61, size=2 * n * 2, dtype=np.int8)
    freq = np.array([-3000.0, 200.0, 5200.0])
    P = np.full(3 * n, 0.25)
    Po = P.copy()
    L.pcorrelator(data.ctypes.data, 2, o.ti, n, freq.ctypes.data, 3, o.crate, m, xc.ctypes.data, P.ctypes.data)
    O.orc_pcorrelator(data.ctypes.data, 2, o.ti, n, freq.ctypes.data, 3, o.crate, m, xc.ctypes.data,
                      Po.ctypes.data)
    assert rel_err(P, Po) < 1e-4
    a = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
    b = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
    a2, conv, convo = a.copy(), np.zeros(n), np.zeros(n)
    L.cpxconv(None, None, a.ctypes.data, b.ctypes.data, m, n, 0, conv.ctypes.data)
    O.orc_cpxconv(a2.ctypes.data, b.ctypes.data, m, n, 0, convo.ctypes.data)
    assert rel_err(conv, convo) < 1e-4
    assert np.abs(a - a2).max() / np.abs(a2).max() < 1e-5      # cpxa holds the inverse transform afterwards


def test_maxvd_meanvd_checkacquisition(gc, orc):
    L, O = gc.lib(), orc.lib()
    rng = np.random.default_rng(2)
    d = rng.uniform(0, 1, 16368)
    d[0] = 2.0
    d[4000] = d[9000] = 1.7
    for exs, exe in ((-1, -1), (0, 40), (16300, 30), (3990, 4010)):
        i1, i2 = C.c_int(), C.c_int()
        assert L.maxvd(d.ctypes.data, len(d), exs, exe, C.byref(i1)) == O.orc_maxvd(d.ctypes.data, len(d), exs, exe, C.byref(i2))
        assert i1.value == i2.value
        assert L.meanvd(d.ctypes.data, len(d), exs, exe) == pytest.approx(O.orc_meanvd(d.ctypes.data, len(d), exs, exe), rel=1e-12)
    ini = gc.sdrini()
    ini.trkcorrn, ini.trkcorrd, ini.trkcorrp = 2, 3, 3
    sdr = gc.SdrCh()
    assert L.initsdrch(1, gc.SYS_GPS, 5, gc.CTYPE_L1CA, 2, 1, 1575.42e6, 16.368e6, 0.0, C.byref(sdr)) == 0
    P = rng.uniform(0.5, 1.0, 71 * 16368)
    P[40 * 16368 + 7] = 30.0
    P[40 * 16368 + 9000] = 6.0
    got = L.checkacquisition(P.ctypes.data, C.byref(sdr))
    res = orc.AcqRes()
    freq = np.ctypeslib.as_array(sdr.acq.freq, shape=(71,)).copy()
    want = O.orc_checkacquisition(P.ctypes.data, 16368, 71, 16, 1e-3, freq.ctypes.data, C.byref(res))
    assert got == want == 1
    assert (sdr.acq.acqcodei, sdr.acq.freqi, sdr.acq.acqfreq) == (res.acqcodei, res.freqi, res.acqfreq)
    assert sdr.acq.peakr == pytest.approx(res.peakr, rel=1e-12) and sdr.acq.cn0 == pytest.approx(res.cn0, rel=1e-12)
    L.freesdrch(C.byref(sdr))


def test_sdracquisition_then_sdrtracking_like_sdrthread(gc, orc, synth, tmp_path):
    """The reference's channel loop (ref src/sdrmain.c:247-316) on the drop-in symbols: file front end ->
    ring -> sdracquisition -> sdrtracking + cumsumcorr + pll/dll, against the oracle doing the same."""
    os.environ["GNSSCORR_ACQSLEEP_MS"] = "0"
    L, O = gc.lib(), orc.lib()
    prn = 14
    codes = {prn: gc.gencode(prn, 1)}
    sats = [dict(prn=prn, doppler=2210.0, codephase=512.7, cn0=48.0, phase=0.3)]
    nblocks, nmore = 8, 4                      # pushed before acquisition / while tracking
    nsamples = nblocks * 65536
    ntotal = (nblocks + nmore) * 65536
    data = synth.make_if(codes, ntotal, f_if=4.092e6, dtype=1, sats=sats, seed=21)
    f = tmp_path / "if.dat"
    data.tofile(f)
    ini = gc.sdrini()
    ini.fend, ini.useif1, ini.useif2 = 10, 1, 0
    ini.file1 = str(f).encode()
    ini.dtype[0], ini.f_sf[0], ini.f_if[0], ini.f_cf[0] = 1, 16.368e6, 4.092e6, 1575.42e6
    ini.trkcorrn, ini.trkcorrd, ini.trkcorrp = 6, 3, 6
    for k, v in (("trkdllb", (5.0, 1.0)), ("trkpllb", (30.0, 10.0)), ("trkfllb", (200.0, 50.0))):
        getattr(ini, k)[0], getattr(ini, k)[1] = v
    ini.fp1 = None
    assert L.rcvinit_file(C.byref(ini)) == 0
    for _ in range(nblocks):
        L.file_pushtomembuf()
    st = gc.sdrstat()
    assert st.buffcnt == nblocks and st.fendbuffsize == 65536

    sdr = gc.SdrCh()
    assert L.initsdrch(1, gc.SYS_GPS, prn, gc.CTYPE_L1CA, 1, 1, 1575.42e6, 16.368e6, 4.092e6, C.byref(sdr)) == 0
    power = np.zeros(71 * 16368)
    buffloc = L.sdracquisition(C.byref(sdr), power.ctypes.data)

    # oracle on the same ring
    ringlen = 5000 * 65536
    big = np.ascontiguousarray(data)
    o = orc.make_chan(prn, dtype=1, f_if=4.092e6, corrn=6, corrd=3, corrp=6)
    xc = orc.codespectrum(o)
    o.xcode = xc.ctypes.data
    ring = orc.Ring()
    ring.buff, ring.ringlen, ring.wrpos = big.ctypes.data, ringlen, nsamples
    opower = np.zeros(71 * 16368)
    it = C.c_int()
    obuffloc = O.orc_sdracquisition(C.byref(o), C.byref(ring), opower.ctypes.data, C.byref(it))
    assert sdr.flagacq == o.flagacq == 1 and buffloc == obuffloc
    assert (sdr.acq.acqcodei, sdr.acq.freqi, sdr.acq.acqfreq) == (o.acq.acqcodei, o.acq.freqi, o.acq.acqfreq)
    assert abs(sdr.acq.acqfreq - 4.092e6 - 2210.0) <= 100.0
    assert rel_err(power, opower) < 1e-4
    assert sdr.trk.carrfreq == o.carrfreq and sdr.trk.codefreq == o.codefreq

    # the grabber keeps delivering blocks while the channel tracks
    for _ in range(nmore):
        L.file_pushtomembuf()
    ring.wrpos = ntotal
    # tracking loop before bit sync: pll/dll every code period (ref src/sdrmain.c:272-276)
    cnt, IP = 0, []
    for _ in range(12):
        L.sdrtracking(C.byref(sdr), buffloc, cnt)
        O.orc_sdrtracking(C.byref(o), C.byref(ring), obuffloc)
        assert sdr.flagtrk == o.flagtrk == 1 and sdr.currnsamp == o.currnsamp
        for t in range(13):
            assert sdr.trk.II[t] == o.II[t] and sdr.trk.QQ[t] == o.QQ[t]
        assert sdr.trk.remcode == o.remcode and sdr.trk.remcarr == o.remcarr
        L.cumsumcorr(C.byref(sdr.trk), 1)
        O.orc_cumsumcorr(C.byref(o), 1)
        L.pll(C.byref(sdr), C.byref(sdr.trk.prm1), sdr.ctime)
        L.dll(C.byref(sdr), C.byref(sdr.trk.prm1), sdr.ctime)
        O.orc_pll(C.byref(o), 0, o.ctime)
        O.orc_dll(C.byref(o), 0, o.ctime)
        assert sdr.trk.carrfreq == o.carrfreq and sdr.trk.codefreq == o.codefreq
        IP.append(sdr.trk.II[0] ** 2 + sdr.trk.QQ[0] ** 2)
        L.clearcumsumcorr(C.byref(sdr.trk))
        O.orc_clearcumsumcorr(C.byref(o))
        buffloc += sdr.currnsamp
        obuffloc += o.currnsamp
        cnt += 1
    # the prompt correlator sits on the signal: power far above the noise floor of an absent PRN
    assert min(IP) > 100 * (8.0 ** 2) * 16368 / 32 ** 2
    # not enough samples buffered yet -> flagtrk 0 and nothing touched
    before = sdr.trk.remcode
    L.sdrtracking(C.byref(sdr), ntotal, cnt)
    assert sdr.flagtrk == 0 and sdr.trk.remcode == before
    L.freesdrch(C.byref(sdr))


def test_sdrtracking_from_32_concurrent_threads(gc, orc, synth, tmp_path):
    """The reference's threading model (ref src/sdrmain.c:144-149): one thread per channel, every one of them
    calling sdrtracking() + cumsumcorr() + pll() + dll() once per code period, concurrently, on its own
    sdrch_t.  32 channels x 40 periods; every channel equal to the oracle bit for bit; the call rate is
    printed (1x real time = 32 000 calls/s)."""
    import threading
    import time
    L, O = gc.lib(), orc.lib()
    nch, nper = 32, 40
    prns = list(range(1, nch + 1))
    codes = {p: gc.gencode(p, 1) for p in prns}
    rng = np.random.default_rng(77)
    sats = [dict(prn=p, doppler=float(rng.uniform(-4000, 4000)), codephase=float(rng.uniform(0, 1023)),
                 cn0=46.0, phase=float(rng.uniform(0, 6.28))) for p in prns[::4]]
    nblocks = 12
    ntotal = nblocks * 65536
    data = synth.make_if(codes, ntotal, f_if=4.092e6, dtype=1, sats=sats, seed=23)
    f = tmp_path / "if32.dat"
    data.tofile(f)
    ini = gc.sdrini()
    ini.fend, ini.useif1, ini.useif2 = 10, 1, 0
    ini.file1 = str(f).encode()
    ini.dtype[0], ini.f_sf[0], ini.f_if[0], ini.f_cf[0] = 1, 16.368e6, 4.092e6, 1575.42e6
    ini.trkcorrn, ini.trkcorrd, ini.trkcorrp = 2, 3, 3
    for k, v in (("trkdllb", (5.0, 1.0)), ("trkpllb", (30.0, 10.0)), ("trkfllb", (200.0, 50.0))):
        getattr(ini, k)[0], getattr(ini, k)[1] = v
    ini.fp1 = None
    assert L.rcvinit_file(C.byref(ini)) == 0
    for _ in range(nblocks):
        L.file_pushtomembuf()
    sdrs, starts = [], []
    for i, p in enumerate(prns):
        sdr = gc.SdrCh()
        assert L.initsdrch(i + 1, gc.SYS_GPS, p, gc.CTYPE_L1CA, 1, 1, 1575.42e6, 16.368e6, 4.092e6, C.byref(sdr)) == 0
        sdr.flagacq = 1
        sdr.acq.acqfreq = 4.092e6 + 200.0 * int(rng.integers(-20, 21))
        sdr.trk.carrfreq, sdr.trk.codefreq = sdr.acq.acqfreq, sdr.crate
        sdrs.append(sdr)
        starts.append(int(rng.integers(0, 16368)))
    hist = [[] for _ in range(nch)]
    errs = []

    def worker(i):
        try:
            sdr, b = sdrs[i], starts[i]
            for cnt in range(nper):
                L.sdrtracking(C.byref(sdr), b, cnt)
                assert sdr.flagtrk == 1
                hist[i].append((list(sdr.trk.II[:5]), list(sdr.trk.QQ[:5]), sdr.currnsamp, sdr.trk.remcode, sdr.trk.remcarr))
                L.cumsumcorr(C.byref(sdr.trk), 1)
                L.pll(C.byref(sdr), C.byref(sdr.trk.prm1), sdr.ctime)
                L.dll(C.byref(sdr), C.byref(sdr.trk.prm1), sdr.ctime)
                L.clearcumsumcorr(C.byref(sdr.trk))
                b += sdr.currnsamp
        except Exception as e:          # noqa: BLE001
            errs.append((i, repr(e)))

    L.sdrtracking(C.byref(gc.SdrCh()), 1 << 60, 0)          # (context creation outside the timed region)
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(nch)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    assert not errs, errs
    rate = nch * nper / dt
    print(f"\nsdrtracking() from {nch} threads: {rate:.0f} calls/s = {rate / 32000:.2f} x real time for 32 channels")
    # every channel against the oracle
    ring = orc.Ring()
    big = np.ascontiguousarray(data)
    ring.buff, ring.ringlen, ring.wrpos = big.ctypes.data, 5000 * 65536, ntotal
    for i, p in enumerate(prns):
        o = orc.make_chan(p, dtype=1, f_if=4.092e6, corrn=2, corrd=3, corrp=3)
        o.acq.acqfreq = sdrs[i].acq.acqfreq
        o.carrfreq, o.codefreq = o.acq.acqfreq, o.crate
        b = starts[i]
        for cnt in range(nper):
            O.orc_sdrtracking(C.byref(o), C.byref(ring), b)
            II, QQ, ns, remc, remp = hist[i][cnt]
            assert ns == o.currnsamp and remc == o.remcode and remp == o.remcarr, (i, cnt)
            assert II == list(o.II[:5]) and QQ == list(o.QQ[:5]), (i, cnt)
            O.orc_cumsumcorr(C.byref(o), 1)
            O.orc_pll(C.byref(o), 0, o.ctime)
            O.orc_dll(C.byref(o), 0, o.ctime)
            O.orc_clearcumsumcorr(C.byref(o))
            b += o.currnsamp
        assert sdrs[i].trk.carrfreq == o.carrfreq and sdrs[i].trk.codefreq == o.codefreq
    for sdr in sdrs:
        L.freesdrch(C.byref(sdr))


def test_sdrtracking_call_rate_from_32_pthreads(gc, orc, synth, tmp_path):
    """The same from 32 pthreads of a C program linked against libgnsscorr.so (tests/host/threads_harness.c) -- the
    call rate the reference's channel threads would see, no interpreter in the way -- with its per-channel
    results checked against the oracle (a weighted checksum of every period's sums + the final frequencies)."""
    import subprocess
    import sys
    nch, nper, nblocks = 32, 200, 56
    prns = list(range(1, nch + 1))
    codes = {p: gc.gencode(p, 1) for p in prns}
    rng = np.random.default_rng(78)
    sats = [dict(prn=p, doppler=float(rng.uniform(-4000, 4000)), codephase=float(rng.uniform(0, 1023)),
                 cn0=46.0, phase=float(rng.uniform(0, 6.28))) for p in prns[::4]]
    ntotal = nblocks * 65536
    data = synth.make_if(codes, ntotal, f_if=4.092e6, dtype=1, sats=sats, seed=24)
    f = tmp_path / "if32.dat"
    data.tofile(f)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "erlangnetwork-gnsslib-sdr_amd")
    exe = str(tmp_path / "threads_harness")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(root, "tests", "host", "threads_harness.c"),
                           "-L" + pkg, "-lgnsscorr", "-lpthread", "-lm", "-Wl,-rpath," + pkg])
    out = subprocess.run([exe, str(f), str(nblocks), str(nch), str(nper)], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    rate = float([l for l in lines if l.startswith("calls_per_s")][0].split()[1])
    print(f"\nsdrtracking() from {nch} pthreads: {rate:.0f} calls/s = {rate / 32000:.2f} x real time for 32 channels")
    if out.stderr.strip():
        print(out.stderr.strip()[-600:])                # (GNSSCORR_CMB_PROF=1: where a combined launch chain's time goes)
    got = {int(l.split()[1]): [float(x) for x in l.split()[2:]] for l in lines if l.startswith("chk")}
    # the harness's deterministic start states
    seed = 12345
    O = orc.lib()
    ring = orc.Ring()
    big = np.ascontiguousarray(data)
    ring.buff, ring.ringlen, ring.wrpos = big.ctypes.data, 5000 * 65536, ntotal
    for i in range(nch):
        seed = (seed * 1103515245 + 12345) & 0xFFFFFFFF
        acqfreq = 4.092e6 + 200.0 * (((seed >> 16) % 41) - 20)
        seed = (seed * 1103515245 + 12345) & 0xFFFFFFFF
        b = (seed >> 8) % 16368
        o = orc.make_chan(i + 1, dtype=1, f_if=4.092e6, corrn=2, corrd=3, corrp=3)
        o.acq.acqfreq = acqfreq
        o.carrfreq, o.codefreq = acqfreq, o.crate
        acc = 0.0
        for cnt in range(nper):
            O.orc_sdrtracking(C.byref(o), C.byref(ring), b)
            assert o.flagtrk == 1
            for t in range(5):
                acc += o.II[t] * (t + 1) + o.QQ[t] * (t + 7)
            O.orc_cumsumcorr(C.byref(o), 1)
            O.orc_pll(C.byref(o), 0, o.ctime)
            O.orc_dll(C.byref(o), 0, o.ctime)
            O.orc_clearcumsumcorr(C.byref(o))
            b += o.currnsamp
        assert got[i] == [acc, o.carrfreq, o.codefreq], (i, got[i], acc, o.carrfreq, o.codefreq)
