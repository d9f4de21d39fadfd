"""CPU tests of the oracle (oracle/gnss_oracle.c) itself: known answers, an
independent numpy restatement of the normative closed forms, the
reference quirks Q1-Q8, and agreement of its two NCO flavours."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
TI = 1 / 16.368e6


def test_ca_first10_chips_known_answer(orc):
    """IS-GPS-200 first-10-chip table, chip +1 <-> bit 1 (ref src/sdrcode.c:146)."""
    kat = json.load(open(os.path.join(HERE, "golden", "ca_first10_octal.json")))["first10_octal"]
    for prn, octal in kat.items():
        code, crate = orc.gencode(int(prn), 1)
        assert len(code) == 1023 and crate == 1.023e6
        bits = "".join("1" if c == 1 else "0" for c in code[:10])
        assert format(int(bits, 2), "o") == octal, prn


def test_code_properties(orc):
    for prn in (1, 17, 32, 120, 138, 210):
        c, _ = orc.gencode(prn, 27 if prn >= 120 else 1)
        c = c.astype(np.int64)
        assert set(np.unique(c)) == {-1, 1}
        assert abs(c.sum()) == 1                       # balanced Gold code
        ac = np.array([np.dot(c, np.roll(c, s)) for s in range(1, 1023)])
        assert set(np.unique(ac)) <= {-65, -1, 63}     # three-valued autocorrelation
    g, crate = orc.gencode(0, 20)
    assert len(g) == 511 and crate == 0.511e6
    g = g.astype(np.int64)
    assert abs(g.sum()) == 1
    assert all(np.dot(g, np.roll(g, s)) == -1 for s in range(1, 511))   # m-sequence
    with pytest.raises(ValueError):
        orc.gencode(1, 2)                               # L1C types are not on this path


def test_carrier_lut(orc):
    cost = np.zeros(32, np.int16)
    sint = np.zeros(32, np.int16)
    orc.lib().orc_carrier_lut(cost.ctypes.data, sint.ctypes.data)
    assert list(cost[:10]) == [32, 31, 30, 27, 23, 18, 12, 6, 0, -6]
    assert list(sint[:9]) == [0, 6, 12, 18, 23, 27, 30, 31, 32]
    assert np.array_equal(cost, np.roll(sint, -8))


@pytest.mark.parametrize("n", [8, 12, 33, 341, 1024, 4092, 32736, 32768])
def test_fft_matches_numpy(orc, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    for sign, ref in ((-1, np.fft.fft(x.astype(np.complex128))), (1, np.fft.ifft(x.astype(np.complex128)) * n)):
        y = x.copy()
        orc.lib().orc_fft(y.ctypes.data, n, sign)
        assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-7


def _np_mix(data, dtype, ti, n, freq, phi0):
    """Normative mixer: exact-arithmetic phase, trunc toward zero, & 31."""
    from fractions import Fraction
    cost = np.array([int(np.floor(np.cos(2 * np.pi / 32 * i) * 32 + 0.5)) for i in range(32)])
    sint = np.array([int(np.floor(np.sin(2 * np.pi / 32 * i) * 32 + 0.5)) for i in range(32)])
    phis = Fraction(phi0 * 32 / (2.0 * 3.1415926535897932))
    ps = Fraction(freq * 32 * ti)
    idx = np.array([int(float(phis + k * ps)) & 31 for k in range(n)])
    d = data.astype(np.int64)
    if dtype == 2:
        dI, dQ = d[0::2][:n], d[1::2][:n]
        return cost[idx] * dI - sint[idx] * dQ, sint[idx] * dI + cost[idx] * dQ
    return cost[idx] * d[:n], sint[idx] * d[:n]


@pytest.mark.parametrize("dtype,freq,phi0", [(2, 1234.5, 0.0), (1, 4.092e6 + 600.0, 0.3), (2, -3.9e6, 6.2),
                                             (2, -3400.0, 0.0)])
def test_mixcarr_flavours_and_numpy(orc, dtype, freq, phi0):
    n = 2000
    rng = np.random.default_rng(1)
    data = rng.integers(-128, 128, size=n * dtype, dtype=np.int8)
    out = {}
    I, Q = np.zeros(n, np.int16), np.zeros(n, np.int16)
    rem = orc.lib().orc_mixcarr_seq(data.ctypes.data, dtype, TI, n, freq, phi0, I.ctypes.data, Q.ctypes.data)
    out["seq"] = (I.astype(np.int64), Q.astype(np.int64), rem)
    nI, nQ = _np_mix(data, dtype, TI, n, freq, phi0)
    assert np.array_equal(out["seq"][0], nI) and np.array_equal(out["seq"][1], nQ)
    if freq > 0:
        assert 0 <= out["seq"][2] <= 2 * np.pi + 1e-12
    else:
        assert out["seq"][2] < 0          # negative phases are never wrapped (ref src/sdrcmn.c:667)


def test_rescode_flavours(orc):
    code, crate = orc.gencode(7, 1)
    L = orc.lib()
    rng = np.random.default_rng(2)
    for coff, smax, dc in ((0.0, 0, 0.0), (100.25, 6, 0.0), (1022.9, 18, 2.5), (0.37, 6, -2.9), (513.5, 3, 1.0)):
        ci = TI * (crate + dc)
        n = 16368
        a = np.zeros(n + 2 * smax, np.int16)
        ra = L.orc_rescode_seq(code.ctypes.data, 1023, coff, smax, ci, n, a.ctypes.data)
        # returned remainder = code phase after n samples, modulo the code length (the wrap is lazy and the
        # value has smax*ci subtracted last: it may come back negative or >= clen, ref src/sdrcmn.c:617-620)
        assert abs(((ra - (coff + n * ci) + 511.5) % 1023) - 511.5) < 1e-6
        # normative closed form: chip = trunc(coff0 + (k + o) * ci) mod clen
        k = np.arange(n + 2 * smax)
        ref = code[np.floor((coff - smax * ci + k * ci) % 1023).astype(int)]
        assert np.count_nonzero(ref != a) <= 2          # fp rounding at chip edges only
    # the remainder may come back >= clen (lazy wrap)
    r = L.orc_rescode_seq(code.ctypes.data, 1023, 1022.99, 0, 0.0625, 1, a.ctypes.data)
    assert r >= 1023


def test_correlator_against_numpy(orc):
    rng = np.random.default_rng(3)
    code, crate = orc.gencode(11, 1)
    for dtype, freq in ((2, 2500.0), (1, 4.092e6 - 1200.0)):
        n = 3000
        data = rng.integers(-100, 101, size=n * dtype, dtype=np.int8)
        s = np.array([3, 6], np.int32)
        coff, phi0, cf = 200.3, 1.1, crate + 0.7
        II, QQ, remc, remp = orc.correlator(data, dtype, TI, n, freq, phi0, cf, coff, s, code)
        mI, mQ = _np_mix(data, dtype, TI, n, freq, phi0)
        ci = TI * cf
        for t, o in enumerate((0, -3, 3, -6, 6)):
            k = np.arange(n)
            chips = code[np.floor((coff + (k + o) * ci) % 1023).astype(int)].astype(np.int64)
            assert II[t] == np.dot(mI, chips) / 32.0
            assert QQ[t] == np.dot(mQ, chips) / 32.0
        assert abs(remc - ((coff + n * ci) % 1023)) < 1e-6


def test_cpxconv_equals_time_domain(orc):
    """pcorrelator (FFT, m = 2n, mixed radix) == direct correlation (SURVEY 8a normative form)."""
    L = orc.lib()
    rng = np.random.default_rng(4)
    ch = orc.make_chan(5, dtype=2, f_if=0.0)
    n, m = ch.nsamp, ch.nfft
    data = rng.integers(-60, 61, size=2 * n * 2, dtype=np.int8)
    freq = np.array([-3400.0, 0.0, 1200.0])
    xc = orc.codespectrum(ch)
    P = np.zeros(3 * n)
    L.orc_pcorrelator(data.ctypes.data, 2, ch.ti, n, freq.ctypes.data, 3, ch.crate, m, xc.ctypes.data,
                      P.ctypes.data)
    Ptd = np.zeros(3 * n)
    code = np.ctypeslib.as_array(ch.code).copy()
    L.orc_pcorrelator_td(data.ctypes.data, 2, ch.ti, n, freq.ctypes.data, 3, m, code.ctypes.data, 1023, ch.ci,
                         100, 140, Ptd.ctypes.data)
    for b in range(3):
        assert rel_err(P[b * n + 100:b * n + 140], Ptd[b * n + 100:b * n + 140]) < 5e-6
    # flagsum accumulates (ref src/sdrcmn.c:244-246)
    P2 = P.copy()
    L.orc_pcorrelator(data.ctypes.data, 2, ch.ti, n, freq.ctypes.data, 3, ch.crate, m, xc.ctypes.data,
                      P2.ctypes.data)
    assert rel_err(P2, 2 * P) < 1e-12


def test_maxvd_meanvd_quirks(orc):
    L = orc.lib()
    d = np.array([5.0, 1.0, 9.0, 9.0, 2.0, 7.0, 3.0, 4.0])
    ind = C.c_int()
    assert L.orc_maxvd(d.ctypes.data, 8, -1, -1, C.byref(ind)) == 9.0 and ind.value == 2   # first max wins
    # element 0 seeds the maximum even when it lies inside the excluded window
    d0 = np.array([50.0, 1.0, 9.0, 3.0, 2.0, 7.0, 3.0, 4.0])
    assert L.orc_maxvd(d0.ctypes.data, 8, 0, 2, C.byref(ind)) == 50.0 and ind.value == 0
    # wrapped window: exclude i >= 6 or i <= 1
    assert L.orc_maxvd(d.ctypes.data, 8, 6, 1, C.byref(ind)) == 9.0
    assert L.orc_meanvd(d.ctypes.data, 8, 6, 1) == pytest.approx((9 + 9 + 2 + 7) / 4)
    assert L.orc_meanvd(d.ctypes.data, 8, 2, 3) == pytest.approx((5 + 1 + 2 + 7 + 3 + 4) / 6)


def test_checkacquisition_crafted(orc):
    n, nf, nsc = 160, 5, 4
    rng = np.random.default_rng(5)
    P = rng.uniform(0.5, 1.0, size=nf * n)
    P[3 * n + 2] = 40.0            # peak near the start: exclusion window wraps
    P[3 * n + 150] = 8.0           # second peak outside the window
    freq = np.arange(nf, dtype=np.float64) * 200.0
    res = orc.AcqRes()
    ok = orc.lib().orc_checkacquisition(P.ctypes.data, n, nf, nsc, 1e-3, freq.ctypes.data, C.byref(res))
    assert ok == 1 and res.acqcodei == 2 and res.freqi == 3 and res.acqfreq == 600.0
    assert res.peakr == pytest.approx(5.0)
    row = P[3 * n:4 * n]
    mask = np.ones(n, bool)
    mask[np.r_[0:11, n - 6:n]] = False            # codei-8 .. codei+8, wrapped
    assert res.cn0 == pytest.approx(10 * np.log10(40.0 / row[mask].mean() / 1e-3))


def test_tracking_driver_quirks(orc, synth):
    """Q1 (II/QQ swap), Q2 (short oldI copy), currnsamp truncation."""
    rng = np.random.default_rng(6)
    n = 16 * 4096
    data = rng.integers(-60, 61, size=(n, 2), dtype=np.int8)
    ring = orc.make_ring(data, n, n)
    o = orc.make_chan(4, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
    o.carrfreq, o.codefreq = 900.0, o.crate
    for i in range(5):
        o.II[i], o.QQ[i] = 100.0 + i, 200.0 + i
    o.oldI[4] = -7.0
    L = orc.lib()
    L.orc_sdrtracking(C.byref(o), C.byref(ring), 31)
    assert o.flagtrk == 1 and o.currnsamp == 16368
    # memcpy of 1+2*corrn*8 = 33 bytes: taps 0..3 copied, tap 4 only its lowest byte
    assert [o.oldI[i] for i in range(4)] == [100.0, 101.0, 102.0, 103.0]
    assert o.oldI[4] != 104.0
    buf = np.zeros(2 * 16368, np.int8)
    L.orc_getbuff(C.byref(ring), 31, 16368, 2, buf.ctypes.data)
    code = np.ctypeslib.as_array(o.code).copy()
    cII, cQQ, _, _ = orc.correlator(buf, 2, o.ti, 16368, 900.0, 0.0, o.crate, 0.0, [3, 6], code)
    assert [o.QQ[i] for i in range(5)] == list(cII)      # trk.QQ <- correlator's II
    assert [o.II[i] for i in range(5)] == list(cQQ)
    # not enough samples buffered: flagtrk stays 0 (ref src/sdrtrk.c:30,50)
    L.orc_sdrtracking(C.byref(o), C.byref(ring), n - 100)
    assert o.flagtrk == 0


def test_loop_filters(orc):
    o = orc.make_chan(1, dtype=1, f_if=4.092e6, corrn=6, corrd=3, corrp=6)
    assert (o.ne, o.nl, o.loopms) == (3, 4, 10)
    assert o.pllw2[0] == pytest.approx((30 / 0.53) ** 2) and o.fllw[1] == pytest.approx(50 / 0.25)
    o.acq.acqfreq = 4.092e6 + 1200.0
    o.carrfreq, o.codefreq = o.acq.acqfreq, o.crate
    o.sumI[0], o.sumQ[0], o.oldsumI[0], o.oldsumQ[0] = 900.0, 300.0, 950.0, 100.0
    o.sumI[3], o.sumQ[3], o.sumI[4], o.sumQ[4] = 500.0, 100.0, 450.0, 80.0
    L = orc.lib()
    L.orc_pll(C.byref(o), 0, 1e-3)
    carr_err = np.arctan2(300.0, 900.0) / np.pi
    freq_err = np.arctan(300.0 / 900.0) - np.arctan(100.0 / 950.0)
    nco = o.pllaw[0] * carr_err + o.pllw2[0] * 1e-3 * carr_err + o.fllw[0] * 1e-3 * freq_err
    assert o.carrNco == pytest.approx(nco) and o.carrfreq == pytest.approx(o.acq.acqfreq + nco)
    L.orc_dll(C.byref(o), 0, 1e-3)
    e, l = np.hypot(500, 100), np.hypot(450, 80)
    code_err = (e - l) / (e + l)
    assert o.codeErr == pytest.approx(code_err)
    cnco = o.dllaw[0] * code_err + o.dllw2[0] * 1e-3 * code_err
    assert o.codefreq == pytest.approx(o.crate - cnco + (o.carrfreq - 4.092e6) / (1575.42e6 / 1.023e6))
    # cumsumcorr / clearcumsumcorr (ref src/sdrtrk.c:64-86)
    for i in range(13):
        o.II[i], o.QQ[i], o.oldI[i], o.oldQ[i] = i, -i, 2 * i, 3 * i
    s0 = o.sumI[5]
    L.orc_cumsumcorr(C.byref(o), 1)
    assert o.sumI[5] == s0 + 5 and o.oldsumQ[5] == 15
    L.orc_clearcumsumcorr(C.byref(o))
    assert o.sumI[5] == 0 and o.oldsumQ[5] == 0


def test_acquisition_driver_on_synthetic_signal(orc, gc, synth):
    """Config 1 in miniature: PRN present -> found at the right Doppler bin in the first iteration;
    PRN absent -> all 10 iterations, no acquisition; returned buffloc per ref src/sdracq.c:51-53."""
    codes = {p: gc.gencode(p, 1) for p in (9,)}
    sats = [dict(prn=9, doppler=-2345.0, codephase=333.3, cn0=50.0, phase=1.0)]
    nsamples = 13 * 16368
    data = synth.make_if(codes, nsamples, f_if=0.0, dtype=2, sats=sats, seed=3)
    ring = orc.make_ring(data, nsamples, nsamples)
    o = orc.make_chan(9, dtype=2, f_if=0.0)
    xc = orc.codespectrum(o)
    o.xcode = xc.ctypes.data
    P = np.zeros(o.nfreq * o.nsamp)
    it = C.c_int()
    b = orc.lib().orc_sdracquisition(C.byref(o), C.byref(ring), P.ctypes.data, C.byref(it))
    assert o.flagacq == 1 and it.value == 1
    assert abs(o.acq.acqfreq - (-2345.0)) <= 100.0 + 1e-9
    assert b == nsamples - 11 * 16368 + o.acq.acqcodei
    # code phase: the replica starts where the generator's code phase wraps to 0
    start = (nsamples - 11 * 16368)
    chips_at_start = (333.3 + 1.023e6 * (1 - 2345.0 / 1575.42e6) * start / 16.368e6) % 1023
    expect = ((1023 - chips_at_start) % 1023) * 16
    assert min(abs(o.acq.acqcodei - expect), 16368 - abs(o.acq.acqcodei - expect)) <= 2
    assert o.carrfreq == o.acq.acqfreq and o.codefreq == o.crate
