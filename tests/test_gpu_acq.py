"""HIP parallel code phase search vs the CPU oracle (parity tests proper, -m gpu).

The oracle runs the reference's algorithm literally (FFT length m = 2*nsamp =
32736, mixed radix, ref src/sdrcmn.c:228-251,738-773); the HIP path uses
L = 32768.  Decision outputs (code phase, Doppler bin, iteration count,
returned buffloc) must be identical; floating point outputs (power, peak ratio,
C/N0) within the north_star tolerance of 1e-4 relative."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

F_SF = 16.368e6


def test_fft16k_matches_numpy(gc, engine):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(0)
    batch = 3
    x = (rng.standard_normal((batch, 16384)) + 1j * rng.standard_normal((batch, 16384))).astype(np.complex64)
    # an impulse and a pure tone exercise index maps exactly
    x[1] = 0
    x[1, 5] = 1.0
    x[2] = np.exp(2j * np.pi * 37 * np.arange(16384) / 16384).astype(np.complex64)
    dx = torch.from_numpy(x).cuda()
    dy = torch.empty_like(dx)
    torch.cuda.synchronize()
    L = gc.lib()
    for sign, ref in ((-1, np.fft.fft(x.astype(np.complex128), axis=1)),
                      (+1, np.fft.ifft(x.astype(np.complex128), axis=1) * 16384)):
        assert L.gnsscorr_fft16k(engine.h, dx.data_ptr(), dy.data_ptr(), sign, batch) == 0
        engine.sync()
        y = dy.cpu().numpy()
        for b in range(batch):
            err = np.abs(y[b] - ref[b]).max() / np.abs(ref[b]).max()
            assert err < 2e-6, (sign, b, err)


def _scenario(gc, synth, dtype, f_if, prns_present, prns_all, seed):
    codes = {p: gc.gencode(p, gc.CTYPE_L1CA) for p in prns_all}
    rng = np.random.default_rng(seed)
    sats = [dict(prn=p, doppler=float(rng.uniform(-4500, 4500)), codephase=float(rng.uniform(0, 1023)),
                 cn0=float(rng.uniform(44, 50)), phase=float(rng.uniform(0, 6.28)),
                 bits=rng.choice([-1.0, 1.0], size=32)) for p in prns_present]
    nsamp = 16 * 16384       # 262144 samples = 16 ms
    data = synth.make_if(codes, nsamp, f_sf=F_SF, f_if=f_if, dtype=dtype, sats=sats, seed=seed)
    return data, sats, nsamp


@pytest.mark.parametrize("dtype,f_if", [(2, 0.0), (1, 4.092e6)])
def test_acquisition_matches_oracle(gc, orc, synth, engine, dtype, f_if):
    prns = [3, 11, 20]                      # 3 and 20 present, 11 absent
    data, sats, nsamples = _scenario(gc, synth, dtype, f_if, [3, 20], prns, seed=7 + dtype)
    engine.ring_create(1, dtype, nsamples)
    engine.ring_push_raw(1, data, nsamples)
    wrpos = 14 * 16368 + 777                # any position with (intg+1)*nsamp samples behind it
    chans = [gc.Channel(p, dtype=dtype, f_if=f_if) for p in prns]
    engine.set_channels(chans)
    engine.acq_run(wrpos)
    res = engine.acq_fetch()

    ring = orc.make_ring(data, nsamples, wrpos)
    for i, p in enumerate(prns):
        o = orc.make_chan(p, dtype=dtype, f_if=f_if)
        xc = orc.codespectrum(o)
        o.xcode = xc.ctypes.data
        power = np.zeros(o.nfreq * o.nsamp)
        iters = C.c_int()
        buffloc = orc.lib().orc_sdracquisition(C.byref(o), C.byref(ring), power.ctypes.data, C.byref(iters))
        r = res[i]
        assert r["flagacq"] == o.flagacq, (p, r, o.acq.peakr)
        assert r["flagacq"] == (1 if p in (3, 20) else 0)
        assert r["iters"] == iters.value
        assert r["acqcodei"] == o.acq.acqcodei and r["freqi"] == o.acq.freqi
        assert r["acqfreq"] == o.acq.acqfreq
        assert r["buffloc"] == buffloc
        assert abs(r["peakr"] - o.acq.peakr) <= 1e-4 * o.acq.peakr
        assert abs(r["cn0"] - o.acq.cn0) <= 1e-4 * abs(o.acq.cn0)
        if p == 3:
            # true Doppler within one 200 Hz bin, code phase consistent with the generator
            s = [s for s in sats if s["prn"] == 3][0]
            assert abs((r["acqfreq"] - f_if) - s["doppler"]) <= 200.0
        if p in (3, 20):        # the full power array of the first and of the last channel
            P = engine.acq_power(i)
            assert P.shape == (o.nfreq, o.nsamp)
            assert rel_err(P.ravel(), power) <= 1e-4


@pytest.mark.parametrize("dtype,f_sf,f_if", [(2, 20e6, 0.0), (1, 20e6, 4.0e6), (2, 26e6, 0.0)])
def test_acquisition_long_periods_65536_point_transform(gc, orc, synth, engine, dtype, f_sf, f_if):
    """20 / 26 Msps front ends (ref frontend/stereo_L1G1.ini:5-13: SF 20 MHz, IF 4 MHz real and zero-IF IQ):
    20000 / 26000 samples per code period, the reference transforms 2*nsamp = 40000 / 52000 points (ref
    src/sdrinit.c:625), the HIP path 65536.  Decisions identical, power / peak ratio / C/N0 to 1e-4."""
    nsamp = int(f_sf * 1e-3)
    prns = [6, 15, 28]                      # 6 and 28 present
    codes = {p: gc.gencode(p, gc.CTYPE_L1CA) for p in prns}
    rng = np.random.default_rng(int(f_sf / 1e6) + dtype)
    sats = [dict(prn=p, doppler=float(rng.uniform(-4000, 4000)), codephase=float(rng.uniform(0, 1023)),
                 cn0=float(rng.uniform(45, 49)), phase=float(rng.uniform(0, 6.28))) for p in (6, 28)]
    nsamples = 14 * nsamp
    data = synth.make_if(codes, nsamples, f_sf=f_sf, f_if=f_if, dtype=dtype, sats=sats, seed=99 + dtype)
    engine.ring_create(1, dtype, nsamples)
    engine.ring_push_raw(1, data, nsamples)
    chans = [gc.Channel(p, dtype=dtype, f_sf=f_sf, f_if=f_if) for p in prns]
    assert chans[0].nsamp == nsamp and chans[0].nfft == 2 * nsamp
    engine.set_channels(chans)
    wrpos = 12 * nsamp + 333
    engine.acq_run(wrpos)
    res = engine.acq_fetch()
    ring = orc.make_ring(data, nsamples, wrpos)
    for i, p in enumerate(prns):
        o = orc.make_chan(p, dtype=dtype, f_sf=f_sf, f_if=f_if)
        assert o.nsamp == nsamp
        xc = orc.codespectrum(o)
        o.xcode = xc.ctypes.data
        power = np.zeros(o.nfreq * o.nsamp)
        iters = C.c_int()
        buffloc = orc.lib().orc_sdracquisition(C.byref(o), C.byref(ring), power.ctypes.data, C.byref(iters))
        r = res[i]
        assert r["flagacq"] == o.flagacq == (1 if p in (6, 28) else 0), (p, r, o.acq.peakr)
        assert r["iters"] == iters.value and r["buffloc"] == buffloc
        assert r["acqcodei"] == o.acq.acqcodei and r["freqi"] == o.acq.freqi and r["acqfreq"] == o.acq.acqfreq
        assert abs(r["peakr"] - o.acq.peakr) <= 1e-4 * o.acq.peakr
        assert abs(r["cn0"] - o.acq.cn0) <= 1e-4 * abs(o.acq.cn0)
        if p == 6:
            s = [s for s in sats if s["prn"] == 6][0]
            assert abs((r["acqfreq"] - f_if) - s["doppler"]) <= 200.0
            P = engine.acq_power(i)
            assert P.shape == (o.nfreq, o.nsamp)
            assert rel_err(P.ravel(), power) <= 1e-4
