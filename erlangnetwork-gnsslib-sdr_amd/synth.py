"""Deterministic synthetic IF generator: int8 samples at f_sf,
real (DTYPE=1) or interleaved I,Q (DTYPE=2), a handful of C/A (or GLONASS)
signals with Doppler, code phase, 50 bps data and AWGN.  Host-side utility for
tests and bench.py; not part of the correlation path."""
import numpy as np

SEED = 20240601


def make_if(codes, nsamp, f_sf=16.368e6, f_if=0.0, dtype=2, sats=None, seed=SEED, noise_sigma=8.0,
            f_cf=1575.42e6, chunk=1 << 22):
    """codes: {prn: (chips int array, chip rate)}.  sats: list of dicts
    (prn, doppler Hz, code phase chips, cn0 dB-Hz, carrier phase rad).
    Returns int8 array of shape (nsamp, 2) for dtype 2 or (nsamp,) for dtype 1."""
    rng = np.random.default_rng(seed)
    if sats is None:
        sats = []
    out = np.empty((nsamp, 2) if dtype == 2 else (nsamp,), dtype=np.int8)
    # amplitude from C/N0: A^2/(2 sigma^2 / f_sf) per real rail
    for s0 in range(0, nsamp, chunk):
        n = min(chunk, nsamp - s0)
        t = (np.arange(s0, s0 + n, dtype=np.float64)) / f_sf
        xi = rng.normal(0.0, noise_sigma, n)
        xq = rng.normal(0.0, noise_sigma, n) if dtype == 2 else None
        for s in sats:
            chips, crate = codes[s["prn"]]
            clen = len(chips)
            fd = s["doppler"]
            cn0 = 10.0 ** (s["cn0"] / 10.0)
            # complex baseband noise power per sample = 2 sigma^2 (IQ) over bandwidth f_sf
            amp = noise_sigma * np.sqrt(2.0 * cn0 / f_sf) if dtype == 2 else noise_sigma * np.sqrt(4.0 * cn0 / f_sf)
            rate = crate * (1.0 + fd / f_cf)
            cph = (s["codephase"] + rate * t) % clen
            c = chips[cph.astype(np.int64)].astype(np.float64)
            bit = s.get("bits")
            if bit is not None:
                c = c * bit[((t * 50.0).astype(np.int64)) % len(bit)]
            ph = 2 * np.pi * (f_if + fd) * t + s.get("phase", 0.0)
            if dtype == 2:
                # the reference's mixer multiplies the samples by exp(+i phi) (ref
                # src/sdrcmn.c:655-656), so an IQ stream carries a carrier at +f as exp(-i 2 pi f t)
                xi += amp * c * np.cos(ph)
                xq -= amp * c * np.sin(ph)
            else:
                xi += amp * c * np.cos(ph)
        if dtype == 2:
            out[s0:s0 + n, 0] = np.clip(np.rint(xi), -127, 127).astype(np.int8)
            out[s0:s0 + n, 1] = np.clip(np.rint(xq), -127, 127).astype(np.int8)
        else:
            out[s0:s0 + n] = np.clip(np.rint(xi), -127, 127).astype(np.int8)
    return out


def default_sats(prns, seed=SEED, npresent=10):
    rng = np.random.default_rng(seed + 1)
    present = sorted(rng.choice(prns, size=min(npresent, len(prns)), replace=False).tolist())
    sats = []
    for p in present:
        sats.append(dict(prn=int(p), doppler=float(rng.uniform(-5000, 5000)),
                         codephase=float(rng.uniform(0, 1023)), cn0=float(rng.uniform(38, 50)),
                         phase=float(rng.uniform(0, 2 * np.pi)),
                         bits=rng.choice([-1.0, 1.0], size=64)))
    return sats


def random_if(nsamp, dtype=2, seed=SEED, amp=40):
    """Uniform random int8 samples (throughput runs, data-independent kernels)."""
    rng = np.random.default_rng(seed)
    shape = (nsamp, 2) if dtype == 2 else (nsamp,)
    return rng.integers(-amp, amp + 1, size=shape, dtype=np.int8)
