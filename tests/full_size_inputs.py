"""Seeded inputs of the full-size parity fixtures (tests/golden/full_size.json): shared by the generator
(tests/golden/make_golden_full.py, oracle side) and the GPU tests (tests/test_gpu_fullsize.py)."""
import numpy as np

F_SF = 16.368e6
NS = 16368
SEED = 20240601
ACQ_MS = 13                      # configs[1]: 11 periods of history + margin
ACQ_WRPOS = 11 * NS + NS // 3 + 700
TRK_MS = 54                      # configs[2]
TRK_EPOCHS = 50
C3_MS = 36                       # configs[3]: acquisition history + 20 periods of tracking
C3_WRPOS = 12 * NS + 1234
C3_EPOCHS = 20


def gps_stream(gc, synth, nms):
    """The bench's stream (bench.py make_signal): ~10 of PRN 1..32 at 38-50 dB-Hz, int8 IQ, zero IF."""
    codes = {p: gc.gencode(p, gc.CTYPE_L1CA) for p in range(1, 33)}
    sats = synth.default_sats(list(range(1, 33)), seed=SEED)
    return synth.make_if(codes, nms * NS, f_sf=F_SF, f_if=0.0, dtype=2, sats=sats, seed=SEED), sats


def trk_states(gc):
    """Seeded mid-track states of 32 channels; every fourth one in the state sdracquisition() leaves (ref
    src/sdracq.c:51-55: code and carrier phase 0, carrier on the 200 Hz grid, nominal chip rate)."""
    rng = np.random.default_rng(SEED + 5)
    st = []
    for i in range(32):
        if i % 4 == 0:
            st.append(dict(carrfreq=200.0 * int(rng.integers(-25, 26)), codefreq=1.023e6, remcode=0.0, remcarr=0.0,
                           buffloc=int(rng.integers(0, NS))))
        else:
            st.append(dict(carrfreq=float(rng.uniform(-5000, 5000)), codefreq=1.023e6 + float(rng.uniform(-3, 3)),
                           remcode=float(rng.uniform(0.0, 0.06)), remcarr=float(rng.uniform(0, 6.2)) if i % 2 else
                           -float(rng.uniform(0, 4000)), buffloc=int(rng.integers(0, NS))))
    return st


def two_streams(gc, synth):
    """Stream 1: the GPS constellation above; stream 2: GLONASS G1 frequency numbers -7..+6, six of them
    present (carrier 1602 MHz + 562.5 kHz k, seen at f_if + 562.5 kHz k: ref src/sdrinit.c:612-615)."""
    d1, sat1 = gps_stream(gc, synth, C3_MS)
    rng = np.random.default_rng(SEED + 9)
    g1 = gc.gencode(1, gc.CTYPE_G1)
    ks = [-7, -4, -1, 0, 3, 6]
    sat2 = [dict(prn=k, doppler=562.5e3 * k + float(rng.uniform(-3000, 3000)), codephase=float(rng.uniform(0, 511)),
                 cn0=float(rng.uniform(42, 49)), phase=float(rng.uniform(0, 6.28))) for k in ks]
    d2 = synth.make_if({k: g1 for k in ks}, C3_MS * NS, f_sf=F_SF, f_if=0.0, dtype=2, sats=sat2, seed=SEED + 9,
                       f_cf=1.602e9 * 1e6)     # huge f_cf: no code Doppler from the folded FDMA offset
    return d1, d2, sat1, sat2


def config3_channels(gc):
    return [gc.Channel(p) for p in range(1, 33)] + [gc.Channel(k, ctype=gc.CTYPE_G1, ftype=2) for k in range(-7, 7)]
