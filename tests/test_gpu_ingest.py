"""IF ingest into the HBM ring (-m gpu): plain int8 blocks through the pinned double buffer, the NSL Stereo packed
byte and the RTL-SDR unsigned bytes expanded on the device (ref src/rcv/stereo/stereo.c:160-205,
src/rcv/rtlsdr/rtlsdr.c:136-143), and rcvgetbuff()'s wrap (ref src/sdrrcv.c:505-532).  Bar: every byte of the
ring equal to the oracle's expansion."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_push_blocks_larger_than_the_staging_buffers_and_read_back(gc, engine):
    rng = np.random.default_rng(1)
    ringlen = 3 * (1 << 22)                                  # 12.6 M samples, IQ: 25 MB
    engine.ring_create(1, 2, ringlen)
    data = rng.integers(-128, 128, size=(ringlen + 5000, 2), dtype=np.int8)
    # 20 MB in one call (staged in 8 MB pieces), then odd-sized blocks that run over the end of the ring
    sizes = [10 * (1 << 20), 1, 65536, 12345, ringlen + 5000 - (10 * (1 << 20) + 1 + 65536 + 12345)]
    pos = 0
    for n in sizes:
        engine.ring_push_raw(1, data[pos:pos + n], n)
        pos += n
    assert engine.ring_wrpos(1) == ringlen + 5000
    # the last `ringlen` samples are in the ring; a read across the wrap comes back in order
    got = engine.ring_read(1, ringlen - 7000, 12000, 2)
    assert np.array_equal(got, data[ringlen - 7000:ringlen + 5000])
    got = engine.ring_read(1, 6000, 100000, 2)
    assert np.array_equal(got, data[6000:106000])


def test_stereo_packed_bytes_feed_both_rings(gc, orc, engine):
    rng = np.random.default_rng(2)
    n = 300000
    ringlen = 1 << 18                                        # the stream laps the ring
    engine.ring_create(1, 1, ringlen)
    engine.ring_create(2, 2, ringlen)
    packed = rng.integers(0, 256, size=n, dtype=np.uint8)
    packed[:256] = np.arange(256, dtype=np.uint8)            # every byte value once
    engine.ring_push_packed(gc.FMT_STEREO, packed[:100001], 100001)
    engine.ring_push_packed(gc.FMT_STEREO, packed[100001:], n - 100001)
    assert engine.ring_wrpos(1) == n and engine.ring_wrpos(2) == n
    e1 = np.zeros(n, np.int8)
    e2 = np.zeros((n, 2), np.int8)
    orc.lib().orc_stereo_exp(packed.ctypes.data, n, 1, e1.ctypes.data)
    orc.lib().orc_stereo_exp(packed.ctypes.data, n, 2, e2.ctypes.data)
    assert set(np.unique(e1)) == {-3, -1, 1, 3} and set(np.unique(e2)) == {-7, -5, -3, -1, 1, 3, 5, 7}
    start = n - ringlen
    assert np.array_equal(engine.ring_read(1, start, ringlen, 1), e1[start:])
    assert np.array_equal(engine.ring_read(2, start, ringlen, 2), e2[start:])


def test_rtlsdr_unsigned_bytes(gc, orc, engine):
    rng = np.random.default_rng(3)
    n = 200000
    engine.ring_create(1, 2, 1 << 18)
    raw = rng.integers(0, 256, size=2 * n, dtype=np.uint8)
    raw[:256] = np.arange(256, dtype=np.uint8)
    engine.ring_push_packed(gc.FMT_RTLSDR, raw, n)
    exp = np.zeros(2 * n, np.int8)
    orc.lib().orc_rtlsdr_exp(raw.ctypes.data, 2 * n, exp.ctypes.data)
    assert exp.min() == -127 and exp.max() == 127
    assert np.array_equal(engine.ring_read(1, 0, n, 2).reshape(-1), exp)


def test_tracking_sees_blocks_pushed_just_before(gc, orc, engine):
    """A block pushed on the copy stream and a tracking batch issued right after it: the batch is ordered
    behind the transfer (no explicit synchronisation by the caller)."""
    import ctypes as C
    rng = np.random.default_rng(4)
    n = 16368 * 6
    data = rng.integers(-90, 91, size=(n, 2), dtype=np.int8)
    engine.ring_create(1, 2, n)
    engine.set_channels([gc.Channel(9, dtype=2, f_if=0.0)])
    st = dict(carrfreq=1800.0, codefreq=1.023e6 + 0.7, remcode=0.02, remcarr=1.0, buffloc=11)
    for rep in range(3):                                     # same ring region rewritten with other samples
        block = (data + rep).astype(np.int8)
        engine._L.gnsscorr_ring_create(engine.h, 1, 2, n, None)
        engine.set_channels([gc.Channel(9, dtype=2, f_if=0.0)])
        engine.ring_push_raw(1, block, n)
        engine.trk_set_state([st])
        engine.trk_run(4)
        II, QQ, ns = engine.trk_fetch()
        o = orc.make_chan(9, dtype=2, f_if=0.0)
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = st["carrfreq"], st["codefreq"], st["remcode"], st["remcarr"]
        ring = orc.make_ring(block, n, n)
        b = st["buffloc"]
        for e in range(4):
            orc.lib().orc_sdrtracking(C.byref(o), C.byref(ring), b)
            assert np.array_equal(II[0, e], np.ctypeslib.as_array(o.II)[:5]), (rep, e)
            b += o.currnsamp
