"""HIP tracking correlators vs the CPU oracle (parity tests proper, -m gpu).

Bar: bit-exact -- E/P/L sums, samples per period and the chained remainders --
against the literal restatement of the reference, whose carrier and code NCOs
advance by one rounded fp64 addition per sample (ref src/sdrcmn.c:608-722)."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

F_SF = 16.368e6


def _setup(gc, orc, engine, dtype, f_if, corrn, corrd, corrp, prns, nsamples, seed, buffloc0, ringlen=None,
           amp=60):
    rng = np.random.default_rng(seed)
    shape = (nsamples, 2) if dtype == 2 else (nsamples,)
    data = rng.integers(-amp, amp + 1, size=shape, dtype=np.int8)
    # include the int8 extremes
    data.reshape(-1)[:4] = [-128, 127, -128, 127]
    ringlen = ringlen or nsamples
    engine.ring_create(1, dtype, ringlen)
    engine.ring_push_raw(1, data, nsamples)
    chans = [gc.Channel(p, dtype=dtype, f_if=f_if, corrn=corrn, corrd=corrd, corrp=corrp) for p in prns]
    engine.set_channels(chans)
    states, ochs = [], []
    for i, c in enumerate(chans):
        # channel 0 starts like a channel fresh out of acquisition (remcode = remcarr = 0, carrfreq on
        # the 200 Hz grid, codefreq = crate: ref src/sdracq.c:51-55), channel 1 with an integer code phase
        # and a non-dyadic chip step (the chip choice hinges on the rounding of the reference's running
        # sum), the others mid-track
        st = dict(carrfreq=f_if + (rng.uniform(-5000, 5000) if i else 2200.0),
                  codefreq=c.crate + (rng.uniform(-3, 3) if i else 0.0),
                  remcode=(rng.uniform(0.01, 0.99) if i > 1 else 0.0), remcarr=rng.uniform(0, 6.2) if i else 0.0,
                  buffloc=buffloc0 + 1000 * i + (i % 3))
        states.append(st)
        o = orc.make_chan(c.prn, dtype=dtype, f_if=f_if, corrn=corrn, corrd=corrd, corrp=corrp)
        ochs.append(o)
    engine.trk_set_state(states)
    return data, chans, states, ochs


def _oracle_run(orc, ochs, states, data, ringlen, wrpos, nepoch):
    ring = orc.make_ring(data, ringlen, wrpos)
    L = orc.lib()
    ntap = 1 + 2 * ochs[0].corrn
    II = np.zeros((len(ochs), nepoch, ntap))
    QQ = np.zeros_like(II)
    ns = np.zeros((len(ochs), nepoch), np.int32)
    fin = []
    for i, (o, st) in enumerate(zip(ochs, states)):
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = st["carrfreq"], st["codefreq"], st["remcode"], st["remcarr"]
        buffloc = st["buffloc"]
        for e in range(nepoch):
            L.orc_sdrtracking(C.byref(o), C.byref(ring), buffloc)
            assert o.flagtrk == 1
            II[i, e] = np.ctypeslib.as_array(o.II)[:ntap]
            QQ[i, e] = np.ctypeslib.as_array(o.QQ)[:ntap]
            ns[i, e] = o.currnsamp
            buffloc += o.currnsamp
        fin.append(dict(remcode=o.remcode, remcarr=o.remcarr, buffloc=buffloc))
    return II, QQ, ns, fin


@pytest.mark.parametrize("dtype,f_if,corrn,corrd,corrp", [
    (2, 0.0, 2, 3, 3),          # BASELINE config 3: int8 IQ, 5 taps
    (1, 4.092e6, 6, 3, 6),      # frontend/iffile.ini: real IF, 13 taps
    (2, 0.0, 1, 8, 8),          # plain E-P-L
    (1, 4.092e6, 2, 3, 3),
])
def test_trk_batch_matches_oracle(gc, orc, engine, dtype, f_if, corrn, corrd, corrp):
    nsamples = 16 * 8192
    nepoch = 6
    data, chans, states, ochs = _setup(gc, orc, engine, dtype, f_if, corrn, corrd, corrp,
                                       prns=[1, 7, 13, 32], nsamples=nsamples, seed=11 + corrn,
                                       buffloc0=5)
    engine.trk_run(nepoch)
    II, QQ, ns = engine.trk_fetch()
    fin = engine.trk_get_state()
    oII, oQQ, ons, ofin = _oracle_run(orc, ochs, states, data, nsamples, nsamples, nepoch)
    assert np.array_equal(ns, ons)
    assert np.array_equal(II, oII)
    assert np.array_equal(QQ, oQQ)
    for a, b in zip(fin, ofin):
        assert a["remcode"] == b["remcode"] and a["remcarr"] == b["remcarr"] and a["buffloc"] == b["buffloc"]
    # cumsumcorr over the batch (ref src/sdrtrk.c:64-76)
    sI, sQ = engine.trk_fetch_sums()
    assert np.array_equal(sI, II.sum(axis=1))
    assert np.array_equal(sQ, QQ.sum(axis=1))


def test_trk_ring_wrap(gc, orc, engine):
    """A code period that straddles the end of the ring (ref src/sdrrcv.c:508-521)."""
    ringlen = 16 * 4096
    nsamples = ringlen
    data, chans, states, ochs = _setup(gc, orc, engine, 2, 0.0, 2, 3, 3, prns=[3, 9], nsamples=nsamples,
                                       seed=5, buffloc0=ringlen - 9000, ringlen=ringlen)
    # pretend the writer went on for another lap and a bit: same bytes, the periods tracked straddle the end
    # of the ring and lie inside what it holds
    engine.ring_commit(1, ringlen + 40000)
    for s in states:
        s["buffloc"] += ringlen
    engine.trk_set_state(states)
    engine.trk_run(2)
    II, QQ, ns = engine.trk_fetch()
    oII, oQQ, ons, _ = _oracle_run(orc, ochs, states, data, ringlen, 2 * ringlen + 40000, 2)
    assert np.array_equal(ns, ons) and np.array_equal(II, oII) and np.array_equal(QQ, oQQ)


def test_correlator_symbol(gc, orc):
    """The reference-named correlator() (ref src/sdrcmn.c:687-722) through the C ABI."""
    L = gc.lib()
    rng = np.random.default_rng(3)
    for dtype, freq in ((2, 1234.5), (1, 4.092e6 - 2200.0), (2, -3.9e6)):
        n = 16369
        data = rng.integers(-100, 101, size=n * dtype, dtype=np.int8)
        code, crate = gc.gencode(5, gc.CTYPE_L1CA)
        s = np.array([3, 6, 9], np.int32)
        II, QQ = np.zeros(7), np.zeros(7)
        remc, remp = C.c_double(), C.c_double()
        code16 = code.astype(np.int16)
        L.correlator(data.ctypes.data, dtype, 1 / F_SF, n, freq, 0.7, crate + 1.5, 100.25, s.ctypes.data, 3,
                     II.ctypes.data, QQ.ctypes.data, C.byref(remc), C.byref(remp), code16.ctypes.data, 1023)
        oII, oQQ, orc_c, orc_p = orc.correlator(data, dtype, 1 / F_SF, n, freq, 0.7, crate + 1.5, 100.25, s,
                                                code16)
        assert np.array_equal(II, oII) and np.array_equal(QQ, oQQ)
        assert remc.value == orc_c and remp.value == orc_p


def test_consecutive_batches_and_lookahead_planner(gc, orc, engine):
    """Batches chained on the device (the planner runs one batch ahead on its own stream): results and
    the chained state must equal one long oracle run; a state change or another batch length drops
    the look-ahead plan."""
    nsamples = 16 * 8192 * 2
    data, chans, states, ochs = _setup(gc, orc, engine, 2, 0.0, 2, 3, 3, prns=[4, 19, 27], nsamples=nsamples,
                                       seed=77, buffloc0=40)
    got_II, got_QQ = [], []
    for nb in (3, 3, 3, 2):                 # 3,3,3 exercises the look-ahead; 2 forces a re-plan
        engine.trk_run(nb)
        II, QQ, _ = engine.trk_fetch()
        got_II.append(II)
        got_QQ.append(QQ)
    II = np.concatenate(got_II, axis=1)
    QQ = np.concatenate(got_QQ, axis=1)
    oII, oQQ, ons, ofin = _oracle_run(orc, ochs, states, data, nsamples, nsamples, 11)
    assert np.array_equal(II, oII) and np.array_equal(QQ, oQQ)
    fin = engine.trk_get_state()
    for a, b in zip(fin, ofin):
        assert a["remcode"] == b["remcode"] and a["remcarr"] == b["remcarr"] and a["buffloc"] == b["buffloc"]
    # closed-loop style: new frequencies between batches (as pll()/dll() would set them)
    for s, f in zip(states, fin):
        s.update(remcode=f["remcode"], remcarr=f["remcarr"], buffloc=f["buffloc"],
                 carrfreq=s["carrfreq"] + 3.5, codefreq=s["codefreq"] - 0.01)
    engine.trk_set_state(states)
    engine.trk_run(3)
    II2, QQ2, _ = engine.trk_fetch()
    oII2, oQQ2, _, _ = _oracle_run(orc, ochs, states, data, nsamples, nsamples, 3)
    assert np.array_equal(II2, oII2) and np.array_equal(QQ2, oQQ2)


def _code_case(case, rng):
    """(code, crate, coff, n): code shapes that take the correlator's different paths."""
    if case == "multilevel":        # steps other than +-2 between chips: the multiplying look-up loop
        return rng.integers(-3, 4, size=1023).astype(np.int16), 1.023e6 + 2.0, 511.75, 16368
    if case == "constant":          # no chip edge at all
        return np.ones(1023, np.int16), 1.023e6, 17.5, 16368
    if case == "sparse":            # a handful of edges: most lanes of the look-up phase idle
        c = np.ones(1023, np.int16)
        c[[5, 6, 400, 1022]] = -1
        return c, 1.023e6 - 1.0, 1000.9, 16370
    if case == "short_fast":        # 0.7 chips per sample over a 300-chip code: two code periods in the call
        return rng.choice(np.array([-1, 1], np.int16), size=300), 0.7 * F_SF, 3.3, 850
    if case == "slow":              # 200 samples per chip
        return rng.choice(np.array([-1, 1], np.int16), size=1023), F_SF / 200.0, 77.2531, 16368
    raise ValueError(case)


@pytest.mark.parametrize("case", ["multilevel", "constant", "sparse", "short_fast", "slow"])
@pytest.mark.parametrize("dtype", [1, 2])
def test_correlator_symbol_code_shapes(gc, orc, case, dtype):
    """correlator() on codes and chip rates beyond GPS L1CA: multi-level chips, no / few chip edges, many
    chips per sample and many samples per chip (ref src/sdrcmn.c:608-621 wraps any code of length len
    as long as the chip step stays below len)."""
    L = gc.lib()
    rng = np.random.default_rng({"multilevel": 1, "constant": 2, "sparse": 3, "short_fast": 4, "slow": 5}[case] * 10 + dtype)
    code, crate, coff, n = _code_case(case, rng)
    data = rng.integers(-128, 128, size=n * dtype, dtype=np.int8)
    s = np.array([2, 5], np.int32)
    II, QQ = np.zeros(5), np.zeros(5)
    remc, remp = C.c_double(), C.c_double()
    freq = 4.092e6 + 777.0 if dtype == 1 else -2345.6
    L.correlator(data.ctypes.data, dtype, 1 / F_SF, n, freq, 1.1, crate, coff, s.ctypes.data, 2,
                 II.ctypes.data, QQ.ctypes.data, C.byref(remc), C.byref(remp), code.ctypes.data, len(code))
    oII, oQQ, orc_c, orc_p = orc.correlator(data, dtype, 1 / F_SF, n, freq, 1.1, crate, coff, s, code)
    assert np.array_equal(II, oII) and np.array_equal(QQ, oQQ)
    assert remc.value == orc_c and remp.value == orc_p


def test_correlator_symbol_refuses_many_code_periods(gc, orc, capfd):
    """A call that spans far more code periods than sdrtracking() ever asks for (one, ref src/sdrtrk.c:31-32)
    needs more NCO pieces than the tables hold: the symbol reports it and leaves the outputs untouched, like
    the reference's own failure path (ref src/sdrcmn.c:697-702)."""
    L = gc.lib()
    rng = np.random.default_rng(8)
    code = rng.choice(np.array([-1, 1], np.int16), size=10)
    n = 9000
    data = rng.integers(-128, 128, size=2 * n, dtype=np.int8)
    s = np.array([2, 5], np.int32)
    II, QQ = np.full(5, 7.5), np.full(5, -7.5)
    remc, remp = C.c_double(1.0), C.c_double(2.0)
    L.correlator(data.ctypes.data, 2, 1 / F_SF, n, 100.0, 0.0, 2.5 * F_SF, 3.3, s.ctypes.data, 2,
                 II.ctypes.data, QQ.ctypes.data, C.byref(remc), C.byref(remp), code.ctypes.data, len(code))
    assert np.all(II == 7.5) and np.all(QQ == -7.5) and remc.value == 1.0 and remp.value == 2.0


def test_prefix_and_replica_forms_agree(gc, orc, engine, tmp_path):
    """The two independent HIP implementations of the correlator -- prefix sums over chip edges (default)
    and the sample-by-sample replica form (GNSSCORR_TRK_ALGO=replica) -- must agree bit for bit."""
    import json
    import os
    import subprocess
    import sys
    nsamples = 16 * 8192
    data, chans, states, ochs = _setup(gc, orc, engine, 2, 0.0, 2, 3, 3, prns=[2, 11, 30], nsamples=nsamples,
                                       seed=404, buffloc0=123)
    engine.trk_run(5)
    II, QQ, ns = engine.trk_fetch()
    script = tmp_path / "replica.py"
    script.write_text(f"""
import json, sys
import numpy as np
sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})
import gnsscorr_loader
gc = gnsscorr_loader.load()
rng = np.random.default_rng(404)
data = rng.integers(-60, 61, size=({nsamples}, 2), dtype=np.int8)
data.reshape(-1)[:4] = [-128, 127, -128, 127]
eng = gc.Engine(0)
eng.ring_create(1, 2, {nsamples})
eng.ring_push_raw(1, data, {nsamples})
eng.set_channels([gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in (2, 11, 30)])
eng.trk_set_state(json.loads({repr(json.dumps(states))}))
eng.trk_run(5)
II, QQ, ns = eng.trk_fetch()
print(json.dumps(dict(II=II.tolist(), QQ=QQ.tolist(), ns=ns.tolist())))
""")
    # ... and so must the default form without its per-period edge table (the correlator then finds the
    # start sample of every chip edge itself, as the closed-loop kernel does) and with the older planner
    # chain (every period certifying its own binade crossings instead of checking discovered ones)
    for env_add in (dict(GNSSCORR_TRK_ALGO="replica"), dict(GNSSCORR_TRK_NOEDGETAB="1"), dict(GNSSCORR_TRK_NOSPEC="1")):
        env = dict(os.environ, **env_add)
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        r = json.loads(out.stdout.strip().splitlines()[-1])
        assert np.array_equal(np.array(r["ns"]), ns), env_add
        assert np.array_equal(np.array(r["II"]), II) and np.array_equal(np.array(r["QQ"]), QQ), env_add


def test_planner_chain_long_batch(gc, orc, engine):
    """400 code periods per channel in one batch: the planner's chained NCO walks (code / carrier
    remainders, currnsamp) must track the literal loops to the last bit over the whole chain -- any
    one-ulp slip in a remainder would show up in the samples-per-period sequence or the final state."""
    nepoch = 400
    nsamples = 16368 * (nepoch + 12)
    data, chans, states, ochs = _setup(gc, orc, engine, 2, 0.0, 2, 3, 3, prns=[1, 6, 14, 23, 31, 9, 18, 27],
                                       nsamples=nsamples, seed=909, buffloc0=3)
    # spread the frequencies: negative and large carrier offsets, code rates off nominal by up to 8 Hz
    rng = np.random.default_rng(910)
    for i, s in enumerate(states):
        s["carrfreq"] = float(rng.uniform(-9000, 9000)) * (-1 if i % 2 else 1)
        s["codefreq"] = chans[i].crate + float(rng.uniform(-8, 8))
    engine.trk_set_state(states)
    stats = np.zeros(8, dtype=np.uint64)
    gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
    engine.trk_run(nepoch)
    II, QQ, ns = engine.trk_fetch()
    fin = engine.trk_get_state()
    # the batch form of the planner (periods discovered side by side with a bracket around their starts, evaluated in
    # the chain) is what serves these channels; the certified step and the walkers take what it declines
    gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
    assert stats[:3].sum() >= len(chans) * nepoch and stats[3:6].sum() >= len(chans) * nepoch, stats
    assert stats[0] >= 0.95 * stats[:3].sum() and stats[3] >= 0.8 * stats[3:6].sum(), stats
    oII, oQQ, ons, ofin = _oracle_run(orc, ochs, states, data, nsamples, nsamples, nepoch)
    assert np.array_equal(ns, ons)
    for a, b in zip(fin, ofin):
        assert a["remcode"] == b["remcode"] and a["remcarr"] == b["remcarr"] and a["buffloc"] == b["buffloc"]
    assert np.array_equal(II, oII) and np.array_equal(QQ, oQQ)


def test_planner_batch_lengths_at_the_kernels_block_edges(gc, orc, engine):
    """Back-to-back batches of 1, 2, 63, 64, 65, 255, 256, 257 and 3 periods (the chain works in blocks of 64 periods,
    the discovery in chunks of 256; every change of length also drops the look-ahead plan and discovery): every
    batch's sums, sample counts and the final state against the oracle's literal loop."""
    lengths = (1, 2, 63, 64, 65, 255, 256, 257, 3)
    total = sum(lengths)
    nsamples = 16368 * (total + 12)
    data, chans, states, ochs = _setup(gc, orc, engine, 2, 0.0, 2, 3, 3, prns=[2, 19, 28], nsamples=nsamples, seed=321, buffloc0=9)
    states[1]["carrfreq"] = -4321.5          # one falling phase (a period inside one binade), one fast rising one
    states[2]["carrfreq"] = 8765.25
    engine.trk_set_state(states)
    oII, oQQ, ons, ofin = _oracle_run(orc, ochs, states, data, nsamples, nsamples, total)
    done = 0
    for n in lengths:
        engine.trk_run(n)
        II, QQ, ns = engine.trk_fetch()
        assert np.array_equal(ns, ons[:, done:done + n]), n
        assert np.array_equal(II, oII[:, done:done + n]) and np.array_equal(QQ, oQQ[:, done:done + n]), n
        done += n
    for a, b in zip(engine.trk_get_state(), ofin):
        assert a["remcode"] == b["remcode"] and a["remcarr"] == b["remcarr"] and a["buffloc"] == b["buffloc"]


def test_planner_brackets_hold_under_the_checks(gc, orc, engine, tmp_path):
    """The batch planner's chain evaluates a period WITHOUT its checks when the period's exact start lies inside the
    bracket the discovery proved the claims for (gnsscorr_plan.hip).  GNSSCORR_PLAN_VERIFY=1 makes the chain run
    every step with the checks: over 3 batches x 16 channels x 500 periods (rising, falling, fast and slow phases,
    the look-ahead discovery in use from the second batch on) no bracketed start may fail them, nearly every period
    must have a bracket that holds its start, and the sums and states must equal the default run's."""
    import json
    import os
    import subprocess
    import sys
    nepoch, nbatch = 500, 3
    nsamples = 16368 * (nepoch * nbatch + 12)
    script = tmp_path / "verify.py"
    script.write_text(f"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})
import gnsscorr_loader
gc = gnsscorr_loader.load()
rng = np.random.default_rng(515)
data = rng.integers(-60, 61, size=({nsamples}, 2), dtype=np.int8)
eng = gc.Engine(0)
eng.ring_create(1, 2, {nsamples})
eng.ring_push_raw(1, data, {nsamples})
chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in range(1, 17)]
eng.set_channels(chans)
states = [dict(carrfreq=float(rng.uniform(-9000, 9000)), codefreq=c.crate + float(rng.uniform(-6, 6)),
               remcode=float(rng.uniform(0.01, 0.99)), remcarr=float(rng.uniform(0, 6.2)), buffloc=5 + 700 * i)
          for i, c in enumerate(chans)]
states[0].update(carrfreq=2200.0, codefreq=chans[0].crate, remcode=0.0, remcarr=0.0)     # fresh out of acquisition
eng.trk_set_state(states)
stats = np.zeros(8, dtype=np.uint64)
gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
sums = []
for b in range({nbatch}):
    eng.trk_run({nepoch})
    II, QQ, ns = eng.trk_fetch()
    sums.append([float(II.sum()), float(QQ.sum()), int(ns.sum()), float(np.abs(II).max())])
fin = eng.trk_get_state()
gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
print(json.dumps(dict(sums=sums, fin=[[f["remcode"].hex(), f["remcarr"].hex(), int(f["buffloc"])] for f in fin], stats=stats.tolist())))
""")
    runs = {}
    for mode in ("0", "1"):
        env = dict(os.environ, GNSSCORR_PLAN_VERIFY=mode)
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        runs[mode] = json.loads(out.stdout.strip().splitlines()[-1])
    for mode, r in runs.items():
        st = r["stats"]
        total = sum(st[:3])                             # (the batch planned ahead of the last run is counted too)
        assert total >= 16 * nepoch * nbatch and sum(st[3:6]) == total, (mode, st)
        assert st[6] == 0, (mode, st)                   # no bracketed start failed a check
        # the brackets hold the exact starts (channel 0's sums hit their thresholds exactly: only brackets a few ulps
        # wide pass at both ends, and those its starts may miss -- it then takes the step with the checks)
        assert st[7] <= total // 16 + total // 1000, (mode, st)
        assert st[0] >= 0.93 * total and st[3] >= 0.93 * total, (mode, st)      # (channel 0 has no brackets: its sums hit every threshold exactly)
    assert runs["0"]["sums"] == runs["1"]["sums"] and runs["0"]["fin"] == runs["1"]["fin"]


def test_planner_chain_wide_correlator_spacing(gc, orc, engine):
    """The shipped front-end files' correlator set (CORRN=6, CORRD=3, CORRP=6: 13 taps, outermost 18 samples,
    ref frontend/iffile.ini) on int8 IQ: the replica starts 18 samples before the period and ends 18 after it,
    which the batch planner's widest tail instance covers -- 120 periods chained, bit for bit, and served by
    the batch form."""
    nepoch = 120
    nsamples = 16368 * (nepoch + 12)
    data, chans, states, ochs = _setup(gc, orc, engine, 2, 0.0, 6, 3, 6, prns=[4, 17, 25, 32], nsamples=nsamples,
                                       seed=1213, buffloc0=40)
    stats = np.zeros(8, dtype=np.uint64)
    gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
    engine.trk_run(nepoch)
    II, QQ, ns = engine.trk_fetch()
    fin = engine.trk_get_state()
    gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
    assert stats[0] >= 0.9 * stats[:3].sum() and stats[:3].sum() >= len(chans) * nepoch, stats
    oII, oQQ, ons, ofin = _oracle_run(orc, ochs, states, data, nsamples, nsamples, nepoch)
    assert np.array_equal(ns, ons)
    for a, b in zip(fin, ofin):
        assert a["remcode"] == b["remcode"] and a["remcarr"] == b["remcarr"] and a["buffloc"] == b["buffloc"]
    assert np.array_equal(II, oII) and np.array_equal(QQ, oQQ)


@pytest.mark.parametrize("f_sf,dtype", [(20e6, 2), (26e6, 2), (26e6, 1)])
def test_trk_20msps_period(gc, orc, engine, f_sf, dtype):
    """A 20 Msps front end (ref frontend/stereo_L1G1.ini): 20000 samples per code period, twenty rounds of the
    correlator (one wavefront each) per period; at 26 Msps the period's 26 rounds no longer fit one workgroup
    (GC_MAXR) and two workgroups share it, for IQ and for real samples."""
    nper = int(f_sf * 1e-3)
    nsamples = nper * 8
    rng = np.random.default_rng(2000)
    data = rng.integers(-100, 101, size=(nsamples, 2) if dtype == 2 else (nsamples,), dtype=np.int8)
    engine.ring_create(1, dtype, nsamples)
    engine.ring_push_raw(1, data, nsamples)
    chans = [gc.Channel(p, dtype=dtype, f_sf=f_sf, f_if=0.0, corrn=2, corrd=4, corrp=4) for p in (7, 24)]
    assert chans[0].nsamp == nper
    engine.set_channels(chans)
    states = [dict(carrfreq=float(rng.uniform(-3000, 3000)), codefreq=c.crate + float(rng.uniform(-1, 1)),
                   remcode=float(rng.uniform(0.1, 0.9)), remcarr=float(rng.uniform(0, 6)), buffloc=11 + 300 * i)
              for i, c in enumerate(chans)]
    engine.trk_set_state(states)
    engine.trk_run(5)
    II, QQ, ns = engine.trk_fetch()
    ochs = [orc.make_chan(c.prn, dtype=dtype, f_sf=f_sf, f_if=0.0, corrn=2, corrd=4, corrp=4) for c in chans]
    oII, oQQ, ons, ofin = _oracle_run(orc, ochs, states, data, nsamples, nsamples, 5)
    assert np.array_equal(ns, ons) and ns[:, 1:].min() >= nper - 1      # (the first period is cut by remcode)
    assert np.array_equal(II, oII) and np.array_equal(QQ, oQQ)
    # (acquisition of such a stream: tests/test_gpu_acq.py::test_acquisition_long_periods_65536_point_transform)


def test_closed_loop_from_acquisition_state(gc, orc, engine, synth):
    """The flow every channel goes through (ref src/sdrmain.c:264-312): tracking starts from what
    sdracquisition() leaves (remcode = remcarr = 0, carrfreq on the 200 Hz grid, codefreq = crate, ref
    src/sdracq.c:51-55) and pll()/dll() move the frequencies after every period (ref src/sdrtrk.c:95-150).
    40 periods on a 48 dB-Hz signal, host-side loop filters, one trk_run per period: sums, samples per
    period and state equal to the literal oracle bit for bit."""
    prns = [3, 11, 22]
    nper = 48
    dop, cph = [1234.0, -2750.0, 4100.0], [100.3, 700.9, 13.0]
    codes = {p: gc.gencode(p, gc.CTYPE_L1CA) for p in prns}
    sats = [dict(prn=p, doppler=d, codephase=c, cn0=48.0, phase=0.3 * i) for i, (p, d, c) in enumerate(zip(prns, dop, cph))]
    sig = synth.make_if(codes, 16368 * nper, f_sf=F_SF, f_if=0.0, dtype=2, sats=sats, seed=31)
    nsamples = sig.shape[0]
    engine.ring_create(1, 2, nsamples)
    engine.ring_push_raw(1, sig, nsamples)
    chans = [gc.Channel(p, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3) for p in prns]
    engine.set_channels(chans)
    L = orc.lib()
    ring = orc.make_ring(sig, nsamples, nsamples)
    for i, (c, f) in enumerate(zip(chans, (1200.0, -2800.0, 4200.0))):
        o = orc.make_chan(c.prn, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
        o.acq.acqfreq = f
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = f, c.crate, 0.0, 0.0
        # code phase of the synthesized signal -> first sample of a code period (what acqcodei gives)
        buffloc = int(round((1023 - cph[i]) * 16)) % 16368
        for e in range(40):
            # (the other channels are parked: a zero chip rate makes their periods empty)
            sts = [dict(carrfreq=0.0, codefreq=0.0, remcode=0.0, remcarr=0.0, buffloc=0) for _ in chans]
            sts[i] = dict(carrfreq=o.carrfreq, codefreq=o.codefreq, remcode=o.remcode, remcarr=o.remcarr, buffloc=buffloc)
            engine.trk_set_state(sts)
            engine.trk_run(1)
            II, QQ, ns = engine.trk_fetch()
            st = engine.trk_get_state()[i]
            L.orc_sdrtracking(C.byref(o), C.byref(ring), buffloc)
            assert o.flagtrk == 1
            assert ns[i, 0] == o.currnsamp, (i, e)
            assert np.array_equal(II[i, 0], np.ctypeslib.as_array(o.II)[:5]), (i, e)
            assert np.array_equal(QQ[i, 0], np.ctypeslib.as_array(o.QQ)[:5]), (i, e)
            assert st["remcode"] == o.remcode and st["remcarr"] == o.remcarr, (i, e)
            L.orc_cumsumcorr(C.byref(o), 1)
            L.orc_pll(C.byref(o), 0, o.ctime)
            L.orc_dll(C.byref(o), 0, o.ctime)
            L.orc_clearcumsumcorr(C.byref(o))
            buffloc += o.currnsamp
        # the loops pulled in: prompt power well above the early/late mean by the end
        assert abs(o.carrfreq - dop[i]) < 150.0


def test_batch_outside_the_ring_is_refused(gc, orc, engine):
    """The reference tracks a period only once it is in the buffer (ref src/sdrtrk.c:26-30) and stops when the
    buffer overruns (ref src/sdrrcv.c:325-349).  A batch that runs past the write position, or over samples
    that were overwritten since, is reported by the fetch instead of handing out sums of other samples."""
    ringlen = 16 * 8192
    data, chans, states, ochs = _setup(gc, orc, engine, 2, 0.0, 2, 3, 3, prns=[3, 9], nsamples=ringlen, seed=6,
                                       buffloc0=100)
    engine.trk_run(7)                       # 7 periods fit 131072 samples
    engine.trk_fetch()
    engine.trk_run(3)                       # ... the next three do not
    with pytest.raises(gc.GnsscorrError, match="outside what the IF ring holds"):
        engine.trk_fetch()
    # overwritten: the writer is more than a ring ahead of the period asked for
    engine.ring_commit(1, 3 * ringlen)
    engine.trk_set_state(states)
    engine.trk_run(1)
    with pytest.raises(gc.GnsscorrError, match="outside what the IF ring holds"):
        engine.trk_fetch()
    # a ring shorter than a code period is refused when the channels are set
    engine.ring_create(1, 2, 8192)
    with pytest.raises(gc.GnsscorrError, match="shorter than a code period"):
        engine.set_channels([gc.Channel(3, dtype=2, f_if=0.0)])
