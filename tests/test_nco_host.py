"""The product's NCO emulation (csrc/gnsscorr_nco.h: piecewise-linear reconstruction of the reference's
sequential fp64 running sums) against the oracle's literal loops, on the CPU.

Bar: every LUT index, every chip index and both returned remainders identical (ref src/sdrcmn.c:608-621,
633-669)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host", "nco_host.cpp")
HDR = os.path.join(os.path.dirname(HERE), "erlangnetwork-gnsslib-sdr_amd", "csrc", "gnsscorr_nco.h")
F_SF = 16.368e6
DPI = 2.0 * 3.1415926535897932


@pytest.fixture(scope="module")
def nco(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("nco") / "nco_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", so])
    L = C.CDLL(so)
    d, i, vp = C.c_double, C.c_int, C.c_void_p
    L.nco_carrier.argtypes = [d, d, d, i, i, vp, C.POINTER(d)]
    L.nco_code.argtypes = [i, d, i, d, i, i, vp, C.POINTER(d)]
    L.nco_chain.argtypes = [d, d, d, i, i, d, i, d, C.POINTER(d), C.POINTER(d)]
    L.nco_chain_fast.argtypes = [d, d, d, d, d, i, d, i, C.POINTER(i), C.POINTER(d), C.POINTER(d)]
    L.nco_chain_cert.argtypes = [d, d, d, d, d, i, d, i, C.POINTER(i), C.POINTER(d), C.POINTER(d), C.POINTER(i)]
    L.nco_code_period.argtypes = [d, d, i, d, i, i, C.POINTER(d)]
    L.nco_carrier_period.argtypes = [d, d, d, i, C.POINTER(d)]
    L.nco_period_tables.argtypes = [d, d, d, d, i, d, i, i, vp, vp, C.POINTER(d), C.POINTER(d)]
    L.nco_claims_chain.argtypes = [d, d, d, d, i, i, d, d, i, d, d, vp, vp, vp, vp]
    L.nco_carrier_fast.argtypes = L.nco_carrier.argtypes
    L.nco_code_fast.argtypes = L.nco_code.argtypes
    return L


def _lut():
    cost = np.floor(np.cos(DPI / 32 * np.arange(32)) * 32 + 0.5).astype(np.int16)
    sint = np.floor(np.sin(DPI / 32 * np.arange(32)) * 32 + 0.5).astype(np.int16)
    return cost, sint


def _carrier_case(nco, orc, phi0, freq, ti, n, cap=48):
    """index sequence from the oracle: real samples of value 1 come out as (cos[idx], sin[idx])"""
    data = np.ones(n, np.int8)
    I, Q = np.zeros(n, np.int16), np.zeros(n, np.int16)
    oprem = orc.lib().orc_mixcarr_seq(data.ctypes.data, 1, ti, n, freq, phi0, I.ctypes.data, Q.ctypes.data)
    idx = np.zeros(n, np.int32)
    prem = C.c_double()
    nseg = nco.nco_carrier(phi0, freq, ti, n, cap, idx.ctypes.data, C.byref(prem))
    assert nseg > 0, f"segment table overflow phi0={phi0} freq={freq}"
    cost, sint = _lut()
    bad = np.flatnonzero((cost[idx] != I) | (sint[idx] != Q))
    assert bad.size == 0, f"phi0={phi0!r} freq={freq!r} n={n}: {bad.size} samples differ, first {bad[:5]}"
    assert prem.value == oprem, f"phi0={phi0!r} freq={freq!r}: prem {prem.value!r} != {oprem!r}"
    # the tabulated fast walker: same indices, same remainder
    idx2 = np.zeros(n, np.int32)
    prem2 = C.c_double()
    nseg2 = nco.nco_carrier_fast(phi0, freq, ti, n, cap + 8, idx2.ctypes.data, C.byref(prem2))
    assert nseg2 > 0 and np.array_equal(idx & 31, idx2 & 31) and prem2.value == oprem, \
        f"fast walker: phi0={phi0!r} freq={freq!r} n={n}"
    return max(nseg, nseg2)


def _code_case(nco, orc, length, coff, smax, ci, n, cap=32):
    code = np.arange(length, dtype=np.int16)
    nt = n + 2 * smax
    rc = np.zeros(nt, np.int16)
    orem = orc.lib().orc_rescode_seq(code.ctypes.data, length, coff, smax, ci, n, rc.ctypes.data)
    chip = np.zeros(nt, np.int32)
    rem = C.c_double()
    nseg = nco.nco_code(length, coff, smax, ci, n, cap, chip.ctypes.data, C.byref(rem))
    if nseg < 0 and nt * ci > 2.5 * length:
        # more than two code periods in one call: outside what sdrtracking() ever asks for (one period,
        # ref src/sdrtrk.c:31-32); the table says so (the C-ABI returns an error) instead of truncating
        assert rem.value == orem
        return 0
    assert nseg > 0, f"segment table overflow coff={coff} ci={ci}"
    bad = np.flatnonzero(chip != rc)
    assert bad.size == 0, f"len={length} coff={coff!r} ci={ci!r} n={n}: {bad.size} chips differ, first {bad[:5]}"
    assert rem.value == orem, f"coff={coff!r} ci={ci!r}: rem {rem.value!r} != {orem!r}"
    chip2 = np.zeros(nt, np.int32)
    rem2 = C.c_double()
    nseg2 = nco.nco_code_fast(length, coff, smax, ci, n, cap + 8, chip2.ctypes.data, C.byref(rem2))
    assert nseg2 > 0 and np.array_equal(chip2, rc) and rem2.value == orem, f"fast walker: coff={coff!r} ci={ci!r} n={n}"
    return max(nseg, nseg2)


def test_carrier_grid_frequencies_from_zero_phase(nco, orc):
    """The state sdracquisition() leaves: remcarr = 0, carrfreq on the 200 Hz grid (ref src/sdracq.c:51-55,
    src/sdrinit.c:633-635) -- zero IF and the 4.092 MHz real IF, both sample rates, acquisition windows too."""
    for f_sf in (16.368e6, 20e6):
        for f_if in (0.0, 4.092e6, 4.0e6):
            for k in range(-35, 36):
                for n in (16368, 32736):
                    _carrier_case(nco, orc, 0.0, f_if + 200.0 * k, 1 / f_sf, n)


def test_carrier_random(nco, orc):
    rng = np.random.default_rng(1)
    worst = 0
    for _ in range(1500):
        f_if = rng.choice([0.0, 4.092e6, -3.94e6, 38.4e3])
        freq = f_if + rng.uniform(-9000, 9000)
        kind = rng.integers(0, 4)
        if kind == 0:
            phi0 = rng.uniform(0, DPI)
        elif kind == 1:
            phi0 = -rng.uniform(0, 1) * 10.0 ** rng.uniform(-3, 6)      # never wrapped when negative (:667)
        elif kind == 2:
            phi0 = 0.0
        else:
            phi0 = rng.uniform(0, 1) * 10.0 ** rng.uniform(-12, 1)
        n = int(rng.integers(16000, 16500))
        worst = max(worst, _carrier_case(nco, orc, float(phi0), float(freq), 1 / F_SF, n))
    assert worst <= 40


def test_carrier_adversarial(nco, orc):
    ti = 1 / F_SF
    cases = [
        (0.0, 0.0), (1.0, 0.0), (-3.0, 0.0),                     # no motion at all
        (0.3, 1e-3), (0.3, -1e-3), (5.0, 1e-9),                  # steps far below the grid of the sum
        (6.0, -5000.0), (0.01, -4.092e6), (DPI, -2200.0),        # positive phase, negative step: through zero
        (-1e-9, 2200.0), (-0.5, 4.092e6), (-100.0, 5000.0), (-1e5, 7.5e6),
        (-1e6, -8.184e6 * 0.999), (1.0, 8.184e6 * 0.999),        # near Nyquist
        (3.0, F_SF / 32), (3.0, F_SF / 64), (0.0, F_SF / 1024), (0.0, -F_SF / 4),   # dyadic steps: exact sums
        (2.0, F_SF / 32 * (1 + 2.0 ** -40)), (0.7, F_SF / 32 * 3 * (1 + 2.0 ** -45)),
    ]
    for phi0, freq in cases:
        for n in (1, 2, 3, 17, 16368, 40000):
            _carrier_case(nco, orc, phi0, freq, ti, n, cap=64)


def test_carrier_ties(nco, orc):
    """Steps that lie exactly half way between two grid points of some binade the sum visits
    (ps = odd * 2^-k): the round-to-even transient of gc_nco_run."""
    rng = np.random.default_rng(7)
    for _ in range(400):
        k = int(rng.integers(40, 53))
        odd = 2 * int(rng.integers(1, 2 ** 20)) + 1
        ps = odd * 2.0 ** -k * rng.choice([-1.0, 1.0])
        freq = ps / 32 * F_SF                                    # freq*32*ti reproduces ps when exact
        phi0 = float(rng.uniform(0, DPI)) * rng.choice([1.0, -30.0])
        _carrier_case(nco, orc, phi0, float(freq), 1 / F_SF, 16368, cap=64)


def test_code_after_acquisition_and_random(nco, orc):
    """coff = 0 with a non-dyadic chip step -- the case in which the closed form picks other chips than the
    running sum (89 of 200 cases in the round-1 review) -- and generic states."""
    rng = np.random.default_rng(2)
    ti = 1 / F_SF
    worst = 0
    for i in range(1200):
        length, crate = (1023, 1.023e6) if i % 3 else (511, 0.511e6)
        codefreq = crate + (rng.uniform(-3, 3) if i % 5 else 0.0)
        ci = ti * codefreq
        coff = [0.0, float(rng.uniform(0, length)), float(rng.integers(0, length)), length - 1e-9][i % 4]
        smax = int(rng.choice([0, 3, 6, 18]))
        n = int((length - (coff % length)) / (codefreq / F_SF))
        if n < 8:
            n = 16368
        worst = max(worst, _code_case(nco, orc, length, coff, smax, ci, n))
    assert worst <= 24


def test_code_adversarial(nco, orc):
    ti = 1 / F_SF
    for length in (1023, 511, 10, 1):
        for ci in (0.0625, ti * 1.023e6, ti * (1.023e6 + 2.5), 0.0625 * (1 + 2.0 ** -44), 0.3, 0.999, 1 / 200.0,
                   3 * 2.0 ** -45, 1e-7):
            if ci >= length:
                continue
            for coff in (0.0, 1e-17, -1e-17, 0.5, length - 1e-13, 2.0 * length - 2.3e-13, 3.5 * length, -0.25):
                for smax in (0, 2, 18):
                    for n in (1, 5, 16368):
                        _code_case(nco, orc, length, coff, smax, ci, n, cap=64)


def test_code_ties(nco, orc):
    rng = np.random.default_rng(9)
    for _ in range(300):
        k = int(rng.integers(40, 56))
        odd = 2 * int(rng.integers(2 ** 30, 2 ** 31)) + 1
        ci = odd * 2.0 ** -k
        if not (1e-3 < ci < 1.0):
            continue
        _code_case(nco, orc, 1023, float(rng.uniform(0, 1023)), 6, ci, 16368, cap=64)


def test_chain_matches_literal_loops_over_many_periods(nco, orc):
    """What the tracking planner chains from period to period (ref src/sdrtrk.c:31-43): currnsamp, remcode,
    remcarr -- 300 periods, every value equal to the literal loops'."""
    rng = np.random.default_rng(5)
    ti = 1 / F_SF
    L = orc.lib()
    for case in range(6):
        carrfreq = [2200.0, -1400.0, 4.0932e6, -3.94e6, 137.77, 5000.0][case]
        codefreq = 1.023e6 + [0.0, 1.4, -2.7, 0.3, 2.9, -0.01][case]
        remcode, remcarr = 0.0, 0.0
        code = np.arange(1023, dtype=np.int16)
        ncert = {}
        for _ in range(300):
            n = int((1023 - remcode) / (codefreq / F_SF))
            data = np.ones(n, np.int8)
            I, Q = np.zeros(n, np.int16), np.zeros(n, np.int16)
            rc = np.zeros(n + 12, np.int16)
            oprem = L.orc_mixcarr_seq(data.ctypes.data, 1, ti, n, carrfreq, remcarr, I.ctypes.data, Q.ctypes.data)
            orem = L.orc_rescode_seq(code.ctypes.data, 1023, remcode, 6, ti * codefreq, n, rc.ctypes.data)
            prem, rem = C.c_double(), C.c_double()
            nco.nco_chain(remcarr, carrfreq, ti, n, 1023, remcode, 6, ti * codefreq, C.byref(prem), C.byref(rem))
            assert prem.value == oprem and rem.value == orem
            # the device planner's own formulation of the step (reciprocal divisions, tabulated walkers)
            n2, prem2, rem2 = C.c_int(), C.c_double(), C.c_double()
            nco.nco_chain_fast(remcarr, carrfreq, ti, F_SF, codefreq, 1023, remcode, 6, C.byref(n2), C.byref(prem2),
                               C.byref(rem2))
            assert n2.value == n and prem2.value == oprem and rem2.value == orem
            # ... and through certified crossings
            used = C.c_int()
            nco.nco_chain_cert(remcarr, carrfreq, ti, F_SF, codefreq, 1023, remcode, 6, C.byref(n2), C.byref(prem2),
                               C.byref(rem2), C.byref(used))
            assert n2.value == n and prem2.value == oprem and rem2.value == orem
            ncert[used.value] = ncert.get(used.value, 0) + 1
            remcode, remcarr = orem, oprem
        # the certified path must be the one that normally runs (both NCOs: used == 3).  Exceptions, all
        # served by the piece walkers: the MHz carriers (phase runs off the 20-binade table), and a carrier
        # exactly on the 200 Hz acquisition grid, whose exact-arithmetic phase hits binade boundaries to
        # the last bit (2200 Hz * 32 / 16.368 MHz * 465 = 2) so that no crossing can be certified there
        if carrfreq in (137.77, 5000.0, -1400.0):
            assert ncert.get(3, 0) >= 295, ncert
        # (the first period starts from code phase exactly 0: its sixth sample lands on the code length to
        # the last bit -- not certifiable either)
        assert ncert.get(3, 0) + ncert.get(2, 0) >= 299, ncert


def _literal_chain(orc, carrfreq, codefreq, remcode, remcarr, nper, smax=6, length=1023):
    L = orc.lib()
    ti = 1 / F_SF
    code = np.arange(length, dtype=np.int16)
    rems, prems, ns = [], [], []
    for _ in range(nper):
        n = int((length - remcode) / (codefreq / F_SF))
        data = np.ones(n, np.int8)
        I, Q = np.zeros(n, np.int16), np.zeros(n, np.int16)
        rc = np.zeros(n + 2 * smax, np.int16)
        remcarr = L.orc_mixcarr_seq(data.ctypes.data, 1, ti, n, carrfreq, remcarr, I.ctypes.data, Q.ctypes.data)
        remcode = L.orc_rescode_seq(code.ctypes.data, length, remcode, smax, ti * codefreq, n, rc.ctypes.data)
        rems.append(remcode)
        prems.append(remcarr)
        ns.append(n)
    return np.array(rems), np.array(prems), np.array(ns)


@pytest.mark.parametrize("shift_code,shift_car", [(0.0, 0.0), (0.03, 0.005), (-3e-9, 2e-9)])
def test_period_steps_on_claims(nco, orc, shift_code, shift_car):
    """The batch planner's branch-free form (gc_code_claims_step / gc_carrier_claims_step): the structure of
    every period discovered from closed-form starts, evaluated and checked from the exact ones.  Every chained
    value equals the literal loops'; with honest starts the evaluation serves nearly every period."""
    ti = 1 / F_SF
    nper = 250
    cases = [(2200.3, 1.023e6 + 1.4, 0.31, 1.7), (-1400.7, 1.023e6 - 2.7, 511.2, 0.0), (137.77, 1.023e6 + 2.9, 0.0, 6.1),
             (4999.1, 1.023e6 - 0.01, 1022.4, 3.3), (-4876.2, 1.023e6 + 3.1, 17.0, -40.0), (0.0, 1.023e6, 0.0, 0.0),
             (6200.0, 1.023e6 + 0.7, 3.0, 0.001), (-12.5, 1.023e6 - 1.1, 800.0, 2.0)]
    for carrfreq, codefreq, remcode0, remcarr0 in cases:
        orem, oprem, on = _literal_chain(orc, carrfreq, codefreq, remcode0, remcarr0, nper)
        rem, prem = np.zeros(nper), np.zeros(nper)
        n, hits = np.zeros(nper, np.int32), np.zeros(6, np.int32)
        nco.nco_claims_chain(ti, F_SF, carrfreq, codefreq, 1023, 6, remcode0, remcarr0, nper, shift_code, shift_car,
                             rem.ctypes.data, prem.ctypes.data, n.ctypes.data, hits.ctypes.data)
        assert np.array_equal(n, on), (carrfreq, codefreq)
        assert np.array_equal(rem, orem), (carrfreq, codefreq, hits)
        assert np.array_equal(prem, oprem), (carrfreq, codefreq, hits)
        # (a slowly falling phase that is still positive shrinks towards zero: not a shape of the period steps,
        # the walkers serve it until it has changed sign)
        # A falling phase is never wrapped: while it is small, a period now and then crosses into the next
        # binade above the evaluation's window and goes to the certified step.
        if shift_code == 0.0 and abs(carrfreq) > 100.0 and codefreq != 1.023e6:
            assert hits[0] >= nper - 5 and hits[1] >= nper - 15, (carrfreq, codefreq, hits)


def test_period_steps_on_claims_random_channels(nco, orc):
    """random channel states (tie binades of the chip step included: about half of all step values have
    one inside the table), 60 periods each"""
    rng = np.random.default_rng(77)
    ti = 1 / F_SF
    nper = 60
    served = np.zeros(2, np.int64)
    for _ in range(40):
        carrfreq = float(rng.uniform(-6000, 6000))
        codefreq = 1.023e6 + float(rng.uniform(-3, 3))
        remcode0, remcarr0 = float(rng.uniform(0, 1023)), float(rng.uniform(0, 6.28))
        orem, oprem, on = _literal_chain(orc, carrfreq, codefreq, remcode0, remcarr0, nper)
        rem, prem = np.zeros(nper), np.zeros(nper)
        n, hits = np.zeros(nper, np.int32), np.zeros(6, np.int32)
        nco.nco_claims_chain(ti, F_SF, carrfreq, codefreq, 1023, 6, remcode0, remcarr0, nper, 0.0, 0.0,
                             rem.ctypes.data, prem.ctypes.data, n.ctypes.data, hits.ctypes.data)
        assert np.array_equal(n, on) and np.array_equal(rem, orem) and np.array_equal(prem, oprem), \
            (carrfreq, codefreq, remcode0, remcarr0, hits)
        served += hits[:2]
    assert served[0] >= 0.97 * 40 * nper and served[1] >= 0.90 * 40 * nper, served


def test_certified_chain_random_states(nco, orc):
    """Certified crossings against the literal loops on random (also unlocked) states: whatever path the
    planner takes, the three chained values are the reference's."""
    rng = np.random.default_rng(77)
    ti = 1 / F_SF
    L = orc.lib()
    code = np.arange(1023, dtype=np.int16)
    used_hist = {}
    for it in range(3000):
        carrfreq = float(rng.choice([0.0, 4.092e6]) + rng.uniform(-9000, 9000))
        codefreq = 1.023e6 + float(rng.uniform(-8, 8)) * (it % 7 != 0)
        remcode = float([0.0, rng.uniform(0, 0.2), rng.uniform(0, 1023), rng.integers(0, 1023)][it % 4])
        remcarr = float([0.0, rng.uniform(0, DPI), -rng.uniform(0, 1e4), rng.uniform(0, 1e-6)][(it // 4) % 4])
        smax = int(rng.choice([3, 6, 18]))
        n = int((1023 - remcode) / (codefreq / F_SF))
        if n < 1:
            continue
        data = np.ones(n, np.int8)
        I, Q = np.zeros(n, np.int16), np.zeros(n, np.int16)
        rc = np.zeros(n + 2 * smax, np.int16)
        oprem = L.orc_mixcarr_seq(data.ctypes.data, 1, ti, n, carrfreq, remcarr, I.ctypes.data, Q.ctypes.data)
        orem = L.orc_rescode_seq(code.ctypes.data, 1023, remcode, smax, ti * codefreq, n, rc.ctypes.data)
        n2, prem2, rem2, used = C.c_int(), C.c_double(), C.c_double(), C.c_int()
        nco.nco_chain_cert(remcarr, carrfreq, ti, F_SF, codefreq, 1023, remcode, smax, C.byref(n2), C.byref(prem2),
                           C.byref(rem2), C.byref(used))
        assert n2.value == n and prem2.value == oprem and rem2.value == orem, (it, carrfreq, codefreq, remcode, remcarr)
        used_hist[used.value] = used_hist.get(used.value, 0) + 1
    assert used_hist.get(3, 0) + used_hist.get(2, 0) > 2000, used_hist


def test_code_period_step(nco, orc):
    """The planner's shape-specialised code step: whenever it applies, its remainder is the literal loop's;
    and for a tracked channel (small remcode, any chip rate near nominal, any tap span) it applies."""
    rng = np.random.default_rng(123)
    ti = 1 / F_SF
    L = orc.lib()
    applied = 0
    total = 0
    for it in range(4000):
        length, crate = (1023, 1.023e6) if it % 4 else (511, 0.511e6)
        codefreq = crate + float(rng.uniform(-8, 8)) * (it % 9 != 0)
        ci = ti * codefreq
        remcode = float([rng.uniform(0, ci), 0.0, rng.uniform(-ci, 2 * ci), rng.uniform(0, length)][it % 4 if it % 16 else 3])
        smax = int(rng.choice([3, 6, 9, 18, 40]))
        n = int((length - remcode) / (codefreq / F_SF))
        if n < 100:
            continue
        code = np.arange(length, dtype=np.int16)
        rc = np.zeros(n + 2 * smax, np.int16)
        orem = L.orc_rescode_seq(code.ctypes.data, length, remcode, smax, ci, n, rc.ctypes.data)
        rem = C.c_double()
        ok = nco.nco_code_period(ti, codefreq, length, remcode, smax, n + 2 * smax, C.byref(rem))
        tracked = 0.0 < remcode < ci
        total += tracked
        if ok:
            assert rem.value == orem, (it, codefreq, remcode, smax, rem.value, orem)
            applied += tracked
    assert total > 800 and applied >= 0.99 * total, (applied, total)


def test_carrier_period_step(nco, orc):
    """The planner's carrier step: its remainder is the literal loop's whenever it applies, and it applies to
    the states a tracked channel has (zero IF and the 4 MHz IF of frontend/iffile.ini, either Doppler sign,
    phases wrapped to [0, 2 pi] or -- negative -- grown for seconds)."""
    rng = np.random.default_rng(321)
    ti = 1 / F_SF
    L = orc.lib()
    applied = total = 0
    for it in range(4000):
        f_if = [0.0, 4.092e6, 0.0, -3.94e6][it % 4]
        freq = f_if + float(rng.uniform(-9000, 9000))
        if freq > 0:
            remcarr = float([rng.uniform(0, DPI), 0.0, rng.uniform(0, 1e-3)][it % 3])
        else:
            remcarr = -float(rng.uniform(0, 1)) * 10.0 ** float(rng.uniform(-2, 5.5))
        n = int(rng.integers(16300, 16400))
        data = np.ones(n, np.int8)
        I, Q = np.zeros(n, np.int16), np.zeros(n, np.int16)
        oprem = L.orc_mixcarr_seq(data.ctypes.data, 1, ti, n, freq, remcarr, I.ctypes.data, Q.ctypes.data)
        prem = C.c_double()
        ok = nco.nco_carrier_period(ti, freq, remcarr, n, C.byref(prem))
        total += 1
        if ok:
            assert prem.value == oprem, (it, freq, remcarr, prem.value, oprem)
            applied += 1
    assert applied >= 0.9 * total, (applied, total)


def test_period_steps_emit_exact_tables(nco, orc):
    """The period steps with table emission (the closed-loop kernel's planner): every LUT index and every chip
    read back from the emitted piece tables equals the literal loops'."""
    rng = np.random.default_rng(55)
    ti = 1 / F_SF
    L = orc.lib()
    cost, sint = _lut()
    code = np.arange(1023, dtype=np.int16)
    seen = {1: 0, 2: 0}
    for it in range(600):
        freq = [0.0, 4.092e6][it % 2] + float(rng.uniform(-6000, 6000))
        remcarr = float(rng.uniform(0, DPI)) if freq > 0 else -float(rng.uniform(0, 3e4))
        codefreq = 1.023e6 + float(rng.uniform(-5, 5))
        ci = ti * codefreq
        remcode = float(rng.uniform(0, ci))
        smax = int(rng.choice([3, 6, 18]))
        n = int((1023 - remcode) / (codefreq / F_SF))
        data = np.ones(n, np.int8)
        I, Q = np.zeros(n, np.int16), np.zeros(n, np.int16)
        rc = np.zeros(n + 2 * smax, np.int16)
        oprem = L.orc_mixcarr_seq(data.ctypes.data, 1, ti, n, freq, remcarr, I.ctypes.data, Q.ctypes.data)
        orem = L.orc_rescode_seq(code.ctypes.data, 1023, remcode, smax, ci, n, rc.ctypes.data)
        idx, chip = np.zeros(n, np.int32), np.zeros(n + 2 * smax, np.int32)
        prem, rem = C.c_double(), C.c_double()
        r = nco.nco_period_tables(ti, freq, remcarr, codefreq, 1023, remcode, smax, n, idx.ctypes.data, chip.ctypes.data,
                                  C.byref(prem), C.byref(rem))
        if r & 1:
            assert prem.value == oprem and np.array_equal(cost[idx & 31], I) and np.array_equal(sint[idx & 31], Q), (it, freq, remcarr)
            seen[1] += 1
        if r & 2:
            assert rem.value == orem and np.array_equal(chip, rc), (it, codefreq, remcode, smax)
            seen[2] += 1
    assert seen[1] > 500 and seen[2] > 590, seen
