"""The identity behind the tracking correlator's prefix-sum form, checked in numpy on the
CPU: for a piecewise-constant resampled code, a tap's sum over a stretch of samples equals the last
chip's value times the stretch total plus one prefix look-up per chip edge,

    sum_k x[k] c[T(k + off)]  =  c_b P(S) + sum_{m=a+1..b} (c_{m-1} - c_m) P(clamp(B_m - off)),

with T(j) = trunc(fma(j, ci, cs)) the chip under replica position j (ref src/sdrcmn.c:608-621 in closed
form) and B_m = min{j : T(j) >= m}.  The kernel applies it per round of 4096 samples with the edge list
of the code; this test restates that decomposition (rounds, edge list, rank table, clamping) in plain
integer numpy and compares it with the direct sample-by-sample sum."""
import numpy as np
import pytest


def chip_T(j, ci, cs):
    # one rounding for j*ci + cs, like the fused operation: extended precision stands in for it
    return np.floor(np.asarray(j, np.longdouble) * np.longdouble(ci) + np.longdouble(cs)).astype(np.int64)


def edge_tables(code):
    clen = len(code)
    edges, rank = [], np.zeros(clen, np.int64)
    for m in range(clen):
        d = int(code[(m - 1) % clen]) - int(code[m])
        if d:
            edges.append((m, d))
        rank[m] = len(edges)
    return edges, rank


def prefix_form(x, code, ci, cs, toffs, rsamp):
    """Per-round evaluation as trk_corr_ps_kernel organises it (klo = 0)."""
    n, clen = len(x), len(code)
    edges, rank = edge_tables(code)
    nedge = len(edges)
    smax2 = max(toffs)
    out = np.zeros(len(toffs), np.int64)
    for kl in range(0, n, rsamp):
        kend = min(kl + rsamp, n)
        seg = np.zeros(rsamp, np.int64)
        seg[:kend - kl] = x[kl:kend]
        P = np.concatenate([[0], np.cumsum(seg)])            # P[e] = sum of the round's first e samples
        ca, cb = int(chip_T(kl, ci, cs)), int(chip_T(kend - 1 + smax2, ci, cs))
        q0 = (ca // clen) * nedge + rank[ca % clen]
        q1 = (cb // clen) * nedge + rank[cb % clen]
        for q in range(q0, q1):
            w, idx = divmod(q, nedge)
            m, d = edges[idx]
            m += w * clen
            # B_m by the estimate + correction the kernel uses
            jc = max(int(np.ceil((m - cs) / ci)), 1)
            B = jc - 1 if chip_T(jc - 1, ci, cs) >= m else (jc if chip_T(jc, ci, cs) >= m else jc + 1)
            assert chip_T(B, ci, cs) >= m and (B == 0 or chip_T(B - 1, ci, cs) < m)
            for t, off in enumerate(toffs):
                e = min(max(B - off - kl, 0), rsamp)
                out[t] += d * P[e]
        out += int(code[cb % clen]) * P[rsamp]
    return out


@pytest.mark.parametrize("case", ["l1ca", "multilevel", "few_edges", "fast_short", "constant"])
def test_edge_sum_equals_sample_sum(case):
    rng = np.random.default_rng({"l1ca": 1, "multilevel": 2, "few_edges": 3, "fast_short": 4, "constant": 5}[case])
    n = 16368 + 37
    x = rng.integers(-8128, 8129, size=n).astype(np.int64)          # carrier-mixed samples, |x| <= 127*64
    if case == "l1ca":
        code, ci, cs = rng.choice([-1, 1], size=1023), 1.023e6 * (1 + 3e-6) / 16.368e6, 511.37
    elif case == "multilevel":
        code, ci, cs = rng.integers(-3, 4, size=1023), 0.0625, 1000.75
    elif case == "few_edges":
        code = np.ones(1023, np.int64)
        code[[3, 4, 700]] = -1
        ci, cs = 0.0624993, 2.5
    elif case == "fast_short":
        code, ci, cs = rng.choice([-1, 1], size=10), 2.5, 3.3
    else:
        code, ci, cs = np.ones(511, np.int64), 0.03125, 17.0
    toffs = [6, 3, 9, 0, 12]                                        # smax + {0, -3, +3, -6, +6}, smax = 6
    clen = len(code)
    direct = np.array([int(np.sum(x * code[chip_T(np.arange(n) + off, ci, cs) % clen])) for off in toffs])
    for rsamp in (4096, 2048):
        assert np.array_equal(prefix_form(x, code, ci, cs, toffs, rsamp), direct)
