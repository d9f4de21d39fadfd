// gnsscorr_nco.h -- exact emulation of the reference's sequential fp64 NCOs.
//
// The reference advances its carrier phase and its code phase by one rounded
// fp64 addition per sample:
//     mixcarr():  phi  += ps    (ref src/sdrcmn.c:653,660)
//     rescode():  coff += ci    (ref src/sdrcmn.c:616, with the lazy wrap of :617)
// and picks the LUT entry / chip from the truncated running sum.  A parallel
// kernel cannot run that loop, but it does not have to: while the running sum x
// stays inside one binade [2^e, 2^(e+1)) every sum is rounded to the same grid
// u = 2^(e-52), so fl(x + s) = x + d with the CONSTANT step d = RN_u(s) (round to
// nearest multiple of u; if s lies exactly half way between two multiples the
// tie goes to the even multiplier, which after at most one step is a constant
// step too).  The sequence is therefore piecewise linear with one piece per
// binade visited: ~6-17 pieces per code period for the carrier, ~14 for the code.
//
// gc_nco_run() gives the length of the piece that starts at x; the walkers below
// chain pieces with one literal addition at every piece boundary, so every value
// they produce is bit-identical to the reference's loop.  The tracking planner
// uses them for the chained remainders (remcode, remcarr), the per-unit expansion
// emits the pieces as segment tables, and the correlator / acquisition kernels
// index the carrier LUT and the code from those tables in integer arithmetic.
//
// Host and device code: tests/test_nco_host.py compiles this header with g++ and
// checks it against the oracle's literal loops.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define GC_HD __host__ __device__ inline
#define GC_HDM __host__ __device__
#else
#define GC_HD static inline
#define GC_HDM
#endif
// no fused multiply-add may be formed from the separate operations written below
// (gcc: compile with -ffp-contract=off)
#if defined(__clang__)
#define GC_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define GC_FP_STRICT
#endif

// correctly rounded fp64 division where the reference divides (device: never the fast reciprocal form)
#if defined(__HIP_DEVICE_COMPILE__)
#define GC_DDIV(a, b) __ddiv_rn((a), (b))
#else
#define GC_DDIV(a, b) ((a) / (b))
#endif

#define GC_NCO_DPI   (2.0*3.1415926535897932)     // DPI with the reference's PI literal (ref src/sdr.h:103-104)
#define GC_NCO_CDIV  32.0

GC_HD uint64_t gc_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
GC_HD double gc_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

// Sequence x_0 = x, x_{i+1} = fl(x_i + s).  Returns m in [0, cap] and *d such that
// x_i = x + i*(*d) EXACTLY (every such value is representable) for i = 0..m.
// m = 0 promises nothing beyond x itself: the caller takes one literal step.
//
// Why: let e = exponent(x), u = 2^(e-52), x = +-A u with A in [2^52, 2^53), s = (b + r) u
// with integer b and |r| <= 1/2.  As long as the exact sum stays inside the binade
// its rounding is to multiples of u, i.e. fl(x + s) = x + b u when |r| < 1/2.  For
// |r| = 1/2 the tie goes to the even multiplier: from an even A the step is the
// even one of {b, b +- 1} (= rint(s/u)) and the result is even again; an odd A is
// left to the caller's literal step.  The run is cut so that every x_i keeps a
// distance of one u from both ends of the binade: then the exact sums are inside it.
GC_HD int64_t gc_nco_run(double x, double s, int64_t cap, double *d)
{
    GC_FP_STRICT
    *d = 0.0;
    if (cap <= 0) return 0;
    if (s == 0.0) return cap;
    const uint64_t ux = gc_d2u(x), us = gc_d2u(s);
    const int ex = (int)((ux >> 52) & 0x7FF), es = (int)((us >> 52) & 0x7FF);
    if (ex == 0 || ex == 0x7FF || es == 0 || es == 0x7FF) return 0;      // zero, subnormal, inf, nan
    // t = s / u = s * 2^(1075 - ex), exact while its exponent stays in range
    const int et = es + 1075 - ex;
    if (et >= 1023 + 51) return 0;               // |s| >= |x| / 4: the binade changes within a few steps
    double b = 0.0;
    bool tie = false;
    if (et >= 1023 - 1) {                        // |t| >= 1/2
        const double t = gc_u2d((us & 0x800FFFFFFFFFFFFFull) | ((uint64_t)et << 52));
        b = rint(t);                             // nearest integer, ties to even
        tie = fabs(t - b) == 0.5;                // (exact: |t| < 2^51)
    }
    const uint64_t A = (ux & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
    if (tie && (A & 1)) return 0;
    if (b == 0.0) return cap;                    // |s| < u/2 (or the tie keeps an even A): the sum never moves
    const double B = fabs(b);
    const bool grow = ((ux >> 63) != 0) == (b < 0.0);
    // multipliers stay in [2^52 + 1, 2^53 - 1]
    double room;
    if (grow) room = (double)(0x001FFFFFFFFFFFFFull - A);
    else room = A > 0x0010000000000000ull ? (double)(A - 0x0010000000000001ull) : -1.0;
    if (room < B) return 0;
    double q = floor(room / B);
    if (fma(-q, B, room) < 0.0) q -= 1.0;        // (all integers below 2^53: exact)
    int64_t m = q >= 9.0e18 ? cap : (int64_t)q;
    if (m > cap) m = cap;
    *d = ldexp(b, ex - 1075);                    // b u, exact
    return m;
}

// ---------------------------------------------------------------------------
// carrier: ref src/sdrcmn.c:649-668
// ---------------------------------------------------------------------------
// start value and step of the running phase in LUT steps (ref :649-650)
GC_HD double gc_carrier_phis(double phi0)
{
    GC_FP_STRICT
    return GC_DDIV(phi0 * GC_NCO_CDIV, GC_NCO_DPI);
}
GC_HD double gc_carrier_ps(double freq, double ti)
{
    GC_FP_STRICT
    return freq * GC_NCO_CDIV * ti;
}

// Walks n samples from x (= phis); emit(k0, x0, d, count) is called once per piece:
// samples k0 .. k0+count-1 hold x0 + i d.  Returns the value after n additions.
template <class Emit>
GC_HD double gc_carrier_walk(double x, double ps, int n, Emit &emit)
{
    GC_FP_STRICT
    int k = 0;
    while (k < n) {
        double d;
        const int64_t m = gc_nco_run(x, ps, (int64_t)(n - 1 - k), &d);
        emit(k, x, d, (int)m + 1);
        x = fma((double)m, d, x);                // exact
        k += (int)m;
        x = x + ps;                              // the reference's own addition
        k += 1;
    }
    return x;
}

struct GcNoEmit {
    GC_HDM void operator()(int, double, double, int) const {}
    GC_HDM void operator()(int, double, double, int, int) const {}
};

// phase remainder: prem = phi*DPI/CDIV; while (prem > DPI) prem -= DPI  (ref :666-668),
// the subtraction loop walked in pieces like the additions above
GC_HD double gc_carrier_prem(double phi)
{
    GC_FP_STRICT
    double p = GC_DDIV(phi * GC_NCO_DPI, GC_NCO_CDIV);
    if (!(p < 1.0e300)) return p;                // the reference would not return either
    while (p > GC_NCO_DPI) {
        double d;
        int64_t m = gc_nco_run(p, -GC_NCO_DPI, (int64_t)1 << 62, &d);
        if (m > 0 && d < 0.0) {
            // step i+1 is taken only while x_i > DPI: at most ceil((p - DPI)/|d|) steps
            const double est = ceil((p - GC_NCO_DPI) / -d);
            if (est < (double)m) m = (int64_t)est;
            while (m > 0 && !(fma((double)(m - 1), d, p) > GC_NCO_DPI)) m--;
            p = fma((double)m, d, p);
            if (m > 0) continue;
        }
        p = p - GC_NCO_DPI;
    }
    return p;
}

// ---------------------------------------------------------------------------
// code: ref src/sdrcmn.c:608-621
// ---------------------------------------------------------------------------
// coff -= smax*ci; coff -= floor(coff/len)*len  (ref :613-614)
GC_HD double gc_code_start(double coff, int smax, double ci, int len)
{
    GC_FP_STRICT
    double cs = coff - (double)smax * ci;
    cs = cs - floor(GC_DDIV(cs, (double)len)) * (double)len;
    return cs;
}

// Walks nt replica positions from c (= gc_code_start); emit(j0, y0, d, count, w) once
// per piece: positions j0 .. j0+count-1 hold y0 + i d (the value the reference
// truncates to a chip index, already wrapped), w = wraps so far.  Returns the
// value after nt additions (the reference returns that minus smax*ci, :620).
template <class Emit>
GC_HD double gc_code_walk(double c, double ci, int len, int nt, Emit &emit)
{
    GC_FP_STRICT
    const double dlen = (double)len;
    int j = 0, w = 0;
    while (j < nt) {
        if (c >= dlen) { c = c - dlen; w++; }    // ref :617
        double d;
        int64_t m = gc_nco_run(c, ci, (int64_t)(nt - 1 - j), &d);
        if (m > 0 && d > 0.0) {                  // every value of the piece stays below len
            const double est = floor((dlen - c) / d);
            if (est < (double)m) m = est > 0.0 ? (int64_t)est : 0;
            while (m > 0 && !(fma((double)m, d, c) < dlen)) m--;
        }
        emit(j, c, d, (int)m + 1, w);
        c = fma((double)m, d, c);
        j += (int)m;
        c = c + ci;                              // ref :619
        j += 1;
    }
    return c;
}

// remainder returned by rescode() (ref :620)
GC_HD double gc_code_rem(double cend, int smax, double ci)
{
    GC_FP_STRICT
    return cend - (double)smax * ci;
}

// ---------------------------------------------------------------------------
// segment tables handed to the kernels
// ---------------------------------------------------------------------------
// Carrier piece in fixed point: 2^64 = one LUT revolution (32 steps), i.e. 59
// fractional bits per step.  LUT index of sample k = (fx + (k - k0) dfx) >> 59.
// A value x with |x| >= 1 is a multiple of 2^-52, so the conversion is exact; the
// truncation toward zero of the reference's (int) cast (ref :654,661) is a floor
// for x >= 0 and a ceiling for x < 0, the latter by adding 2^59 - 1 before the
// shift.  |x| < 1 truncates to 0; |x| >= 2^31 leaves the range of int, where the
// reference's cast is undefined -- x86-64's cvttsd2si yields INT_MIN there, whose
// low five bits are 0: both cases are emitted as an all-zero piece.
#define GC_FX_BIAS ((1ULL << 59) - 1)
struct GcCarSeg { uint64_t fx, dfx; };

GC_HD GcCarSeg gc_carseg_make(double x, double d)
{
    GcCarSeg s;
    s.fx = 0;
    s.dfx = 0;
    const uint64_t ux = gc_d2u(x);
    const int e = (int)((ux >> 52) & 0x7FF) - 1023;
    if (e < 0 || e >= 31) return s;
    const uint64_t A = (ux & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;   // |x| = A 2^(e-52)
    const int sh = e + 7;                                                     // -> A 2^(e-52+59)
    uint64_t fx = sh < 64 ? A << sh : 0;
    // d = +-B 2^(e-52) with an integer B < 2^51 (gc_nco_run), or 0
    const uint64_t B = (uint64_t)ldexp(fabs(d), 52 - e);
    uint64_t dfx = sh < 64 ? B << sh : 0;
    if (d < 0.0) dfx = (uint64_t)0 - dfx;
    if (ux >> 63) fx = (uint64_t)0 - fx + GC_FX_BIAS;
    s.fx = fx;
    s.dfx = dfx;
    return s;
}

// Emitter that fills k0[] / seg[] (at most cap pieces; adjacent all-zero pieces are
// merged).  n pieces emitted so far; overflow is flagged, never written past cap.
struct GcCarTable {
    int *k0;
    GcCarSeg *seg;
    int cap, n, overflow;
    GC_HDM void operator()(int k, double x, double d, int /*count*/)
    {
        const GcCarSeg s = gc_carseg_make(x, d);
        if (n > 0 && s.fx == 0 && s.dfx == 0 && seg[n - 1].fx == 0 && seg[n - 1].dfx == 0) return;
        if (n >= cap) { overflow = 1; return; }
        k0[n] = k;
        seg[n] = s;
        n++;
    }
};

// Code piece: positions j0 .. j0+cnt-1 hold y0 + i d (chip = (int) of that), w wraps in front.
struct GcCodeSeg {
    double y0, d, inv, ylast;   // inv = 1/d (0 for a single position), ylast = value of the last position
    int j0, cnt, w, pad;
};

struct GcCodeTable {
    GcCodeSeg *seg;
    int cap, n, overflow;
    GC_HDM void operator()(int j, double y, double d, int count, int w)
    {
        GC_FP_STRICT
        // positions whose value truncates to the same chip as their predecessor's piece need no
        // piece of their own: |y| < 1 after a wrap (chip 0) is the common case -- merge those
        const double yl = fma((double)(count - 1), d, y);
        if (n > 0 && seg[n - 1].w == w && seg[n - 1].y0 > -1.0 && seg[n - 1].ylast < 1.0 && y > -1.0 && yl < 1.0) {
            seg[n - 1].cnt += count;
            seg[n - 1].ylast = yl;
            seg[n - 1].d = 0.0;                  // d = 0 with y0 in (-1, 1): every position is chip 0
            seg[n - 1].inv = 0.0;
            return;
        }
        if (n >= cap) { overflow = 1; return; }
        GcCodeSeg s;
        s.y0 = y;
        s.d = count > 1 ? d : 0.0;
        s.inv = (count > 1 && d != 0.0) ? 1.0 / d : 0.0;
        s.ylast = yl;
        s.j0 = j;
        s.cnt = count;
        s.w = w;
        s.pad = 0;
        seg[n++] = s;
    }
};

// chip index (before the modulo by the code length is needed: always < len) of replica
// position j, and the wrap count in front of it
GC_HD int gc_code_chip_at(const GcCodeSeg *seg, int nseg, int j, int *w, int *piece = nullptr)
{
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {                            // last piece with j0 <= j
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid].j0 <= j) lo = mid; else hi = mid - 1;
    }
    const GcCodeSeg &s = seg[lo];
    if (w) *w = s.w;
    if (piece) *piece = lo;
    int i = j - s.j0;
    if (i < 0) i = 0;
    if (i >= s.cnt) i = s.cnt - 1;
    if (s.d == 0.0) return (int)s.y0;            // one position, or a run of positions inside chip 0
    return (int)fma((double)i, s.d, s.y0);
}

// LUT index of sample k from a carrier table (binary search; the kernels' fast paths keep a
// running piece index instead)
GC_HD int gc_carrier_idx_at(const int *k0, const GcCarSeg *seg, int nseg, int k)
{
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (k0[mid] <= k) lo = mid; else hi = mid - 1;
    }
    return (int)((seg[lo].fx + (uint64_t)(int64_t)(k - k0[lo]) * seg[lo].dfx) >> 59);
}
