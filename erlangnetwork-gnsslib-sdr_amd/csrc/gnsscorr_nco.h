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
#define GC_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define GC_HD static inline
#define GC_HDM
#define GC_HD_NOINLINE static
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

// debug builds (-DGC_LOOP_DEBUG, tools/debug) drop progress marks into host-visible memory
#ifndef GC_DBG_MARK
#define GC_DBG_MARK(slot, value) do { } while (0)
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

// ---------------------------------------------------------------------------
// fast walkers
// ---------------------------------------------------------------------------
// The planner chains one walk per (channel, period) and sits on the critical path
// of a tracking batch; gc_nco_run() costs a division and a dozen bit operations per
// piece.  For the usual case -- a running sum that grows away from zero, i.e. sum
// and addend of one sign -- everything that depends only on the addend s and on the
// binade is tabulated once per frequency: the rounded step d_i = RN_u(s), its
// reciprocal and the tie flag for the binades 2^(es+2) .. 2^(es+2+GC_NB) (es =
// exponent of s).  A piece then costs: distance to the top of the binade (exact),
// quotient by reciprocal multiplication, one exact remainder to settle it.  The loops
// over the table are written with constant indices so that the table lives in
// registers.  Anything else -- the first few steps next to zero, a sum that shrinks
// toward zero, ties that start from an odd multiplier, binades beyond the table --
// falls through to one gc_nco_run() piece and the fast pass is tried again: both
// produce exact values, so mixing them is free of consequences.
#define GC_NB 20
struct GcNcoFast {
    double s;
    double inv_s;               // RN(1/|s|)
    int    ex0;                 // biased exponent of table entry 0
    unsigned tie;               // bit i: s lies exactly half way between two grid points of binade i
    double d[GC_NB];            // RN_u(s), signed
    double inv[GC_NB];          // 1/|d| (0 when d = 0)
};

// with_inv = false leaves inv[] zero (the period steps need only inv_s and, for the code, inv[itop])
GC_HD void gc_fast_init(GcNcoFast &f, double s, bool with_inv = true)
{
    GC_FP_STRICT
    const uint64_t us = gc_d2u(s);
    const int es = (int)((us >> 52) & 0x7FF);
    f.s = s;
    f.inv_s = 1.0 / fabs(s);
    f.tie = 0;
    f.ex0 = 0x7FFFFFF;          // no table: zero, subnormal, inf, nan or exponents out of range
    const bool ok = es > 60 && es < 0x7FF - GC_NB - 4;
    if (ok) f.ex0 = es + 2;
#pragma unroll
    for (int i = 0; i < GC_NB; i++) {
        f.d[i] = 0.0;
        f.inv[i] = 0.0;
        if (!ok) continue;
        const int ex = es + 2 + i;
        const int et = es + 1075 - ex;                  // exponent of t = s/u, < 1023 + 51 by construction
        double b = 0.0;
        if (et >= 1023 - 1) {
            const double t = gc_u2d((us & 0x800FFFFFFFFFFFFFull) | ((uint64_t)et << 52));
            b = rint(t);
            if (fabs(t - b) == 0.5) f.tie |= 1u << i;
        }
        f.d[i] = ldexp(b, ex - 1075);
        if (b != 0.0 && with_inv) f.inv[i] = 1.0 / fabs(f.d[i]);
    }
}

// one table piece: x in binade ex0 + i, growing.  lim_below: values must stay below it (code: the code
// length; carrier: +inf).  Returns the run length m <= cap with x + j d exact for j <= m.
GC_HD int gc_piece_len(double d, double inv, double x, int cap, double lim_below);
GC_HD int gc_fast_piece(const GcNcoFast &f, int i, double x, int cap, double lim_below)
{
    return gc_piece_len(f.d[i], f.inv[i], x, cap, lim_below);
}

// d, inv: the step RN_u(s) of x's binade and RN(1/|d|)
GC_HD int gc_piece_len(double d, double inv, double x, int cap, double lim_below)
{
    GC_FP_STRICT
    // (written without branches: on the device this runs on one lane of a wavefront, where a taken
    // branch costs more than the whole piece)
    const uint64_t ux = gc_d2u(x);
    const double top = fabs(gc_u2d(ux | 0x000FFFFFFFFFFFFFull));           // largest magnitude of the binade
    const double u = gc_u2d((uint64_t)(((ux >> 52) & 0x7FF) - 52) << 52);   // its grid
    const double lim = lim_below <= top ? lim_below - u : top;              // below lim_below, on the grid
    const double R = lim - fabs(x);                                         // exact (same binade)
    const double dabs = fabs(d);
    double q = floor(R * inv);
    const double r = fma(-q, dabs, R);                                      // exact
    q += r < 0.0 ? -1.0 : (r >= dabs ? 1.0 : 0.0);
    q = q < 0.0 ? 0.0 : q;
    const double dc = (double)cap;
    q = (q > dc || dabs == 0.0) ? dc : q;                                   // d = 0: the sum never moves
    q = R < 0.0 ? 0.0 : q;
    return (int)q;
}

template <class Emit>
GC_HD double gc_fast_carrier_walk(const GcNcoFast &f, double x, int n, Emit &emit)
{
    GC_FP_STRICT
    const double s = f.s;
    const bool table = f.ex0 != 0x7FFFFFF;
    int k = 0;
    while (k < n) {
        const int kin = k;
        // next to zero (fewer than ~4 steps per binade) every sample is its own piece: the reference's
        // own additions, in a loop of their own
        while (table && k < n && (x == 0.0 || (int)((gc_d2u(x) >> 52) & 0x7FF) < f.ex0)) {
            emit(k, x, 0.0, 1);
            x = x + s;
            k += 1;
        }
        if (k < n && x != 0.0 && (gc_d2u(x) >> 63) == (gc_d2u(s) >> 63)) {
            const int i0 = (int)((gc_d2u(x) >> 52) & 0x7FF) - f.ex0;
#pragma unroll
            for (int i = 0; i < GC_NB; i++) {
                if (i < i0) continue;
                const uint64_t ux = gc_d2u(x);
                if (!(k < n && (int)((ux >> 52) & 0x7FF) == f.ex0 + i)) break;
                if (!(((f.tie >> i) & 1) && (ux & 1))) {
                    const int m = gc_fast_piece(f, i, x, n - 1 - k, INFINITY);
                    emit(k, x, f.d[i], m + 1);
                    x = fma((double)m, f.d[i], x);
                    k += m;
                    x = x + s;
                    k += 1;
                }
            }
        }
        if (k != kin) continue;
        double d;
        const int64_t m = gc_nco_run(x, s, (int64_t)(n - 1 - k), &d);
        emit(k, x, d, (int)m + 1);
        x = fma((double)m, d, x);
        k += (int)m;
        x = x + s;
        k += 1;
    }
    return x;
}

template <class Emit>
GC_HD double gc_fast_code_walk(const GcNcoFast &f, double c, int len, int nt, Emit &emit)
{
    GC_FP_STRICT
    const double ci = f.s, dlen = (double)len;
    const bool table = f.ex0 != 0x7FFFFFF && ci > 0.0;
    int j = 0, w = 0;
    while (j < nt) {
        if (c >= dlen) { c = c - dlen; w++; }
        const int jin = j;
        while (table && j < nt && c < dlen && (c == 0.0 || (int)((gc_d2u(c) >> 52) & 0x7FF) < f.ex0)) {
            emit(j, c, 0.0, 1, w);
            c = c + ci;
            j += 1;
        }
        if (table && j < nt && c > 0.0 && c < dlen) {
            const int i0 = (int)((gc_d2u(c) >> 52) & 0x7FF) - f.ex0;
#pragma unroll
            for (int i = 0; i < GC_NB; i++) {
                if (i < i0) continue;
                const uint64_t uc = gc_d2u(c);
                if (!(j < nt && c < dlen && (int)((uc >> 52) & 0x7FF) == f.ex0 + i)) break;
                if (!(((f.tie >> i) & 1) && (uc & 1))) {
                    const int m = gc_fast_piece(f, i, c, nt - 1 - j, dlen);
                    emit(j, c, f.d[i], m + 1, w);
                    c = fma((double)m, f.d[i], c);
                    j += m;
                    c = c + ci;
                    j += 1;
                }
            }
        }
        if (j != jin) continue;
        double d;
        int64_t m = gc_nco_run(c, ci, (int64_t)(nt - 1 - j), &d);
        if (m > 0 && d > 0.0) {
            const double est = floor((dlen - c) / d);
            if (est < (double)m) m = est > 0.0 ? (int64_t)est : 0;
            while (m > 0 && !(fma((double)m, d, c) < dlen)) m--;
        }
        emit(j, c, d, (int)m + 1, w);
        c = fma((double)m, d, c);
        j += (int)m;
        c = c + ci;
        j += 1;
    }
    return c;
}

// prem (ref src/sdrcmn.c:666-668) with the subtraction loop run through a table for the addend -DPI
// (gc_fast_init(f, -GC_NCO_DPI)): a phase of thousands of radians (a 4 MHz IF over one code period)
// comes down one binade per piece; the last few subtractions are the reference's own.
GC_HD double gc_fast_prem(const GcNcoFast &f, double phi)
{
    GC_FP_STRICT
    double p = GC_DDIV(phi * GC_NCO_DPI, GC_NCO_CDIV);
    if (!(p < 1.0e300)) return p;
    // table binades start at 2^(ex0 - 1023) = 16 > DPI: inside them every value exceeds DPI, so the loop
    // condition holds for every step of a piece
    bool more = (int)((gc_d2u(p) >> 52) & 0x7FF) >= f.ex0 && p > 0.0;      // below 16: the plain loop at the end
    while (more) {
        more = false;
#pragma unroll
        for (int i = GC_NB - 1; i >= 0; i--) {
            const uint64_t up = gc_d2u(p);
            if ((int)((up >> 52) & 0x7FF) == f.ex0 + i && !(up >> 63) && !(((f.tie >> i) & 1) && (up & 1))) {
                // down to the multiplier 2^52 + 1 of this binade
                const double lo = gc_u2d((up & 0xFFF0000000000000ull) | 1ull);
                const double R = p - lo;
                const double dabs = fabs(f.d[i]);
                if (R >= dabs && dabs > 0.0) {
                    double q = floor(R * f.inv[i]);
                    const double r = fma(-q, dabs, R);
                    if (r < 0.0) q -= 1.0;
                    else if (r >= dabs) q += 1.0;
                    p = fma(q, f.d[i], p);
                }
                p = p - GC_NCO_DPI;              // (p > DPI here)
                more = true;
            }
        }
        if ((int)((gc_d2u(p) >> 52) & 0x7FF) >= f.ex0 + GC_NB) return gc_carrier_prem(phi);   // beyond the table
    }
    while (p > GC_NCO_DPI) p = p - GC_NCO_DPI;
    return p;
}

// x / b for many x and one b, y = RN(1/b) given: q = RN(x y), r = x - b q (exact by fma), RN(q + r y) is
// the correctly rounded quotient (Markstein) -- three dependent operations instead of the hardware's
// division sequence; the planner divides by 2 pi and by the chips-per-sample ratio once per period.
// (Checked against true division: 2.5e8 random operands for b = 2 pi and for chip-per-sample ratios,
// no mismatch; tests/test_nco_host.py repeats a sample of it.)
GC_HD double gc_div_y(double x, double b, double y)
{
    GC_FP_STRICT
    const double q = x * y;
    const double r = fma(-q, b, x);
    return fma(r, y, q);
}

// gc_code_start without its division while -len <= coff - smax*ci < len (the neighbours of -1, 0 and 1
// are more than an ulp away from the quotient there, so its floor is -1 or 0 by sign alone)
GC_HD double gc_code_start_fast(double coff, double smaxci, int len)
{
    GC_FP_STRICT
    const double dlen = (double)len;
    double cs = coff - smaxci;
    double fl = cs < 0.0 ? -1.0 : 0.0;
    if (!(cs >= -dlen && cs < dlen)) fl = floor(GC_DDIV(cs, dlen));
    return cs - fl * dlen;
}

// ---------------------------------------------------------------------------
// certified crossings: the planner's form of the walk
// ---------------------------------------------------------------------------
// On the device the planner's chain runs on one wavefront per channel and pays ~10 clocks per
// instruction whatever it does, so what has to be sequential is cut to two additions per binade:
//
//   * WHERE the running sum crosses each binade boundary b (the first index k with |x_k| >= b) does
//     not need the chain.  The exact-arithmetic sum x0 + k s differs from the reference's running
//     sum by the accumulated rounding, at most E(b) = sum over the binades below b of (steps in the
//     binade) * ulp/2 <= b (b/|s| + 4) 2^-53.  If x0 + (k-1) s and x0 + k s lie on either side of b
//     by more than E(b) plus the evaluation error, k is the crossing of the reference's sum too --
//     for all boundaries at once, one per lane (gc_cert_crossing).  Sums that involve no rounding at
//     all (a step that is a multiple of the coarsest grid: the nominal chip rate 1/16) are compared
//     exactly instead.
//   * WHAT the sum is there does need the chain, but with the crossings known it is one fma (the
//     steps inside the binade, all equal to RN_u(s)) and one addition (the reference's own, across
//     the boundary) per binade: gc_cert_chain.
//
// A boundary that cannot be certified (the sum passes within ~1e-9 of it: about once in 1e7
// periods), a tie binade entered on an odd multiplier, or a walk that is not of the usual shape
// makes the caller fall back to the piece walkers above; every path yields the reference's values.
#define GC_CERT_FAIL (-1)
#define GC_CERT_FAR  0x3fffffff     // certified: not reached within n steps

// first k in [1, n] with a0 + k*sabs >= b for the reference's rounded running sum (a0 < b), given
// mg = 0 when no addition of the walk rounds (then the products below are exact) and the bound of the
// accumulated rounding otherwise.  inv = RN(1/sabs).
GC_HD int gc_cert_crossing(double a0, double sabs, double inv, double b, double mg, int n)
{
    GC_FP_STRICT
    double k = ceil((b - a0) * inv);
    k = k < 1.0 ? 1.0 : k;
    const double dn = (double)n;
    if (k > dn + 1.0) k = dn + 1.0;
    double lo = fma(k - 1.0, sabs, a0), hi = fma(k, sabs, a0);
    // the quotient estimate may be one off
    const bool down = lo >= b, up = hi < b;
    k = down ? k - 1.0 : (up ? k + 1.0 : k);
    const double lo2 = down ? fma(k - 1.0, sabs, a0) : (up ? hi : lo);
    const double hi2 = down ? lo : (up ? fma(k, sabs, a0) : hi);
    lo = lo2;
    hi = hi2;
    if (k > dn) {           // not within n steps, if the last sum stays clear of b
        const double last = fma(dn, sabs, a0);
        return last < b - mg ? GC_CERT_FAR : GC_CERT_FAIL;
    }
    if (k < 1.0) return GC_CERT_FAIL;
    const bool ok = mg == 0.0 ? (lo < b && hi >= b) : (lo < b - mg && hi >= b + mg);
    return ok ? (int)k : GC_CERT_FAIL;
}

// rounding accumulated by the reference's sum below b (generous), plus the error of evaluating
// a0 + k*sabs once
GC_HD double gc_cert_margin(double b, double inv)
{
    GC_FP_STRICT
    return b * (b * inv + 8.0) * 1.1102230246251565e-16;       // 2^-53
}

// no addition of a walk from x0 with step s up to magnitude `top` rounds: x0 and s are multiples of
// ulp(top)
GC_HD bool gc_cert_exact(double x0, double s, double top)
{
    const int et = (int)((gc_d2u(top) >> 52) & 0x7FF);
    if (et < 64) return false;
    const double sc = gc_u2d((uint64_t)(2046 - (et - 52)) << 52);           // 1/ulp(top)
    const double a = x0 * sc, b = s * sc;
    return a == rint(a) && b == rint(b) && fabs(a) < 9.0e15 && fabs(b) < 9.0e15;
}

// The chain: from x (sample index k, |x| inside table binade i0) over the table binades to index n.
// K[i] (i > i0) = certified index of the first sample in binade i or above (relative to index 0 of
// the same x0 the crossings were computed from); Klim/blim: the same for the walk's upper limit
// (code: the code length, carrier: none = GC_CERT_FAR), which lies in binade ilim.
// On return *kout = index reached: n (then the value is x_n), or the index of the first sample at
// or above the limit (then the value is that sample's).  Returns false when a tie binade was
// entered on an odd multiplier (caller falls back); values are exact otherwise.
GC_HD bool gc_cert_chain(const GcNcoFast &f, double *px, int *pk, int n, int i0, const int *K, int ilim, int Klim)
{
    GC_FP_STRICT
    double x = *px;
    int k = *pk;
    const double s = f.s;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < GC_NB; i++) {
        if (i < i0 || i > ilim || k >= n) continue;
        // last index inside this binade: before the next boundary, before the limit, before n
        int last = i + 1 < GC_NB ? K[i + 1] - 1 : n - 1;
        if (i == ilim) last = Klim - 1;
        if (last > n - 1) last = n - 1;
        int m = last - k;
        if (m < 0) { ok = false; continue; }
        if (((f.tie >> i) & 1) && (gc_d2u(x) & 1) && m > 0) {
            // a tie binade entered on an odd multiplier: this one step rounds the other way (to the even
            // neighbour); from there on the step is f.d[i]
            x = x + s;
            m -= 1;
        }
        x = fma((double)m, f.d[i], x);
        k = last;
        x = x + s;
        k += 1;
    }
    *px = x;
    *pk = k;
    return ok;
}

// Crossings of one growing stretch, computed "per lane": boundary index i in (i0, GC_NB) is the bottom
// of table binade i, boundary GC_NB is the walk's limit (the code length; carrier: none).  On the host
// this is a loop, on the device lane i computes K[i] (gnsscorr_trk.hip) -- same function per boundary.
struct GcCertCtx {
    double a0, sabs, inv;   // |x0|, |s|, RN(1/|s|)
    int    n;               // steps available
    bool   exact;           // no addition of the stretch rounds
    int    ex0;
};

GC_HD int gc_cert_lane(const GcCertCtx &c, int i, double lim)
{
    // boundary value: bottom of table binade i, or the limit for i == GC_NB
    const double b = i < GC_NB ? gc_u2d((uint64_t)(c.ex0 + i) << 52) : lim;
    if (!(b > c.a0)) return 0;
    if (!(b < 1.0e300)) return GC_CERT_FAR;
    return gc_cert_crossing(c.a0, c.sabs, c.inv, b, c.exact ? 0.0 : gc_cert_margin(b, c.inv), c.n);
}

// One growing stretch through the table by certified crossings: x at index k (same sign as s, inside
// the table, below lim) -> index n or the first sample at/above lim.  gc_cert_setup checks the shape
// and prepares the per-boundary context; then one gc_cert_lane per boundary (host: a loop, device: one
// lane each); then gc_cert_chain.  false: not done (caller falls back to the piece walkers).
GC_HD bool gc_cert_setup(const GcNcoFast &f, double x, int k, int n, double lim, GcCertCtx *c, int *pi0, int *pilim)
{
    GC_FP_STRICT
    const int ex = (int)((gc_d2u(x) >> 52) & 0x7FF);
    const int i0 = ex - f.ex0;
    if (f.ex0 == 0x7FFFFFF || i0 < 0 || i0 >= GC_NB || !(fabs(x) < lim)) return false;
    if ((gc_d2u(x) >> 63) != (gc_d2u(f.s) >> 63)) return false;
    c->a0 = fabs(x);
    c->sabs = fabs(f.s);
    c->inv = f.inv_s;
    c->n = n - k;
    c->ex0 = f.ex0;
    int ilim = GC_NB - 1;
    double reach = fma((double)c->n, c->sabs, c->a0);
    if (lim < 1.0e300) {
        // binade of the largest value below the limit
        const double below = gc_u2d(gc_d2u(lim) - 1);
        ilim = (int)((gc_d2u(below) >> 52) & 0x7FF) - f.ex0;
        if (ilim < i0 || ilim >= GC_NB) return false;
        // the sample that reaches the limit must still be inside binade ilim (lim + |s| below its top)
        if (!(lim + c->sabs < gc_u2d((uint64_t)(f.ex0 + ilim + 1) << 52))) return false;
        if (reach > lim + c->sabs) reach = lim + c->sabs;
    } else if (!(reach < gc_u2d((uint64_t)(f.ex0 + GC_NB) << 52))) {
        return false;                       // runs off the table
    }
    c->exact = gc_cert_exact(x, f.s, reach);
    *pi0 = i0;
    *pilim = ilim;
    return true;
}

GC_HD bool gc_cert_stretch(const GcNcoFast &f, double *px, int *pk, int n, double lim, int *K)
{
    GcCertCtx c;
    int i0, ilim;
    if (!gc_cert_setup(f, *px, *pk, n, lim, &c, &i0, &ilim)) return false;
    for (int i = i0 + 1; i <= GC_NB; i++) {     // (device: one lane each, gnsscorr_trk.hip)
        K[i] = (i <= ilim || i == GC_NB) ? gc_cert_lane(c, i, lim) : GC_CERT_FAR;
        if (K[i] == GC_CERT_FAIL) return false;
    }
    double y = *px;
    int k = 0;
    if (!gc_cert_chain(f, &y, &k, c.n, i0, K, ilim, K[GC_NB])) return false;
    *px = y;
    *pk += k;
    return true;
}

// The planner's code walk (end value only): literal steps next to zero, certified stretches through the
// table, wraps in between; false when a stretch could not be certified (caller: gc_fast_code_walk).
GC_HD bool gc_plan_code_walk(const GcNcoFast &f, double c, int len, int nt, int *K, double *cend)
{
    GC_FP_STRICT
    const double ci = f.s, dlen = (double)len;
    if (f.ex0 == 0x7FFFFFF || !(ci > 0.0)) return false;
    int j = 0;
    while (j < nt) {
        if (c >= dlen) c = c - dlen;
        while (j < nt && c < dlen && (c == 0.0 || (int)((gc_d2u(c) >> 52) & 0x7FF) < f.ex0)) {
            c = c + ci;
            j += 1;
        }
        if (j >= nt) break;
        if (c >= dlen) continue;
        if (!(c > 0.0) || !gc_cert_stretch(f, &c, &j, nt, dlen, K)) return false;
    }
    *cend = c;
    return true;
}

// A sum far above its addend (a negative carrier phase is never wrapped, ref src/sdrcmn.c:667: after a
// fraction of a second it is thousands of LUT steps while the step stays ~0.01) spends the whole period
// inside one binade: n equal steps d = RN_u(s), if the last one stays below the top of the binade.
GC_HD bool gc_one_binade_walk(double x, double s, int n, double *xn, double *dout = nullptr, bool *tie_out = nullptr)
{
    GC_FP_STRICT
    const uint64_t ux = gc_d2u(x), us = gc_d2u(s);
    if (tie_out) *tie_out = false;
    const int ex = (int)((ux >> 52) & 0x7FF), es = (int)((us >> 52) & 0x7FF);
    if (ex == 0 || ex == 0x7FF || es == 0 || es == 0x7FF || (ux >> 63) != (us >> 63)) return false;
    const int et = es + 1075 - ex;
    if (et >= 1023 + 51) return false;
    double b = 0.0;
    if (et >= 1023 - 1) {
        const double t = gc_u2d((us & 0x800FFFFFFFFFFFFFull) | ((uint64_t)et << 52));
        b = rint(t);
        if (tie_out) *tie_out = fabs(t - b) == 0.5;             // (the step then depends on the multiplier's parity)
        if (fabs(t - b) == 0.5 && (ux & 1)) return false;      // tie from an odd multiplier
    }
    const double d = ldexp(b, ex - 1075);
    const double y = fma((double)n, d, x);
    const double top = fabs(gc_u2d(ux | 0x000FFFFFFFFFFFFFull));
    if (!(fabs(y) <= top)) return false;
    *xn = y;
    if (dout) *dout = d;
    return true;
}

// The planner's carrier walk (value after n additions); false: caller uses gc_fast_carrier_walk.
GC_HD bool gc_plan_carrier_walk(const GcNcoFast &f, double x, int n, int *K, double *xn)
{
    GC_FP_STRICT
    if (f.ex0 == 0x7FFFFFF) return false;
    if ((int)((gc_d2u(x) >> 52) & 0x7FF) >= f.ex0 + GC_NB) return gc_one_binade_walk(x, f.s, n, xn);
    int k = 0;
    while (k < n && (x == 0.0 || (int)((gc_d2u(x) >> 52) & 0x7FF) < f.ex0)) {
        x = x + f.s;
        k += 1;
    }
    if (k < n && !gc_cert_stretch(f, &x, &k, n, INFINITY, K)) return false;
    if (k != n) return false;
    *xn = x;
    return true;
}

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------
// device planner: one wavefront per channel, lane i = binade boundary i
// ---------------------------------------------------------------------------
// device form of gc_cert_stretch (gnsscorr_nco.h): the crossings one per lane.  Ks: LDS, GC_NB + 2 ints.
__device__ __forceinline__ bool cert_stretch_dev(const GcNcoFast &f, double *px, int *pk, int n, double lim, int *Ks,
                                                 int lane)
{
    GcCertCtx c;
    int i0, ilim;
    if (!gc_cert_setup(f, *px, *pk, n, lim, &c, &i0, &ilim)) return false;        // (wave-uniform)
    int Kl = GC_CERT_FAR;
    if (lane > i0 && lane <= GC_NB && (lane <= ilim || lane == GC_NB)) Kl = gc_cert_lane(c, lane, lim);
    if (__any(Kl == GC_CERT_FAIL)) return false;
    if (lane <= GC_NB) Ks[lane] = Kl;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                         // (one wavefront: LDS order suffices)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double y = *px;
    int k = 0;
    const bool ok = gc_cert_chain(f, &y, &k, c.n, i0, Ks, ilim, Ks[GC_NB]);
    __builtin_amdgcn_wave_barrier();                                               // before Ks is rewritten
    if (!ok) return false;
    *px = y;
    *pk += k;
    return true;
}

// gc_plan_code_walk / gc_plan_carrier_walk with the lanes at work
__device__ __forceinline__ bool plan_code_dev(const GcNcoFast &f, double c, int len, int nt, int *Ks, int lane, double *cend)
{
    const double ci = f.s, dlen = (double)len;
    if (f.ex0 == 0x7FFFFFF || !(ci > 0.0)) return false;
    int j = 0;
    while (j < nt) {
        if (c >= dlen) c = __dsub_rn(c, dlen);
        while (j < nt && c < dlen && (c == 0.0 || (int)((gc_d2u(c) >> 52) & 0x7FF) < f.ex0)) {
            c = __dadd_rn(c, ci);
            j += 1;
        }
        if (j >= nt) break;
        if (c >= dlen) continue;
        if (!(c > 0.0) || !cert_stretch_dev(f, &c, &j, nt, dlen, Ks, lane)) return false;
    }
    *cend = c;
    return true;
}

__device__ __forceinline__ bool plan_carrier_dev(const GcNcoFast &f, double x, int n, int *Ks, int lane, double *xn)
{
    if (f.ex0 == 0x7FFFFFF) return false;
    if ((int)((gc_d2u(x) >> 52) & 0x7FF) >= f.ex0 + GC_NB) return gc_one_binade_walk(x, f.s, n, xn);
    int k = 0;
    while (k < n && (x == 0.0 || (int)((gc_d2u(x) >> 52) & 0x7FF) < f.ex0)) {
        x = __dadd_rn(x, f.s);
        k += 1;
    }
    if (k < n && !cert_stretch_dev(f, &x, &k, n, INFINITY, Ks, lane)) return false;
    if (k != n) return false;
    *xn = x;
    return true;
}

#endif

// ---------------------------------------------------------------------------
// the planner's period step, specialised to the shape a tracked channel has
// ---------------------------------------------------------------------------
// Code NCO of one period of a tracked channel (ref src/sdrcmn.c:613-620 as driven by
// src/sdrtrk.c:31-43): the replica starts 2*smax samples before the end of the previous code period,
// i.e. in the binade that holds the code length ("head": a few equal steps up to the wrap), wraps,
// climbs through every binade from ~4 ci to the code length again ("climb"), wraps a second time and
// ends after another ~2*smax samples next to zero ("tail").  gc_code_period() walks exactly that:
//   head   one exact quotient (the values are c0 + j d_top),
//   climb  crossings of the binade boundaries certified one per lane (`fill`), then a chain of one fma
//          and one addition per binade with nothing else on its dependency path,
//   tail   the reference's own additions.
// Anything that does not fit (period not ending in the tail, first sample outside the top binade, a
// crossing that cannot be certified, ...) returns false and the caller takes the general walkers.
struct GcCodePlan {
    GcNcoFast f;            // table for the addend ci
    double dlen;            // code length
    double limtop;          // largest value below dlen on its binade's grid
    double smaxci;          // smax*ci (ref :613)
    int    itop;            // table binade that holds limtop
    int    it;              // the table binade (<= itop) in which ci is a tie, or -1
    bool   exact;           // ci is a multiple of ulp(dlen): no addition of the climb rounds
    bool   ok;              // the table covers the code (else: general walkers only)
};

GC_HD void gc_code_plan_init(GcCodePlan &P, double ci, int len, int smax, bool with_inv = true)
{
    GC_FP_STRICT
    gc_fast_init(P.f, ci, with_inv);
    P.dlen = (double)len;
    P.smaxci = (double)smax * ci;
    P.limtop = gc_u2d(gc_d2u(P.dlen) - 1);
    P.itop = (int)((gc_d2u(P.limtop) >> 52) & 0x7FF) - P.f.ex0;
    P.ok = P.f.ex0 != 0x7FFFFFF && ci > 0.0 && P.itop >= 1 && P.itop < GC_NB &&
           P.dlen + ci < gc_u2d((uint64_t)(P.f.ex0 + P.itop + 1) << 52) && !((P.f.tie >> P.itop) & 1);
    P.exact = P.ok && gc_cert_exact(P.limtop, ci, P.dlen);
    P.it = -1;
#pragma unroll
    for (int i = 0; i < GC_NB; i++)
        if (((P.f.tie >> i) & 1) && i <= P.itop) P.it = i;
    if (P.ok && !with_inv) {
#pragma unroll
        for (int i = 0; i < GC_NB; i++)
            if (i == P.itop && P.f.d[i] != 0.0) P.f.inv[i] = 1.0 / fabs(P.f.d[i]);
    }
}

// fill(K, ctx, i0, itop, lim): K[i] = certified crossing of boundary i (i0 < i <= itop: bottom of table
// binade i; i == GC_NB: lim) relative to ctx.a0 -- gc_cert_lane per boundary; returns false on any
// GC_CERT_FAIL.  (host: GcFillLoop below; device: one lane per boundary + readlane)
// The crossings handed back are strictly increasing over i0 < i <= itop, and K[GC_NB] above K[itop] (a
// fill that finds them otherwise returns false).
struct GcFillLoop {
    GC_HDM bool operator()(int *K, const GcCertCtx &c, int i0, int itop, double lim) const
    {
        int prev = 0;
        for (int i = i0 + 1; i <= GC_NB; i++) {
            K[i] = (i <= itop || i == GC_NB) ? gc_cert_lane(c, i, lim) : GC_CERT_FAR;
            if (K[i] == GC_CERT_FAIL) return false;
            if (i <= itop || i == GC_NB) {
                if (K[i] <= prev && !(K[i] == GC_CERT_FAR && (i != GC_NB || !(lim < 1.0e300)))) return false;    // (several may be out of reach)
                prev = K[i];
            }
        }
        return true;
    }
};

// The climb's chain for a table whose binade ITOP holds the code length: written for a compile-time ITOP
// so that nothing but one fma and one addition per binade sits on the dependency path (the loop bounds,
// the table entries and the tie test are constants or scalar work beside it).
template <int ITOP, class Emit>
GC_HD bool gc_code_climb(const GcNcoFast &f, const int *K, int i0, int cn, double ci, double *py, int *pk, int jbase, Emit &emit)
{
    GC_FP_STRICT
    double y = *py;
    int k = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i <= ITOP; i++) {
        const bool active = i >= i0;                // (i0 is 0 or 1)
        const int Kn = i == ITOP ? K[GC_NB] : K[i + 1];
        int last = Kn - 1;
        last = last > cn - 1 ? cn - 1 : last;
        int m = last - k;
        ok = ok && (m >= 0 || !active);
        int kb = k;
        if ((f.tie >> i) & 1) {                     // (at most one table binade: a scalar branch)
            const bool odd = active && (gc_d2u(y) & 1) && m > 0;
            if (odd) emit(jbase + kb, y, 0.0, 1, 1);
            const double yl = y + ci;
            y = odd ? yl : y;
            m -= odd ? 1 : 0;
            kb += odd ? 1 : 0;
        }
        m = active ? m : 0;
        if (active && m >= 0) emit(jbase + kb, y, f.d[i], m + 1, 1);
        const double yn = fma((double)m, f.d[i], y) + ci;
        y = active ? yn : y;
        k = active ? last + 1 : k;
    }
    *py = y;
    *pk = k;
    return ok;
}

// The same with the second wrap inside the period (always, for a tracked channel) and the crossings known
// to increase: no test is left beside the two operations, except in the one binade (if any: `it`, else -1)
// in which the addend is a tie -- entered on an odd multiplier, its first step is the reference's own
// addition (it rounds to the even neighbour), the rest are the table's.
template <int ITOP, class Emit>
GC_HD void gc_code_climb_lean(const GcNcoFast &f, const int *K, int i0, int it, double ci, double *py, int jbase, Emit &emit)
{
    GC_FP_STRICT
    double y = *py;
#pragma unroll
    for (int i = 0; i <= ITOP; i++) {
        if (i == 0 && i0 != 0) continue;
        int ks = (i == i0) ? 0 : K[i];
        const int ke = i == ITOP ? K[GC_NB] : K[i + 1];
        if (i == it) {
            if ((gc_d2u(y) & 1) && ke - 1 - ks > 0) {
                emit(jbase + ks, y, 0.0, 1, 1);
                y = y + ci;
                ks += 1;
            }
        }
        emit(jbase + ks, y, f.d[i], ke - ks, 1);
        y = fma((double)(ke - 1 - ks), f.d[i], y) + ci;
    }
    *py = y;
}

// emit(j0, y0, d, count, w) receives the pieces (as gc_code_walk's emitter does) when the step applies;
// on a false return the emitter may have seen some pieces already: reset it before the fallback.
template <int ITOP, class Fill, class Emit>
GC_HD bool gc_code_period_body(const GcCodePlan &P, double remcode, int nt, Fill &fill, double *remcode_out, Emit &emit)
{
    GC_FP_STRICT
    const GcNcoFast &f = P.f;
    const double ci = f.s, dlen = P.dlen;
    // ---- start value (ref :613-614) and head: c0 + j d_top < len for j <= q
    double cs = remcode - P.smaxci;
    const double fl = cs < 0.0 ? -1.0 : 0.0;
    if (!(cs >= -dlen && cs < dlen)) return false;
    const double c0 = cs - fl * dlen;
    if ((int)((gc_d2u(c0) >> 52) & 0x7FF) != f.ex0 + ITOP || !(c0 < dlen)) return false;
    const double dtop = f.d[ITOP];
    double y;
    int j;
    {
        const double R = P.limtop - c0;
        double q = floor(R * f.inv[ITOP]);
        const double r = fma(-q, dtop, R);
        q += r < 0.0 ? -1.0 : (r >= dtop ? 1.0 : 0.0);
        if (!(q >= 0.0 && q < (double)(nt - 2))) return false;
        y = fma(q + 1.0, dtop, c0) - dlen;          // first sample at or above len, wrapped (exact)
        j = (int)q + 1;
        emit(0, c0, dtop, j, 0);
    }
    // ---- next to zero: the reference's own additions up to the table
    const double b0 = gc_u2d((uint64_t)f.ex0 << 52);
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const bool lit = y < b0;
        if (lit) emit(j, y, 0.0, 1, 1);
        const double yl = y + ci;
        y = lit ? yl : y;
        j += lit ? 1 : 0;
    }
    if (!(y >= b0) || j >= nt - 2) return false;
    // ---- climb: crossings (one boundary per lane), then the chain
    GcCertCtx c;
    c.a0 = y;
    c.sabs = ci;
    c.inv = f.inv_s;
    c.n = nt - j;
    c.exact = P.exact;
    c.ex0 = f.ex0;
    const int i0 = (int)((gc_d2u(y) >> 52) & 0x7FF) - f.ex0;
    if (i0 < 0 || i0 > 1) return false;
    int K[GC_NB + 1];
    if (!fill(K, c, i0, ITOP, dlen)) return false;
    if (K[GC_NB] >= c.n) return false;              // (the period must end in the tail)
    gc_code_climb_lean<ITOP>(f, K, i0, P.it, ci, &y, j, emit);
    j += K[GC_NB];
    if (j >= nt || !(y >= dlen)) return false;
    // ---- second wrap and tail
    y = y - dlen;
    int t = nt - j;
    for (; t >= 4; t -= 4) {                        // (the reference's own additions, four per trip)
        emit(nt - t, y, 0.0, 1, 2);
        y = y + ci;
        emit(nt - t + 1, y, 0.0, 1, 2);
        y = y + ci;
        emit(nt - t + 2, y, 0.0, 1, 2);
        y = y + ci;
        emit(nt - t + 3, y, 0.0, 1, 2);
        y = y + ci;
    }
    for (; t > 0; t--) {
        emit(nt - t, y, 0.0, 1, 2);
        y = y + ci;
    }
    *remcode_out = y - P.smaxci;
    return true;
}

// (out of line: inlined six times into the closed-loop kernel the step made that kernel hang -- a
// compiler-sensitive failure)
#ifdef GC_CODE_PERIOD_INLINE        // (tools/debug: the r2 build in which the closed-loop kernel stalled)
#define GC_PERIOD_ATTR GC_HD
#else
#define GC_PERIOD_ATTR GC_HD_NOINLINE
#endif
template <int ITOP, class Fill, class Emit>
GC_PERIOD_ATTR bool gc_code_period_t(const GcCodePlan &P, double remcode, int nt, Fill &fill, double *remcode_out, Emit &emit)
{
    return gc_code_period_body<ITOP>(P, remcode, nt, fill, remcode_out, emit);
}

template <class Fill, class Emit>
GC_HD bool gc_code_period(const GcCodePlan &P, double remcode, int nt, Fill &fill, double *remcode_out, Emit &emit)
{
    if (!P.ok) return false;
    switch (P.itop) {           // (GPS / GLONASS codes at 2..64 samples per chip: 7..12)
    case 7:  return gc_code_period_t<7>(P, remcode, nt, fill, remcode_out, emit);
    case 8:  return gc_code_period_t<8>(P, remcode, nt, fill, remcode_out, emit);
    case 9:  return gc_code_period_t<9>(P, remcode, nt, fill, remcode_out, emit);
    case 10: return gc_code_period_t<10>(P, remcode, nt, fill, remcode_out, emit);
    case 11: return gc_code_period_t<11>(P, remcode, nt, fill, remcode_out, emit);
    case 12: return gc_code_period_t<12>(P, remcode, nt, fill, remcode_out, emit);
    default: return false;
    }
}

template <class Fill>
GC_HD bool gc_code_period(const GcCodePlan &P, double remcode, int nt, Fill &fill, double *remcode_out)
{
    GcNoEmit ne;
    return gc_code_period(P, remcode, nt, fill, remcode_out, ne);
}

// Carrier NCO of one period (ref src/sdrcmn.c:649-668): phase remainder in, phase remainder out.
struct GcCarPlan {
    GcNcoFast f;            // table for the addend ps
    GcNcoFast fprem;        // table for -DPI (the remainder loop of a phase of thousands of radians)
    double ydpi;            // RN(1/DPI)
};

GC_HD void gc_car_plan_init(GcCarPlan &P, double ps, bool with_inv = true, bool with_prem = true)
{
    GC_FP_STRICT
    gc_fast_init(P.f, ps, with_inv);
    if (with_prem) gc_fast_init(P.fprem, -GC_NCO_DPI);
    P.ydpi = 1.0 / GC_NCO_DPI;
}

template <class Fill, class Emit>
GC_HD bool gc_carrier_period(const GcCarPlan &P, double remcarr, int n, Fill &fill, double *remcarr_out, Emit &emit)
{
    GC_FP_STRICT
    const GcNcoFast &f = P.f;
    if (f.ex0 == 0x7FFFFFF || n < 1) return false;
    const double s = f.s;
    double x = gc_div_y(remcarr * GC_NCO_CDIV, GC_NCO_DPI, P.ydpi);      // ref :649
    int k = 0;
    bool done = false;
    if ((int)((gc_d2u(x) >> 52) & 0x7FF) >= f.ex0 + GC_NB) {
        const double x0 = x;
        double d1;
        if (!gc_one_binade_walk(x, s, n, &x, &d1)) return false;
        emit(0, x0, d1, n);
        done = true;
    }
    if (!done) {
        // next to zero: the reference's own additions (a channel fresh out of acquisition starts at phase 0)
        for (int t = 0; t < 8 && k < n && (x == 0.0 || (int)((gc_d2u(x) >> 52) & 0x7FF) < f.ex0); t++) {
            emit(k, x, 0.0, 1);
            x = x + s;
            k += 1;
        }
        if (k < n) {
            const int i0 = (int)((gc_d2u(x) >> 52) & 0x7FF) - f.ex0;
            if (i0 < 0 || i0 >= GC_NB || (gc_d2u(x) >> 63) != (gc_d2u(s) >> 63)) return false;
            GcCertCtx c;
            c.a0 = fabs(x);
            c.sabs = fabs(s);
            c.inv = f.inv_s;
            c.n = n - k;
            c.exact = false;
            c.ex0 = f.ex0;
            if (!(fma((double)c.n, c.sabs, c.a0) < gc_u2d((uint64_t)(f.ex0 + GC_NB) << 52))) return false;
            int K[GC_NB + 1];
            if (!fill(K, c, i0, GC_NB - 1, INFINITY)) return false;
            int kk = 0;
            bool ok = true;
#pragma unroll
            for (int i = 0; i < GC_NB; i++) {
                if (i < i0 || kk >= c.n) continue;
                int last = (i + 1 < GC_NB ? K[i + 1] : GC_CERT_FAR) - 1;
                last = last > c.n - 1 ? c.n - 1 : last;
                int m = last - kk;
                ok = ok && m >= 0;
                int kb = kk;
                if ((f.tie >> i) & 1) {
                    const bool odd = (gc_d2u(x) & 1) && m > 0;
                    if (odd) emit(k + kb, x, 0.0, 1);
                    const double xl = x + s;
                    x = odd ? xl : x;
                    m -= odd ? 1 : 0;
                    kb += odd ? 1 : 0;
                }
                if (m >= 0) emit(k + kb, x, f.d[i], m + 1);
                x = fma((double)m, f.d[i], x);
                x = x + s;
                kk = last + 1;
            }
            if (!ok || kk != c.n) return false;
        }
    }
    // phase remainder (ref :666-668)
    double p = x * GC_NCO_DPI * (1.0 / GC_NCO_CDIV);        // (/32: an exact scaling)
    if (!(p < 1.0e300)) { *remcarr_out = p; return true; }
    if ((int)((gc_d2u(p) >> 52) & 0x7FF) >= P.fprem.ex0 && p > 0.0) p = gc_fast_prem(P.fprem, x);
    else while (p > GC_NCO_DPI) p = p - GC_NCO_DPI;
    *remcarr_out = p;
    return true;
}

template <class Fill>
GC_HD bool gc_carrier_period(const GcCarPlan &P, double remcarr, int n, Fill &fill, double *remcarr_out)
{
    GcNoEmit ne;
    return gc_carrier_period(P, remcarr, n, fill, remcarr_out, ne);
}

// Closed-form start of period e of a batch whose frequencies are held (what the speculation pass works
// from; never used for a result).  In real numbers the code phase at sample N is remcode0 + N ci, period e
// starts at sample N_e = floor((e len - remcode0)/spc), and the carrier phase there is remcarr0 + N_e w
// brought into (0, DPI] by whole subtractions of DPI when it exceeds DPI (ref src/sdrcmn.c:666-668; a
// falling phase is never wrapped).  The reference's running sums differ from these by their accumulated
// rounding (~1e-9 after a thousand periods): enough to find the binade crossings of almost every period,
// and the chain checks every one of them.
GC_HD void gc_spec_start(double remcode0, double remcarr0, double ci, double spc, double ps, double dlen, int e,
                         double *remcode_e, double *remcarr_e, int *n_e)
{
    GC_FP_STRICT
    const double de = (double)e;
    const double N0 = e == 0 ? 0.0 : floor((de * dlen - remcode0) / spc);
    const double N1 = floor(((de + 1.0) * dlen - remcode0) / spc);
    const double dn = N1 - N0;
    *n_e = (dn > 0.0 && dn < 2147483648.0) ? (int)dn : 0;
    const double p = N0 * ci, pe = fma(N0, ci, -p);             // product and its rounding error
    *remcode_e = (remcode0 + (p - de * dlen)) + pe;
    const double w = ps * GC_NCO_DPI * (1.0 / GC_NCO_CDIV);
    const double q = N0 * w, qe = fma(N0, w, -q);
    double phi = remcarr0 + q;
    if (phi > GC_NCO_DPI && phi < 1.0e300) {
        const double M = ceil(phi / GC_NCO_DPI) - 1.0;
        phi = fma(-M, GC_NCO_DPI, phi);
    }
    *remcarr_e = phi + qe;
}

// ---------------------------------------------------------------------------
// period steps on claims: the planner's batch form
// ---------------------------------------------------------------------------
// On the device the chain of a channel runs on one wavefront and pays a few clocks per INSTRUCTION whatever
// it does (and ~25 per taken branch), so the batch planner splits a period step in two:
//
//   discover  (one lane per period, all periods of a batch side by side: trk_spec_kernel) works from the
//             closed-form period start (gc_spec_start) and finds the STRUCTURE of the step: how many equal
//             steps the head takes, how many of the reference's own additions follow next to zero, how many
//             equal steps each binade takes, how many subtractions the phase remainder needs -- integers,
//             the "claims" of the period;
//   evaluate  (the sequential chain: trk_plan4_kernel, and the closed loop's tail) runs the step from the exact
//             period start as one fixed, branch-free sequence of positions -- literal additions whose addend is
//             the step or zero, binade segments y -> fma(dm, d, y + pre) + step -- with the claims as its data.
//             The same function checks every claim against the values it produces: a segment's last value below
//             the top of its binade and its successor at or above, a literal addition only below the table, the
//             period's samples adding up.  The checks are the definitions of the claimed numbers, so a step that
//             passes them has produced the reference's values.
//   who checks   every operation of the step is monotone in the period's start and every check compares such a
//             value with a constant, so claims that pass from both ends of an interval of starts pass from every
//             start inside it: the discovery runs the checks at the ends of a bracket around its estimate of the
//             start (gnsscorr_plan.hip, trk_spec_kernel), and a chain whose exact start lies inside the bracket
//             calls the step with its verdict unused -- the compiler then drops the checks and the values
//             remain.  A start outside its bracket takes the step with its checks, then the certified step
//             above, then the walkers.
//
// Both are the same function (template parameter DISCOVER), so there is one statement of the step's shape.
// In the binade where the addend is a tie (at most one) the segment always starts with one of the
// reference's own additions (pre): from an odd multiplier that step rounds to the even neighbour, from an
// even one it equals the table's step -- either way the rest of the segment starts even, where the table's
// step holds.
#define GC_CLAIM_ROW   24           // ints per (channel, period) row of either NCO
#define GC_CLAIM_LIT   5            // code: literal additions after the first wrap, at most (as gc_code_period_body)
#define GC_CLAIM_TAIL  15           // code: literal additions after the second wrap, at most (tap offsets up to 7 samples: 8 positions)
#define GC_CLAIM_TAIL2 32           // ... for correlator spacings up to 30 samples (the shipped CORRN=6, CORRD=3: 18)
#define GC_CLAIM_CLIT  8            // carrier: literal additions next to zero, at most (as gc_carrier_period)
#define GC_CLAIM_CSEG  13           // carrier: binade segments, at most
#define GC_CLAIM_PREM  12           // carrier: subtractions of DPI in the remainder loop, at most (12: up to ~11 kHz at 1 ms)

struct GcCodeClaims {               // GC_CLAIM_ROW ints
    int tag;                        // 1: claims present
    int i0, q, nl, jsum;            // entry binade of the climb, head steps, literal additions, samples before the tail
    int dm[13];                     // equal steps per table binade 0..ITOP (after the tie binade's own addition)
    int n, pad;                     // bracket form: the period's samples
    double lo, hi;                  // bracket form: period starts (remcode) the claims were proved for, ends included
};

struct GcCarClaims {                // GC_CLAIM_ROW ints
    int tag;                        // 1: table walk, 2: one binade far above the table (no claims needed)
    int nl, i0, nseg, kprem;        // literal additions (bracket form: the period's samples), entry binade, binade segments, subtractions of DPI
    int dm[GC_CLAIM_CSEG];
    int pad[2];
    double lo, hi;                  // bracket form: period starts (remcarr) the claims were proved for
};

GC_HD int gc_expo(double x) { return (int)((gc_d2u(x) >> 52) & 0x7FF); }
GC_HD uint32_t gc_hi32(double x) { return (uint32_t)(gc_d2u(x) >> 32); }

// per-channel constants of the code step, as the evaluation wants them: every one a plain value (on the device
// they are pinned to vector registers, GC_PIN_V: the chain's wavefront has few scalar registers to spare)
#if defined(__HIP_DEVICE_COMPILE__)
#define GC_PIN_V(x) asm volatile("" : "+v"(x))
#else
#define GC_PIN_V(x) do { } while (0)
#endif
// The evaluating form runs with every lane of a wavefront on the same period: a count it indexes registers with
// is the same in all lanes, and saying so (readfirstlane) lets the index go through the scalar unit.  (Never
// in the discovering form, whose lanes are different periods.)
#if defined(__HIP_DEVICE_COMPILE__)
#define GC_UNIFORM_INT(x) __builtin_amdgcn_readfirstlane(x)
#else
#define GC_UNIFORM_INT(x) (x)
#endif
// a[idx] for an index that is the same in every lane: on the device through a vector value (indexed register
// move), never through memory
template <int N>
GC_HD double gc_pick(const double (&a)[N], int idx)
{
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(N <= 48, "gc_pick: at most 48 values");
    if constexpr (N <= 16) {
        typedef double gc_vecn __attribute__((ext_vector_type(N)));
        gc_vecn v;
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = a[i];
        return v[idx < N ? idx : N - 1];
    } else {
        // sixteen at a time, then the chunk (the index is the same in every lane: scalar selects)
        typedef double gc_vec16 __attribute__((ext_vector_type(16)));
        constexpr int NC = (N + 15) / 16;
        double r = 0.0;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            gc_vec16 v;
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = a[c * 16 + i < N ? c * 16 + i : N - 1];
            const double rc = v[idx & 15];
            r = (idx >> 4) == c ? rc : r;
        }
        return r;
    }
#else
    return a[idx];
#endif
}

template <int ITOP>
struct GcCodeStepC {
    double d[ITOP + 1], pre[ITOP + 1];                     // step, the tie binade's own addition (ci or 0)
    double ci, dlen, b0, smaxci;
    int    ex_top;                                          // biased exponent the start value must have (binade ITOP's)
};

template <int ITOP>
GC_HD void gc_code_stepc_init(GcCodeStepC<ITOP> &C, const GcCodePlan &P)
{
    C.ci = P.f.s;
    C.dlen = P.dlen;
    C.b0 = gc_u2d((uint64_t)P.f.ex0 << 52);
    C.smaxci = P.smaxci;
    C.ex_top = P.f.ex0 + ITOP;
#pragma unroll
    for (int i = 0; i <= ITOP; i++) {
        C.d[i] = P.f.d[i];
        C.pre[i] = i == P.it ? P.f.s : 0.0;
        GC_PIN_V(C.d[i]);
        GC_PIN_V(C.pre[i]);
    }
    GC_PIN_V(C.ci);
    GC_PIN_V(C.dlen);
    GC_PIN_V(C.b0);
    GC_PIN_V(C.smaxci);
}

// PERLANE: the lanes of the calling wavefront hold different periods (always so when discovering; and when the
// discovery pass checks claims at the other end of a bracket)
template <int ITOP, int TMAX, bool DISCOVER, bool PERLANE = DISCOVER>
GC_HD bool gc_code_claims_step(const GcCodePlan &P, const GcCodeStepC<ITOP> &C, double remcode, int nt, GcCodeClaims &cl,
                               double *remcode_out, const double *dmd = nullptr)
{
    // dmd (evaluating a value only): the claimed counts of the climb as doubles, converted by the caller -- cl.dm is
    // then not read by anything the value depends on
    GC_FP_STRICT
    const double ci = C.ci, dlen = C.dlen, dtop = C.d[ITOP];
    bool ok = DISCOVER || cl.tag == 1;
    // ---- start value (ref src/sdrcmn.c:613-614) and head, as gc_code_period_body
    const double cs = remcode - C.smaxci;
    const double c0 = cs < 0.0 ? cs + dlen : cs;
    // (c0 positive, in the code length's binade and below the code length: with it -dlen <= cs < dlen, the range
    // in which the reference's floor(cs/dlen) is -1 or 0)
    ok = ok && (int)(gc_hi32(c0) >> 20) == C.ex_top && c0 < dlen;
    int q;
    if (DISCOVER) {
        const double R = P.limtop - c0;
        double qd = floor(R * P.f.inv[ITOP]);
        const double r = fma(-qd, dtop, R);
        qd += r < 0.0 ? -1.0 : (r >= dtop ? 1.0 : 0.0);
        q = (qd >= 0.0 && qd < 1.0e9) ? (int)qd : -1;
        cl.q = q;
    } else {
        q = cl.q;
    }
    const double dq = (double)q;
    const double yq = fma(dq, dtop, c0);            // last sample below the code length ...
    double y = fma(dq + 1.0, dtop, c0);             // ... and the first at or above it
    ok = ok && q < nt - 2 && yq < dlen && y >= dlen;         // (a negative q fails the last comparison)
    y = y - dlen;
    // ---- next to zero: the reference's own additions up to the table
    const double b0 = C.b0;
    int nl;
    if (DISCOVER) {
        nl = 0;
        double t = y;
        for (int k = 0; k < GC_CLAIM_LIT; k++)
            if (t < b0) { t = t + ci; nl++; }
        cl.nl = nl;
    } else {
        nl = cl.nl;
    }
    {
        // all GC_CLAIM_LIT sums, then the claimed one: the values rise, so "the last addition started below the
        // table and ended in it" covers the ones before it
        double ys[GC_CLAIM_LIT + 1];
        ys[0] = y;
#pragma unroll
        for (int k = 0; k < GC_CLAIM_LIT; k++) ys[k + 1] = ys[k] + ci;
        const int nlu = PERLANE ? nl : GC_UNIFORM_INT(nl);
        const bool inr = nlu >= 0 && nlu <= GC_CLAIM_LIT;
        const int ix = inr ? nlu : 0;
        const double yprev = gc_pick(ys, ix > 0 ? ix - 1 : 0);
        y = gc_pick(ys, ix);
        ok = ok && inr && (ix == 0 || yprev < b0);
    }
    ok = ok && y >= b0 && q + 1 + nl < nt - 2;
    const int i0 = gc_expo(y) - (C.ex_top - ITOP);
    if (DISCOVER) cl.i0 = i0;
    ok = ok && i0 == cl.i0 && i0 >= 0 && i0 <= 1;
    // ---- climb: one segment per table binade (only binade 0 can be skipped: i0 is 0 or 1)
    int j = q + 1 + nl;
    int dmor = 0;                                   // (no claimed count negative: one test for all of them)
#pragma unroll
    for (int i = 0; i <= ITOP; i++) {
        const bool active = i > 0 || i0 == 0;
        const double y1 = y + (active ? C.pre[i] : 0.0);
        int dm;
        if (DISCOVER) {
            dm = active ? gc_fast_piece(P.f, i, y1, 1 << 24, i == ITOP ? dlen : INFINITY) : 0;
            cl.dm[i] = dm;
            j += active ? (C.pre[i] != 0.0 ? 1 : 0) + dm + 1 : 0;
        } else {
            dm = cl.dm[i];
            GC_PIN_V(dm);
        }
        const double yl = fma(dmd ? dmd[i] : (double)dm, C.d[i], y1);
        const double yn = yl + ci;
        // last value of the segment below the top of its binade, the next one at or above it: for the table's
        // binades (tops are powers of two, the values positive) a comparison of the exponent fields
        bool in;
        if (i == ITOP) {
            in = yl < dlen && yn >= dlen;
        } else {
            const uint32_t ktop = (uint32_t)(C.ex_top - ITOP + i + 1) << 20;
            in = gc_hi32(yl) < ktop && gc_hi32(yn) >= ktop;
        }
        ok = ok && (!active || in);
        dmor |= dm;
        y = active ? yn : y;
    }
    ok = ok && dmor >= 0;
    if (DISCOVER) {
#pragma unroll
        for (int i = ITOP + 1; i < 13; i++) cl.dm[i] = 0;
        cl.jsum = j;
    } else {
        j = cl.jsum;                                // (the sum the discovering run of this function formed of the same claims)
    }
    // ---- second wrap and tail
    y = y - dlen;
    const int t = nt - j;
    ok = ok && t >= 1 && t <= TMAX;
    {
        double ys[TMAX + 1];
        ys[0] = y;
#pragma unroll
        for (int k = 0; k < TMAX; k++) ys[k + 1] = ys[k] + ci;
        const int tu = PERLANE ? t : GC_UNIFORM_INT(t);
        y = gc_pick(ys, (tu >= 1 && tu <= TMAX) ? tu : 0);
    }
    *remcode_out = y - C.smaxci;
    if (DISCOVER) cl.tag = ok ? 1 : 0;
    return ok;
}

template <int ITOP, bool DISCOVER>
GC_HD bool gc_code_claims_itop(const GcCodePlan &P, double remcode, int nt, GcCodeClaims &cl, double *remcode_out)
{
    GcCodeStepC<ITOP> C;
    gc_code_stepc_init(C, P);
    return gc_code_claims_step<ITOP, GC_CLAIM_TAIL2, DISCOVER, true>(P, C, remcode, nt, cl, remcode_out);     // (the widest tail: claims carry counts, not the instance)
}

template <bool DISCOVER>
GC_HD bool gc_code_claims(const GcCodePlan &P, double remcode, int nt, GcCodeClaims &cl, double *remcode_out)
{
    if (DISCOVER) cl.tag = 0;
    if (!P.ok) return false;
    switch (P.itop) {           // (GPS / GLONASS codes at 2..64 samples per chip: 7..12)
    case 7:  return gc_code_claims_itop<7, DISCOVER>(P, remcode, nt, cl, remcode_out);
    case 8:  return gc_code_claims_itop<8, DISCOVER>(P, remcode, nt, cl, remcode_out);
    case 9:  return gc_code_claims_itop<9, DISCOVER>(P, remcode, nt, cl, remcode_out);
    case 10: return gc_code_claims_itop<10, DISCOVER>(P, remcode, nt, cl, remcode_out);
    case 11: return gc_code_claims_itop<11, DISCOVER>(P, remcode, nt, cl, remcode_out);
    case 12: return gc_code_claims_itop<12, DISCOVER>(P, remcode, nt, cl, remcode_out);
    default: return false;
    }
}

// Carrier step on claims.  A tracked channel's phase (in LUT steps) starts a period inside (0, 32] -- or, for a
// falling phase, which is never wrapped, anywhere below zero -- and climbs a few binades in magnitude.  Two
// shapes are evaluated:
//   window      the period starts and ends inside the GC_CLAIM_CWIN table binades that hold the largest phase a
//               rising period of this channel can reach (per channel: GcCarStepC.ilo): one position per binade,
//               constants in registers;
//   one binade  the whole period inside one binade, whichever (gc_one_binade_walk: the usual case for a phase
//               far from zero).
// Anything else -- a start next to zero, a start below the window -- is left to the certified step and the
// walkers.
#define GC_CLAIM_CWIN 11
struct GcCarStepC {
    double d[GC_CLAIM_CWIN], pre[GC_CLAIM_CWIN];   // step, the tie binade's own addition (s or 0)
    double s;
    int    ilo, ex0;                // first binade of the window; biased exponent of table binade 0
};

// nmax: the longest period the channel is expected to have (samples); a longer one only loses the fast path
GC_HD void gc_car_stepc_init(GcCarStepC &C, const GcCarPlan &P, int nmax)
{
    GC_FP_STRICT
    const GcNcoFast &f = P.f;
    C.s = f.s;
    C.ex0 = f.ex0;
    C.ilo = 0;
    if (f.ex0 != 0x7FFFFFF) {
        const double xmax = fma((double)nmax, fabs(f.s), GC_NCO_CDIV);
        int imax = gc_expo(xmax) - f.ex0;
        imax = imax < GC_CLAIM_CWIN - 1 ? GC_CLAIM_CWIN - 1 : (imax > GC_NB - 1 ? GC_NB - 1 : imax);
        C.ilo = imax - (GC_CLAIM_CWIN - 1);
    }
#pragma unroll
    for (int p = 0; p < GC_CLAIM_CWIN; p++) {
        C.d[p] = 0.0;
        C.pre[p] = 0.0;
#pragma unroll
        for (int i = 0; i < GC_NB; i++) {           // (static indices into the table)
            if (i != C.ilo + p) continue;
            C.d[p] = f.d[i];
            C.pre[p] = ((f.tie >> i) & 1) ? f.s : 0.0;
        }
        GC_PIN_V(C.d[p]);
        GC_PIN_V(C.pre[p]);
    }
    GC_PIN_V(C.s);
}

// SHAPE (evaluating): 0 the row says which (cl.tag), 1 / 2 the caller knows it is the window / one binade
template <bool DISCOVER, bool PERLANE = DISCOVER, int SHAPE = 0>
GC_HD bool gc_carrier_claims_step(const GcCarPlan &P, const GcCarStepC &C, double remcarr, int n, GcCarClaims &cl, double *remcarr_out,
                                  const double *dmd = nullptr)
{
    GC_FP_STRICT
    const double s = C.s;
    if (DISCOVER) {
        cl.tag = 0;
        cl.nl = cl.i0 = cl.nseg = cl.kprem = 0;
        for (int j = 0; j < GC_CLAIM_CSEG; j++) cl.dm[j] = 0;
        cl.pad[0] = cl.pad[1] = 0;
    }
    if (C.ex0 == 0x7FFFFFF || n < 1) return false;
    double x = gc_div_y(remcarr * GC_NCO_CDIV, GC_NCO_DPI, P.ydpi);      // ref src/sdrcmn.c:649
    bool ok = true;
    if (DISCOVER) {
        double xe;
        cl.tag = gc_one_binade_walk(x, s, n, &xe) ? 2 : 1;
    }
    if (SHAPE == 2 || (SHAPE == 0 && cl.tag == 2)) {
        if (!gc_one_binade_walk(x, s, n, &x)) { if (DISCOVER) cl.tag = 0; return false; }
    } else {
        ok = cl.tag == 1;
        const int i0 = gc_expo(x) - C.ex0, p0 = i0 - C.ilo;
        ok = ok & (p0 >= 0) & (p0 < GC_CLAIM_CWIN) & ((gc_d2u(x) >> 63) == (gc_d2u(s) >> 63)) & (x != 0.0);
        if (DISCOVER) {
            // the segments, as gc_fast_carrier_walk steps them; dm by window position
            cl.i0 = i0;
            int k = 0, nseg = 0;
            double t = x;
            bool fits = ok;
            for (int p = 0; p < GC_CLAIM_CWIN; p++) {
                if (!fits || p < p0 || k >= n) continue;
                if (gc_expo(t) != C.ex0 + C.ilo + p) { fits = false; continue; }
                if (C.pre[p] != 0.0) {
                    if (k >= n - 1) { fits = false; continue; }     // (the tie binade's addition would be the period's last: not here)
                    t = t + s;
                    k += 1;
                    if (gc_expo(t) != C.ex0 + C.ilo + p) { fits = false; continue; }
                }
                const int m = gc_piece_len(C.d[p], 1.0 / fabs(C.d[p]), t, n - 1 - k, INFINITY);
                cl.dm[p] = m;
                t = fma((double)m, C.d[p], t);
                k += m;
                t = t + s;
                k += 1;
                nseg++;
            }
            if (!fits || k != n) ok = false;
            cl.nseg = nseg;
            cl.nl = k;                              // samples the claimed segments add up to
        }
        ok = ok & (i0 == cl.i0) & (cl.nseg >= 1) & (p0 + cl.nseg <= GC_CLAIM_CWIN) & (cl.nl == n);
        const int plast = p0 + cl.nseg - 1;
        int dmor = 0;                               // (no claimed count negative: one test for all of them)
#pragma unroll
        for (int p = 0; p < GC_CLAIM_CWIN; p++) {
            const bool active = (p >= p0) & (p <= plast), last = p == plast;
            int dm = cl.dm[p];
            if (!DISCOVER) GC_PIN_V(dm);            // (stays in its vector register: the chain is short of scalar ones)
            const double x1 = x + C.pre[p];
            const double xl = fma(dmd ? dmd[p] : (double)dm, C.d[p], x1);
            const double xn = xl + s;
            // (magnitudes against the top of the binade, a power of two: exponent fields)
            const uint32_t ktop = (uint32_t)(C.ex0 + C.ilo + p + 1) << 20;
            const uint32_t hl = gc_hi32(xl) & 0x7FFFFFFFu, hn = gc_hi32(xn) & 0x7FFFFFFFu;
            ok = ok & ((int)!active | ((int)(hl < ktop) & ((int)last | (int)(hn >= ktop))));
            dmor |= dm;
            x = active ? xn : x;
        }
        ok = ok & (dmor >= 0);
    }
    // ---- phase remainder (ref :666-668): the reference's own subtractions, the last one in the last position
    double p = x * GC_NCO_DPI * (1.0 / GC_NCO_CDIV);
    ok = ok & (p < 1.0e300);
    int kp;
    if (DISCOVER) {
        kp = 0;
        double t = p;
        for (int k = 0; k < GC_CLAIM_PREM; k++)
            if (t > GC_NCO_DPI) { t = t - GC_NCO_DPI; kp++; }
        cl.kprem = kp;
    } else {
        kp = cl.kprem;
    }
    {
        // all GC_CLAIM_PREM differences, then the claimed one (the values fall: if the last subtraction was
        // called for, so were the ones before it)
        double ps[GC_CLAIM_PREM + 1];
        ps[0] = p;
#pragma unroll
        for (int k = 0; k < GC_CLAIM_PREM; k++) ps[k + 1] = ps[k] - GC_NCO_DPI;
        const int ku = PERLANE ? kp : GC_UNIFORM_INT(kp);
        const bool inr = ku >= 0 && ku <= GC_CLAIM_PREM;
        const int ix = inr ? ku : 0;
        const double pprev = gc_pick(ps, ix > 0 ? ix - 1 : 0);
        p = gc_pick(ps, ix);
        ok = ok & inr & ((ix == 0) | (pprev > GC_NCO_DPI)) & !(p > GC_NCO_DPI);
    }
    *remcarr_out = p;
    if (DISCOVER && !ok) cl.tag = 0;
    return ok;
}

#if defined(__HIPCC__)
// crossings one boundary per lane, handed to every lane by readlane (wave-uniform afterwards)
struct GcFillLanes {
    int lane;
    __device__ bool operator()(int *K, const GcCertCtx &c, int i0, int itop, double lim) const
    {
        int Kl = GC_CERT_FAR;
        const bool mine = lane > i0 && lane <= GC_NB && (lane <= itop || lane == GC_NB);
        if (mine) Kl = gc_cert_lane(c, lane, lim);
        // strictly increasing: against the lane below (lane GC_NB against lane itop)
        const int src = lane == GC_NB ? itop : lane - 1;
        const int below = __shfl(Kl, src < 0 ? 0 : src, 64);
        const bool bad = mine && (Kl == GC_CERT_FAIL || (src > i0 && Kl <= below && !(Kl == GC_CERT_FAR && (lane != GC_NB || !(lim < 1.0e300)))) || Kl <= 0);
        if (__any(bad)) return false;
#pragma unroll
        for (int i = 0; i <= GC_NB; i++) K[i] = __builtin_amdgcn_readlane(Kl, i);
        return true;
    }
};
#endif
