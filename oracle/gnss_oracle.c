/*
 * gnss_oracle.c -- CPU restatement of GNSS-SDRLIB's acquisition + tracking
 * correlation path (see gnss_oracle.h: TEST INFRASTRUCTURE ONLY, parity
 * unpinned).  Every function cites the reference file:line it follows
 * (paths relative to the reference root).
 */
#include "gnss_oracle.h"

#include <complex.h>
#undef I            /* parameter names below use I/Q for the two rails */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* PRN code generators                                                       */
/* ------------------------------------------------------------------------- */

/* G2 tap delays for PRN 1..210, ref src/sdrcode.c:103-125 (IS-GPS-200 /
 * SBAS / QZSS assignment table). */
static const short ca_delay[210] = {
      5,   6,   7,   8,  17,  18, 139, 140, 141, 251,
    252, 254, 255, 256, 257, 258, 469, 470, 471, 472,
    473, 474, 509, 512, 513, 514, 515, 516, 859, 860,
    861, 862, 863, 950, 947, 948, 950,  67, 103,  91,
     19, 679, 225, 625, 946, 638, 161,1001, 554, 280,
    710, 709, 775, 864, 558, 220, 397,  55, 898, 759,
    367, 299,1018, 729, 695, 780, 801, 788, 732,  34,
    320, 327, 389, 407, 525, 405, 221, 761, 260, 326,
    955, 653, 699, 422, 188, 438, 959, 539, 879, 677,
    586, 153, 792, 814, 446, 264,1015, 278, 536, 819,
    156, 957, 159, 712, 885, 461, 248, 713, 126, 807,
    279, 122, 197, 693, 632, 771, 467, 647, 203, 145,
    175,  52,  21, 237, 235, 886, 657, 634, 762, 355,
   1012, 176, 603, 130, 359, 595,  68, 386, 797, 456,
    499, 883, 307, 127, 211, 121, 118, 163, 628, 853,
    484, 289, 811, 202,1021, 463, 568, 904, 670, 230,
    911, 684, 309, 644, 932,  12, 314, 891, 212, 185,
    675, 503, 150, 395, 345, 846, 798, 992, 357, 995,
    877, 112, 144, 476, 193, 109, 445, 291,  87, 399,
    292, 901, 339, 208, 711, 189, 263, 537, 663, 942,
    173, 900,  30, 500, 935, 556, 373,  85, 652, 310
};

/* ref src/sdrcode.c:101-154.  The reference keeps the two 10-stage registers
 * in +-1 form initialised to -1 and multiplies taps; here the same registers
 * are kept as bits (1 <-> -1) and taps are XORed.  code = -G1*G2(delayed),
 * i.e. chip +1 where the XOR of the two output bits is 1. */
static int gen_l1ca(int prn, short *code, int *len, double *crate)
{
    unsigned char g1[1023], g2[1023];
    unsigned r1 = 0x3FF, r2 = 0x3FF;  /* bit s (0-based) = stage s+1 */
    int i;
    if (prn < 1 || prn > 210) return -1;
    for (i = 0; i < 1023; i++) {
        unsigned f1, f2;
        g1[i] = (r1 >> 9) & 1;
        g2[i] = (r2 >> 9) & 1;
        f1 = ((r1 >> 2) ^ (r1 >> 9)) & 1;                       /* R1[2]*R1[9] */
        f2 = ((r2 >> 1) ^ (r2 >> 2) ^ (r2 >> 5) ^ (r2 >> 7) ^
              (r2 >> 8) ^ (r2 >> 9)) & 1;           /* R2[1,2,5,7,8,9] product */
        r1 = ((r1 << 1) | f1) & 0x3FF;
        r2 = ((r2 << 1) | f2) & 0x3FF;
    }
    for (i = 0; i < 1023; i++) {
        int j = (i + 1023 - ca_delay[prn - 1]) % 1023;
        code[i] = (g1[i] ^ g2[j]) ? 1 : -1;
    }
    *len = 1023;
    *crate = 1.023e6;
    return 0;
}

/* ref src/sdrcode.c:426-444: 9-stage register, all -1 (bit 1) at start,
 * code[i] = -R[6], feedback R[4]*R[8]. */
static int gen_g1(short *code, int *len, double *crate)
{
    unsigned r = 0x1FF;
    int i;
    for (i = 0; i < 511; i++) {
        unsigned f = ((r >> 4) ^ (r >> 8)) & 1;
        code[i] = ((r >> 6) & 1) ? 1 : -1;
        r = ((r << 1) | f) & 0x1FF;
    }
    *len = 511;
    *crate = 0.511e6;
    return 0;
}

/* ref src/sdrcode.c:523-539.  The reference's switch has no CTYPE_G1 case
 * (its G1 generator is unreachable); the oracle exposes it because the build
 * needs GLONASS channels (SURVEY hard part 8). */
int orc_gencode(int prn, int ctype, short *code, int *len, double *crate)
{
    switch (ctype) {
    case ORC_CTYPE_L1CA:
    case ORC_CTYPE_L1SBAS: return gen_l1ca(prn, code, len, crate);
    case ORC_CTYPE_G1:     return gen_g1(code, len, crate);
    default:               return -1;
    }
}

/* ------------------------------------------------------------------------- */
/* carrier wipe-off                                                          */
/* ------------------------------------------------------------------------- */

/* ref src/sdrcmn.c:643-648 */
void orc_carrier_lut(short *cost, short *sint)
{
    int i;
    for (i = 0; i < ORC_CDIV; i++) {
        cost[i] = (short)floor(cos(ORC_DPI / ORC_CDIV * i) / ORC_CSCALE + 0.5);
        sint[i] = (short)floor(sin(ORC_DPI / ORC_CDIV * i) / ORC_CSCALE + 0.5);
    }
}

static inline void mix_one(const signed char *data, int dtype, int k, int idx,
                           const short *cost, const short *sint, short *I,
                           short *Q)
{
    if (dtype == 2) {          /* ref src/sdrcmn.c:652-657 */
        int d0 = data[2 * k], d1 = data[2 * k + 1];
        I[k] = (short)(cost[idx] * d0 - sint[idx] * d1);
        Q[k] = (short)(sint[idx] * d0 + cost[idx] * d1);
    } else {                   /* ref src/sdrcmn.c:659-664 */
        int d0 = data[k];
        I[k] = (short)(cost[idx] * d0);
        Q[k] = (short)(sint[idx] * d0);
    }
}

/* ref src/sdrcmn.c:666-668: one-sided wrap of the phase remainder */
static double phase_remainder(double phi)
{
    double prem = phi * ORC_DPI / ORC_CDIV;
    while (prem > ORC_DPI) prem -= ORC_DPI;
    return prem;
}

/* ref src/sdrcmn.c:633-669, literal: phi += ps once per sample */
double orc_mixcarr_seq(const signed char *data, int dtype, double ti, int n,
                       double freq, double phi0, short *I, short *Q)
{
    short cost[ORC_CDIV], sint[ORC_CDIV];
    double phi = phi0 * ORC_CDIV / ORC_DPI;
    double ps = freq * ORC_CDIV * ti;
    int k;
    orc_carrier_lut(cost, sint);
    for (k = 0; k < n; k++, phi += ps)
        mix_one(data, dtype, k, ((int)phi) & (ORC_CDIV - 1), cost, sint, I, Q);
    return phase_remainder(phi);
}

/* ------------------------------------------------------------------------- */
/* code resampling                                                           */
/* ------------------------------------------------------------------------- */

/* ref src/sdrcmn.c:608-621, literal */
double orc_rescode_seq(const short *code, int len, double coff, int smax,
                       double ci, int n, short *rcode)
{
    int j;
    coff -= smax * ci;
    coff -= floor(coff / len) * len;
    for (j = 0; j < n + 2 * smax; j++, coff += ci) {
        if (coff >= len) coff -= len;
        rcode[j] = code[(int)coff];
    }
    return coff - smax * ci;
}

/* ------------------------------------------------------------------------- */
/* tracking correlator                                                       */
/* ------------------------------------------------------------------------- */

/* dot_23 / dot_22, ref src/sdrcmn.c:307-354: two input rows against three / two replica rows in one pass, double
 * accumulators of short*short products, the accumulations written out one by one as in the reference (so that the
 * CPU baseline runs the loops the reference's code runs). */
static void orc_dot_23(const short *a1, const short *a2, const short *b1, const short *b2, const short *b3, int n,
                       double *d1, double *d2)
{
    const short *p1 = a1, *p2 = a2, *q1 = b1, *q2 = b2, *q3 = b3;
    d1[0] = d1[1] = d1[2] = d2[0] = d2[1] = d2[2] = 0.0;
    for (; p1 < a1 + n; p1++, p2++, q1++, q2++, q3++) {
        d1[0] += (*p1) * (*q1);
        d1[1] += (*p1) * (*q2);
        d1[2] += (*p1) * (*q3);
        d2[0] += (*p2) * (*q1);
        d2[1] += (*p2) * (*q2);
        d2[2] += (*p2) * (*q3);
    }
}

static void orc_dot_22(const short *a1, const short *a2, const short *b1, const short *b2, int n, double *d1, double *d2)
{
    const short *p1 = a1, *p2 = a2, *q1 = b1, *q2 = b2;
    d1[0] = d1[1] = d2[0] = d2[1] = 0.0;
    for (; p1 < a1 + n; p1++, p2++, q1++, q2++) {
        d1[0] += (*p1) * (*q1);
        d1[1] += (*p1) * (*q2);
        d2[0] += (*p2) * (*q1);
        d2[1] += (*p2) * (*q2);
    }
}

/* ref src/sdrcmn.c:687-722 */
void orc_correlator(const signed char *data, int dtype, double ti, int n,
                    double freq, double phi0, double crate, double coff,
                    const int *s, int ns, double *II, double *QQ,
                    double *remc, double *remp, const short *codein, int coden)
{
    int smax = s[ns - 1], i;
    short *dI = (short *)malloc(sizeof(short) * (size_t)(n + 64));
    short *dQ = (short *)malloc(sizeof(short) * (size_t)(n + 64));
    short *ce = (short *)malloc(sizeof(short) * (size_t)(n + 2 * smax + 1));
    const short *code;
    if (!dI || !dQ || !ce) { free(dI); free(dQ); free(ce); return; }
    code = ce + smax;

    *remp = orc_mixcarr_seq(data, dtype, ti, n, freq, phi0, dI, dQ);
    *remc = orc_rescode_seq(codein, coden, coff, smax, ti * crate, n, ce);
    /* P, E1, L1 then (Ei, Li) pairs: ref :712-715 */
    orc_dot_23(dI, dQ, code, code - s[0], code + s[0], n, II, QQ);
    for (i = 1; i < ns; i++)
        orc_dot_22(dI, dQ, code - s[i], code + s[i], n, II + 1 + i * 2, QQ + 1 + i * 2);
    for (i = 0; i < 1 + 2 * ns; i++) {
        II[i] *= ORC_CSCALE;
        QQ[i] *= ORC_CSCALE;
    }
    free(dI); free(dQ); free(ce);
}

/* ------------------------------------------------------------------------- */
/* FFT-based parallel code phase search                                      */
/* ------------------------------------------------------------------------- */

/* ref src/sdrcmn.c:185-195 */
void orc_cpxcpx(const short *I, const short *Q, double scale, int n, float *cpx)
{
    int i;
    for (i = 0; i < n; i++) {
        cpx[2 * i]     = I[i] * (float)scale;
        cpx[2 * i + 1] = Q ? Q[i] * (float)scale : 0.0f;
    }
}

/* Mixed-radix decimation-in-time DFT for any length (the reference's FFT
 * length 2*nsamp = 32736 = 2^5*3*11*31 is not a power of two).  Arithmetic in
 * double, result rounded to float once: this stands in for FFTW3f, whose
 * rounding cannot be reproduced (library absent, version unpinned). */
typedef struct { int n, sign; double complex *w; } twtab_t;
static __thread twtab_t g_tw[4];

static const double complex *roots(int n, int sign)
{
    int i, slot = -1;
    for (i = 0; i < 4; i++)
        if (g_tw[i].w && g_tw[i].n == n && g_tw[i].sign == sign) return g_tw[i].w;
    for (i = 0; i < 4; i++) if (!g_tw[i].w) { slot = i; break; }
    if (slot < 0) { slot = 0; free(g_tw[0].w); g_tw[0].w = NULL; }
    g_tw[slot].w = (double complex *)malloc(sizeof(double complex) * (size_t)n);
    g_tw[slot].n = n; g_tw[slot].sign = sign;
    for (i = 0; i < n; i++) {
        double a = ORC_DPI * (double)i / (double)n;
        g_tw[slot].w[i] = cos(a) + _Complex_I * (sign * sin(a));
    }
    return g_tw[slot].w;
}

static int small_factor(int n)
{
    int p;
    if (n % 4 == 0) return 4;
    if (n % 2 == 0) return 2;
    for (p = 3; p * p <= n; p += 2) if (n % p == 0) return p;
    return n;
}

/* out[0..n) = DFT of in[0], in[stride], ...; scratch has n entries */
static void dft_rec(const double complex *in, int stride, int n,
                    double complex *out, double complex *scratch,
                    const double complex *w, int wn)
{
    int p, m, r, k, q;
    if (n == 1) { out[0] = in[0]; return; }
    p = small_factor(n);
    m = n / p;
    for (r = 0; r < p; r++)
        dft_rec(in + (size_t)r * stride, stride * p, m, scratch + (size_t)r * m,
                out + (size_t)r * m, w, wn);
    {
        int ws = wn / n;              /* w[ws*t] = root_n^t   */
        int wp = wn / p;              /* w[wp*t] = root_p^t   */
        double complex t[64];
        if (p > 64) return;           /* primes above 64 are not needed here */
        for (k = 0; k < m; k++) {
            for (r = 0; r < p; r++)
                t[r] = scratch[(size_t)r * m + k] * w[(size_t)ws * ((size_t)r * k % n)];
            if (p == 2) {
                out[k] = t[0] + t[1];
                out[k + m] = t[0] - t[1];
            } else if (p == 4) {
                double complex j1 = w[wp];         /* root_4^1 = -+i */
                double complex a = t[0] + t[2], b = t[0] - t[2];
                double complex c = t[1] + t[3], d = (t[1] - t[3]) * j1;
                out[k] = a + c;
                out[k + m] = b + d;
                out[k + 2 * m] = a - c;
                out[k + 3 * m] = b - d;
            } else {
                for (q = 0; q < p; q++) {
                    double complex acc = t[0];
                    for (r = 1; r < p; r++)
                        acc += t[r] * w[(size_t)wp * ((r * q) % p)];
                    out[k + (size_t)q * m] = acc;
                }
            }
        }
    }
}

void orc_fft(float *cpx, int n, int sign)
{
    double complex *in = (double complex *)malloc(sizeof(double complex) * 3 * (size_t)n);
    double complex *out = in + n, *scr = in + 2 * (size_t)n;
    const double complex *w = roots(n, sign < 0 ? -1 : 1);
    int i;
    for (i = 0; i < n; i++) in[i] = cpx[2 * i] + _Complex_I * (double)cpx[2 * i + 1];
    dft_rec(in, 1, n, out, scr, w, n);
    for (i = 0; i < n; i++) {
        cpx[2 * i]     = (float)creal(out[i]);
        cpx[2 * i + 1] = (float)cimag(out[i]);
    }
    free(in);
}

/* ref src/sdrcmn.c:228-251 */
void orc_cpxconv(float *a, const float *b, int m, int n, int flagsum,
                 double *conv)
{
    float m2 = (float)m * m, re;
    int i;
    orc_fft(a, m, -1);
    for (i = 0; i < m; i++) {           /* -A*conj(B): ref :236-240 */
        float *p = a + 2 * i;
        const float *q = b + 2 * i;
        re   = -p[0] * q[0] - p[1] * q[1];
        p[1] =  p[0] * q[1] - p[1] * q[0];
        p[0] = re;
    }
    orc_fft(a, m, +1);
    for (i = 0; i < n; i++) {
        const float *p = a + 2 * i;
        double v = (p[0] * p[0] + p[1] * p[1]) / m2;
        if (flagsum) conv[i] += v; else conv[i] = v;
    }
}

/* ref src/sdrcmn.c:261-276 */
void orc_cpxpspec(float *cpx, int n, int flagsum, double *pspec)
{
    int i;
    orc_fft(cpx, n, -1);
    for (i = 0; i < n; i++) {
        const float *p = cpx + 2 * i;
        double v = (p[0] * p[0] + p[1] * p[1]);
        if (flagsum) pspec[i] += v; else pspec[i] = v;
    }
}

/* ref src/sdrinit.c:645-655 */
void orc_codespectrum(const short *code, int clen, double ci, int nsamp,
                      int nfft, float *xcode)
{
    short *rc = (short *)calloc((size_t)nfft, sizeof(short));
    orc_rescode_seq(code, clen, 0.0, 0, ci, nsamp, rc);
    orc_cpxcpx(rc, NULL, 1.0, nfft, xcode);
    orc_fft(xcode, nfft, -1);
    free(rc);
}

/* ref src/sdrcmn.c:738-773.  dataR is a full copy of the 2n input samples
 * (m = 2n: the "zero padding" memset is overwritten completely). */
void orc_pcorrelator(const signed char *data, int dtype, double ti, int n,
                     const double *freq, int nfreq, double crate, int m,
                     const float *codex, double *P)
{
    signed char *dR = (signed char *)calloc((size_t)m * dtype, 1);
    short *dI = (short *)malloc(sizeof(short) * (size_t)(m + 64));
    short *dQ = (short *)malloc(sizeof(short) * (size_t)(m + 64));
    float *dx = (float *)malloc(sizeof(float) * 2 * (size_t)m);
    int i;
    (void)crate;
    memcpy(dR, data, (size_t)2 * n * dtype);
    for (i = 0; i < nfreq; i++) {
        orc_mixcarr_seq(dR, dtype, ti, m, freq[i], 0.0, dI, dQ);
        orc_cpxcpx(dI, dQ, ORC_CSCALE / m, m, dx);
        orc_cpxconv(dx, codex, m, n, 1, &P[(size_t)i * n]);
    }
    free(dR); free(dI); free(dQ); free(dx);
}

/* Direct evaluation of what pcorrelator computes (normative closed
 * form): no FFT involved, used to cross-check the FFT route on a few lags. */
void orc_pcorrelator_td(const signed char *data, int dtype, double ti, int n,
                        const double *freq, int nfreq, int m,
                        const short *code, int clen, double ci,
                        int k0, int k1, double *P)
{
    short *dI = (short *)malloc(sizeof(short) * (size_t)(m + 64));
    short *dQ = (short *)malloc(sizeof(short) * (size_t)(m + 64));
    short *rc = (short *)malloc(sizeof(short) * (size_t)n);
    int b, k, j;
    orc_rescode_seq(code, clen, 0.0, 0, ci, n, rc);
    for (b = 0; b < nfreq; b++) {
        orc_mixcarr_seq(data, dtype, ti, m, freq[b], 0.0, dI, dQ);
        for (k = k0; k < k1; k++) {
            double sr = 0.0, si = 0.0, sc = 32.0 * (double)m;
            for (j = 0; j < n; j++) {
                sr += dI[k + j] * rc[j];
                si += dQ[k + j] * rc[j];
            }
            P[(size_t)b * n + k] += (sr * sr + si * si) / (sc * sc);
        }
    }
    free(dI); free(dQ); free(rc);
}

/* ------------------------------------------------------------------------- */
/* acquisition decision                                                      */
/* ------------------------------------------------------------------------- */

static int outside(int i, int exinds, int exinde)
{
    return (exinds <= exinde && (i < exinds || i > exinde)) ||
           (exinds >  exinde && (i < exinds && i > exinde));
}

/* ref src/sdrcmn.c:461-476: element 0 seeds the maximum even when it lies in
 * the excluded window; strict '<' keeps the first maximum. */
double orc_maxvd(const double *d, int n, int exinds, int exinde, int *ind)
{
    double mx = d[0];
    int i;
    *ind = 0;
    for (i = 1; i < n; i++)
        if (outside(i, exinds, exinde) && mx < d[i]) { mx = d[i]; *ind = i; }
    return mx;
}

/* ref src/sdrcmn.c:487-497 */
double orc_meanvd(const double *d, int n, int exinds, int exinde)
{
    double mean = 0.0;
    int i, ne = 0;
    for (i = 0; i < n; i++) {
        if (outside(i, exinds, exinde)) mean += d[i]; else ne++;
    }
    return mean / (n - ne);
}

/* ref src/sdracq.c:71-95 (+ ind2sub src/sdrcmn.c:574-578) */
int orc_checkacquisition(const double *P, int nsamp, int nfreq, int nsampchip,
                         double ctime, const double *freq, orc_acqres_t *res)
{
    int maxi, codei, freqi, exinds, exinde, dummy;
    double maxP, maxP2, meanP;
    maxP = orc_maxvd(P, nsamp * nfreq, -1, -1, &maxi);
    codei = maxi % nsamp;
    freqi = nfreq * maxi / (nsamp * nfreq);
    exinds = codei - 2 * nsampchip; if (exinds < 0) exinds += nsamp;
    exinde = codei + 2 * nsampchip; if (exinde >= nsamp) exinde -= nsamp;
    meanP = orc_meanvd(&P[(size_t)freqi * nsamp], nsamp, exinds, exinde);
    res->cn0 = 10 * log10(maxP / meanP / ctime);
    maxP2 = orc_maxvd(&P[(size_t)freqi * nsamp], nsamp, exinds, exinde, &dummy);
    res->peakr = maxP / maxP2;
    res->acqcodei = codei;
    res->freqi = freqi;
    res->acqfreq = freq[freqi];
    res->acquired = res->peakr > 3.0;   /* ACQTH, ref src/sdr.h:148 */
    return res->acquired;
}

/* ------------------------------------------------------------------------- */
/* drivers                                                                   */
/* ------------------------------------------------------------------------- */

/* ref src/sdrrcv.c:505-532 */
void orc_getbuff(const orc_ring_t *ring, uint64_t buffloc, int n, int dtype,
                 signed char *out)
{
    uint64_t rb = (uint64_t)dtype * ring->ringlen;
    uint64_t loc = ((uint64_t)dtype * buffloc) % rb;
    int nb = dtype * n;
    int nout = (int)((int64_t)(loc + (uint64_t)nb) - (int64_t)rb);
    if (nout > 0) {
        memcpy(out, ring->buff + loc, (size_t)(nb - nout));
        memcpy(out + (nb - nout), ring->buff, (size_t)nout);
    } else {
        memcpy(out, ring->buff + loc, (size_t)nb);
    }
}

/* ref src/sdrinit.c:583-657 with :385-394 (acq), :402-425 (loop constants),
 * :432-480 (taps).  GLONASS FDMA offsets as :612-615. */
int orc_initchan(orc_chan_t *ch, int prn, int ctype, int dtype, double f_cf,
                 double f_sf, double f_if, int corrn, int corrd, int corrp,
                 const double *dllb, const double *pllb, const double *fllb)
{
    int i;
    memset(ch, 0, sizeof(*ch));
    ch->dtype = dtype;
    ch->f_sf = f_sf;
    ch->f_if = f_if;
    ch->ti = 1 / f_sf;
    if (orc_gencode(prn, ctype, ch->code, &ch->clen, &ch->crate) < 0) return -1;
    ch->ci = ch->ti * ch->crate;
    ch->ctime = ch->clen / ch->crate;
    ch->nsamp = (int)(f_sf * ch->ctime);
    ch->nsampchip = (int)(ch->nsamp / ch->clen);
    if (ctype == ORC_CTYPE_G1) {
        ch->f_cf = 1.60200E9 + 0.56250E6 * prn;
        ch->foffset = 0.56250E6 * prn;
    } else {
        ch->f_cf = f_cf;
        ch->foffset = 0.0;
    }
    ch->intg = 10;                       /* ACQINTG_*            */
    ch->nfreq = 2 * (7000 / 200) + 1;    /* ACQHBAND / ACQSTEP   */
    ch->nfft = 2 * ch->nsamp;
    for (i = 0; i < ch->nfreq; i++)
        ch->freq[i] = ch->f_if + ((i - (ch->nfreq - 1) / 2) * 200.0) + ch->foffset;
    ch->corrn = corrn;
    for (i = 0; i < corrn; i++) {
        ch->corrp[i] = corrd * (i + 1);
        if (ch->corrp[i] == corrp) { ch->ne = 2 * (i + 1) - 1; ch->nl = 2 * (i + 1); }
    }
    ch->loopms = (ctype == ORC_CTYPE_L1SBAS ? 2 : 10) * (int)(ch->ctime * 1000);
    ch->prn = prn;                                                                  /* nav->sdreph.prn, ref src/sdrinit.c:506 */
    ch->rate = ctype == ORC_CTYPE_G1 ? 10 : (ctype == ORC_CTYPE_L1SBAS ? 2 : 20);   /* NAVRATE_*, ref src/sdr.h:159-171 */
    for (i = 0; i < 2; i++) {
        ch->dllw2[i] = (dllb[i] / 0.53) * (dllb[i] / 0.53);
        ch->dllaw[i] = 1.414 * (dllb[i] / 0.53);
        ch->pllw2[i] = (pllb[i] / 0.53) * (pllb[i] / 0.53);
        ch->pllaw[i] = 1.414 * (pllb[i] / 0.53);
        ch->fllw[i]  = fllb[i] / 0.25;
    }
    return 0;
}

/* ref src/sdracq.c:14-62 */
uint64_t orc_sdracquisition(orc_chan_t *ch, const orc_ring_t *ring,
                            double *power, int *iters_done)
{
    signed char *data = (signed char *)malloc((size_t)2 * ch->nsamp * ch->dtype);
    uint64_t buffloc = ring->wrpos - (uint64_t)(ch->intg + 1) * ch->nsamp;
    int i;
    for (i = 0; i < ch->intg; i++) {
        orc_getbuff(ring, buffloc, 2 * ch->nsamp, ch->dtype, data);
        buffloc += ch->nsamp;
        orc_pcorrelator(data, ch->dtype, ch->ti, ch->nsamp, ch->freq, ch->nfreq,
                        ch->crate, ch->nfft, ch->xcode, power);
        if (orc_checkacquisition(power, ch->nsamp, ch->nfreq, ch->nsampchip,
                                 ch->ctime, ch->freq, &ch->acq)) {
            ch->flagacq = 1;
            break;
        }
    }
    if (iters_done) *iters_done = (i < ch->intg) ? i + 1 : ch->intg;
    if (ch->flagacq) {
        buffloc += (uint64_t)(-(int64_t)(i + 1) * ch->nsamp + ch->acq.acqcodei);
        ch->carrfreq = ch->acq.acqfreq;
        ch->codefreq = ch->crate;
    }
    free(data);
    return buffloc;
}

/* ref src/sdrtrk.c:15-54.  trk.QQ is handed to the correlator as its "II"
 * and trk.II as its "QQ" (:42), and the old-value memcpy copies
 * 1+2*corrn*sizeof(double) bytes (:35-36); both kept. */
uint64_t orc_sdrtracking(orc_chan_t *ch, const orc_ring_t *ring,
                         uint64_t buffloc)
{
    uint64_t bufflocnow = ring->wrpos - (uint64_t)ch->nsamp;
    ch->flagtrk = 0;
    if (bufflocnow > buffloc) {
        signed char *data =
            (signed char *)malloc((size_t)(ch->nsamp + 100) * ch->dtype);
        ch->currnsamp = (int)((ch->clen - ch->remcode) / (ch->codefreq / ch->f_sf));
        orc_getbuff(ring, buffloc, ch->currnsamp, ch->dtype, data);
        memcpy(ch->oldI, ch->II, 1 + 2 * ch->corrn * sizeof(double));
        memcpy(ch->oldQ, ch->QQ, 1 + 2 * ch->corrn * sizeof(double));
        ch->oldremcode = ch->remcode;
        ch->oldremcarr = ch->remcarr;
        orc_correlator(data, ch->dtype, ch->ti, ch->currnsamp, ch->carrfreq,
                       ch->oldremcarr, ch->codefreq, ch->oldremcode, ch->corrp,
                       ch->corrn, ch->QQ, ch->II, &ch->remcode, &ch->remcarr,
                       ch->code, ch->clen);
        ch->flagtrk = 1;
        free(data);
    }
    return bufflocnow;
}

/* ref src/sdrtrk.c:64-76 */
void orc_cumsumcorr(orc_chan_t *ch, int polarity)
{
    int i;
    for (i = 0; i < 1 + 2 * ch->corrn; i++) {
        ch->II[i] *= polarity;
        ch->QQ[i] *= polarity;
        ch->oldsumI[i] += ch->oldI[i];
        ch->oldsumQ[i] += ch->oldQ[i];
        ch->sumI[i] += ch->II[i];
        ch->sumQ[i] += ch->QQ[i];
    }
}

/* ref src/sdrtrk.c:77-86 */
void orc_clearcumsumcorr(orc_chan_t *ch)
{
    int i;
    for (i = 0; i < 1 + 2 * ch->corrn; i++)
        ch->oldsumI[i] = ch->oldsumQ[i] = ch->sumI[i] = ch->sumQ[i] = 0;
}

/* ref src/sdrtrk.c:95-126 */
void orc_pll(orc_chan_t *ch, int prm, double dt)
{
    double IP = ch->sumI[0], QP = ch->sumQ[0];
    double oldIP = ch->oldsumI[0], oldQP = ch->oldsumQ[0];
    double carrErr, freqErr, f1, f2;
    if (IP > 0) carrErr = atan2(QP, IP) / ORC_PI;
    else        carrErr = atan2(-QP, -IP) / ORC_PI;
    f1 = (IP == 0)    ? ORC_PI / 2 : atan(QP / IP);
    f2 = (oldIP == 0) ? ORC_PI / 2 : atan(oldQP / oldIP);
    freqErr = f1 - f2;
    if (freqErr >  ORC_PI / 2) freqErr =  ORC_PI - freqErr;
    if (freqErr < -ORC_PI / 2) freqErr = -ORC_PI - freqErr;
    ch->carrNco += ch->pllaw[prm] * (carrErr - ch->carrErr) +
                   ch->pllw2[prm] * dt * carrErr + ch->fllw[prm] * dt * freqErr;
    ch->carrfreq = ch->acq.acqfreq + ch->carrNco;
    ch->carrErr = carrErr;
    ch->freqErr = freqErr;
}

/* ref src/sdrtrk.c:135-150 */
void orc_dll(orc_chan_t *ch, int prm, double dt)
{
    double IE = ch->sumI[ch->ne], IL = ch->sumI[ch->nl];
    double QE = ch->sumQ[ch->ne], QL = ch->sumQ[ch->nl];
    double codeErr = (sqrt(IE * IE + QE * QE) - sqrt(IL * IL + QL * QL)) /
                     (sqrt(IE * IE + QE * QE) + sqrt(IL * IL + QL * QL));
    ch->codeNco += ch->dllaw[prm] * (codeErr - ch->codeErr) +
                   ch->dllw2[prm] * dt * codeErr;
    ch->codefreq = ch->crate - ch->codeNco +
                   (ch->carrfreq - ch->f_if - ch->foffset) / (ch->f_cf / ch->crate);
    ch->codeErr = codeErr;
}

/* maxvi(), ref src/sdrcmn.c:407-422 */
static int orc_maxvi(const int *data, int n, int exinds, int exinde, int *ind)
{
    int i, max = data[0];
    *ind = 0;
    for (i = 1; i < n; i++) {
        if ((exinds <= exinde && (i < exinds || i > exinde)) || (exinds > exinde && (i < exinds && i > exinde))) {
            if (max < data[i]) {
                max = data[i];
                *ind = i;
            }
        }
    }
    return max;
}

#define ORC_NAVSYNCTH 50    /* ref src/sdr.h:157 */

/* checksync(), ref src/sdrnav.c:198-233.  nav->sdreph.prn is the channel's PRN (ref src/sdrinit.c:506): every PRN
 * above 5 takes the shift-register branch; nav->ocode is all ones for L1CA / SBAS / G1 (ref src/sdrinit.c:520-521,
 * 542-543, 557-558). */
int orc_checksync(double IP, double IPold, orc_chan_t *ch)
{
    int i, corr = 0, maxi;
    if (ch->prn > 5) {
        /* shiftdata(&bitsync[0], &bitsync[1], sizeof(int), rate-1): ref src/sdrcmn.c:587-596 */
        memmove(&ch->bitsync[0], &ch->bitsync[1], sizeof(int) * (size_t)(ch->rate - 1));
        ch->bitsync[ch->rate - 1] = (IP < 0 ? -1 : 1);
        for (i = 0; i < ch->rate; i++) corr += 1 * ch->bitsync[i];
        if (abs(corr) == ch->rate) {
            ch->synci = ch->biti;
            return 1;
        }
    } else {
        if (IPold * IP < 0) {
            ch->bitsync[ch->biti] += 1;
            maxi = orc_maxvi(ch->bitsync, ch->rate, -1, -1, &ch->synci);
            if (maxi > ORC_NAVSYNCTH) {
                ch->synci--;
                if (ch->synci < 0) ch->synci = ch->rate - 1;
                return 1;
            }
        }
    }
    return 0;
}

/* checkbit(), ref src/sdrnav.c:241-282; nav->cnt is navcnt here.  The frame bit buffer (fbits, :272-275) belongs
 * to the frame decoder, which is outside the path: the decided bit is left in ch->bit / ch->swsync. */
int orc_checkbit(double IP, int loopms, orc_chan_t *ch)
{
    int diffi = ch->biti - ch->synci, syncflag = 1, polarity = 1;
    ch->swreset = 0;
    ch->swsync = 0;
    if (diffi == 1 || diffi == -ch->rate + 1) {
        ch->bitIP = IP;
        ch->swreset = 1;
        ch->navcnt = 1;
    } else {
        ch->bitIP += IP;
        if (ch->bitIP * IP < 0) syncflag = 0;
    }
    if (ch->navcnt % loopms == 0) ch->swloop = 1;
    else ch->swloop = 0;
    if (diffi == 0) {
        if (ch->flagpol) polarity = -1;
        else polarity = 1;
        ch->bit = (ch->bitIP < 0) ? -polarity : polarity;
        ch->swsync = 1;
    }
    ch->navcnt++;
    return syncflag;
}

/* sdrnavigation() up to the frame decoder, ref src/sdrnav.c:18-36 */
void orc_sdrnavigation_sync(orc_chan_t *ch, uint64_t cnt)
{
    ch->biti = (int)(cnt % (uint64_t)ch->rate);
    if (ch->rate == 1 && cnt > 2000 / (ch->ctime * 1000)) {
        ch->synci = 0;
        ch->flagsync = 1;
    }
    if (!ch->flagsync && cnt > 2000 / (ch->ctime * 1000))
        ch->flagsync = orc_checksync(ch->II[0], ch->oldI[0], ch);
    if (ch->flagsync) orc_checkbit(ch->II[0], ch->loopms, ch);
}

/* ref src/sdrmain.c:264-312; sdrnavigation() is called from sdrtracking() after the correlator (src/sdrtrk.c:46);
 * nav.ocode is all ones (src/sdrinit.c:520-521) */
int orc_sdrthread_step(orc_chan_t *ch, const orc_ring_t *ring, uint64_t *buffloc)
{
    orc_sdrtracking(ch, ring, *buffloc);
    if (!ch->flagtrk) return 0;
    orc_sdrnavigation_sync(ch, ch->cnt);
    orc_cumsumcorr(ch, 1);
    ch->flagloopfilter = 0;
    if (!ch->flagsync) {
        orc_pll(ch, 0, ch->ctime);
        orc_dll(ch, 0, ch->ctime);
        ch->flagloopfilter = 1;
    } else if (ch->swloop) {
        orc_pll(ch, 1, (double)ch->loopms / 1000);
        orc_dll(ch, 1, (double)ch->loopms / 1000);
        ch->flagloopfilter = 2;
        /* ref src/sdrmain.c:283-288: SNSMOOTHMS = 100 (src/sdr.h:198) */
        orc_setobsdata(ch, *buffloc, ch->cnt, ch->loopcnt % (uint64_t)(100 / ch->loopms) == 0 ? 1 : 0);
        ch->loopcnt++;
    }
    if (ch->flagloopfilter) orc_clearcumsumcorr(ch);
    ch->cnt++;
    *buffloc += (uint64_t)ch->currnsamp;
    return 1;
}

/* ref src/sdrnav_gps.c:141-164 */
static int orc_paritycheck_l1ca(const int *bits)
{
    int i, stat = 0, pbits[6];
    pbits[0] = bits[0] * bits[2] * bits[3] * bits[4] * bits[6] * bits[7] * bits[11] * bits[12] *
        bits[13] * bits[14] * bits[15] * bits[18] * bits[19] * bits[21] * bits[24];
    pbits[1] = bits[1] * bits[3] * bits[4] * bits[5] * bits[7] * bits[8] * bits[12] * bits[13] *
        bits[14] * bits[15] * bits[16] * bits[19] * bits[20] * bits[22] * bits[25];
    pbits[2] = bits[0] * bits[2] * bits[4] * bits[5] * bits[6] * bits[8] * bits[9] * bits[13] *
        bits[14] * bits[15] * bits[16] * bits[17] * bits[20] * bits[21] * bits[23];
    pbits[3] = bits[1] * bits[3] * bits[5] * bits[6] * bits[7] * bits[9] * bits[10] * bits[14] *
        bits[15] * bits[16] * bits[17] * bits[18] * bits[21] * bits[22] * bits[24];
    pbits[4] = bits[1] * bits[2] * bits[4] * bits[6] * bits[7] * bits[8] * bits[10] * bits[11] *
        bits[15] * bits[16] * bits[17] * bits[18] * bits[19] * bits[22] * bits[23] * bits[25];
    pbits[5] = bits[0] * bits[4] * bits[6] * bits[7] * bits[9] * bits[10] * bits[11] * bits[12] *
        bits[14] * bits[16] * bits[20] * bits[23] * bits[24] * bits[25];
    for (i = 0; i < 6; i++) stat += (pbits[i] - bits[26 + i]);
    return stat == 0;
}

/* ref src/sdrnav.c:325-346 (L1CA branch) */
static int orc_paritycheck(const orc_frame_t *f)
{
    int i, j, stat = 0, bits[302];
    for (i = 0; i < 302; i++) bits[i] = f->polarity * f->fbitsdec[i];
    for (i = 0; i < 10; i++) {
        if (bits[i * 30 + 1] == -1)
            for (j = 2; j < 26; j++) bits[i * 30 + j] *= -1;
        stat += orc_paritycheck_l1ca(&bits[i * 30]);
    }
    return stat == 10;
}

/* ref src/sdrnav.c:373-411 (L1CA branch); pre_l1ca: src/sdrinit.c:492 */
static int orc_findpreamble(orc_frame_t *f)
{
    static const int pre[8] = {1, -1, -1, -1, 1, -1, 1, 1};
    int i, corr = 0;
    for (i = 0; i < 8; i++) corr += f->fbitsdec[2 + i] * pre[i];
    if (abs(corr) == 8) {
        f->polarity = corr > 0 ? 1 : -1;
        if (orc_paritycheck(f)) return 1;
    }
    return 0;
}

/* ref src/sdrnav_gps.c:170-190, :123-135, :18 (tow) with bits2byte (src/sdrnav.c:154-171: -1 => 1) and RTKLIB's
 * getbitu (most significant bit first) */
static int orc_decode_l1ca(orc_frame_t *f)
{
    int i, j, id = 0;
    unsigned tow = 0;
    for (i = 0; i < 10; i++)
        if (f->fbitsdec[i * 30 + 1] == -1)
            for (j = 2; j < 26; j++) f->fbitsdec[i * 30 + j] *= -1;
    for (i = 0; i < 3; i++) id = (id << 1) | (f->fbitsdec[2 + 49 + i] < 0 ? 1 : 0);
    for (i = 0; i < 17; i++) tow = (tow << 1) | (f->fbitsdec[2 + 30 + i] < 0 ? 1u : 0u);
    if (id >= 1 && id <= 5) f->tow_gpst = tow * 6.0;      /* (decode_frame_l1ca decodes subframes 1..5 only) */
    return id;
}

/* ref src/sdrnav.c:41-82 */
void orc_navframe_l1ca(orc_frame_t *f, int bit, uint64_t buffloc, uint64_t cnt)
{
    int i;
    if (!bit) return;                                   /* (swsync off: nothing behind checkbit() runs) */
    for (i = 0; i + 1 < 302; i++) f->fbits[i] = f->fbits[i + 1];       /* checkbit(), ref :274-277 */
    f->fbits[301] = bit;
    if (!f->flagtow) memcpy(f->fbitsdec, f->fbits, sizeof(f->fbits));   /* predecodefec */
    if (!f->flagtow) f->flagsyncf = orc_findpreamble(f);
    if (f->flagsyncf && !f->flagtow) {
        f->firstsf = buffloc;
        f->firstsfcnt = cnt;
        f->flagtow = 1;
    }
    if (f->flagtow) {
        if ((int)(cnt - f->firstsfcnt) % 6000 == 0) {   /* nav->update = flen * rate = 300 * 20 */
            memcpy(f->fbitsdec, f->fbits, sizeof(f->fbits));
            f->sfid = orc_decode_l1ca(f);
            if (f->tow_gpst == 0) {
                f->flagsyncf = 0;
                f->flagtow = 0;
            } else if (cnt - f->firstsfcnt == 0) {
                f->flagdec = 1;
                f->firstsftow = f->tow_gpst;
            }
        }
    }
}

/* ref src/sdrtrk.c:160-209 */
void orc_setobsdata(orc_chan_t *ch, uint64_t buffloc, uint64_t cnt, int snrflag)
{
    const double dpi = 2.0 * 3.1415926535897932;                     /* DPI, ref src/sdr.h:103-104 */
    ch->obs_tow = ch->firstsftow + (double)(cnt - ch->firstsfcnt) * ch->ctime;       /* :170-171 */
    ch->obs_codei = buffloc;
    ch->obs_cntout = cnt;
    ch->obs_remcout = ch->oldremcode * ch->f_sf / ch->codefreq;     /* :174 */
    ch->obs_D = -(ch->carrfreq - ch->f_if - ch->foffset);           /* :177 */
    if (!ch->flagremcarradd) {                                      /* :180-184 */
        ch->obs_L -= ch->remcarr / dpi;
        ch->flagremcarradd = 1;
    }
    if (ch->flagsyncf && !ch->flagpolarityadd) {                    /* :186-194 */
        if (ch->polarity == 1) ch->obs_L += 0.5;
        ch->flagpolarityadd = 1;
    }
    ch->obs_L += ch->obs_D * (ch->loopms * ch->currnsamp / ch->f_sf);       /* :196 */
    ch->obs_Isum += fabs(ch->sumI[0]);                              /* :198 */
    if (snrflag) {                                                  /* :199-208 */
        ch->obs_S = 10 * log(ch->obs_Isum / 100.0 / 100.0) + log(500.0) + 5;
        ch->obs_codeisum = buffloc;
        ch->obs_Isum = 0;
        ch->obs_nsnr++;
    }
    ch->obs_n++;
}

/* ref src/rcv/stereo/stereo.c:160-205 */
void orc_stereo_exp(const unsigned char *buf, int n, int dtype, signed char *expbuf)
{
    static const signed char base1[4] = {-3, -1, +1, +3};                  /* 2 bits */
    static const signed char base2[8] = {+1, +3, +5, +7, -7, -5, -3, -1};  /* 3 bits */
    int i;
    if (dtype == 1) {
        for (i = 0; i < n; i++) expbuf[i] = base1[(buf[i] >> 6) & 0x03];
    } else {
        for (i = 0; i < n; i++) {
            expbuf[2 * i]     = base2[(buf[i] >> 3) & 0x07];
            expbuf[2 * i + 1] = base2[buf[i] & 0x07];
        }
    }
}

/* ref src/rcv/rtlsdr/rtlsdr.c:136-143 */
void orc_rtlsdr_exp(const unsigned char *buf, int n, signed char *expbuf)
{
    int i;
    for (i = 0; i < n; i++) expbuf[i] = (signed char)((buf[i] - 127.5));
}
