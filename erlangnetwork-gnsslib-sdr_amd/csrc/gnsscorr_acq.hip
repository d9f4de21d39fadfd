// gnsscorr_acq.hip -- parallel code phase acquisition for gfx950 (MI355X).
//
// Replaces sdracquisition()'s loop (ref src/sdracq.c:29-43): per iteration
// pcorrelator() (ref src/sdrcmn.c:738-773) = mixcarr + cpxcpx + cpxconv per
// Doppler bin, accumulated into P, then checkacquisition() (ref
// src/sdracq.c:71-95).
//
// The reference's FFT length m = 2*nsamp (32736) is replaced by L = 32768:
// the replica has only nsamp non-zero samples and the data window 2*nsamp, so
// lags 0..nsamp-1 never wrap for any L >= 2*nsamp-1 and the correlation values
// are the same numbers (hard part 3).  A length-32768 transform is
// done as two LDS-resident 16384-point transforms (gnsscorr_fft.h) plus one
// radix-2 stage; spectra stay in the FFT's pass order (no reordering anywhere):
//
//   acq_fwd  (per Doppler bin, iteration; shared by all SVs of a grid):
//            carrier wipe-off of the 2*nsamp window straight from the HBM
//            ring, FFT of the even and of the odd samples, radix-2 combine,
//            spectrum stored split by parity of the frequency index.
//   acq_code (per channel, once): same for the zero-padded +-1 replica.
//   acq_corr (per channel x Doppler bin): for every iteration, X*conj(C) on
//            load, inverse FFT of the even-index and odd-index halves,
//            y[k] = E[k] + w^-k O[k], |y|^2/L^2 added to fp64 accumulators that
//            live in registers across the iterations; after each iteration
//            the row statistics checkacquisition() needs are reduced in the
//            workgroup.  P only leaves the chip when the caller asks for it.
//   acq_final (per channel): the decision of checkacquisition() after each
//            iteration, first success wins (ref src/sdracq.c:39-42).
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "gnsscorr_ctx.h"
#include "gnsscorr_fft.h"

#define GC_L        32768
#define GC_LH       16384

using gcfft::cadd;
using gcfft::cmul;
using gcfft::cmulc;
using gcfft::csub;

// Carrier NCO of one Doppler bin over the 2*nsamp acquisition window, phase 0 at its first sample
// (ref src/sdrcmn.c:761): the reference's running sum as a piece table (gnsscorr_nco.h)
struct GcAcqCar {
    int n, pad;
    int k0[GC_NCAR];
    GcCarSeg seg[GC_NCAR];
};

#define GC_ACQ_CARLDS (((GC_NCAR * 4 + 15) & ~15) + GC_NCAR * (int)sizeof(GcCarSeg))

struct GcAcqWork {
    GcAcqCar *car = nullptr;    // [grid][bin]
    int *car_overflow = nullptr;
    float2 *tw16k = nullptr;    // exp(-2 pi i t/16384), t < 16384
    float2 *tw32k = nullptr;    // exp(-2 pi i t/32768), t < 16384
    float2 *tw32p = nullptr;    // the same twiddles in pass order: exp(-2 pi i freq_of(p)/32768), p < 16384
    float2 *tw64p1 = nullptr;   // exp(-2 pi i freq_of(p)/65536) and
    float2 *tw64p3 = nullptr;   // exp(-2 pi i 3 freq_of(p)/65536), pass order (the 65536-point path)
    int L = GC_L;               // transform length of this channel set: 32768, or 65536 when a period exceeds 16384 samples
    float2 *X = nullptr;        // [grid][iter][bin][2][16384]: X[f] and X[f + 16384] at the pass position of f
    size_t  X_elems = 0;
    float2 *C = nullptr;        // [ch][2][16384]
    GcAcqRow *rows = nullptr;   // [ch][iter][bin]
    int *arrive = nullptr;      // [ch][iter] workgroups done with the iteration, then [ch] "acquired" flags
    int *iters = nullptr;       // [ch] iteration limit for acq_corr
    gnsscorr_acqres_t *res = nullptr;   // [ch]
    double *P = nullptr;        // one channel's power array (on demand)
    size_t P_elems = 0;
    int ngrid = 0, maxfreq = 0, maxintg = 0;
    std::vector<int> grid_chan;         // a representative channel per grid
    int *d_grid_chan = nullptr;         // device copy of grid_chan
    uint64_t *d_grid_wrpos = nullptr;   // ring write position seen by each grid
    bool code_ready = false;
    uint64_t last_wrpos[2] = {0, 0};
    bool ran = false;
};

namespace {

// carrier LUT (ref src/sdrcmn.c:643-648)
__constant__ signed char aCos32[32] = {32, 31, 30, 27, 23, 18, 12, 6, 0, -6, -12, -18, -23, -27, -30, -31,
                                       -32, -31, -30, -27, -23, -18, -12, -6, 0, 6, 12, 18, 23, 27, 30, 31};
__constant__ signed char aSin32[32] = {0, 6, 12, 18, 23, 27, 30, 31, 32, 31, 30, 27, 23, 18, 12, 6,
                                       0, -6, -12, -18, -23, -27, -30, -31, -32, -31, -30, -27, -23, -18, -12, -6};

__global__ void tw64_init_kernel(float2 *tw64p1, float2 *tw64p3)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= GC_LH) return;
    double s, c;
    sincospi(-2.0 * (double)gcfft::freq_of(t) / 65536.0, &s, &c);
    tw64p1[t] = make_float2((float)c, (float)s);
    sincospi(-2.0 * 3.0 * (double)gcfft::freq_of(t) / 65536.0, &s, &c);
    tw64p3[t] = make_float2((float)c, (float)s);
}

__global__ void tw_init_kernel(float2 *tw16k, float2 *tw32k, float2 *tw32p)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= GC_LH) return;
    double s, c;
    sincospi(-2.0 * (double)t / 16384.0, &s, &c);
    tw16k[t] = make_float2((float)c, (float)s);
    sincospi(-2.0 * (double)t / 32768.0, &s, &c);
    tw32k[t] = make_float2((float)c, (float)s);
    sincospi(-2.0 * (double)gcfft::freq_of(t) / 32768.0, &s, &c);
    tw32p[t] = make_float2((float)c, (float)s);
}

// Forward 32768-point transform of a sequence given by a per-sample functor, decimated in time:
// Ee = FFT16k(x[2j]), Oo = FFT16k(x[2j+1]), X[f] = Ee[f] + w^f Oo[f], X[f + 16384] = Ee[f] - w^f Oo[f]
// (w = exp(-2 pi i/32768), f < 16384).  out[0][p] / out[1][p] = X[f(p)] / X[f(p) + 16384], p in pass
// order: exactly the pairs the inverse transform of the correlation wants side by side (acq_corr).
// Ee waits in out[0] while Oo is computed (each lane reads back what it wrote itself).
template <class F>
__device__ __forceinline__ void fwd32k_store(F sample, float2 *lds, const float2 *__restrict__ tw16k,
                                             const float2 *__restrict__ tw32p, float2 *__restrict__ out,
                                             int tid)
{
    gcfft::dif<-1>([&](int j) { return sample(2 * j); },
                   [out](int p, float2 x0, float2 x1, float2 x2, float2 x3) {
                       float4 *d = reinterpret_cast<float4 *>(out + p);
                       d[0] = make_float4(x0.x, x0.y, x1.x, x1.y);
                       d[1] = make_float4(x2.x, x2.y, x3.x, x3.y);
                   },
                   lds, tw16k, tid);
    __syncthreads();
    gcfft::dif<-1>([&](int j) { return sample(2 * j + 1); },
                   [out, tw32p](int p, float2 x0, float2 x1, float2 x2, float2 x3) {
                       float4 *lo = reinterpret_cast<float4 *>(out + p);
                       float4 *hi = reinterpret_cast<float4 *>(out + GC_LH + p);
                       const float4 ea = lo[0], eb = lo[1];
                       const float4 ta = *reinterpret_cast<const float4 *>(tw32p + p);
                       const float4 tb = *reinterpret_cast<const float4 *>(tw32p + p + 2);
                       const float2 o0 = cmul(x0, make_float2(ta.x, ta.y)), o1 = cmul(x1, make_float2(ta.z, ta.w));
                       const float2 o2 = cmul(x2, make_float2(tb.x, tb.y)), o3 = cmul(x3, make_float2(tb.z, tb.w));
                       lo[0] = make_float4(ea.x + o0.x, ea.y + o0.y, ea.z + o1.x, ea.w + o1.y);
                       lo[1] = make_float4(eb.x + o2.x, eb.y + o2.y, eb.z + o3.x, eb.w + o3.y);
                       hi[0] = make_float4(ea.x - o0.x, ea.y - o0.y, ea.z - o1.x, ea.w - o1.y);
                       hi[1] = make_float4(eb.x - o2.x, eb.y - o2.y, eb.z - o3.x, eb.w - o3.y);
                   },
                   lds, tw16k, tid);
}

// Forward 65536-point transform (code periods of 16385..32768 samples: 20 / 26 Msps front ends, ref
// frontend/stereo_L1G1.ini), decimated in time by four: E_r = FFT16k(x[4j + r]),
//     X[f + 16384 q] = sum_r (-i)^(r q) w^(r f) E_r[f],   w = exp(-2 pi i/65536), f < 16384,
// stored as out[q][p] (p = pass position of f): the four values the inverse transform wants side by side.
// With a_r = w^(r f) E_r:  P+- = E_0 +- a_2, Q+- = a_1 +- a_3,
//     X_0 = P+ + Q+,  X_2 = P+ - Q+,  X_1 = P- - i Q-,  X_3 = P- + i Q-.
// The partial results wait in `out` between the four sub-transforms (each lane reads back what it wrote).
template <class F>
__device__ __forceinline__ void fwd64k_store(F sample, float2 *lds, const float2 *__restrict__ tw16k,
                                             const float2 *__restrict__ tw32p, const float2 *__restrict__ tw64p1,
                                             const float2 *__restrict__ tw64p3, float2 *__restrict__ out, int tid)
{
    auto put = [](float2 *dst, float2 x0, float2 x1, float2 x2, float2 x3) {
        float4 *d = reinterpret_cast<float4 *>(dst);
        d[0] = make_float4(x0.x, x0.y, x1.x, x1.y);
        d[1] = make_float4(x2.x, x2.y, x3.x, x3.y);
    };
    auto get = [](const float2 *src, float2 (&v)[4]) {
        const float4 *d = reinterpret_cast<const float4 *>(src);
        const float4 a = d[0], b = d[1];
        v[0] = make_float2(a.x, a.y); v[1] = make_float2(a.z, a.w);
        v[2] = make_float2(b.x, b.y); v[3] = make_float2(b.z, b.w);
    };
    // r = 0: E_0 -> out[0]
    gcfft::dif<-1>([&](int j) { return sample(4 * j); },
                   [&](int p, float2 x0, float2 x1, float2 x2, float2 x3) { put(out + p, x0, x1, x2, x3); },
                   lds, tw16k, tid);
    __syncthreads();
    // r = 2: P+ -> out[0], P- -> out[1]
    gcfft::dif<-1>([&](int j) { return sample(4 * j + 2); },
                   [&](int p, float2 x0, float2 x1, float2 x2, float2 x3) {
                       float2 e[4], t[4];
                       get(out + p, e);
                       get(tw32p + p, t);
                       const float2 a[4] = {cmul(x0, t[0]), cmul(x1, t[1]), cmul(x2, t[2]), cmul(x3, t[3])};
                       put(out + p, cadd(e[0], a[0]), cadd(e[1], a[1]), cadd(e[2], a[2]), cadd(e[3], a[3]));
                       put(out + GC_LH + p, csub(e[0], a[0]), csub(e[1], a[1]), csub(e[2], a[2]), csub(e[3], a[3]));
                   },
                   lds, tw16k, tid);
    __syncthreads();
    // r = 1: a_1 -> out[2]
    gcfft::dif<-1>([&](int j) { return sample(4 * j + 1); },
                   [&](int p, float2 x0, float2 x1, float2 x2, float2 x3) {
                       float2 t[4];
                       get(tw64p1 + p, t);
                       put(out + 2 * GC_LH + p, cmul(x0, t[0]), cmul(x1, t[1]), cmul(x2, t[2]), cmul(x3, t[3]));
                   },
                   lds, tw16k, tid);
    __syncthreads();
    // r = 3: the four outputs
    gcfft::dif<-1>([&](int j) { return sample(4 * j + 3); },
                   [&](int p, float2 x0, float2 x1, float2 x2, float2 x3) {
                       float2 t[4], a1[4], pp[4], pm[4];
                       get(tw64p3 + p, t);
                       get(out + 2 * GC_LH + p, a1);
                       get(out + p, pp);
                       get(out + GC_LH + p, pm);
                       const float2 a3[4] = {cmul(x0, t[0]), cmul(x1, t[1]), cmul(x2, t[2]), cmul(x3, t[3])};
                       float2 X0[4], X1[4], X2[4], X3[4];
#pragma unroll
                       for (int i = 0; i < 4; i++) {
                           const float2 qp = cadd(a1[i], a3[i]), qm = csub(a1[i], a3[i]);
                           const float2 iqm = make_float2(-qm.y, qm.x);            // i Q-
                           X0[i] = cadd(pp[i], qp);
                           X2[i] = csub(pp[i], qp);
                           X1[i] = csub(pm[i], iqm);
                           X3[i] = cadd(pm[i], iqm);
                       }
                       put(out + p, X0[0], X0[1], X0[2], X0[3]);
                       put(out + GC_LH + p, X1[0], X1[1], X1[2], X1[3]);
                       put(out + 2 * GC_LH + p, X2[0], X2[1], X2[2], X2[3]);
                       put(out + 3 * GC_LH + p, X3[0], X3[1], X3[2], X3[3]);
                   },
                   lds, tw16k, tid);
}

// one lane per (grid, bin): the bin's carrier piece table
__global__ void acq_nco_kernel(const GcChan *__restrict__ chan, const int *__restrict__ grid_chan,
                               const double *__restrict__ freqs, GcAcqCar *__restrict__ car, int ngrid, int maxfreq,
                               int *__restrict__ overflow)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ngrid * maxfreq) return;
    const GcChan &c = chan[grid_chan[i / maxfreq]];
    const int bin = i % maxfreq;
    GcAcqCar *t = car + i;
    t->n = 0;
    if (bin >= c.nfreq) return;
    GcCarTable ct{t->k0, t->seg, GC_NCAR, 0, 0};
    gc_carrier_walk(gc_carrier_phis(0.0), gc_carrier_ps(freqs[c.freq_off + bin], c.ti), 2 * c.nsamp, ct);
    t->n = ct.n;
    if (ct.overflow) atomicAdd(overflow, 1);
}

// acq_fwd: grid (bin, iteration, grid group)
__global__ __launch_bounds__(GC_FFT_THREADS) void acq_fwd_kernel(
    const GcChan *__restrict__ chan, const int *__restrict__ grid_chan, const GcAcqCar *__restrict__ car,
    const uint64_t *__restrict__ grid_wrpos, const float2 *__restrict__ tw16k,
    const float2 *__restrict__ tw32p, const float2 *__restrict__ tw64p1, const float2 *__restrict__ tw64p3,
    float2 *__restrict__ X, int maxfreq, int maxintg, int L)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    const int bin = blockIdx.x, it = blockIdx.y, g = blockIdx.z, tid = threadIdx.x;
    const GcChan &c = chan[grid_chan[g]];
    if (bin >= c.nfreq || it >= c.intg) return;
    const int n = c.nsamp, n2 = 2 * n, dtype = c.dtype;
    // window of iteration `it`: ref src/sdracq.c:24-32
    const uint64_t buffloc = grid_wrpos[g] - (uint64_t)(c.intg + 1) * n + (uint64_t)it * n;
    const uint64_t base = buffloc % c.ringlen;
    const gc_gptr_i8 ring = (gc_gptr_i8)c.ring;
    const uint64_t ringlen = c.ringlen;
    // the bin's carrier table (phase starts at 0 for every bin and iteration, :761) behind the FFT image
    int *lk0 = reinterpret_cast<int *>(smem + GC_FFT_LDS + 256);
    GcCarSeg *lseg = reinterpret_cast<GcCarSeg *>(smem + GC_FFT_LDS + 256 + ((GC_NCAR * 4 + 15) & ~15));
    const GcAcqCar *gt = car + (size_t)g * maxfreq + bin;
    const int ncar = gt->n;
    if (tid < ncar) { lk0[tid] = gt->k0[tid]; lseg[tid] = gt->seg[tid]; }
    __syncthreads();
    const float sc = (float)((1.0 / 32.0) / (double)c.nfft);     // CSCALE/m, ref src/sdrcmn.c:764

    auto sample = [&](int s) -> float2 {
        if (s >= n2) return make_float2(0.f, 0.f);
        uint64_t pos = base + (uint64_t)s;
        if (pos >= ringlen) pos -= ringlen;
        const int idx = gc_carrier_idx_at(lk0, lseg, ncar, s);
        const int cs_ = aCos32[idx], sn_ = aSin32[idx];
        int I, Q;
        if (dtype == 2) {
            const int d0 = ring[2 * pos], d1 = ring[2 * pos + 1];
            I = cs_ * d0 - sn_ * d1;
            Q = sn_ * d0 + cs_ * d1;
        } else {
            const int d0 = ring[pos];
            I = cs_ * d0;
            Q = sn_ * d0;
        }
        return make_float2((float)I * sc, (float)Q * sc);
    };
    float2 *out = X + (((size_t)g * maxintg + it) * maxfreq + bin) * (size_t)L;
    if (L == 2 * GC_L) fwd64k_store(sample, lds, tw16k, tw32p, tw64p1, tw64p3, out, tid);
    else fwd32k_store(sample, lds, tw16k, tw32p, out, tid);
}

// acq_code: grid (channel)
__global__ __launch_bounds__(GC_FFT_THREADS) void acq_code_kernel(const GcChan *__restrict__ chan,
                                                                  const float2 *__restrict__ tw16k,
                                                                  const float2 *__restrict__ tw32p,
                                                                  const float2 *__restrict__ tw64p1,
                                                                  const float2 *__restrict__ tw64p3,
                                                                  float2 *__restrict__ C, int L)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    const int ch = blockIdx.x, tid = threadIdx.x;
    const GcChan &c = chan[ch];
    const int n = c.nsamp, clen = c.clen;
    const double ci = __dmul_rn(c.ti, c.crate);      // sdr->ci, ref src/sdrinit.c:605
    const gc_gptr_i8 code = (gc_gptr_i8)c.code;
    // rescode(code, clen, 0, 0, ci, nsamp) zero-padded (ref src/sdrinit.c:650-651)
    auto sample = [&](int s) -> float2 {
        if (s >= n) return make_float2(0.f, 0.f);
        long long t = (long long)__dmul_rn((double)s, ci);
        while (t >= clen) t -= clen;
        return make_float2((float)code[(int)t], 0.f);
    };
    if (L == 2 * GC_L) fwd64k_store(sample, lds, tw16k, tw32p, tw64p1, tw64p3, C + (size_t)ch * L, tid);
    else fwd32k_store(sample, lds, tw16k, tw32p, C + (size_t)ch * L, tid);
}

// ---- workgroup reductions used by acq_corr --------------------------------
struct MaxIdx { double v; int k; };

__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b)
{   // larger value wins, first index on ties (maxvd's strict '<', ref src/sdrcmn.c:461-476)
    return (b.v > a.v || (b.v == a.v && b.k < a.k)) ? b : a;
}

template <int NT>
__device__ __forceinline__ MaxIdx wg_argmax(MaxIdx m, double *sd, int *si, int tid)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MaxIdx t;
        t.v = __shfl_xor(m.v, o, 64);
        t.k = __shfl_xor(m.k, o, 64);
        m = better(m, t);
    }
    __syncthreads();
    if ((tid & 63) == 0) { sd[tid >> 6] = m.v; si[tid >> 6] = m.k; }
    __syncthreads();
    MaxIdx r; r.v = sd[0]; r.k = si[0];
#pragma unroll
    for (int w = 1; w < NT / 64; w++) { MaxIdx t; t.v = sd[w]; t.k = si[w]; r = better(r, t); }
    return r;
}

template <int NT>
__device__ __forceinline__ void wg_sum_max(double &s, double &m, double *sd, int tid)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        m = fmax(m, __shfl_xor(m, o, 64));
    }
    __syncthreads();
    if ((tid & 63) == 0) { sd[tid >> 6] = s; sd[16 + (tid >> 6)] = m; }
    __syncthreads();
    s = sd[0]; m = sd[16];
#pragma unroll
    for (int w = 1; w < NT / 64; w++) { s += sd[w]; m = fmax(m, sd[16 + w]); }
}

struct RawXC { float4 la, lb, ha, hb, cla, clb, cha, chb; };     // X[f], X[f+L/2], C[f], C[f+L/2] at 4 positions
struct RawXCT { RawXC r; float4 ta, tb; };                        // + their 32768-point twiddles

// acq_corr: one workgroup of NT lanes per (bin, channel).
// y = IFFT_L(Y), Y = X conj(C) (ref src/sdrcmn.c:236-246; the reference's extra minus sign vanishes
// under |.|^2), split by lag parity so that each 16384-point transform ends in final values:
//   y[2j]   = IFFT_16k( Y[f] + Y[f + L/2] )[j]
//   y[2j+1] = IFFT_16k( (Y[f] - Y[f + L/2]) exp(+2 pi i f/L) )[j],   f < L/2.
// Lags k < nsamp <= 16384 are wanted, i.e. j < 8192: of a lane's outputs j = o + 1024 q those with
// q < 8.  Nothing has to wait for the other transform, so the only long-lived registers are the
// power accumulators.
#define GC_ACQ_G 4              // channels that share an XCD's forward spectra (acq_corr_kernel)
template <int NT>
__global__ __launch_bounds__(NT) void acq_corr_kernel(
    const GcChan *__restrict__ chan, const float2 *__restrict__ tw16k, const float2 *__restrict__ tw32p,
    const float2 *__restrict__ X, const float2 *__restrict__ C, const int *__restrict__ iters,
    GcAcqRow *rows, double *__restrict__ Pout, int pout_ch, int maxfreq, int maxintg, int nchg,
    int *arrive, int *done)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    double *sd = reinterpret_cast<double *>(smem + GC_FFT_LDS);       // 32 doubles
    int *si = reinterpret_cast<int *>(smem + GC_FFT_LDS + 256);       // 16 ints
    // Workgroup order: the Doppler bins of one channel are consecutive blocks, so they run side by side (71
    // bins on 256 CUs: three to four channels at a time) and reach the end of each iteration together --
    // what lets the channel stop at the iteration that acquires it (below).  The channel's code spectrum
    // (256 KB) is then held by every XCD's L2, the forward spectra stream from the Infinity Cache, which
    // holds all of them (186 MB).  (Pout: one channel's power array, blocks = its bins.)
    const int tid0 = threadIdx.x;
    const int slot = blockIdx.x & 7, qq = blockIdx.x >> 3;
    // Search: blocks b and b + 8 tend to share an XCD (speed only).  XCD `slot` takes the bins slot, slot + 8, ... of
    // GC_ACQ_G channels at a time: the ~9 spectra X[iteration][bin] it needs per iteration (2.3 MB) and the G code
    // spectra (1 MB) fit its 4 MB L2, so a spectrum leaves the Infinity Cache once per G channels (and both lag
    // parities) instead of once per channel and parity; chip-wide the G channels' bins all run side by side, which is
    // what lets a channel stop at the iteration that acquires it.
    constexpr int G = GC_ACQ_G;
    const int bpx = (maxfreq + 7) >> 3;                 // bins per XCD slot
    const int grp = qq / (bpx * G), rem = qq - grp * (bpx * G);
    const int bin = Pout ? qq * 8 + slot : (rem / G) * 8 + slot;
    const int ch = Pout ? pout_ch : grp * G + rem % G;
    if (bin >= maxfreq || (!Pout && ch >= nchg)) return;
    const GcChan &c = chan[ch];
    if (bin >= c.nfreq) return;
    const int n = c.nsamp, nit = iters[ch], nsc2 = 2 * c.nsampchip;
    const float2 *Cc = C + (size_t)ch * GC_L;
    const float invL2 = 1.0f / ((float)GC_L * (float)GC_L);

    // lane `tid` owns lags k = 2 (tid + NT*h + 1024*q) + par (par < 2, h < 1024/NT, q < 8):
    // register index s = NPH*par + 8*h + q
    constexpr int NPH = 8 * (1024 / NT), NP = 2 * NPH;
    constexpr int CHUNK = 2;                       // butterflies whose operands are in flight together
    double P[NP];
#pragma unroll
    for (int s = 0; s < NP; s++) P[s] = 0.0;
    auto lag_of = [](int tid, int s) { return 2 * (tid + NT * ((s % NPH) >> 3) + 1024 * (s & 7)) + s / NPH; };

    for (int it = 0; it < nit; it++) {
        // Opaque copy of the lane id: keeps the address computations of one iteration from being
        // hoisted out of the loop (they would be spilled to scratch, not kept in VGPRs).
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const float2 *Xb = X + (((size_t)c.grid * maxintg + it) * maxfreq + bin) * GC_L;
        auto load8 = [Xb, Cc](int p) {
            RawXC r;
            r.la = *reinterpret_cast<const float4 *>(Xb + p);
            r.lb = *reinterpret_cast<const float4 *>(Xb + p + 2);
            r.ha = *reinterpret_cast<const float4 *>(Xb + GC_LH + p);
            r.hb = *reinterpret_cast<const float4 *>(Xb + GC_LH + p + 2);
            r.cla = *reinterpret_cast<const float4 *>(Cc + p);
            r.clb = *reinterpret_cast<const float4 *>(Cc + p + 2);
            r.cha = *reinterpret_cast<const float4 *>(Cc + GC_LH + p);
            r.chb = *reinterpret_cast<const float4 *>(Cc + GC_LH + p + 2);
            return r;
        };
        // Y[f] = X[f] conj(C[f]) at the four positions of a butterfly, low and high half of the spectrum
#define GC_YLO(r, k) cmulc(make_float2((k) < 2 ? ((k) & 1 ? (r).la.z : (r).la.x) : ((k) & 1 ? (r).lb.z : (r).lb.x), \
                                       (k) < 2 ? ((k) & 1 ? (r).la.w : (r).la.y) : ((k) & 1 ? (r).lb.w : (r).lb.y)), \
                           make_float2((k) < 2 ? ((k) & 1 ? (r).cla.z : (r).cla.x) : ((k) & 1 ? (r).clb.z : (r).clb.x), \
                                       (k) < 2 ? ((k) & 1 ? (r).cla.w : (r).cla.y) : ((k) & 1 ? (r).clb.w : (r).clb.y)))
#define GC_YHI(r, k) cmulc(make_float2((k) < 2 ? ((k) & 1 ? (r).ha.z : (r).ha.x) : ((k) & 1 ? (r).hb.z : (r).hb.x), \
                                       (k) < 2 ? ((k) & 1 ? (r).ha.w : (r).ha.y) : ((k) & 1 ? (r).hb.w : (r).hb.y)), \
                           make_float2((k) < 2 ? ((k) & 1 ? (r).cha.z : (r).cha.x) : ((k) & 1 ? (r).chb.z : (r).chb.x), \
                                       (k) < 2 ? ((k) & 1 ? (r).cha.w : (r).cha.y) : ((k) & 1 ? (r).chb.w : (r).chb.y)))
        // even lags
        gcfft::dit<+1, NT, CHUNK>(load8,
                       [&](const RawXC &r, float2 &x0, float2 &x1, float2 &x2, float2 &x3) {
                           x0 = cadd(GC_YLO(r, 0), GC_YHI(r, 0));
                           x1 = cadd(GC_YLO(r, 1), GC_YHI(r, 1));
                           x2 = cadd(GC_YLO(r, 2), GC_YHI(r, 2));
                           x3 = cadd(GC_YLO(r, 3), GC_YHI(r, 3));
                       },
                       [&](int h, int, float2 (&a)[16]) {
#pragma unroll
                           for (int q = 0; q < 8; q++) {
                               const float pw = fmaf(a[q].x, a[q].x, a[q].y * a[q].y) * invL2;
                               P[8 * h + q] += (double)pw;      // (ref :244-246 with the m-point scaling folded)
                           }
                       },
                       lds, tw16k, tid);
        __syncthreads();
        asm volatile("" : "+v"(tid));            // (as above: the second transform recomputes its addresses)
        // odd lags; exp(+2 pi i f/L) = conj of the forward twiddle
        gcfft::dit<+1, NT, CHUNK>([&](int p) {
                           RawXCT t;
                           t.r = load8(p);
                           t.ta = *reinterpret_cast<const float4 *>(tw32p + p);
                           t.tb = *reinterpret_cast<const float4 *>(tw32p + p + 2);
                           return t;
                       },
                       [&](const RawXCT &t, float2 &x0, float2 &x1, float2 &x2, float2 &x3) {
                           x0 = cmulc(csub(GC_YLO(t.r, 0), GC_YHI(t.r, 0)), make_float2(t.ta.x, t.ta.y));
                           x1 = cmulc(csub(GC_YLO(t.r, 1), GC_YHI(t.r, 1)), make_float2(t.ta.z, t.ta.w));
                           x2 = cmulc(csub(GC_YLO(t.r, 2), GC_YHI(t.r, 2)), make_float2(t.tb.x, t.tb.y));
                           x3 = cmulc(csub(GC_YLO(t.r, 3), GC_YHI(t.r, 3)), make_float2(t.tb.z, t.tb.w));
                       },
                       [&](int h, int, float2 (&a)[16]) {
#pragma unroll
                           for (int q = 0; q < 8; q++) {
                               const float pw = fmaf(a[q].x, a[q].x, a[q].y * a[q].y) * invL2;
                               P[NPH + 8 * h + q] += (double)pw;
                           }
                       },
                       lds, tw16k, tid);
#undef GC_YLO
#undef GC_YHI

        // row statistics for checkacquisition()
        MaxIdx m; m.v = -1.0; m.k = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < NP; s++) {
            const int k = lag_of(tid, s);
            if (k < n) { MaxIdx t; t.v = P[s]; t.k = k; m = better(m, t); }
        }
        m = wg_argmax<NT>(m, sd, si, tid);
        int exs = m.k - nsc2; if (exs < 0) exs += n;
        int exe = m.k + nsc2; if (exe >= n) exe -= n;
        double so = 0.0, mo = -1.0;
#pragma unroll
        for (int s = 0; s < NP; s++) {
            const int k = lag_of(tid, s);
            if (k < n) {
                const bool outside = (exs <= exe) ? (k < exs || k > exe) : (k < exs && k > exe);
                if (outside) so += P[s];
                if (outside || k == 0) mo = fmax(mo, P[s]);    // element 0 seeds maxvd()
            }
        }
        wg_sum_max<NT>(so, mo, sd, tid);
        if (tid == 0) {
            GcAcqRow r;
            r.rowmax = m.v; r.sum_out = so; r.max_out = mo; r.argmax = m.k; r.pad = 0;
            rows[((size_t)ch * maxintg + it) * maxfreq + bin] = r;
        }
        // The reference stops a channel at the first iteration whose peak ratio passes the threshold
        // (ref src/sdracq.c:39-42); so does this: the workgroup that finishes an iteration of its channel
        // last takes the decision acq_final will take from the same rows, and the channel's workgroups
        // leave at the next iteration boundary they reach after it.  (Iterations past the acquiring one
        // are never looked at: what they would have written is not missed.)
        if (arrive && tid < 64) {
            int last = 0;
            if (tid == 0) {
                __threadfence();
                last = atomicAdd(&arrive[ch * maxintg + it], 1) == c.nfreq - 1;
            }
            last = __shfl(last, 0);
            if (last) {
                __threadfence();
                const GcAcqRow *row = rows + ((size_t)ch * maxintg + it) * maxfreq;
                double bv = -1.0;
                int bi = 0x7fffffff;
                for (int b = tid; b < c.nfreq; b += 64) {
                    const double v = row[b].rowmax;
                    if (v > bv) { bv = v; bi = b; }
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const double ov = __shfl_xor(bv, d);
                    const int oi = __shfl_xor(bi, d);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                const GcAcqRow w = row[bi];
                if (tid == 0 && w.rowmax / w.max_out > 3.0)                 // ACQTH, as acq_final
                    __hip_atomic_store(&done[ch], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) si[15] = __hip_atomic_load(&done[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();      // LDS image and reduction scratch are reused next iteration
        if (arrive && si[15]) break;
    }
    if (Pout) {
#pragma unroll
        for (int s = 0; s < NP; s++) {
            const int k = lag_of(tid0, s);
            if (k < n) Pout[(size_t)bin * n + k] = P[s];
        }
    }
}

// acq_corr for the 65536-point transform: one workgroup of 512 lanes per (bin, channel).  The inverse
// transform split by lag mod 4 -- with F = f + 16384 q and k = 4 j + s,
//     y[4j + s] = IFFT_16k( w^(-f s) sum_q i^(q s) Y[f + 16384 q] )[j],   Y = X conj(C), w = exp(-2 pi i/65536)
// -- so that, as in the 32768-point kernel, every 16384-point transform ends in final lags and the
// only long-lived registers are the power accumulators: lane `tid` owns lags 4 (tid + 512 h + 1024 q) + s,
// h < 2, q < 8, s < 4 (lags below nsamp <= 32768), register index 16 s + 8 h + q.
struct RawXC4 { float4 x[4][2], c[4][2]; };

__global__ __launch_bounds__(512) void acq_corr64_kernel(
    const GcChan *__restrict__ chan, const float2 *__restrict__ tw16k, const float2 *__restrict__ tw32p,
    const float2 *__restrict__ tw64p1, const float2 *__restrict__ tw64p3,
    const float2 *__restrict__ X, const float2 *__restrict__ C, const int *__restrict__ iters,
    GcAcqRow *__restrict__ rows, double *__restrict__ Pout, int pout_ch, int maxfreq, int maxintg, int nchg)
{
    constexpr int NT = 512, L = 2 * GC_L, NP = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    double *sd = reinterpret_cast<double *>(smem + GC_FFT_LDS);       // 32 doubles
    int *si = reinterpret_cast<int *>(smem + GC_FFT_LDS + 256);       // 16 ints
    const int tid0 = threadIdx.x;
    const int slot = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int bin = Pout ? qq * 8 + slot : qq % maxfreq;
    const int ch = Pout ? pout_ch : (qq / maxfreq) * 8 + slot;
    if (bin >= maxfreq || (!Pout && ch >= nchg)) return;
    const GcChan &c = chan[ch];
    if (bin >= c.nfreq) return;
    const int n = c.nsamp, nit = iters[ch], nsc2 = 2 * c.nsampchip;
    const float2 *Cc = C + (size_t)ch * L;
    const float invL2 = 1.0f / ((float)L * (float)L);
    double P[NP];
#pragma unroll
    for (int s = 0; s < NP; s++) P[s] = 0.0;
    auto lag_of = [](int tid, int s) { return 4 * (tid + 512 * ((s >> 3) & 1) + 1024 * (s & 7)) + (s >> 4); };
    for (int it = 0; it < nit; it++) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const float2 *Xb = X + (((size_t)c.grid * maxintg + it) * maxfreq + bin) * (size_t)L;
        auto load = [Xb, Cc](int p) {
            RawXC4 r;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                r.x[q][0] = *reinterpret_cast<const float4 *>(Xb + q * GC_LH + p);
                r.x[q][1] = *reinterpret_cast<const float4 *>(Xb + q * GC_LH + p + 2);
                r.c[q][0] = *reinterpret_cast<const float4 *>(Cc + q * GC_LH + p);
                r.c[q][1] = *reinterpret_cast<const float4 *>(Cc + q * GC_LH + p + 2);
            }
            return r;
        };
        // Y_q at position k (0..3) of the butterfly
        auto Y = [](const RawXC4 &r, int q, int k) {
            const float4 xv = r.x[q][k >> 1], cv = r.c[q][k >> 1];
            return (k & 1) ? cmulc(make_float2(xv.z, xv.w), make_float2(cv.z, cv.w))
                           : cmulc(make_float2(xv.x, xv.y), make_float2(cv.x, cv.y));
        };
        auto residue = [&](auto s_tag) {
            constexpr int S = decltype(s_tag)::value;
            gcfft::dit<+1, NT, 1>(
                [&](int p) { return load(p); },
                [&](const RawXC4 &r, float2 &x0, float2 &x1, float2 &x2, float2 &x3) {
                    float2 z[4];
                    // (the pass position of this butterfly: recomputed from the lane, dit() fetched it at 4*(tid + NT*i))
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float2 y0 = Y(r, 0, k), y1 = Y(r, 1, k), y2 = Y(r, 2, k), y3 = Y(r, 3, k);
                        float2 v;
                        if (S == 0) v = cadd(cadd(y0, y2), cadd(y1, y3));
                        else if (S == 2) v = csub(cadd(y0, y2), cadd(y1, y3));
                        else {
                            const float2 d = csub(y1, y3), id = make_float2(-d.y, d.x);      // i (y1 - y3)
                            v = S == 1 ? cadd(csub(y0, y2), id) : csub(csub(y0, y2), id);
                        }
                        z[k] = v;
                    }
                    x0 = z[0]; x1 = z[1]; x2 = z[2]; x3 = z[3];
                },
                [&](int h, int, float2 (&a)[16]) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const float pw = fmaf(a[q].x, a[q].x, a[q].y * a[q].y) * invL2;
                        P[16 * S + 8 * h + q] += (double)pw;
                    }
                },
                lds, tw16k, tid);
        };
        (void)residue;
        // The twiddle w^(-f s) has to be applied per input: dit() hands `make` only the raw operands, so the
        // twiddled variants fetch their factors with the operands.
        struct RawT { RawXC4 r; float4 ta, tb; };
        auto residue_tw = [&](auto s_tag, const float2 *__restrict__ twp) {
            constexpr int S = decltype(s_tag)::value;
            gcfft::dit<+1, NT, 1>(
                [&](int p) {
                    RawT t;
                    t.r = load(p);
                    t.ta = *reinterpret_cast<const float4 *>(twp + p);
                    t.tb = *reinterpret_cast<const float4 *>(twp + p + 2);
                    return t;
                },
                [&](const RawT &t, float2 &x0, float2 &x1, float2 &x2, float2 &x3) {
                    float2 z[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float2 y0 = Y(t.r, 0, k), y1 = Y(t.r, 1, k), y2 = Y(t.r, 2, k), y3 = Y(t.r, 3, k);
                        float2 v;
                        if (S == 2) v = csub(cadd(y0, y2), cadd(y1, y3));
                        else {
                            const float2 d = csub(y1, y3), id = make_float2(-d.y, d.x);
                            v = S == 1 ? cadd(csub(y0, y2), id) : csub(csub(y0, y2), id);
                        }
                        const float2 w = k == 0 ? make_float2(t.ta.x, t.ta.y) : k == 1 ? make_float2(t.ta.z, t.ta.w)
                                       : k == 2 ? make_float2(t.tb.x, t.tb.y) : make_float2(t.tb.z, t.tb.w);
                        z[k] = cmulc(v, w);          // exp(+2 pi i f s/65536) = conj of the forward twiddle
                    }
                    x0 = z[0]; x1 = z[1]; x2 = z[2]; x3 = z[3];
                },
                [&](int h, int, float2 (&a)[16]) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const float pw = fmaf(a[q].x, a[q].x, a[q].y * a[q].y) * invL2;
                        P[16 * S + 8 * h + q] += (double)pw;
                    }
                },
                lds, tw16k, tid);
        };
        residue(std::integral_constant<int, 0>{});
        __syncthreads();
        asm volatile("" : "+v"(tid));
        residue_tw(std::integral_constant<int, 1>{}, tw64p1);
        __syncthreads();
        asm volatile("" : "+v"(tid));
        residue_tw(std::integral_constant<int, 2>{}, tw32p);
        __syncthreads();
        asm volatile("" : "+v"(tid));
        residue_tw(std::integral_constant<int, 3>{}, tw64p3);

        // row statistics for checkacquisition()
        MaxIdx m; m.v = -1.0; m.k = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < NP; s++) {
            const int k = lag_of(tid, s);
            if (k < n) { MaxIdx t; t.v = P[s]; t.k = k; m = better(m, t); }
        }
        m = wg_argmax<NT>(m, sd, si, tid);
        int exs = m.k - nsc2; if (exs < 0) exs += n;
        int exe = m.k + nsc2; if (exe >= n) exe -= n;
        double so = 0.0, mo = -1.0;
#pragma unroll
        for (int s = 0; s < NP; s++) {
            const int k = lag_of(tid, s);
            if (k < n) {
                const bool outside = (exs <= exe) ? (k < exs || k > exe) : (k < exs && k > exe);
                if (outside) so += P[s];
                if (outside || k == 0) mo = fmax(mo, P[s]);    // element 0 seeds maxvd()
            }
        }
        wg_sum_max<NT>(so, mo, sd, tid);
        if (tid == 0) {
            GcAcqRow r;
            r.rowmax = m.v; r.sum_out = so; r.max_out = mo; r.argmax = m.k; r.pad = 0;
            rows[((size_t)ch * maxintg + it) * maxfreq + bin] = r;
        }
        __syncthreads();
    }
    if (Pout) {
#pragma unroll
        for (int s = 0; s < NP; s++) {
            const int k = lag_of(tid0, s);
            if (k < n) Pout[(size_t)bin * n + k] = P[s];
        }
    }
}

// acq_final: one wavefront per channel; the lanes share the bins of an iteration
__global__ __launch_bounds__(64) void acq_final_kernel(const GcChan *__restrict__ chan, const double *__restrict__ freqs,
                                                       const GcAcqRow *__restrict__ rows,
                                                       const uint64_t *__restrict__ grid_wrpos,
                                                       gnsscorr_acqres_t *__restrict__ res, int *__restrict__ iters_out,
                                                       int nch, int maxfreq, int maxintg)
{
    const int ch = blockIdx.x, lane = threadIdx.x;
    if (ch >= nch) return;
    const GcChan &c = chan[ch];
    const int n = c.nsamp;
    gnsscorr_acqres_t r;
    r.acqcodei = 0; r.freqi = 0; r.acqfreq = 0; r.cn0 = 0; r.peakr = 0; r.flagacq = 0; r.iters = c.intg;
    const uint64_t b0 = grid_wrpos[c.grid] - (uint64_t)(c.intg + 1) * n;
    int it = 0;
    for (; it < c.intg; it++) {
        const GcAcqRow *row = rows + ((size_t)ch * maxintg + it) * maxfreq;
        // best row: largest rowmax, lowest flat index (= lowest bin) on ties
        double bv = -1.0;
        int bi = 0x7fffffff;
        for (int b = lane; b < c.nfreq; b += 64) {
            const double v = row[b].rowmax;
            if (v > bv) { bv = v; bi = b; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double ov = __shfl_xor(bv, d);
            const int oi = __shfl_xor(bi, d);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        const int fi = bi;
        const GcAcqRow w = row[fi];
        const int ne = 4 * c.nsampchip + 1;                     // samples inside the excluded window
        const double meanP = w.sum_out / (double)(n - ne);
        r.cn0 = 10.0 * log10(w.rowmax / meanP / c.ctime);
        r.peakr = w.rowmax / w.max_out;
        r.acqcodei = w.argmax;
        r.freqi = fi;
        r.acqfreq = freqs[c.freq_off + fi];
        if (r.peakr > 3.0) { r.flagacq = 1; break; }            // ACQTH, ref src/sdr.h:148
    }
    r.iters = r.flagacq ? it + 1 : c.intg;
    // ref src/sdracq.c:51-53 / :62
    r.buffloc = r.flagacq ? b0 + (uint64_t)r.acqcodei : b0 + (uint64_t)c.intg * n;
    if (lane == 0) {
        res[ch] = r;
        if (iters_out) iters_out[ch] = r.iters;
    }
}

__global__ void fill_int_kernel(int *p, const GcChan *__restrict__ chan, int nch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nch) p[i] = chan[i].intg;
}

// stand-alone batch FFT (op-level entry point and tests): grid (batch); natural order in and out
// (the pass-order result is scattered to its frequency on the way out)
template <int S>
__global__ __launch_bounds__(GC_FFT_THREADS) void fft16k_kernel(const float2 *__restrict__ in,
                                                                float2 *__restrict__ out,
                                                                const float2 *__restrict__ tw16k)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const float2 *x = in + (size_t)blockIdx.x * GC_LH;
    float2 *y = out + (size_t)blockIdx.x * GC_LH;
    gcfft::dif<S>([&](int j) { return x[j]; },
                  [&](int p, float2 x0, float2 x1, float2 x2, float2 x3) {
                      y[gcfft::freq_of(p)] = x0; y[gcfft::freq_of(p + 1)] = x1;
                      y[gcfft::freq_of(p + 2)] = x2; y[gcfft::freq_of(p + 3)] = x3;
                  },
                  lds, tw16k, tid);
}

// power spectrum of a 16384- or 32768-point sequence (cpxpspec, ref src/sdrcmn.c:261-276)
__global__ __launch_bounds__(GC_FFT_THREADS) void pspec_kernel(const float2 *__restrict__ in, int n,
                                                               const float2 *__restrict__ tw16k,
                                                               const float2 *__restrict__ tw32k,
                                                               double *__restrict__ pspec, int flagsum)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    auto power_to = [&](int mul, int add) {
        return [=](int p, float2 x0, float2 x1, float2 x2, float2 x3) {
            const float2 xs[4] = {x0, x1, x2, x3};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int f = mul * gcfft::freq_of(p + i) + add;
                const double pw = (double)fmaf(xs[i].x, xs[i].x, xs[i].y * xs[i].y);
                pspec[f] = flagsum ? pspec[f] + pw : pw;
            }
        };
    };
    if (n == GC_LH) {
        gcfft::dif<-1>([&](int j) { return in[j]; }, power_to(1, 0), lds, tw16k, tid);
    } else {
        gcfft::dif<-1>([&](int j) { return cadd(in[j], in[j + GC_LH]); }, power_to(2, 0), lds, tw16k, tid);
        __syncthreads();
        gcfft::dif<-1>([&](int j) { return cmul(csub(in[j], in[j + GC_LH]), tw32k[j]); }, power_to(2, 1), lds,
                       tw16k, tid);
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
void gc_acq_free(gnsscorr_ctx *ctx)
{
    GcAcqWork *w = ctx->acq;
    if (!w) return;
    hipFree(w->tw16k); hipFree(w->tw32k); hipFree(w->tw32p); hipFree(w->tw64p1); hipFree(w->tw64p3);
    hipFree(w->X); hipFree(w->C); hipFree(w->rows); hipFree(w->arrive);
    hipFree(w->iters); hipFree(w->res); hipFree(w->P);
    hipFree(w->d_grid_chan); hipFree(w->d_grid_wrpos); hipFree(w->car); hipFree(w->car_overflow);
    delete w;
    ctx->acq = nullptr;
}

static int acq_tables(gnsscorr_ctx *ctx)
{
    if (!ctx->acq) ctx->acq = new GcAcqWork();
    GcAcqWork *w = ctx->acq;
    if (w->tw16k) return GNSSCORR_OK;
    GC_HIP(hipMalloc((void **)&w->tw16k, sizeof(float2) * GC_LH));
    GC_HIP(hipMalloc((void **)&w->tw32k, sizeof(float2) * GC_LH));
    GC_HIP(hipMalloc((void **)&w->tw32p, sizeof(float2) * GC_LH));
    GC_HIP(hipMalloc((void **)&w->tw64p1, sizeof(float2) * GC_LH));
    GC_HIP(hipMalloc((void **)&w->tw64p3, sizeof(float2) * GC_LH));
    hipLaunchKernelGGL(tw_init_kernel, dim3(GC_LH / 256), dim3(256), 0, ctx->stream, w->tw16k, w->tw32k, w->tw32p);
    hipLaunchKernelGGL(tw64_init_kernel, dim3(GC_LH / 256), dim3(256), 0, ctx->stream, w->tw64p1, w->tw64p3);
    GC_HIP(hipGetLastError());
    {
        const int lds = GC_FFT_LDS + 256;
        GC_HIP(hipFuncSetAttribute((const void *)acq_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds + GC_ACQ_CARLDS));
        GC_HIP(hipFuncSetAttribute((const void *)acq_code_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        GC_HIP(hipFuncSetAttribute((const void *)acq_corr_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 256));
        GC_HIP(hipFuncSetAttribute((const void *)acq_corr_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 256));
        GC_HIP(hipFuncSetAttribute((const void *)acq_corr64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 256));
        GC_HIP(hipFuncSetAttribute((const void *)fft16k_kernel<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        GC_HIP(hipFuncSetAttribute((const void *)fft16k_kernel<+1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        GC_HIP(hipFuncSetAttribute((const void *)pspec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    return GNSSCORR_OK;
}

static int acq_prepare(gnsscorr_ctx *ctx)
{
    int rc = acq_tables(ctx);
    if (rc) return rc;
    GcAcqWork *w = ctx->acq;
    if (w->C) return GNSSCORR_OK;
    const int nch = ctx->nch;
    w->ngrid = 0; w->maxfreq = 0; w->maxintg = 0;
    w->L = GC_L;
    w->grid_chan.clear();
    for (int i = 0; i < nch; i++) {
        const GcChan &c = ctx->hchan[i];
        // the window of 2*nsamp samples must fit the transform: 32768 points up to 16384 samples per code
        // period, 65536 (20 / 26 Msps front ends) up to 32768
        if (c.nsamp > GC_L)
            return gc_fail(GNSSCORR_EINVAL, "acquisition: nsamp %d needs an FFT longer than 65536", c.nsamp);
        if (c.nsamp > GC_LH) w->L = 2 * GC_L;
        if (c.grid >= w->ngrid) { w->ngrid = c.grid + 1; w->grid_chan.push_back(i); }
        if (c.nfreq > w->maxfreq) w->maxfreq = c.nfreq;
        if (c.intg > w->maxintg) w->maxintg = c.intg;
    }
    w->X_elems = (size_t)w->ngrid * w->maxintg * w->maxfreq * w->L;
    GC_HIP(hipMalloc((void **)&w->X, sizeof(float2) * w->X_elems));
    GC_HIP(hipMalloc((void **)&w->C, sizeof(float2) * (size_t)nch * w->L));
    GC_HIP(hipMalloc((void **)&w->rows, sizeof(GcAcqRow) * (size_t)nch * w->maxintg * w->maxfreq));
    GC_HIP(hipMalloc((void **)&w->arrive, sizeof(int) * ((size_t)nch * w->maxintg + nch)));
    GC_HIP(hipMalloc((void **)&w->iters, sizeof(int) * nch));
    GC_HIP(hipMalloc((void **)&w->res, sizeof(gnsscorr_acqres_t) * nch));
    GC_HIP(hipMalloc((void **)&w->d_grid_chan, sizeof(int) * w->ngrid));
    GC_HIP(hipMalloc((void **)&w->d_grid_wrpos, sizeof(uint64_t) * w->ngrid));
    GC_HIP(hipMemcpyAsync(w->d_grid_chan, w->grid_chan.data(), sizeof(int) * w->ngrid, hipMemcpyHostToDevice,
                          ctx->stream));
    GC_HIP(hipMalloc((void **)&w->car, sizeof(GcAcqCar) * (size_t)w->ngrid * w->maxfreq));
    GC_HIP(hipMalloc((void **)&w->car_overflow, sizeof(int)));
    GC_HIP(hipMemsetAsync(w->car_overflow, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(acq_nco_kernel, dim3((w->ngrid * w->maxfreq + 63) / 64), dim3(64), 0, ctx->stream, ctx->dchan,
                       w->d_grid_chan, ctx->dfreqs, w->car, w->ngrid, w->maxfreq, w->car_overflow);
    GC_HIP(hipGetLastError());
    {
        GcTimed t(ctx, "acq_code");
        hipLaunchKernelGGL(acq_code_kernel, dim3(nch), dim3(GC_FFT_THREADS), GC_FFT_LDS + 256, ctx->stream,
                           ctx->dchan, w->tw16k, w->tw32p, w->tw64p1, w->tw64p3, w->C, w->L);
    }
    GC_HIP(hipGetLastError());
    int over = 0;
    GC_HIP(hipMemcpyAsync(&over, w->car_overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    if (over) {
        hipFree(w->C); w->C = nullptr;       // not prepared
        return gc_fail(GNSSCORR_EINVAL, "acquisition: %d Doppler bins need more carrier NCO pieces than the tables hold", over);
    }
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_acq_run(gnsscorr_ctx *ctx, uint64_t wrpos)
{
    if (!ctx) return gc_fail(GNSSCORR_EINVAL, "null context");
    if (!ctx->nch) return gc_fail(GNSSCORR_ESTATE, "acq_run: no channels set");
    GC_HIP(hipSetDevice(ctx->device));
    int rc = acq_prepare(ctx);
    if (rc) return rc;
    rc = gc_ingest_fence(ctx);
    if (rc) return rc;
    GcAcqWork *w = ctx->acq;
    std::vector<uint64_t> gw(w->ngrid);
    for (int g = 0; g < w->ngrid; g++) {
        const GcChan &c = ctx->hchan[w->grid_chan[g]];
        const int ft = ctx->hdesc[w->grid_chan[g]].ftype;
        const uint64_t wp = wrpos ? wrpos : ctx->ring[ft - 1].wrpos;
        if (wp < (uint64_t)(c.intg + 1) * c.nsamp)
            return gc_fail(GNSSCORR_ESTATE, "acq_run: ring %d holds %llu samples, %llu needed", ft,
                           (unsigned long long)wp, (unsigned long long)((uint64_t)(c.intg + 1) * c.nsamp));
        if (c.ringlen < (uint64_t)(c.intg + 1) * c.nsamp)
            return gc_fail(GNSSCORR_EINVAL, "acq_run: ring %d (%llu samples) is shorter than the %d code periods the search looks back",
                           ft, (unsigned long long)c.ringlen, c.intg + 1);
        gw[g] = wp;
    }
    // pageable source: the copy is staged before the call returns
    GC_HIP(hipMemcpyAsync(w->d_grid_wrpos, gw.data(), sizeof(uint64_t) * w->ngrid, hipMemcpyHostToDevice,
                          ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    const int lds = GC_FFT_LDS + 256;
    {
        GcTimed t(ctx, "acq_fwd");
        hipLaunchKernelGGL(acq_fwd_kernel, dim3(w->maxfreq, w->maxintg, w->ngrid), dim3(GC_FFT_THREADS), lds + GC_ACQ_CARLDS,
                           ctx->stream, ctx->dchan, w->d_grid_chan, w->car, w->d_grid_wrpos, w->tw16k, w->tw32p,
                           w->tw64p1, w->tw64p3, w->X, w->maxfreq, w->maxintg, w->L);
    }
    GC_HIP(hipGetLastError());
    hipLaunchKernelGGL(fill_int_kernel, dim3((ctx->nch + 63) / 64), dim3(64), 0, ctx->stream, w->iters,
                       ctx->dchan, ctx->nch);
    // arrival counters per (channel, iteration) and the channels' "acquired" flags
    GC_HIP(hipMemsetAsync(w->arrive, 0, sizeof(int) * ((size_t)ctx->nch * w->maxintg + ctx->nch), ctx->stream));
    {
        GcTimed t(ctx, "acq_corr");
        static const int nt = getenv("GNSSCORR_ACQ_NT") ? atoi(getenv("GNSSCORR_ACQ_NT")) : 512;
        // (bins per XCD slot) x (channels, in groups of GC_ACQ_G) x 8 slots
        const unsigned acq_grid = 8u * (unsigned)((w->maxfreq + 7) / 8) * (unsigned)(GC_ACQ_G * ((ctx->nch + GC_ACQ_G - 1) / GC_ACQ_G));
        if (w->L == 2 * GC_L)
            hipLaunchKernelGGL(acq_corr64_kernel, dim3(8 * ((ctx->nch + 7) / 8) * w->maxfreq), dim3(512), lds + 256,
                               ctx->stream, ctx->dchan, w->tw16k, w->tw32p, w->tw64p1, w->tw64p3, w->X, w->C, w->iters,
                               w->rows, (double *)nullptr, 0, w->maxfreq, w->maxintg, ctx->nch);
        else if (nt == 1024)
            hipLaunchKernelGGL(acq_corr_kernel<1024>, dim3(acq_grid), dim3(1024), lds + 256,
                               ctx->stream, ctx->dchan, w->tw16k, w->tw32p, w->X, w->C, w->iters, w->rows,
                               (double *)nullptr, 0, w->maxfreq, w->maxintg, ctx->nch, w->arrive, w->arrive + (size_t)ctx->nch * w->maxintg);
        else
            hipLaunchKernelGGL(acq_corr_kernel<512>, dim3(acq_grid), dim3(512), lds + 256,
                               ctx->stream, ctx->dchan, w->tw16k, w->tw32p, w->X, w->C, w->iters, w->rows,
                               (double *)nullptr, 0, w->maxfreq, w->maxintg, ctx->nch, w->arrive, w->arrive + (size_t)ctx->nch * w->maxintg);
    }
    GC_HIP(hipGetLastError());
    {
        GcTimed t(ctx, "acq_final");
        hipLaunchKernelGGL(acq_final_kernel, dim3(ctx->nch), dim3(64), 0, ctx->stream, ctx->dchan,
                           ctx->dfreqs, w->rows, w->d_grid_wrpos, w->res, w->iters, ctx->nch, w->maxfreq,
                           w->maxintg);
    }
    GC_HIP(hipGetLastError());
    w->ran = true;
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_acq_fetch(gnsscorr_ctx *ctx, gnsscorr_acqres_t *res)
{
    if (!ctx || !ctx->acq || !ctx->acq->ran) return gc_fail(GNSSCORR_ESTATE, "acq_fetch: no acq_run yet");
    if (!res) return gc_fail(GNSSCORR_EINVAL, "acq_fetch: null result array");
    GC_HIP(hipSetDevice(ctx->device));
    GC_HIP(hipMemcpyAsync(res, ctx->acq->res, sizeof(gnsscorr_acqres_t) * ctx->nch, hipMemcpyDeviceToHost,
                          ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GNSSCORR_OK;
}

// Acquired channels start tracking where sdracquisition() leaves them (ref src/sdracq.c:51-55:
// trk.carrfreq = acq.acqfreq, trk.codefreq = crate, code and carrier remainders zero, tracking from the
// returned buffloc); channels that were not acquired keep their state.
__global__ void acq_to_trk_kernel(const GcChan *__restrict__ chan, const gnsscorr_acqres_t *__restrict__ res,
                                  GcTrkState *__restrict__ state, int nch)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nch || !res[ch].flagacq) return;
    GcTrkState s;
    s.carrfreq = res[ch].acqfreq;
    s.codefreq = chan[ch].crate;
    s.remcode = 0.0;
    s.remcarr = 0.0;
    s.buffloc = res[ch].buffloc;
    state[ch] = s;
}

extern "C" int gnsscorr_trk_start_from_acq(gnsscorr_ctx *ctx)
{
    if (!ctx || !ctx->acq || !ctx->acq->ran) return gc_fail(GNSSCORR_ESTATE, "trk_start_from_acq: no acq_run yet");
    GC_HIP(hipSetDevice(ctx->device));
    if (ctx->stream2) GC_HIP(hipStreamSynchronize(ctx->stream2));     // a look-ahead plan may be in flight
    if (ctx->stream3) GC_HIP(hipStreamSynchronize(ctx->stream3));
    ctx->ahead_valid = false;                                         // ... and is dropped
    ctx->state_touched = true;
    hipLaunchKernelGGL(acq_to_trk_kernel, dim3((ctx->nch + 63) / 64), dim3(64), 0, ctx->stream, ctx->dchan,
                       ctx->acq->res, ctx->dstate2[ctx->state_cur], ctx->nch);
    GC_HIP(hipGetLastError());
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_acq_power(gnsscorr_ctx *ctx, int ch, double *power)
{
    if (!ctx || !ctx->acq || !ctx->acq->ran) return gc_fail(GNSSCORR_ESTATE, "acq_power: no acq_run yet");
    if (ch < 0 || ch >= ctx->nch || !power) return gc_fail(GNSSCORR_EINVAL, "acq_power: channel %d", ch);
    GC_HIP(hipSetDevice(ctx->device));
    GcAcqWork *w = ctx->acq;
    const GcChan &c = ctx->hchan[ch];
    const size_t elems = (size_t)c.nfreq * c.nsamp;
    if (elems > w->P_elems) {
        hipFree(w->P); w->P = nullptr; w->P_elems = 0;
        GC_HIP(hipMalloc((void **)&w->P, sizeof(double) * elems));
        w->P_elems = elems;
    }
    // iteration count of the last run is still in w->iters[ch]; rows of this channel are rewritten
    // with identical values
    if (w->L == 2 * GC_L)
        hipLaunchKernelGGL(acq_corr64_kernel, dim3(8 * ((c.nfreq + 7) / 8)), dim3(512), GC_FFT_LDS + 512,
                           ctx->stream, ctx->dchan, w->tw16k, w->tw32p, w->tw64p1, w->tw64p3, w->X, w->C, w->iters,
                           w->rows, w->P, ch, w->maxfreq, w->maxintg, 1);
    else
        hipLaunchKernelGGL(acq_corr_kernel<512>, dim3(8 * ((c.nfreq + 7) / 8)), dim3(512), GC_FFT_LDS + 512,
                           ctx->stream, ctx->dchan, w->tw16k, w->tw32p, w->X, w->C, w->iters, w->rows, w->P, ch,
                           w->maxfreq, w->maxintg, 1, (int *)nullptr, (int *)nullptr);
    GC_HIP(hipGetLastError());
    GC_HIP(hipMemcpyAsync(power, w->P, sizeof(double) * elems, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_fft16k(gnsscorr_ctx *ctx, const void *in, void *out, int sign, int batch)
{
    if (!ctx || !in || !out || batch <= 0) return gc_fail(GNSSCORR_EINVAL, "fft16k: bad arguments");
    GC_HIP(hipSetDevice(ctx->device));
    int rc = acq_tables(ctx);
    if (rc) return rc;
    GcAcqWork *w = ctx->acq;
    GcTimed t(ctx, "fft16k");
    if (sign < 0)
        hipLaunchKernelGGL(fft16k_kernel<-1>, dim3(batch), dim3(GC_FFT_THREADS), GC_FFT_LDS + 256, ctx->stream,
                           (const float2 *)in, (float2 *)out, w->tw16k);
    else
        hipLaunchKernelGGL(fft16k_kernel<+1>, dim3(batch), dim3(GC_FFT_THREADS), GC_FFT_LDS + 256, ctx->stream,
                           (const float2 *)in, (float2 *)out, w->tw16k);
    GC_HIP(hipGetLastError());
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_pspec(gnsscorr_ctx *ctx, const float *cpx, int n, int flagsum, double *pspec)
{
    if (!ctx || !cpx || !pspec) return gc_fail(GNSSCORR_EINVAL, "pspec: bad arguments");
    if (n != GC_LH && n != GC_L) return gc_fail(GNSSCORR_EINVAL, "pspec: n %d (16384 or 32768 supported)", n);
    GC_HIP(hipSetDevice(ctx->device));
    int rc = acq_tables(ctx);
    if (rc) return rc;
    GcAcqWork *w = ctx->acq;
    float2 *din = nullptr;
    double *dps = nullptr;
    GC_HIP(hipMalloc((void **)&din, sizeof(float2) * n));
    if (hipMalloc((void **)&dps, sizeof(double) * n) != hipSuccess) { hipFree(din); return gc_fail(GNSSCORR_EHIP, "pspec: hipMalloc"); }
    hipMemcpyAsync(din, cpx, sizeof(float2) * n, hipMemcpyHostToDevice, ctx->stream);
    if (flagsum) hipMemcpyAsync(dps, pspec, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream);
    hipLaunchKernelGGL(pspec_kernel, dim3(1), dim3(GC_FFT_THREADS), GC_FFT_LDS + 256, ctx->stream, din, n,
                       w->tw16k, w->tw32k, dps, flagsum);
    hipError_t e = hipGetLastError();
    hipMemcpyAsync(pspec, dps, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    hipFree(din); hipFree(dps);
    if (e != hipSuccess) return gc_fail_hip(e, "pspec_kernel", __FILE__, __LINE__);
    if (e2 != hipSuccess) return gc_fail_hip(e2, "pspec sync", __FILE__, __LINE__);
    return GNSSCORR_OK;
}
