// gnsscorr_fft.h -- LDS-resident 16384-point complex FFT for one 512-thread
// workgroup on gfx950, the building block of the parallel code phase search
// (it stands where the reference calls FFTW: cpxfft/cpxifft, ref
// src/sdrcmn.c:134-175).
//
// 16384 = 4 * 16 * 16 * 16.  Stockham auto-sort, four passes over a single
// 128 KiB LDS image; each thread owns 32 points, so a pass reads all of its
// operands into registers, meets the workgroup barrier, and only then writes
// the image back in the permuted order (in place, no second buffer).
//   pass 0  radix-4 , no twiddles, operands come from the caller's registers
//   pass 1  radix-16, sub-transform length 64
//   pass 2  radix-16, sub-transform length 1024
//   pass 3  radix-16, sub-transform length 16384, results stay in registers
// Twiddles w^r (r = 1..15) are built from one table entry w = exp(-2*pi*i*t/16384)
// by a product tree of depth <= 4, so each factor carries <= 4 float roundings.
#pragma once

#include <hip/hip_runtime.h>

#define GC_FFT_N        16384
#define GC_FFT_THREADS  512
#define GC_FFT_LDS      (GC_FFT_N * 8)

namespace gcfft {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
// multiply by S*i  (S = -1 forward, +1 backward)
template <int S>
__device__ __forceinline__ float2 muli(float2 a)
{
    return S > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
template <int S>
__device__ __forceinline__ float2 tw(float2 t)     // table holds forward twiddles
{
    return S > 0 ? make_float2(t.x, -t.y) : t;
}

// 4-point DFT in place: X[k] = sum_r a[r] exp(S*2*pi*i*r*k/4)
template <int S>
__device__ __forceinline__ void dft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3)
{
    const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2);
    const float2 s13 = cadd(a1, a3), d13 = muli<S>(csub(a1, a3));
    a0 = cadd(s02, s13);
    a1 = cadd(d02, d13);
    a2 = csub(s02, s13);
    a3 = csub(d02, d13);
}

// 16-point DFT in place, natural order in and out.
template <int S>
__device__ __forceinline__ void dft16(float2 (&a)[16])
{
    constexpr float C1 = 0.92387953251128673848f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;   // sin(pi/8)
    constexpr float H = 0.70710678118654752440f;    // sqrt(1/2)
    // step 1: for each r1, DFT4 over r2 of a[r1 + 4*r2]  -> B[r1][k2] stored at a[r1 + 4*k2]
#pragma unroll
    for (int r1 = 0; r1 < 4; r1++) dft4<S>(a[r1], a[r1 + 4], a[r1 + 8], a[r1 + 12]);
    // step 2: multiply B[r1][k2] by W16^(r1*k2), W16 = exp(S*2*pi*i/16)
    const float sg = (float)S;
    a[1 + 4]  = cmul(a[1 + 4],  make_float2(C1, sg * S1));     // W^1
    a[2 + 4]  = cmul(a[2 + 4],  make_float2(H, sg * H));       // W^2
    a[3 + 4]  = cmul(a[3 + 4],  make_float2(S1, sg * C1));     // W^3
    a[1 + 8]  = cmul(a[1 + 8],  make_float2(H, sg * H));       // W^2
    a[2 + 8]  = muli<S>(a[2 + 8]);                             // W^4
    a[3 + 8]  = cmul(a[3 + 8],  make_float2(-H, sg * H));      // W^6
    a[1 + 12] = cmul(a[1 + 12], make_float2(S1, sg * C1));     // W^3
    a[2 + 12] = cmul(a[2 + 12], make_float2(-H, sg * H));      // W^6
    a[3 + 12] = cmul(a[3 + 12], make_float2(-C1, -sg * S1));   // W^9
    // step 3: for each k2, DFT4 over r1 of a[r1 + 4*k2] -> X[4*k1 + k2] stored at a[k1 + 4*k2]
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) dft4<S>(a[4 * k2], a[4 * k2 + 1], a[4 * k2 + 2], a[4 * k2 + 3]);
    // transpose to natural order: X[4*k1 + k2] currently at a[k1 + 4*k2]
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i + 1; j < 4; j++) {
            const float2 t = a[i + 4 * j];
            a[i + 4 * j] = a[j + 4 * i];
            a[j + 4 * i] = t;
        }
}

// a[r] *= w^r for r = 1..15
__device__ __forceinline__ void twiddle16(float2 (&a)[16], float2 w1)
{
    const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
    const float2 w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3), w8 = cmul(w4, w4);
    a[1] = cmul(a[1], w1);  a[2] = cmul(a[2], w2);  a[3] = cmul(a[3], w3);  a[4] = cmul(a[4], w4);
    a[5] = cmul(a[5], w5);  a[6] = cmul(a[6], w6);  a[7] = cmul(a[7], w7);  a[8] = cmul(a[8], w8);
    a[9] = cmul(a[9], cmul(w8, w1));   a[10] = cmul(a[10], cmul(w8, w2));
    a[11] = cmul(a[11], cmul(w8, w3)); a[12] = cmul(a[12], cmul(w8, w4));
    a[13] = cmul(a[13], cmul(w8, w5)); a[14] = cmul(a[14], cmul(w8, w6));
    a[15] = cmul(a[15], cmul(w8, w7));
}

// One radix-16 pass for the butterfly with index j (0..1023) whose operands
// are in a[16]: twiddle (sub-transform position k = j mod NS, period 16*NS),
// transform, and return the LDS index of output r as base + r*NS.
template <int S, int NS>
__device__ __forceinline__ int pass16(float2 (&a)[16], int j, const float2 *__restrict__ tw16k)
{
    const int k = j & (NS - 1);
    if (NS > 1) {
        const float2 w1 = tw<S>(tw16k[k * (GC_FFT_N / (16 * NS))]);
        twiddle16(a, w1);
    }
    dft16<S>(a);
    return (j - k) * 16 + k;
}

// Full transform of x[j] = load(j), j < 16384.  On exit v[s] = X[tid + 512*s]
// (s < 32).  `lds` is the 128 KiB image; the caller must not touch it between
// entry and exit, and a workgroup barrier is required before the image is
// reused after exit (the last pass only reads).
template <int S, class Load>
__device__ __forceinline__ void fft16k(Load load, float2 (&v)[32], float2 *lds,
                                       const float2 *__restrict__ tw16k, int tid)
{
    // pass 0: radix 4, NS = 1: operands x[j + 4096*r], results y[4*j + r]
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const int j = tid + GC_FFT_THREADS * b;
        float2 x0 = load(j), x1 = load(j + 4096), x2 = load(j + 8192), x3 = load(j + 12288);
        dft4<S>(x0, x1, x2, x3);
        float4 *dst = reinterpret_cast<float4 *>(lds + 4 * j);
        dst[0] = make_float4(x0.x, x0.y, x1.x, x1.y);
        dst[1] = make_float4(x2.x, x2.y, x3.x, x3.y);
    }
    __syncthreads();

    float2 a0[16], a1[16];
    // pass 1: NS = 4
#pragma unroll
    for (int r = 0; r < 16; r++) { a0[r] = lds[tid + 1024 * r]; a1[r] = lds[tid + 512 + 1024 * r]; }
    __syncthreads();
    {
        const int o0 = pass16<S, 4>(a0, tid, tw16k), o1 = pass16<S, 4>(a1, tid + 512, tw16k);
#pragma unroll
        for (int r = 0; r < 16; r++) { lds[o0 + 4 * r] = a0[r]; lds[o1 + 4 * r] = a1[r]; }
    }
    __syncthreads();
    // pass 2: NS = 64
#pragma unroll
    for (int r = 0; r < 16; r++) { a0[r] = lds[tid + 1024 * r]; a1[r] = lds[tid + 512 + 1024 * r]; }
    __syncthreads();
    {
        const int o0 = pass16<S, 64>(a0, tid, tw16k), o1 = pass16<S, 64>(a1, tid + 512, tw16k);
#pragma unroll
        for (int r = 0; r < 16; r++) { lds[o0 + 64 * r] = a0[r]; lds[o1 + 64 * r] = a1[r]; }
    }
    __syncthreads();
    // pass 3: NS = 1024, outputs X[j + 1024*r] stay in registers
#pragma unroll
    for (int r = 0; r < 16; r++) { a0[r] = lds[tid + 1024 * r]; a1[r] = lds[tid + 512 + 1024 * r]; }
    pass16<S, 1024>(a0, tid, tw16k);
    pass16<S, 1024>(a1, tid + 512, tw16k);
#pragma unroll
    for (int r = 0; r < 16; r++) { v[2 * r] = a0[r]; v[2 * r + 1] = a1[r]; }
}

}  // namespace gcfft
