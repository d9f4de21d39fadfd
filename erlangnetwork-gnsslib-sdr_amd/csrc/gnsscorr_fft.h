// gnsscorr_fft.h -- LDS-resident 16384-point complex FFT for one 512-thread
// workgroup on gfx950, the building block of the parallel code phase search
// (it stands where the reference calls FFTW: cpxfft/cpxifft, ref
// src/sdrcmn.c:134-175).
//
// 16384 = 16 * 16 * 16 * 4, in place in one (padded) 136 KiB LDS image.
//   dif<S> : natural-order input  -> output in "pass order" (digit reversed)
//            passes: radix-16 on sub-transform sizes 16384, 1024, 64, then radix-4
//   dit<S> : pass-order input     -> natural-order output
//            passes: radix-4, then radix-16 on sizes 64, 1024, 16384
// A spectrum produced by dif is consumed by dit without any reordering, which
// is all the correlation needs: X and conj(C) meet element by element in pass
// order.  Every butterfly reads and writes its own 16 (or 4) image slots, so a
// lane holds only one butterfly (32 VGPRs) at a time and the only barriers are
// the ones between passes.  The first pass takes its operands from a functor
// (global memory / on-the-fly samples), the last pass hands its results to a
// functor, so neither touches LDS twice.
//
// Position p = 1024*q1 + 64*q2 + 4*q3 + q4 of the pass-order image holds
// frequency f = q1 + 16*q2 + 256*q3 + 4096*q4.
//
// Image padding: 4 float2 after every 64 (index i -> i + 4*(i>>6)); with it the
// stride-4 pass touches 64 different banks per 32 lanes instead of 8.
// Twiddles w^r (r = 1..15) come from one table entry w = exp(-2*pi*i*t/16384)
// through a product tree of depth <= 4.
#pragma once

#include <hip/hip_runtime.h>

#define GC_FFT_N        16384
#define GC_FFT_THREADS  512
#define GC_FFT_LDS      ((GC_FFT_N + 4 * (GC_FFT_N >> 6)) * 8)      // 139264 bytes

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
__device__ __forceinline__ int padi(int i) { return i + ((i >> 6) << 2); }

// frequency held by pass-order position p
__device__ __forceinline__ int freq_of(int p)
{
    return (p >> 10) + (((p >> 6) & 15) << 4) + (((p >> 2) & 15) << 8) + ((p & 3) << 12);
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

// Geometry of radix-16 butterfly b (0..1023) of the pass working on
// sub-transforms of size M: element r lives at image index base + r*STEP
// (padding included), o is the position inside the sub-transform.
template <int M>
struct Geo {
    static constexpr int STRIDE = M / 16;
    static constexpr int STEP = STRIDE >= 64 ? STRIDE + 4 * (STRIDE >> 6) : STRIDE;
    __device__ static __forceinline__ int o(int b) { return b & (STRIDE - 1); }
    __device__ static __forceinline__ int base(int b) { return padi((b / STRIDE) * M + (b & (STRIDE - 1))); }
    __device__ static __forceinline__ float2 w1(const float2 *__restrict__ tw16k, int b)
    {
#ifdef GC_ABLATE_TW
        return make_float2(1.0f, 0.0f);
#else
        return tw16k[o(b) * (GC_FFT_N / M)];
#endif
    }
};

// ---- decimation in frequency: natural in, pass order out --------------------
template <int S, int M>
__device__ __forceinline__ void dif16_lds(float2 *lds, const float2 *__restrict__ tw16k, int b)
{
    float2 a[16];
    const int base = Geo<M>::base(b);
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = lds[base + r * Geo<M>::STEP];
    dft16<S>(a);
    twiddle16(a, tw<S>(Geo<M>::w1(tw16k, b)));
#pragma unroll
    for (int r = 0; r < 16; r++) lds[base + r * Geo<M>::STEP] = a[r];
}

// load(j) -> x[j], j < 16384 natural order;  store(p, v0..v3) <- pass-order positions p..p+3
template <int S, int NT = GC_FFT_THREADS, class Load, class Store>
__device__ __forceinline__ void dif(Load load, Store store, float2 *lds, const float2 *__restrict__ tw16k,
                                    int tid)
{
#pragma unroll
    for (int h = 0; h < 1024 / NT; h++) {   // size 16384: operands straight from the functor
        const int b = tid + NT * h;
        float2 a[16];
#pragma unroll
        for (int r = 0; r < 16; r++) a[r] = load(b + 1024 * r);
        dft16<S>(a);
        twiddle16(a, tw<S>(tw16k[b]));
        const int base = Geo<16384>::base(b);
#pragma unroll
        for (int r = 0; r < 16; r++) lds[base + r * Geo<16384>::STEP] = a[r];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 1024 / NT; h++) dif16_lds<S, 1024>(lds, tw16k, tid + NT * h);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 1024 / NT; h++) dif16_lds<S, 64>(lds, tw16k, tid + NT * h);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4096 / NT; i++) {   // radix 4 on 4 adjacent slots, results leave through the functor
        const int p = 4 * (tid + NT * i);
        const float4 *src = reinterpret_cast<const float4 *>(lds + padi(p));
        const float4 u = src[0], v = src[1];
        float2 x0 = make_float2(u.x, u.y), x1 = make_float2(u.z, u.w);
        float2 x2 = make_float2(v.x, v.y), x3 = make_float2(v.z, v.w);
        dft4<S>(x0, x1, x2, x3);
        store(p, x0, x1, x2, x3);
    }
}

// ---- decimation in time: pass order in, natural out --------------------------
template <int S, int M>
__device__ __forceinline__ void dit16_lds(float2 *lds, const float2 *__restrict__ tw16k, int b)
{
    float2 a[16];
    const int base = Geo<M>::base(b);
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = lds[base + r * Geo<M>::STEP];
    twiddle16(a, tw<S>(Geo<M>::w1(tw16k, b)));
    dft16<S>(a);
#pragma unroll
    for (int r = 0; r < 16; r++) lds[base + r * Geo<M>::STEP] = a[r];
}

// fetch(p) -> raw operands of pass-order inputs p..p+3 (memory loads only, any struct);
// make(raw, x0..x3) turns them into the four inputs.  The loads of CH butterflies are issued
// back to back before the first use, so a lane has 4*CH 16-byte loads in flight instead of
// paying the memory latency once per butterfly.
// sink(h, o, a) <- results y[o + 1024*q] = a[q], q < 16,
// called for h = 0 .. 1024/NT-1 with o = tid + NT*h (h is a compile-time constant after unrolling,
// so the caller can index register arrays with it).  NT = lanes in the workgroup (512 or 1024).
template <int S, int NT = GC_FFT_THREADS, int CH = 4, class Fetch, class Make, class Sink>
__device__ __forceinline__ void dit(Fetch fetch, Make make, Sink sink, float2 *lds,
                                    const float2 *__restrict__ tw16k, int tid)
{
    constexpr int NB = 4096 / NT;
    static_assert(NB % CH == 0, "chunk must divide the butterflies per lane");
#pragma unroll
    for (int i0 = 0; i0 < NB; i0 += CH) {
        decltype(fetch(0)) raw[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) raw[c] = fetch(4 * (tid + NT * (i0 + c)));
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int p = 4 * (tid + NT * (i0 + c));
            float2 x0, x1, x2, x3;
            make(raw[c], x0, x1, x2, x3);
            dft4<S>(x0, x1, x2, x3);
            float4 *dst = reinterpret_cast<float4 *>(lds + padi(p));
            dst[0] = make_float4(x0.x, x0.y, x1.x, x1.y);
            dst[1] = make_float4(x2.x, x2.y, x3.x, x3.y);
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 1024 / NT; h++) dit16_lds<S, 64>(lds, tw16k, tid + NT * h);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 1024 / NT; h++) dit16_lds<S, 1024>(lds, tw16k, tid + NT * h);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 1024 / NT; h++) {
        const int b = tid + NT * h;
        float2 a[16];
        const int base = Geo<16384>::base(b);
#pragma unroll
        for (int r = 0; r < 16; r++) a[r] = lds[base + r * Geo<16384>::STEP];
        twiddle16(a, tw<S>(tw16k[b]));
        dft16<S>(a);
        sink(h, b, a);
    }
}

}  // namespace gcfft
