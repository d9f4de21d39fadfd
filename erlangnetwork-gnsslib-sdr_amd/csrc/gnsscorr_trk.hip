// gnsscorr_trk.hip -- E/P/L tracking correlators with carrier wipe-off for
// gfx950 (MI355X).
//
// Replaces correlator() = mixcarr + rescode + dot_22/dot_23 of the reference
// (ref src/sdrcmn.c:608-722) as driven by sdrtracking() (ref
// src/sdrtrk.c:31-43), for every (channel, code period) of a batch in one
// launch.
//
//   trk_plan : one lane per channel walks the batch's code periods and emits
//              (buffloc, currnsamp, code phase, carrier phase) per period --
//              the closed-form NCO chain of sdrtracking()/mixcarr()/rescode().
//   trk_corr : one 256-thread workgroup per (channel, period).  The period's
//              resampled +-1 replica is built once in LDS, the int8 IF window
//              is streamed from the HBM ring with 16-byte coalesced loads
//              (aligned down; head/tail samples masked), the 32-step carrier
//              LUT is applied with v_dot4_i32_i8 straight on the packed
//              samples, and the 2*(1+2*corrn) int32 accumulators are reduced
//              across the wavefront and the workgroup.  All sums are exact
//              integers (|sum| < 2^31), scaled by 1/32 at the end like the
//              reference's CSCALE.
#include "gnsscorr_internal.h"

namespace {

// ---------------------------------------------------------------------------
// closed-form NCO chain (must stay operation-for-operation identical to
// oracle/gnss_oracle.c: orc_mixcarr_cf / orc_rescode_cf / orc_sdrtracking)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double gc_carrier_rem(double phi0, double freq, double ti, int n)
{
    const double phis = __ddiv_rn(__dmul_rn(phi0, (double)GC_CDIV), GC_DPI);
    const double ps = __dmul_rn(__dmul_rn(freq, (double)GC_CDIV), ti);
    double prem = __ddiv_rn(__dmul_rn(__fma_rn((double)n, ps, phis), GC_DPI), (double)GC_CDIV);
    if (prem > GC_DPI) prem = __fma_rn(-floor(__ddiv_rn(prem, GC_DPI)), GC_DPI, prem);
    return prem;
}

// start offset of the resampled replica: coff - smax*ci reduced to [0,len)
__device__ __forceinline__ double gc_code_start(double coff, int smax, double ci, int len)
{
    double cs = __dsub_rn(coff, __dmul_rn((double)smax, ci));
    cs = __dsub_rn(cs, __dmul_rn(floor(__ddiv_rn(cs, (double)len)), (double)len));
    return cs;
}

__device__ __forceinline__ double gc_code_rem(double coff, int smax, double ci, int len, int n)
{
    const double cs = gc_code_start(coff, smax, ci, len);
    const int nt = n + 2 * smax;
    double wraps = 0.0;
    if (nt > 0) wraps = (double)((long long)__fma_rn((double)(nt - 1), ci, cs) / len);
    const double cend = __dsub_rn(__fma_rn((double)nt, ci, cs), __dmul_rn(wraps, (double)len));
    return __dsub_rn(cend, __dmul_rn((double)smax, ci));
}

__global__ void trk_plan_kernel(const GcChan *__restrict__ chan, GcTrkState *__restrict__ state,
                                GcTrkPlan *__restrict__ plan, int nch, int nepoch)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nch) return;
    const GcChan c = chan[ch];
    GcTrkState s = state[ch];
    const double ci = __dmul_rn(c.ti, s.codefreq);
    const double spc = __ddiv_rn(s.codefreq, c.f_sf);      // chips per sample
    for (int e = 0; e < nepoch; e++) {
        // ref src/sdrtrk.c:31-32
        const int n = (int)__ddiv_rn(__dsub_rn((double)c.clen, s.remcode), spc);
        GcTrkPlan p;
        p.buffloc = s.buffloc;
        p.coff = s.remcode;
        p.phi0 = s.remcarr;
        p.carrfreq = s.carrfreq;
        p.codefreq = s.codefreq;
        p.n = n;
        p.pad = 0;
        plan[(size_t)ch * nepoch + e] = p;
        s.remcarr = gc_carrier_rem(s.remcarr, s.carrfreq, c.ti, n);
        s.remcode = gc_code_rem(s.remcode, c.smax, ci, c.clen, n);
        s.buffloc += (uint64_t)(int64_t)n;
    }
    state[ch] = s;
}

// ---------------------------------------------------------------------------
// correlator
// ---------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// carrier LUT: cost[i] = floor(32 cos(2 pi i/32) + 0.5) (ref src/sdrcmn.c:643-648)
__constant__ signed char kCos32[32] = {32, 31, 30, 27, 23, 18, 12, 6, 0, -6, -12, -18, -23, -27, -30, -31,
                                       -32, -31, -30, -27, -23, -18, -12, -6, 0, 6, 12, 18, 23, 27, 30, 31};
__constant__ signed char kSin32[32] = {0, 6, 12, 18, 23, 27, 30, 31, 32, 31, 30, 27, 23, 18, 12, 6,
                                       0, -6, -12, -18, -23, -27, -30, -31, -32, -31, -30, -27, -23, -18, -12, -6};

template <int DTYPE, int NTAP>
__global__ __launch_bounds__(256) void trk_corr_kernel(const GcChan *__restrict__ chan,
                                                       const GcTrkPlan *__restrict__ plan,
                                                       double *__restrict__ corrI,
                                                       double *__restrict__ corrQ,
                                                       int *__restrict__ nsamp_out, int nepoch,
                                                       int ntap_stride, int ntap_lo, int max_n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int e = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    const GcChan &c = chan[ch];
    const int ntap = c.ntap;
    // this instantiation serves channels with ntap in (ntap_lo, NTAP] and this dtype
    if (c.dtype != DTYPE || ntap > NTAP || ntap <= ntap_lo) return;

    const GcTrkPlan p = plan[(size_t)ch * nepoch + e];
    const int n = p.n, smax = c.smax, clen = c.clen;
    const size_t obase = ((size_t)ch * nepoch + e) * ntap_stride;
    if (tid == 0 && nsamp_out) nsamp_out[(size_t)ch * nepoch + e] = n;
    if (n <= 0 || n > max_n) {          // outside the reference's (nsamp+100) scratch
        if (tid < ntap) { corrI[obase + tid] = 0.0; corrQ[obase + tid] = 0.0; }
        return;
    }

    // LDS carve (all offsets multiples of 16)
    uint2 *lut = reinterpret_cast<uint2 *>(smem);                    // 32 x 8 B
    int *red = reinterpret_cast<int *>(smem + 256);                  // 4 x 2*NTAP ints
    constexpr int RED_BYTES = ((4 * 2 * NTAP * 4) + 15) & ~15;
    signed char *chips = reinterpret_cast<signed char *>(smem + 256 + RED_BYTES);  // clen (<= 1024)
    signed char *rc = chips + 1024;                                  // nt + 2*GC_RCPAD

    if (tid < 32) {
        const int cs_ = kCos32[tid], sn_ = kSin32[tid];
        uint2 v;
        if (DTYPE == 2) {   // bytes [c,-s] -> I ; [s,c] -> Q for one IQ sample
            v.x = (unsigned)(cs_ & 0xFF) | ((unsigned)((-sn_) & 0xFF) << 8);
            v.y = (unsigned)(sn_ & 0xFF) | ((unsigned)(cs_ & 0xFF) << 8);
        } else {
            v.x = (unsigned)(cs_ & 0xFF);
            v.y = (unsigned)(sn_ & 0xFF);
        }
        lut[tid] = v;
    }
    for (int i = tid; i < clen; i += 256) chips[i] = c.code[i];
    __syncthreads();

    // resampled replica, ref src/sdrcmn.c:608-621 in closed form
    const double ci = __dmul_rn(c.ti, p.codefreq);
    const double cs = gc_code_start(p.coff, smax, ci, clen);
    const int nt = n + 2 * smax;
    for (int j = tid; j < nt + 2 * GC_RCPAD; j += 256) {
        const int jj = j - GC_RCPAD;
        signed char v = 0;
        if (jj >= 0 && jj < nt) {
            long long t = (long long)__fma_rn((double)jj, ci, cs);
            while (t >= clen) t -= clen;
            v = chips[(int)t];
        }
        rc[j] = v;
    }
    __syncthreads();

    // carrier NCO, ref src/sdrcmn.c:649-650
    const double phis = __ddiv_rn(__dmul_rn(p.phi0, (double)GC_CDIV), GC_DPI);
    const double ps = __dmul_rn(__dmul_rn(p.carrfreq, (double)GC_CDIV), c.ti);

    int accI[NTAP], accQ[NTAP], toff[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; t++) {
        accI[t] = 0;
        accQ[t] = 0;
        toff[t] = GC_RCPAD + smax + (t < ntap ? c.tapoff[t] : 0);
    }

    const uint64_t ringbytes = c.ringlen * (uint64_t)DTYPE;
    const uint64_t a0 = (p.buffloc % c.ringlen) * (uint64_t)DTYPE;
    const uint64_t a_al = a0 & ~(uint64_t)15;
    const int head = (int)(a0 - a_al);
    const int G = (head + n * DTYPE + 15) >> 4;
    constexpr int SPG = 16 / DTYPE;       // samples per 16-byte group
    const int8_t *ring = c.ring;

    for (int g = tid; g < G; g += 256) {
        uint64_t addr = a_al + (uint64_t)g * 16;
        if (addr >= ringbytes) addr -= ringbytes;
        const uint4 v = *reinterpret_cast<const uint4 *>(ring + addr);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        const int kb = (g * 16 - head) / DTYPE;   // exact: head is a multiple of DTYPE
#pragma unroll
        for (int i = 0; i < SPG; i++) {
            const int k = kb + i;
            const bool valid = (unsigned)k < (unsigned)n;
            const double phi = __fma_rn((double)k, ps, phis);
            const int idx = ((int)phi) & (GC_CDIV - 1);
            const uint2 l = lut[idx];
            int I, Q;
            if (DTYPE == 2) {
                const int sh = (i & 1) * 16;
                const unsigned bI = valid ? (l.x << sh) : 0u, bQ = valid ? (l.y << sh) : 0u;
                I = __builtin_amdgcn_sdot4((int)w[i >> 1], (int)bI, 0, false);
                Q = __builtin_amdgcn_sdot4((int)w[i >> 1], (int)bQ, 0, false);
            } else {
                const int sh = (i & 3) * 8;
                const unsigned bI = valid ? (l.x << sh) : 0u, bQ = valid ? (l.y << sh) : 0u;
                I = __builtin_amdgcn_sdot4((int)w[i >> 2], (int)bI, 0, false);
                Q = __builtin_amdgcn_sdot4((int)w[i >> 2], (int)bQ, 0, false);
            }
#pragma unroll
            for (int t = 0; t < NTAP; t++) {
                const int pc = rc[toff[t] + k];
                accI[t] = __mul24(pc, I) + accI[t];
                accQ[t] = __mul24(pc, Q) + accQ[t];
            }
        }
    }

    // wavefront then workgroup reduction
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int t = 0; t < NTAP; t++) {
        const int si = wave_sum(accI[t]), sq = wave_sum(accQ[t]);
        if (lane == 0) {
            red[wv * 2 * NTAP + t] = si;
            red[wv * 2 * NTAP + NTAP + t] = sq;
        }
    }
    __syncthreads();
    if (tid < ntap) {
        int si = 0, sq = 0;
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) {
            si += red[w4 * 2 * NTAP + tid];
            sq += red[w4 * 2 * NTAP + NTAP + tid];
        }
        corrI[obase + tid] = (double)si * (1.0 / 32.0);   // CSCALE, ref src/sdrcmn.c:716-719
        corrQ[obase + tid] = (double)sq * (1.0 / 32.0);
    }
}

// cumsumcorr() over the batch, ref src/sdrtrk.c:64-76 (polarity +1: ocode is
// all ones, ref src/sdrinit.c:520-521); sums in epoch order.
__global__ void trk_sums_kernel(const double *__restrict__ corrI, const double *__restrict__ corrQ,
                                double *__restrict__ sumI, double *__restrict__ sumQ, int nch,
                                int nepoch, int ntap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nch * ntap) return;
    const int ch = i / ntap, t = i % ntap;
    double si = 0.0, sq = 0.0;
    for (int e = 0; e < nepoch; e++) {
        si += corrI[((size_t)ch * nepoch + e) * ntap + t];
        sq += corrQ[((size_t)ch * nepoch + e) * ntap + t];
    }
    sumI[i] = si;
    sumQ[i] = sq;
}

template <int DTYPE, int NTAP>
int launch_corr(hipStream_t st, const GcChan *chan, const GcTrkPlan *plan, double *corrI,
                double *corrQ, int *nsamp_out, int nch, int nepoch, int ntap_stride, int ntap_lo,
                int max_n, int smax_max)
{
    constexpr int RED_BYTES = ((4 * 2 * NTAP * 4) + 15) & ~15;
    const size_t lds = 256 + RED_BYTES + 1024 + (((size_t)max_n + 2 * smax_max + 2 * GC_RCPAD + 15) & ~(size_t)15);
    dim3 grid(nepoch, nch), block(256);
    hipLaunchKernelGGL((trk_corr_kernel<DTYPE, NTAP>), grid, block, lds, st, chan, plan, corrI, corrQ,
                       nsamp_out, nepoch, ntap_stride, ntap_lo, max_n);
    GC_HIP(hipGetLastError());
    return 0;
}

template <int DTYPE>
int launch_corr_taps(hipStream_t st, const GcChan *chan, const GcTrkPlan *plan, double *corrI,
                     double *corrQ, int *nsamp_out, int nch, int nepoch, int ntap_stride, int ntap,
                     int max_n, int smax_max)
{
    // smallest instantiation that holds ntap accumulators; it serves (lo, NTAP]
    if (ntap <= 3)  return launch_corr<DTYPE, 3>(st, chan, plan, corrI, corrQ, nsamp_out, nch, nepoch, ntap_stride, 0, max_n, smax_max);
    if (ntap <= 5)  return launch_corr<DTYPE, 5>(st, chan, plan, corrI, corrQ, nsamp_out, nch, nepoch, ntap_stride, 3, max_n, smax_max);
    if (ntap <= 7)  return launch_corr<DTYPE, 7>(st, chan, plan, corrI, corrQ, nsamp_out, nch, nepoch, ntap_stride, 5, max_n, smax_max);
    if (ntap <= 9)  return launch_corr<DTYPE, 9>(st, chan, plan, corrI, corrQ, nsamp_out, nch, nepoch, ntap_stride, 7, max_n, smax_max);
    if (ntap <= 13) return launch_corr<DTYPE, 13>(st, chan, plan, corrI, corrQ, nsamp_out, nch, nepoch, ntap_stride, 9, max_n, smax_max);
    if (ntap <= 21) return launch_corr<DTYPE, 21>(st, chan, plan, corrI, corrQ, nsamp_out, nch, nepoch, ntap_stride, 13, max_n, smax_max);
    return launch_corr<DTYPE, 33>(st, chan, plan, corrI, corrQ, nsamp_out, nch, nepoch, ntap_stride, 21, max_n, smax_max);
}

}  // namespace

int gc_launch_trk_plan(hipStream_t st, const GcChan *chan, GcTrkState *state, GcTrkPlan *plan,
                       int nch, int nepoch)
{
    hipLaunchKernelGGL(trk_plan_kernel, dim3((nch + 63) / 64), dim3(64), 0, st, chan, state, plan, nch,
                       nepoch);
    GC_HIP(hipGetLastError());
    return 0;
}

// One launch serves every channel whose (dtype, tap bucket) matches; callers
// invoke it once per distinct (dtype, ntap) present in the channel set.
int gc_launch_trk_corr(hipStream_t st, const GcChan *chan, const GcTrkPlan *plan, double *corrI,
                       double *corrQ, int *nsamp_out, int nch, int nepoch, int ntap_stride,
                       int dtype, int ntap, int max_n, int smax_max)
{
    if (dtype == 2)
        return launch_corr_taps<2>(st, chan, plan, corrI, corrQ, nsamp_out, nch, nepoch, ntap_stride, ntap, max_n, smax_max);
    if (dtype == 1)
        return launch_corr_taps<1>(st, chan, plan, corrI, corrQ, nsamp_out, nch, nepoch, ntap_stride, ntap, max_n, smax_max);
    return gc_fail(GNSSCORR_EINVAL, "trk_corr: dtype %d not 1 or 2", dtype);
}

int gc_launch_trk_sums(hipStream_t st, const double *corrI, const double *corrQ, double *sumI,
                       double *sumQ, int nch, int nepoch, int ntap)
{
    const int total = nch * ntap;
    hipLaunchKernelGGL(trk_sums_kernel, dim3((total + 127) / 128), dim3(128), 0, st, corrI, corrQ, sumI,
                       sumQ, nch, nepoch, ntap);
    GC_HIP(hipGetLastError());
    return 0;
}
