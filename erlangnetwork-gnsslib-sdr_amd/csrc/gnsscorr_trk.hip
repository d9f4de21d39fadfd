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
#include <cstdlib>
#include <type_traits>

#include "gnsscorr_internal.h"

namespace {

// ---------------------------------------------------------------------------
// closed-form NCO chain (the test-side CPU checker restates these operations
// one for one; any change here changes the normative closed form)
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

__global__ void trk_plan_kernel(const GcChan *__restrict__ chan, const GcTrkState *__restrict__ state_in,
                                GcTrkState *__restrict__ state_out, GcTrkPlan *__restrict__ plan, int nch,
                                int nepoch)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nch) return;
    const GcChan c = chan[ch];
    GcTrkState s = state_in[ch];
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
    state_out[ch] = s;
}

// ---------------------------------------------------------------------------
// per-unit constants
// ---------------------------------------------------------------------------
// One lane per (channel, epoch): everything the correlator workgroups would
// otherwise each recompute (ring offset, NCO start values).
__global__ void trk_expand_kernel(const GcChan *__restrict__ chan, const GcTrkPlan *__restrict__ plan,
                                  GcTrkUnit *__restrict__ unit, int *__restrict__ nsamp_out, int nch,
                                  int nepoch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nch * nepoch) return;
    const GcChan &c = chan[i / nepoch];
    const GcTrkPlan p = plan[i];
    GcTrkUnit u;
    const uint64_t a0 = (p.buffloc % c.ringlen) * (uint64_t)c.dtype;
    u.a_al = a0 & ~(uint64_t)15;
    u.head = (int)(a0 - u.a_al);
    u.n = p.n;
    u.G = (u.head + p.n * c.dtype + 15) >> 4;
    u.nt = p.n + 2 * c.smax;
    u.ci = __dmul_rn(c.ti, p.codefreq);
    u.cs = gc_code_start(p.coff, c.smax, u.ci, c.clen);
    gc_carrier_fx(p.phi0, p.carrfreq, c.ti, &u.phi_fx, &u.ps_fx, &u.kflip, &u.neg);
    unit[i] = u;
    if (nsamp_out) nsamp_out[i] = p.n;
}

// ---------------------------------------------------------------------------
// correlator
// ---------------------------------------------------------------------------
// sum over the wavefront, valid in lane 63 (row scans, then row broadcasts)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_add(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ int wave_sum63(int v)
{
    v = dpp_add<0x111, 0xF>(v);     // row_shr:1
    v = dpp_add<0x112, 0xF>(v);     // row_shr:2
    v = dpp_add<0x114, 0xF>(v);     // row_shr:4
    v = dpp_add<0x118, 0xF>(v);     // row_shr:8
    v = dpp_add<0x142, 0xA>(v);     // row_bcast:15
    v = dpp_add<0x143, 0xC>(v);     // row_bcast:31
    return v;
}

// carrier LUT: cost[i] = floor(32 cos(2 pi i/32) + 0.5) (ref src/sdrcmn.c:643-648)
__constant__ signed char kCos32[32] = {32, 31, 30, 27, 23, 18, 12, 6, 0, -6, -12, -18, -23, -27, -30, -31,
                                       -32, -31, -30, -27, -23, -18, -12, -6, 0, 6, 12, 18, 23, 27, 30, 31};
__constant__ signed char kSin32[32] = {0, 6, 12, 18, 23, 27, 30, 31, 32, 31, 30, 27, 23, 18, 12, 6,
                                       0, -6, -12, -18, -23, -27, -30, -31, -32, -31, -30, -27, -23, -18, -12, -6};

typedef short gc_s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dot2(unsigned a, unsigned b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(gc_s2, a), __builtin_bit_cast(gc_s2, b), c, false);
}
// a . b over four int8 lanes with a literal zero accumulator (VOP3P form: no register to clear)
__device__ __forceinline__ int dot4z(unsigned a, unsigned b)
{
    return __builtin_amdgcn_sdot4((int)a, (int)b, 0, false);
}

// Workgroup (seg, epoch, channel): correlates the 16-byte sample groups
// [seg*256*NIT, (seg+1)*256*NIT) of one code period against every tap and
// writes its 2*ntap int32 partial sums.
//
// LDS holds the period's resampled replica for this segment as one dword per
// sample position, (chip(j), chip(j+1)) as two int16, so that the pair a tap
// needs for samples (k, k+1) is a single aligned dword whatever the tap
// offset: the taps then cost one v_dot2_i32_i16 per two samples and rail.
// The image is stored transposed -- position p at row p%8, column p/8 -- because
// lane l works on samples 8l..8l+7: for a given tap and sample pair the 64 lanes
// then read 64 consecutive dwords of one row (no bank conflicts) instead of
// dwords 8 apart (8-way conflict).
template <int DTYPE, int NTAP, int NIT>
__global__ __launch_bounds__(256) void trk_corr_kernel(const GcChan *__restrict__ chan,
                                                       const GcTrkUnit *__restrict__ unit,
                                                       int *__restrict__ partial, int nch, int nepoch, int nseg,
                                                       int ntap_stride, int ntap_lo, int max_n, int ablate)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPG = 16 / DTYPE;                 // samples per 16-byte group
    constexpr int SEGG = 256 * NIT;                 // groups per segment
    constexpr int SEGS = SEGG * SPG;                // samples per segment
    // Workgroup order (speed only, never correctness): blocks b and b+8 tend to share an XCD, so
    // every 8th block walks one epoch's channels and segments back to back -- the epoch's IF
    // window is then fetched from HBM once and served to the other channels by that XCD's L2.
    const int tid = threadIdx.x;
    const int per_epoch = nch * nseg;
    const int slot = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int e = (qq / per_epoch) * 8 + slot, rr = qq % per_epoch;
    const int ch = rr / nseg, seg = rr % nseg;
    if (e >= nepoch) return;
    const GcChan &c = chan[ch];
    const int ntap = c.ntap;
    // this instantiation serves channels with ntap in (ntap_lo, NTAP] and this dtype
    if (c.dtype != DTYPE || ntap > NTAP || ntap <= ntap_lo) return;

    const GcTrkUnit u = unit[(size_t)ch * nepoch + e];
    const int n = u.n, smax = c.smax, clen = c.clen, head = u.head, G = u.G;
    int *pout = partial + (((size_t)ch * nepoch + e) * nseg + seg) * 2 * ntap_stride;
    const int g0 = seg * SEGG;
    // outside the reference's (nsamp+100) scratch, or nothing left for this segment
    if (n <= 0 || n > max_n || g0 >= G) {
        if (tid < 2 * ntap_stride) pout[tid] = 0;
        return;
    }
    const int klo = (g0 * 16 - head) / DTYPE;       // first sample index of the segment (may be < 0)

    // LDS carve (all offsets multiples of 16)
    // carrier LUT, one copy per sample position inside a dword (2 for IQ, 4 for real samples), each
    // entry = the two int8 operand words that turn v_dot4 on the packed samples into I and Q
    constexpr int LUTPOS = DTYPE == 2 ? 2 : 4;
    constexpr int LUT_BYTES = 32 * 8 * LUTPOS;
    uint2 *lut = reinterpret_cast<uint2 *>(smem);
    int *red = reinterpret_cast<int *>(smem + LUT_BYTES);            // 4 x 2*NTAP ints
    constexpr int RED_BYTES = ((4 * 2 * NTAP * 4) + 15) & ~15;
    unsigned *rcp = reinterpret_cast<unsigned *>(smem + LUT_BYTES + RED_BYTES);

    if (tid < 32 * LUTPOS) {
        const int idx = tid & 31, pos = tid >> 5;
        const int cs_ = kCos32[idx], sn_ = kSin32[idx];
        uint2 v;
        if (DTYPE == 2) {   // bytes [c,-s] -> I ; [s,c] -> Q for one IQ sample
            v.x = ((unsigned)(cs_ & 0xFF) | ((unsigned)((-sn_) & 0xFF) << 8)) << (16 * pos);
            v.y = ((unsigned)(sn_ & 0xFF) | ((unsigned)(cs_ & 0xFF) << 8)) << (16 * pos);
        } else {
            v.x = (unsigned)(cs_ & 0xFF) << (8 * pos);
            v.y = (unsigned)(sn_ & 0xFF) << (8 * pos);
        }
        lut[tid] = v;
    }

    // IF samples of this lane's groups: issued now, consumed after the replica is built, so the
    // HBM/L2 latency hides behind the fill phase
    const gc_gptr_i8 ring = (gc_gptr_i8)c.ring;
    const uint64_t ringbytes = c.ringlen * (uint64_t)DTYPE;
    uint4 vdata[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int g = g0 + tid + 256 * it;
        uint64_t addr = u.a_al + (uint64_t)(g < G ? g : g0) * 16;
        if (addr >= ringbytes) addr -= ringbytes;
        const gc_u4v t4 = *(gc_gptr_u4)(ring + addr);
        vdata[it] = make_uint4(t4.x, t4.y, t4.z, t4.w);
    }

    // ---- resampled replica, ref src/sdrcmn.c:608-621 in closed form --------------------------
    // position j of the replica (j = smax + k + tap offset) lives at rcp[j - klo]; chip index
    // T(j) = trunc(fma(j, ci, cs)) is non-decreasing in j, so a 17-position task needs two
    // evaluations plus a 4-step bisection when it holds one chip edge.
    const double ci = u.ci, cs = u.cs;
    const int nt = u.nt;
    const gc_gptr_i8 code = (gc_gptr_i8)c.code;
    const int npos = SEGS + 2 * smax + 1;           // positions this segment can touch
    constexpr int RS = SEGS / 8 + 64;               // row stride (dwords) of the transposed image
    auto chipT = [&](int j) -> int { return (int)(long long)__fma_rn((double)j, ci, cs); };
    auto chipS = [&](int T) -> int { while (T >= clen) T -= clen; return (int)code[T]; };
    for (int q = tid; q * 16 < npos && !(ablate & 1); q += 256) {
        const int j0 = klo + q * 16;
        unsigned w[16];
        if (j0 >= 0 && j0 + 16 < nt) {
            const int T0 = chipT(j0), T1 = chipT(j0 + 16);
            if (T1 - T0 <= 1) {
                int lo = 0, hi = 16;                 // T(j0+lo) == T0, T(j0+hi) == T1
                if (T1 != T0) {
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        const int mid = (lo + hi) >> 1;
                        if (chipT(j0 + mid) == T0) lo = mid; else hi = mid;
                    }
                } else {
                    hi = 17;
                }
                const unsigned sa = (unsigned)chipS(T0) & 0xFFFFu, sb = (unsigned)chipS(T1) & 0xFFFFu;
                const unsigned AA = sa | (sa << 16), AB = sa | (sb << 16), BB = sb | (sb << 16);
#pragma unroll
                for (int i = 0; i < 16; i++)         // pair (s(j0+i), s(j0+i+1)); first index holding sb is hi
                    w[i] = (i + 1 < hi) ? AA : (i + 1 == hi ? AB : BB);
            } else {
                int prev = chipS(T0);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int nx = chipS(chipT(j0 + i + 1));
                    w[i] = ((unsigned)prev & 0xFFFFu) | ((unsigned)nx << 16);
                    prev = nx;
                }
            }
        } else {                                    // task touches the ends of the replica
            int prev = (j0 >= 0 && j0 < nt) ? chipS(chipT(j0)) : 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int j = j0 + i + 1;
                const int nx = (j >= 0 && j < nt) ? chipS(chipT(j)) : 0;
                w[i] = ((unsigned)prev & 0xFFFFu) | ((unsigned)nx << 16);
                prev = nx;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) rcp[(i & 7) * RS + 2 * q + (i >> 3)] = w[i];   // position 16q+i
    }
    __syncthreads();

    // carrier NCO: the truncation bias of a negative phase is folded into the start value unless
    // the phase changes sign inside this period (then it is chosen per sample)
    const bool flip = u.kflip < n;
    const unsigned long long ps = u.ps_fx;
    const unsigned long long phi0 = u.phi_fx + ((!flip && (u.neg & 1)) ? GC_FX_BIAS : 0ULL);
    int accI[NTAP], accQ[NTAP], toff[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; t++) {
        accI[t] = 0;
        accQ[t] = 0;
        toff[t] = smax + (t < ntap ? c.tapoff[t] : 0);
    }

    // the sign-flip variant (a carrier phase that crosses zero inside the period) is a separate copy
    // of the loop, chosen once per workgroup, so the common copy stays one basic block per group
    auto run = [&](auto flip_tag) {
        constexpr bool FLIP = decltype(flip_tag)::value;
    #pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int gl = tid + 256 * it, g = g0 + gl;
            if (g >= G || (ablate & 2)) break;
            uint4 v = vdata[it];
            const int kb = (g * 16 - head) / DTYPE;     // exact: head is a multiple of DTYPE
            const bool edge = kb < 0 || kb + SPG > n;
            if (__ballot(edge) != 0ULL) {               // only the period's first / last wavefront
                if (edge) {                             // blank the samples outside [0, n)
                    unsigned m[4];
    #pragma unroll
                    for (int d = 0; d < 4; d++) {
                        m[d] = 0;
    #pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int k = kb + (d * 4 + b) / DTYPE;
                            if (k >= 0 && k < n) m[d] |= 0xFFu << (8 * b);
                        }
                    }
                    v.x &= m[0]; v.y &= m[1]; v.z &= m[2]; v.w &= m[3];
                }
            }
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
            unsigned ip[SPG / 2], qp[SPG / 2];
            unsigned long long phi = phi0 + (unsigned long long)(long long)kb * ps;
    #pragma unroll
            for (int i = 0; i < SPG; i += 2) {
                int I[2], Q[2];
    #pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    unsigned long long ph = phi;
                    if (FLIP) ph += (((kb + i + s2 < u.kflip) ? (u.neg & 1) : (u.neg >> 1)) ? GC_FX_BIAS : 0ULL);
                    const int pos = DTYPE == 2 ? ((i + s2) & 1) : ((i + s2) & 3);
                    const uint2 l = lut[32 * pos + (int)(ph >> 59)];
                    const unsigned wd = w[DTYPE == 2 ? (i + s2) >> 1 : (i + s2) >> 2];
                    I[s2] = dot4z(wd, l.x);
                    Q[s2] = dot4z(wd, l.y);
                    phi += ps;
                }
                ip[i >> 1] = __builtin_amdgcn_perm((unsigned)I[1], (unsigned)I[0], 0x05040100u);
                qp[i >> 1] = __builtin_amdgcn_perm((unsigned)Q[1], (unsigned)Q[0], 0x05040100u);
            }
            const unsigned *rb = rcp + gl * (SPG / 8);
    #pragma unroll
            for (int t = 0; t < NTAP; t++) {
    #pragma unroll
                for (int j = 0; j < SPG / 2; j++) {
                    const int pj = toff[t] + 2 * j;                     // position relative to the group
                    const unsigned cp = rb[(pj & 7) * RS + (pj >> 3)];
                    accI[t] = dot2(ip[j], cp, accI[t]);
                    accQ[t] = dot2(qp[j], cp, accQ[t]);
                }
            }
        }

    };
    if (flip) run(std::true_type{}); else run(std::false_type{});

    // wavefront then workgroup reduction
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int t = 0; t < NTAP; t++) {
        const int si = wave_sum63(accI[t]), sq = wave_sum63(accQ[t]);
        if (lane == 63) {
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
        pout[tid] = si;
        pout[ntap_stride + tid] = sq;
    }
}

// Sums the segment partials of every (channel, epoch) into the correlator
// outputs (x CSCALE = 1/32, ref src/sdrcmn.c:716-719) and accumulates them over
// the batch like cumsumcorr() (ref src/sdrtrk.c:64-76; polarity +1: ocode is
// all ones, ref src/sdrinit.c:520-521).  All sums are exact integers.
__global__ __launch_bounds__(256) void trk_finish_kernel(const int *__restrict__ partial,
                                                         double *__restrict__ corrI,
                                                         double *__restrict__ corrQ,
                                                         double *__restrict__ sumI, double *__restrict__ sumQ,
                                                         int nepoch, int nseg, int ntap)
{
    __shared__ long long acc[2 * GNSSCORR_MAXTAPS];
    const int ch = blockIdx.x, tid = threadIdx.x;
    if (tid < 2 * ntap) acc[tid] = 0;
    __syncthreads();
    long long loc[2 * GNSSCORR_MAXTAPS];
    for (int t = 0; t < 2 * ntap; t++) loc[t] = 0;
    for (int e = tid; e < nepoch; e += 256) {
        const int *pp = partial + ((size_t)ch * nepoch + e) * nseg * 2 * ntap;
        for (int t = 0; t < 2 * ntap; t++) {
            int s = 0;
            for (int g = 0; g < nseg; g++) s += pp[g * 2 * ntap + t];
            const double v = (double)s * (1.0 / 32.0);
            if (t < ntap) corrI[((size_t)ch * nepoch + e) * ntap + t] = v;
            else corrQ[((size_t)ch * nepoch + e) * ntap + (t - ntap)] = v;
            loc[t] += s;
        }
    }
    for (int t = 0; t < 2 * ntap; t++) atomicAdd((unsigned long long *)&acc[t], (unsigned long long)loc[t]);
    __syncthreads();
    if (tid < ntap) {
        sumI[ch * ntap + tid] = (double)acc[tid] * (1.0 / 32.0);
        sumQ[ch * ntap + tid] = (double)acc[ntap + tid] * (1.0 / 32.0);
    }
}

int g_trk_nit = 0;      // groups per lane per segment workgroup (2, 4 or 8); 0 = not yet chosen

template <int DTYPE, int NTAP, int NIT>
int launch_corr(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, int *partial, int nch, int nepoch,
                int nseg, int ntap_stride, int ntap_lo, int max_n, int smax_max)
{
    constexpr int RED_BYTES = ((4 * 2 * NTAP * 4) + 15) & ~15;
    constexpr int SEGS = 256 * NIT * (16 / DTYPE);
    constexpr int RS = SEGS / 8 + 64;
    const int lds = (int)((DTYPE == 2 ? 512 : 1024) + RED_BYTES + 8 * RS * 4);
    static const int ablate = getenv("GNSSCORR_TRK_ABLATE") ? atoi(getenv("GNSSCORR_TRK_ABLATE")) : 0;
    if (lds > 64 * 1024)
        GC_HIP(hipFuncSetAttribute((const void *)trk_corr_kernel<DTYPE, NTAP, NIT>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const long long total = 8LL * ((nepoch + 7) / 8) * nch * nseg;
    if (total > 0x7fffffffLL) return gc_fail(GNSSCORR_EINVAL, "trk_corr: batch too large (%lld workgroups)", total);
    dim3 grid((unsigned)total), block(256);
    hipLaunchKernelGGL((trk_corr_kernel<DTYPE, NTAP, NIT>), grid, block, lds, st, chan, unit, partial, nch, nepoch,
                       nseg, ntap_stride, ntap_lo, max_n, ablate);
    GC_HIP(hipGetLastError());
    return 0;
}

template <int DTYPE, int NTAP>
int launch_corr_nit(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, int *partial, int nch, int nepoch,
                    int nseg, int ntap_stride, int ntap_lo, int max_n, int smax_max)
{
    switch (g_trk_nit) {
    default: return launch_corr<DTYPE, NTAP, 2>(st, chan, unit, partial, nch, nepoch, nseg, ntap_stride, ntap_lo, max_n, smax_max);
    case 8: return launch_corr<DTYPE, NTAP, 8>(st, chan, unit, partial, nch, nepoch, nseg, ntap_stride, ntap_lo, max_n, smax_max);
    case 4: return launch_corr<DTYPE, NTAP, 4>(st, chan, unit, partial, nch, nepoch, nseg, ntap_stride, ntap_lo, max_n, smax_max);
    }
}

template <int DTYPE>
int launch_corr_taps(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, int *partial, int nch, int nepoch,
                     int nseg, int ntap_stride, int ntap, int max_n, int smax_max)
{
    // smallest instantiation that holds ntap accumulators; it serves (lo, NTAP]
#define GC_LC(N, LO) return launch_corr_nit<DTYPE, N>(st, chan, unit, partial, nch, nepoch, nseg, ntap_stride, LO, max_n, smax_max)
    if (ntap <= 3)  GC_LC(3, 0);
    if (ntap <= 5)  GC_LC(5, 3);
    if (ntap <= 7)  GC_LC(7, 5);
    if (ntap <= 13) GC_LC(13, 7);
    if (ntap <= 21) GC_LC(21, 13);
    GC_LC(33, 21);
#undef GC_LC
}

void trk_pick_nit()
{
    if (g_trk_nit) return;
    const char *e = getenv("GNSSCORR_TRK_NIT");
    g_trk_nit = e ? atoi(e) : 2;
    if (g_trk_nit != 2 && g_trk_nit != 4 && g_trk_nit != 8) g_trk_nit = 2;
}

}  // namespace

int gc_trk_nseg(int dtype, int max_n)
{
    trk_pick_nit();
    const int groups = (15 + max_n * dtype + 15) / 16 + 1;
    return (groups + 256 * g_trk_nit - 1) / (256 * g_trk_nit);
}

int gc_launch_trk_plan(hipStream_t st, const GcChan *chan, const GcTrkState *state_in, GcTrkState *state_out,
                       GcTrkPlan *plan, int nch, int nepoch)
{
    hipLaunchKernelGGL(trk_plan_kernel, dim3((nch + 63) / 64), dim3(64), 0, st, chan, state_in, state_out, plan,
                       nch, nepoch);
    GC_HIP(hipGetLastError());
    return 0;
}

int gc_launch_trk_expand(hipStream_t st, const GcChan *chan, const GcTrkPlan *plan, GcTrkUnit *unit,
                         int *nsamp_out, int nch, int nepoch)
{
    const int total = nch * nepoch;
    hipLaunchKernelGGL(trk_expand_kernel, dim3((total + 255) / 256), dim3(256), 0, st, chan, plan, unit, nsamp_out,
                       nch, nepoch);
    GC_HIP(hipGetLastError());
    return 0;
}

// One launch serves every channel whose (dtype, tap bucket) matches; callers
// invoke it once per distinct dtype present in the channel set.
int gc_launch_trk_corr(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, int *partial, int nch,
                       int nepoch, int nseg, int ntap_stride, int dtype, int ntap, int max_n, int smax_max)
{
    trk_pick_nit();
    if (smax_max > 64) return gc_fail(GNSSCORR_EINVAL, "trk_corr: tap offset %d samples (<= 64 supported)", smax_max);
    if (dtype == 2)
        return launch_corr_taps<2>(st, chan, unit, partial, nch, nepoch, nseg, ntap_stride, ntap, max_n, smax_max);
    if (dtype == 1)
        return launch_corr_taps<1>(st, chan, unit, partial, nch, nepoch, nseg, ntap_stride, ntap, max_n, smax_max);
    return gc_fail(GNSSCORR_EINVAL, "trk_corr: dtype %d not 1 or 2", dtype);
}

int gc_launch_trk_finish(hipStream_t st, const int *partial, double *corrI, double *corrQ, double *sumI,
                         double *sumQ, int nch, int nepoch, int nseg, int ntap)
{
    hipLaunchKernelGGL(trk_finish_kernel, dim3(nch), dim3(256), 0, st, partial, corrI, corrQ, sumI, sumQ, nepoch,
                       nseg, ntap);
    GC_HIP(hipGetLastError());
    return 0;
}
