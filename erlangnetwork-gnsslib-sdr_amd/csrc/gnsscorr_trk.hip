// gnsscorr_trk.hip -- E/P/L tracking correlators with carrier wipe-off for
// gfx950 (MI355X).
//
// Replaces correlator() = mixcarr + rescode + dot_22/dot_23 of the reference
// (ref src/sdrcmn.c:608-722) as driven by sdrtracking() (ref
// src/sdrtrk.c:31-43), for every (channel, code period) of a batch in one
// launch.
//
//   (the planner -- the exact NCO chain from period to period -- is gnsscorr_plan.hip)
//   trk_expand / trk_edges : per (channel, period) the NCO piece tables, ring offsets, chip-edge ranges and
//              the start samples of the chip edges.
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

#ifdef GC_LOOP_DEBUG
// progress marks of the closed-loop kernel in host-visible memory (tools/debug/loop_marks.py)
#include <hip/hip_runtime.h>
__device__ unsigned long long *gc_dbg_marks = nullptr;
#define GC_DBG_MARK(slot, value) do { if (gc_dbg_marks && blockIdx.x == 0 && (threadIdx.x & 63) == 0) { \
        __hip_atomic_store(&gc_dbg_marks[slot], (unsigned long long)(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } } while (0)
extern "C" int gnsscorr_debug_marks(void **host)
{
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, 4096, hipHostMallocMapped) != hipSuccess) return -1;
    for (int i = 0; i < 512; i++) ((unsigned long long *)h)[i] = 0;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(gc_dbg_marks), &d, sizeof(d)) != hipSuccess) return -1;
    *host = h;
    return 0;
}
#endif

#include "gnsscorr_internal.h"

namespace {


// rounds per workgroup of the prefix-sum correlator: a whole period when it fits GC_MAXR rounds
__host__ __device__ inline int trk_ps_rounds(int dtype, int max_n, int nit)
{
    const int groups = (15 + max_n * dtype + 15) / 16 + 1;
    const int rounds = (groups + 256 * nit - 1) / (256 * nit);
    const int nseg = (rounds + GC_MAXR - 1) / GC_MAXR;
    return (rounds + nseg - 1) / nseg;
}
// groups per lane and round: real (1-byte) samples carry 16 running sums per group, one group keeps
// the LDS image small
__host__ __device__ inline int trk_ps_nit(int dtype, int nit) { return dtype == 1 ? 1 : nit; }

// ---------------------------------------------------------------------------
// per-unit constants and NCO tables
// ---------------------------------------------------------------------------
// One lane per (channel, epoch): ring offset, the two NCOs of the period as piece tables, and per
// round of the correlator the chip edges its samples can touch.
__global__ void trk_expand_kernel(const GcChan *__restrict__ chan, const GcTrkPlan *__restrict__ plan,
                                  GcTrkUnit *__restrict__ unit, GcUnitSegs *__restrict__ segs,
                                  int *__restrict__ nsamp_out, int nch, int nepoch,
                                  GcRound *__restrict__ rounds, int nseg, int max_n, int nit,
                                  int *__restrict__ nco_overflow)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nch * nepoch) return;
    const GcChan &c = chan[i / nepoch];
    const GcTrkPlan p = plan[i];
    GcUnitSegs *sg = segs + i;
    GcTrkUnit u;
    const uint64_t a0 = (p.buffloc % c.ringlen) * (uint64_t)c.dtype;
    u.a_al = a0 & ~(uint64_t)15;
    u.head = (int)(a0 - u.a_al);
    u.n = p.n;
    u.G = (u.head + p.n * c.dtype + 15) >> 4;
    u.nt = p.n + 2 * c.smax;
    u.ncar = 0;
    u.ncode = 0;
    u.eq0 = -1;
    u.eq1 = -1;
    if (nsamp_out) nsamp_out[i] = p.n;
    const double ci = __dmul_rn(c.ti, p.codefreq);
    // outside the reference's (nsamp+100) scratch (ref src/sdrtrk.c:23), or a chip step for which its
    // one-subtraction code wrap (src/sdrcmn.c:617) is undefined: nothing is correlated
    if (!(p.n > 0 && p.n <= max_n && ci > 0.0 && ci < (double)c.clen)) {
        u.n = 0;
        unit[i] = u;
        return;
    }
    GcCarTable ct{sg->carK0, sg->car, GC_NCAR, 0, 0};
    GcCodeTable dt{sg->code, GC_NCODE, 0, 0};
    {
        GcNcoFast f;
        gc_fast_init(f, gc_carrier_ps(p.carrfreq, c.ti));
        gc_fast_carrier_walk(f, gc_carrier_phis(p.phi0), p.n, ct);
        gc_fast_init(f, ci);
        gc_fast_code_walk(f, gc_code_start(p.coff, c.smax, ci, c.clen), c.clen, u.nt, dt);
    }
    u.ncar = ct.n;
    u.ncode = dt.n;
    // what the correlator's scans rely on: carrier pieces start at sample 0 and at increasing samples, code pieces
    // are non-empty, contiguous and cover the nt replica positions (anything else is reported, never correlated)
    bool bad = ct.overflow || dt.overflow || ct.n < 1 || dt.n < 1;
    if (!bad) {
        bad = sg->carK0[0] != 0;
        for (int q = 0; q + 1 < ct.n; q++) bad = bad || !(sg->carK0[q] < sg->carK0[q + 1]);
        for (int q = 0; q < dt.n; q++)
            bad = bad || sg->code[q].cnt <= 0 || sg->code[q].j0 + sg->code[q].cnt != (q + 1 < dt.n ? sg->code[q + 1].j0 : u.nt);
    }
    if (bad) {
        if (nco_overflow) atomicAdd(nco_overflow, 1);
        u.n = 0;
        unit[i] = u;
        return;
    }
    unit[i] = u;

    // rounds of the prefix-sum correlator (same geometry as trk_corr_ps_kernel): round r of workgroup
    // seg covers samples [kl, kl + rsamp) of the period and can touch the chips T(first sample) ..
    // T(last sample + 2 smax); rank[] turns those into positions in the code's edge list
    if (!rounds) return;
    const int nitc = trk_ps_nit(c.dtype, nit), rgrp = 256 * nitc, rsamp = rgrp * (16 / c.dtype);
    const int rpw = trk_ps_rounds(c.dtype, max_n, nitc);
    const unsigned short *rank = (const unsigned short *)(c.code + 1024);
    int eq0 = 0x7fffffff, eq1 = -1;
    for (int seg = 0; seg < nseg; seg++) {
        const int g0 = seg * rgrp * rpw;
        if (g0 >= u.G) break;
        const int klo = (g0 * 16 - u.head) / c.dtype;
        for (int r = 0; r < rpw && g0 + r * rgrp < u.G; r++) {
            const int kl = klo + r * rsamp;
            const int kfirst = kl > 0 ? kl : 0;
            const int kend = (kl + rsamp < p.n ? kl + rsamp : p.n);
            int wa = 0, wb = 0, hint = 0;
            const int ma = gc_code_chip_at(sg->code, dt.n, kfirst, &wa, &hint);
            const int mb = gc_code_chip_at(sg->code, dt.n, kend - 1 + 2 * c.smax, &wb, nullptr);
            GcRound ro;
            ro.q0 = wa * c.nedge + (int)rank[ma];
            ro.q1 = wb * c.nedge + (int)rank[mb];
            ro.clast = (short)c.code[mb];
            ro.w0 = (short)wa;
            ro.hint = hint;
            rounds[((size_t)i * nseg + seg) * GC_MAXR + r] = ro;
            eq0 = ro.q0 < eq0 ? ro.q0 : eq0;
            eq1 = ro.q1 > eq1 ? ro.q1 : eq1;
        }
    }
    // the edges whose start samples trk_edges tabulates (a period longer than the table, or replica positions
    // beyond 16 bits: the correlator finds them itself)
    if (eq1 > eq0 && eq1 - eq0 <= GC_EDGTAB && u.nt < 65535) {
        unit[i].eq0 = eq0;
        unit[i].eq1 = eq1;
    }
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

// Phase time stamps of sampled workgroups (debug builds only: -DGC_TRK_TRACE; tools/trk_trace.py)
#ifdef GC_TRK_TRACE
#define GC_TRACE_N 4096
__device__ unsigned long long gc_trk_trace[GC_TRACE_N * 12];
#define GC_STAMP(i) do { if (tr) tr[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int gnsscorr_debug_trk_trace(unsigned long long *dst)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(gc_trk_trace), sizeof(unsigned long long) * GC_TRACE_N * 12) == hipSuccess ? 0 : -1;
}
#else
#define GC_STAMP(i) do { } while (0)
#endif

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
                                                       const GcUnitSegs *__restrict__ segs,
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
#ifdef GC_TRK_TRACE
    unsigned long long *tr = nullptr;
    if (tid == 0 && (blockIdx.x % 31) == 0 && blockIdx.x / 31 < GC_TRACE_N) {
        tr = gc_trk_trace + (blockIdx.x / 31) * 12;
        tr[8] = wall_clock64();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tr[10] = ((unsigned long long)xcc << 32) | hw;
        tr[11] = blockIdx.x;
    }
#endif
    GC_STAMP(0);
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
    GC_STAMP(1);

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

    GC_STAMP(2);
    // ---- resampled replica, ref src/sdrcmn.c:608-621 -----------------------------------------
    // position j of the replica (j = smax + k + tap offset) lives at rcp[j - klo]; the unwrapped chip
    // index T(j) (the reference's truncated running sum, from the unit's code table) is non-decreasing
    // in j, so a 17-position task needs two evaluations plus a 4-step bisection when it holds one
    // chip edge.
    const int nt = u.nt;
    const gc_gptr_i8 code = (gc_gptr_i8)c.code;
    const int npos = SEGS + 2 * smax + 1;           // positions this segment can touch
    constexpr int RS = SEGS / 8 + 64;               // row stride (dwords) of the transposed image
    // the unit's NCO tables, behind the replica image
    int *sk0 = reinterpret_cast<int *>(smem + LUT_BYTES + RED_BYTES + 8 * RS * 4);
    GcCarSeg *scar = reinterpret_cast<GcCarSeg *>(sk0 + ((GC_NCAR + 4) & ~3));
    GcCodeSeg *scode = reinterpret_cast<GcCodeSeg *>(scar + GC_NCAR);
    const int ncar = u.ncar, ncode = u.ncode;
    {
        const GcUnitSegs *gs = segs + ((size_t)ch * nepoch + e);
        if (tid < ncar) { sk0[tid] = gs->carK0[tid]; scar[tid] = gs->car[tid]; }
        if (tid >= 64 && tid - 64 < ncode) scode[tid - 64] = gs->code[tid - 64];
    }
    __syncthreads();
    auto chipT = [&](int j) -> int {
        int w = 0;
        const int chip = gc_code_chip_at(scode, ncode, j, &w);
        return chip + w * clen;
    };
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
    GC_STAMP(3);
    __syncthreads();
    GC_STAMP(4);

    int accI[NTAP], accQ[NTAP], toff[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; t++) {
        accI[t] = 0;
        accQ[t] = 0;
        toff[t] = smax + (t < ntap ? c.tapoff[t] : 0);
    }

    // (this form looks every sample's LUT index up in the carrier table on its own: it is the
    // independent cross-check of the production kernel, not a fast path)
    auto run = [&]() {
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
    #pragma unroll
            for (int i = 0; i < SPG; i += 2) {
                int I[2], Q[2];
    #pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    const int pos = DTYPE == 2 ? ((i + s2) & 1) : ((i + s2) & 3);
                    const int kk = kb + i + s2;
                    const int idx = gc_carrier_idx_at(sk0, scar, ncar, kk < 0 ? 0 : kk);
                    const uint2 l = lut[32 * pos + idx];
                    const unsigned wd = w[DTYPE == 2 ? (i + s2) >> 1 : (i + s2) >> 2];
                    I[s2] = dot4z(wd, l.x);
                    Q[s2] = dot4z(wd, l.y);
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
    run();
    GC_STAMP(5);

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
    GC_STAMP(6);
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
#ifdef GC_TRK_TRACE
    if (tr) { tr[7] = __builtin_readcyclecounter(); tr[9] = wall_clock64(); }
#endif
}

// ---------------------------------------------------------------------------
// correlator, prefix-sum form
// ---------------------------------------------------------------------------
// The resampled code is piecewise constant (one chip lasts 1/ci samples), so a
// tap's sum over a stretch of S samples is
//     sum_k x[k] c[T(k + off)] = c_b P(S) + sum_{m=a+1..b} (c_{m-1} - c_m) P(B_m - off),
// where x is the carrier-mixed sample, P(e) the sum of the stretch's first e
// mixed samples, B_m = min{j : T(j) >= m} the replica position at which chip m
// starts, and a..b the chips the stretch touches.  All terms are integers and
// the identity is exact (Abel summation; sums wrap mod 2^32 and the true result
// fits), so the result is bit-identical to the sample-by-sample correlator --
// but the taps cost one prefix look-up per chip EDGE (a chip at which the code
// changes value) instead of one multiply-add per sample: the per-sample work
// left is the carrier mixing, whose chained v_dot4 accumulators ARE the
// running sums.
//
// One workgroup serves one (channel, epoch) [or a long period's share of it]
// in rounds of 256*NIT sample groups that reuse one LDS image:
//   phase A  lane L mixes its NIT consecutive groups (two chained dot4 per
//            sample) and stores the running sums loc[p][L] (p samples into
//            the lane's span; row 0 is constant zero); a DPP scan over the
//            wavefront and the per-wave totals turn the lane totals into
//            lbase[L], the sum in front of the lane's span.
//   phase B  one chip edge per thread: B_m from the closed-form code NCO (a
//            reciprocal estimate, corrected against T itself), then per tap
//            P = loc + lbase at the clamped sample position.
// Accumulators stay in registers over the rounds; one reduction at the end.
// cost[i] of the carrier LUT from immediates (no table load on the workgroup's critical path):
// eight dwords of four int8 entries each
__device__ __forceinline__ int lut_cos(int i)
{
    auto pk = [](int a, int b, int c, int d) -> unsigned {
        return (unsigned)(a & 0xFF) | ((unsigned)(b & 0xFF) << 8) | ((unsigned)(c & 0xFF) << 16) | ((unsigned)(d & 0xFF) << 24);
    };
    const unsigned w0 = pk(32, 31, 30, 27), w1 = pk(23, 18, 12, 6), w2 = pk(0, -6, -12, -18), w3 = pk(-23, -27, -30, -31);
    const unsigned w4 = pk(-32, -31, -30, -27), w5 = pk(-23, -18, -12, -6), w6 = pk(0, 6, 12, 18), w7 = pk(23, 27, 30, 31);
    const int h = i >> 2;
    const unsigned lo = (h & 2) ? ((h & 1) ? w3 : w2) : ((h & 1) ? w1 : w0);
    const unsigned hi = (h & 2) ? ((h & 1) ? w7 : w6) : ((h & 1) ? w5 : w4);
    const unsigned w = (h & 4) ? hi : lo;
    return (int)(signed char)((w >> (8 * (i & 3))) & 0xFF);
}

// start position of the chip an edge-list entry names: B = min{j : T(j) >= m in code period w}, T = the
// reference's truncated running sum.  The code table holds that sum as pieces y0 + i d: the first piece (from
// `hint` on) whose last value reaches m holds B, and inside it i = ceil((m - y0)/d), settled by two exact
// evaluations.  scode: the unit's pieces (LDS).
__device__ __forceinline__ int gc_edge_start(const GcCodeSeg *scode, int ncode, int ed, int w, int hint)
{
    const int m = (int)(short)(ed & 0xFFFF);
    const double thr = m ? (double)m : -0.5;        // chip 0: any value above -1 truncates to it
    int sp = hint;
    bool hit = false;
    while (true) {
        const int sw = scode[sp].w;
        hit = sw > w || (sw == w && scode[sp].ylast >= thr);
        if (hit || sp + 1 >= ncode) break;
        sp++;
    }
    const int j0 = scode[sp].j0;
    if (!hit) return j0 + scode[sp].cnt;            // past the replica: clamped away by the look-ups
    const double d = scode[sp].d, y0 = scode[sp].y0;
    int i = 0;
    if (scode[sp].w == w && d != 0.0 && thr > y0) {
        i = (int)ceil((thr - y0) * scode[sp].inv);
        if (i < 1) i = 1;
        if (__fma_rn((double)(i - 1), d, y0) >= thr) i--;
        else if (__fma_rn((double)i, d, y0) < thr) i++;
    }
    return j0 + i;
}

// Edge table: one workgroup per (channel, period) tabulates the start samples of the chip edges [eq0, eq1) the
// period's rounds can touch (uint16 each; GC_EDGTAB per unit), so that the correlator's look-up phase reads
// them instead of searching the code table edge by edge.  Lanes take consecutive edges: neighbours share a piece.
__global__ __launch_bounds__(256) void trk_edges_kernel(const GcChan *__restrict__ chan, const GcTrkUnit *__restrict__ unit,
                                                        const GcUnitSegs *__restrict__ segs, unsigned short *__restrict__ etab,
                                                        int nch, int nepoch)
{
    __shared__ __attribute__((aligned(16))) GcCodeSeg scode[GC_NCODE];
    const int ui = blockIdx.x, tid = threadIdx.x;
    if (ui >= nch * nepoch) return;
    const GcTrkUnit u = unit[ui];
    if (u.n <= 0 || u.eq0 < 0) return;
    const GcChan &c = chan[ui / nepoch];
    if (tid < u.ncode) scode[tid] = segs[ui].code[tid];
    __syncthreads();
    const int __attribute__((address_space(1))) *edges = (const int __attribute__((address_space(1))) *)((gc_gptr_i8)c.code + 3072);
    const int nedge = c.nedge;
    unsigned short *out = etab + (size_t)ui * GC_EDGTAB;
    for (int q = u.eq0 + tid; q < u.eq1; q += 256) {
        const int w = q / nedge, idx = q - w * nedge;
        // the piece that holds the edge: pieces are ordered by (code period, value), the test is monotone
        const int ed = edges[idx];
        const int m = (int)(short)(ed & 0xFFFF);
        const double thr = m ? (double)m : -0.5;
        int lo = 0, hi = u.ncode - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const int sw = scode[mid].w;
            if (sw > w || (sw == w && scode[mid].ylast >= thr)) hi = mid; else lo = mid + 1;
        }
        const int js = gc_edge_start(scode, u.ncode, ed, w, lo);
        out[q - u.eq0] = (unsigned short)(js < 0 ? 0 : (js > 65535 ? 65535 : js));
    }
}

template <int DTYPE, int NIT>
struct PsLayout {
    static constexpr int SPG = 16 / DTYPE;                      // samples per 16-byte group
    static constexpr int LSP = NIT * SPG;                       // samples per lane and round
    static constexpr int RGRP = 256 * NIT;                      // groups per round
    static constexpr int RSAMP = 256 * LSP;                     // samples per round
    static constexpr int LPAD = LSP + 1;                        // image stride per lane: odd in 8-byte units
    static constexpr int MAXR = GC_MAXR;                        // rounds per workgroup, at most
    static constexpr int LUTPOS = DTYPE == 2 ? 2 : 4;
    static constexpr int LUT_BYTES = 32 * 8 * LUTPOS;
    static constexpr int WT_OFF = LUT_BYTES;                    // wpre[2][8] int2 (two rounds in flight)
    static constexpr int LB_OFF = WT_OFF + 128;
    static constexpr int LOC_OFF = LB_OFF + ((257 * 8 + 15) & ~15);
    // the unit's NCO tables: carrier piece starts (+ closing sentinel), carrier pieces, code pieces
    static constexpr int K0_OFF = LOC_OFF + (((256 * LPAD + 1) * 8 + 15) & ~15);
    static constexpr int CAR_OFF = K0_OFF + (((GC_NCAR + 1) * 4 + 15) & ~15);
    static constexpr int CODE_OFF = CAR_OFF + GC_NCAR * (int)sizeof(GcCarSeg);
    static constexpr int RED_OFF = CODE_OFF + GC_NCODE * (int)sizeof(GcCodeSeg);
    static constexpr int bytes(int ntap) { return RED_OFF + 4 * 2 * ntap * 4 + 16; }
};

__device__ __forceinline__ int wave_scan(int v)     // inclusive prefix sum over the 64 lanes
{
    v = dpp_add<0x111, 0xF>(v);
    v = dpp_add<0x112, 0xF>(v);
    v = dpp_add<0x114, 0xF>(v);
    v = dpp_add<0x118, 0xF>(v);
    v = dpp_add<0x142, 0xA>(v);
    v = dpp_add<0x143, 0xC>(v);
    return v;
}

// One (channel, period) unit [or a long period's share `seg` of it] on one 256-lane workgroup: the body of
// trk_corr_ps_kernel, also called period by period from the closed-loop kernel.  smem: PsLayout bytes.
// Every lane of the workgroup must call it (it synchronises the workgroup).
template <int DTYPE, int NTAP, int NIT>
__device__ __forceinline__ void ps_unit(const GcChan &c, const GcTrkUnit &u, const GcUnitSegs *__restrict__ gs,
                                        const GcRound *__restrict__ myrounds, int *__restrict__ pout, int ntap_stride,
                                        int max_n, int rpw, int seg, int ablate, char *smem, int tid,
                                        const unsigned short *__restrict__ etab_u = nullptr)
{
    using L = PsLayout<DTYPE, NIT>;
    constexpr int SPG = L::SPG, LSP = L::LSP, RGRP = L::RGRP, RSAMP = L::RSAMP, LPAD = L::LPAD;
    const int ntap = c.ntap;
    const int n = u.n, smax = c.smax, head = u.head, G = u.G;
    const int g0 = seg * RGRP * rpw;
    // nothing to correlate (trk_expand: outside the reference's scratch, undefined chip step, NCO table
    // overflow) or nothing left for this workgroup
    if (n <= 0 || n > max_n || g0 >= G) {
        if (tid < 2 * ntap_stride) pout[tid] = 0;
        return;
    }
    const int klo = (g0 * 16 - head) / DTYPE;       // first sample index of the workgroup (may be < 0)
    int nround = (G - g0 + RGRP - 1) / RGRP;
    if (nround > rpw) nround = rpw;

    constexpr int LUTPOS = L::LUTPOS;
    uint2 *lut = reinterpret_cast<uint2 *>(smem);
    int *wpre = reinterpret_cast<int *>(smem + L::WT_OFF);            // [2][8][2]: sums of the waves in front
    int2 *lbase = reinterpret_cast<int2 *>(smem + L::LB_OFF);         // [256 + 1]
    int2 *loc = reinterpret_cast<int2 *>(smem + L::LOC_OFF);          // [256 lanes][LPAD] + closing entry
    int *red = reinterpret_cast<int *>(smem + L::RED_OFF);            // 4 x 2*NTAP
    int *sk0 = reinterpret_cast<int *>(smem + L::K0_OFF);             // [ncar] + INT_MAX
    GcCarSeg *scar = reinterpret_cast<GcCarSeg *>(smem + L::CAR_OFF);
    GcCodeSeg *scode = reinterpret_cast<GcCodeSeg *>(smem + L::CODE_OFF);

    const gc_gptr_i8 ring = (gc_gptr_i8)c.ring;
    const uint64_t ringbytes = c.ringlen * (uint64_t)DTYPE;
    const int wv = tid >> 6, lane = tid & 63;
    // A round whose 16-byte groups do not run over the end of the ring (all but one per ring
    // revolution) is loaded from a wave-uniform base plus the lane's offset, groups past the period's
    // end included: they stay inside the ring and are blanked below.
    auto load_round = [&](int r, uint4 *dst) {
        uint64_t rb = u.a_al + (uint64_t)(g0 + r * RGRP) * 16;
        if (rb >= ringbytes) rb -= ringbytes;
        if (rb + (uint64_t)RGRP * 16 <= ringbytes) {
            const gc_gptr_i8 base = ring + rb;
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const gc_u4v t4 = *(gc_gptr_u4)(base + (unsigned)(tid * NIT + it) * 16u);
                dst[it] = make_uint4(t4.x, t4.y, t4.z, t4.w);
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int g = g0 + r * RGRP + tid * NIT + it;
                uint64_t addr = u.a_al + (uint64_t)(g < G ? g : g0) * 16;
                if (addr >= ringbytes) addr -= ringbytes;
                const gc_u4v t4 = *(gc_gptr_u4)(ring + addr);
                dst[it] = make_uint4(t4.x, t4.y, t4.z, t4.w);
            }
        }
    };
    uint4 vA[NIT], vB[NIT];
    load_round(0, vA);                                  // in flight while the tables are set up

    // ---- chip edges (ref src/sdrcmn.c:608-621 in closed form) --------------------------------
    // The replica position of chip M's first sample is B_M = min{j : T(j) >= M},
    // T(j) = trunc(fma(j, ci, cs)).  Only chips at which the code changes value matter; they are
    // numbered q = period * nedge + list index, and rank[] converts a chip number into that
    // numbering; trk_expand prepared, per round, the edges [q0, q1) its samples can touch.
    const gc_gptr_i8 code = (gc_gptr_i8)c.code;
    const int __attribute__((address_space(1))) *edges = (const int __attribute__((address_space(1))) *)(code + 3072);
    const int nedge = c.nedge;
    const int ncar = u.ncar, ncode = u.ncode;
    if (gs) {           // (the closed-loop kernel's planner writes the tables straight into the LDS image)
        if (tid < ncar) { sk0[tid] = gs->carK0[tid]; scar[tid] = gs->car[tid]; }
        if (tid == ncar) sk0[tid] = 0x7fffffff;
        if (tid >= 64 && tid - 64 < ncode) scode[tid - 64] = gs->code[tid - 64];
    }
    if (tid < 32 * LUTPOS) {
        const int idx = tid & 31, pos = tid >> 5;
        const int cs_ = lut_cos(idx), sn_ = lut_cos((idx - 8) & 31);     // sin(i) = cos(i - 8)
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
    // constant part of the prefix image: entry 0 of every lane (nothing summed yet) and of the closing lane
    loc[tid * LPAD] = make_int2(0, 0);
    if (tid == 0) { loc[256 * LPAD] = make_int2(0, 0); lbase[256] = make_int2(0, 0); }   // there P = slot 4 = total
    for (int x = tid; x < 4 * 2 * NTAP; x += 256) red[x] = 0;        // waves without a chip edge skip the reduction
    if (tid < 32) wpre[tid] = 0;
    __syncthreads();

    const bool pm1 = c.pm1 != 0;
    unsigned accI[NTAP], accQ[NTAP], finI = 0, finQ = 0;
    int toff[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; t++) {
        accI[t] = 0;
        accQ[t] = 0;
        toff[t] = smax + (t < ntap ? c.tapoff[t] : 0) + klo;
    }
    int wseg = 0;                                       // wave-uniform: carrier piece of the wave's first sample
    bool busy = false;                                  // wave-uniform: this wave owned an edge in some round
    // start sample of edge q: from the unit's edge table (trk_edges) when there is one, else searched here
    const bool have_etab = etab_u != nullptr && u.eq0 >= 0;
    auto edge_js = [&](int q, int ed, int w, int hint) -> int {
        if (have_etab) return (int)etab_u[q - u.eq0];
        return gc_edge_start(scode, ncode, ed, w, hint);
    };
    auto edge_load = [&](int q, int w0, int *w) -> int {  // w0: code periods in front of the round's first edge
        q -= w0 * nedge;
        if (q >= nedge) {                                 // a round seldom spans a code period
            const int wq = q / nedge;
            q -= wq * nedge;
            w0 += wq;
        }
        *w = w0;
        return edges[q];
    };

    auto round = [&](int r, uint4 *vdata, uint4 *vnext) {
        // opaque copy of the lane id: per-lane address arithmetic stays inside the round instead of being
        // hoisted out of the loop over rounds into registers that would then spill
        int tl = tid;
        asm volatile("" : "+v"(tl));
        if (r + 1 < nround) load_round(r + 1, vnext);
        const GcRound ro = myrounds[r];
        const int rq0 = ro.q0, rq1 = ro.q1, rlast = ro.clast, rw0 = ro.w0, rhint = ro.hint;
        int q = rq0 + tl, ew = 0, ed = 0;
        const int q1 = (ablate & 1) ? 0 : rq1;
        if (q < q1) ed = edge_load(q, rw0, &ew);       // in flight during the mixing phase
        busy = busy || (rq0 + wv * 64 < q1);
        const int kl = klo + r * RSAMP;

        // ---- phase A: carrier mixing (ref src/sdrcmn.c:643-662) and running sums ------------
        int aI = 0, aQ = 0, js = 0;
        const int roff = r * RSAMP;
        const int kw = kl + wv * 64 * LSP;
        // only the wavefronts that hold the period's first or last sample see samples outside [0, n) (one in
        // the first round, one or two in the last): the others skip the blanking test altogether
        const bool ragged = !(ablate & 4) && (kw < 0 || kw + 64 * LSP > n || g0 + r * RGRP + (wv + 1) * 64 * NIT > G);
        // carrier pieces: the wave's 64 * LSP samples start in piece wseg; when no other piece starts
        // inside them (the common case -- a piece is a whole binade of the running phase) every lane
        // steps the same piece, otherwise each lane finds its own and switches where the next one starts
        // (every scan over the piece starts is bounded by the piece count: it never depends on the closing
        // sentinel alone -- an LDS read past the table returns 0 and would keep an unbounded scan going for ever)
#ifdef GC_UNBOUNDED_SCANS       // (tools/debug: the round-2 form, kept to reproduce its stall)
        while (sk0[wseg + 1] <= kw) wseg++;
#else
        while (wseg + 1 < ncar && sk0[wseg + 1] <= kw) wseg++;
#endif
        const bool onepiece = (ablate & 8) || wseg + 1 >= ncar || sk0[wseg + 1] >= kw + 64 * LSP;
        auto run = [&](auto multi_tag) {
            constexpr bool MULTI = decltype(multi_tag)::value;
            int sp = wseg, knext = 0x7fffffff;
            const int kb0 = kl + tl * LSP;
            if (MULTI) {
#ifdef GC_UNBOUNDED_SCANS
                while (sk0[sp + 1] <= kb0) sp++;
#else
                while (sp + 1 < ncar && sk0[sp + 1] <= kb0) sp++;
#endif
                knext = sp + 1 < ncar ? sk0[sp + 1] : 0x7fffffff;
            }
            unsigned long long dfx = scar[sp].dfx;
            unsigned long long phi = scar[sp].fx + (unsigned long long)(long long)(kb0 - sk0[sp]) * dfx;
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int gl = tl * NIT + it, g = g0 + r * RGRP + gl;
                uint4 v = vdata[it];
                const int kb = kl + gl * SPG;
                if (ragged) {
                    const bool edge = kb < 0 || kb + SPG > n || g >= G;
                    if (__ballot(edge) != 0ULL) {
                        if (edge) {                     // blank the samples outside [0, n)
                            unsigned m[4];
#pragma unroll
                            for (int d = 0; d < 4; d++) {
                                m[d] = 0;
#pragma unroll
                                for (int b = 0; b < 4; b++) {
                                    const int k = kb + (d * 4 + b) / DTYPE;
                                    if (k >= 0 && k < n && g < G) m[d] |= 0xFFu << (8 * b);
                                }
                            }
                            v.x &= m[0]; v.y &= m[1]; v.z &= m[2]; v.w &= m[3];
                        }
                    }
                }
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
                // the group's LUT entries first, all in flight together (the image writes below could alias
                // them as far as the compiler knows, and would otherwise serialise read - wait - write per sample)
                uint2 l[SPG];
#pragma unroll
                for (int i = 0; i < SPG; i++) {
                    if (MULTI) {
                        if (kb + i == knext) {          // the next piece starts at this sample (sp + 1 < ncar: knext is its start)
                            sp++;
                            phi = scar[sp].fx;
                            dfx = scar[sp].dfx;
                            knext = sp + 1 < ncar ? sk0[sp + 1] : 0x7fffffff;
                        }
                    }
                    const int pos = DTYPE == 2 ? (i & 1) : (i & 3);
                    l[i] = lut[32 * pos + (int)(phi >> 59)];
                    phi += dfx;
                }
#pragma unroll
                for (int i = 0; i < SPG; i++) {
                    const unsigned wd = w[DTYPE == 2 ? i >> 1 : i >> 2];
                    aI = __builtin_amdgcn_sdot4((int)wd, (int)l[i].x, aI, false);
                    aQ = __builtin_amdgcn_sdot4((int)wd, (int)l[i].y, aQ, false);
                    const int p = it * SPG + i + 1;
                    if (p < LSP) loc[tl * LPAD + p] = make_int2(aI, aQ);
                }
            }
        };
        if (!(ablate & 2)) { if (onepiece) run(std::false_type{}); else run(std::true_type{}); }
        // the start sample of this lane's chip edge (no LDS involved: overlaps the image writes)
        if (q < q1) js = edge_js(q, ed, ew, rhint) - roff;
        const int sI = wave_scan(aI), sQ = wave_scan(aQ);
        // lanes 60..63 add this wave's total into the "waves in front" sums of the later waves and
        // the grand total (slot 4): one LDS atomic per rail instead of a pass over all totals
        int *wp = wpre + (r & 1) * 16;
        {
            const int tI = __builtin_amdgcn_readlane(sI, 63), tQ = __builtin_amdgcn_readlane(sQ, 63);
            const int slot = wv + 1 + (lane - 60);
            if (lane >= 60 && slot <= 4) {
                atomicAdd(&wp[2 * slot], tI);
                atomicAdd(&wp[2 * slot + 1], tQ);
            }
        }
        lbase[tl] = make_int2(sI - aI, sQ - aQ);       // sum in front of this lane's span inside its wave
        __syncthreads();
        {
            const int2 tv = *reinterpret_cast<const int2 *>(&wp[8]);
            const int ti = __builtin_amdgcn_readfirstlane(tv.x), tq = __builtin_amdgcn_readfirstlane(tv.y);
            finI += (unsigned)rlast * (unsigned)ti;         // c_b P(S), the term of the round's last chip
            finQ += (unsigned)rlast * (unsigned)tq;
            if (tl < 16) wpre[((r + 1) & 1) * 16 + tl] = 0;   // the other copy, for the next round
        }

        // ---- phase B: one prefix look-up per chip edge and tap -------------------------------
        // (taps past ntap repeat tap 0 and are never written out; the +-1 code variant adds or
        // subtracts and doubles at the end, the general one multiplies by the step)
        auto lookups = [&](auto pm1_tag) {
            constexpr bool PM1 = decltype(pm1_tag)::value;
            while (q < q1) {
                const int dd = ed >> 16;
                const unsigned sg = (unsigned)(dd >> 31);
#pragma unroll
                for (int t = 0; t < NTAP; t++) {
                    // many taps: a compiler barrier every four keeps their look-ups from all being issued
                    // (and held in registers) before the first one is consumed
                    if (NTAP > 7 && t % 4 == 0 && t) asm volatile("" ::: "memory");
                    int ee = js - toff[t];
                    ee = ee < 0 ? 0 : (ee > RSAMP ? RSAMP : ee);
                    const int col = ee / LSP;             // the lane that owns sample ee; its image entry is ee + col
                    // running sum inside the lane + lanes in front inside the wave + waves in front
                    const int2 a = loc[ee + col], b = lbase[col];
                    const int2 w = *reinterpret_cast<const int2 *>(&wp[2 * (col >> 6)]);
                    const unsigned pI = (unsigned)(a.x + b.x + w.x), pQ = (unsigned)(a.y + b.y + w.y);
                    if (PM1) {
                        accI[t] += (pI ^ sg) - sg;
                        accQ[t] += (pQ ^ sg) - sg;
                    } else {
                        // 32-bit products kept apart from the adds: fused into v_mad_u64_u32 they would
                        // turn every accumulator into a 64-bit register pair
                        unsigned mI = (unsigned)dd * pI, mQ = (unsigned)dd * pQ;
                        asm volatile("" : "+v"(mI), "+v"(mQ));
                        accI[t] += mI;
                        accQ[t] += mQ;
                    }
                }
                q += 256;
                if (q < q1) { ed = edge_load(q, rw0, &ew); js = edge_js(q, ed, ew, rhint) - roff; }
            }
        };
        if (pm1) lookups(std::true_type{}); else lookups(std::false_type{});
        if (r + 1 < nround) __syncthreads();            // look-ups done before the image is rewritten
    };
    for (int r = 0; r < nround; r += 2) {
        round(r, vA, vB);
        if (r + 1 < nround) round(r + 1, vB, vA);
    }

    // wavefront then workgroup reduction (waves without an edge leave red[] at its initial zero)
    if (busy) {
#pragma unroll
        for (int t = 0; t < NTAP; t++) {
            const int si = wave_sum63((int)accI[t]), sq = wave_sum63((int)accQ[t]);
            if (lane == 63) {
                red[wv * 2 * NTAP + t] = si;
                red[wv * 2 * NTAP + NTAP + t] = sq;
            }
        }
    }
    __syncthreads();
    if (tid < ntap) {
        unsigned si = 0, sq = 0;
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) {
            si += (unsigned)red[w4 * 2 * NTAP + tid];
            sq += (unsigned)red[w4 * 2 * NTAP + NTAP + tid];
        }
        if (pm1) { si *= 2u; sq *= 2u; }
        pout[tid] = (int)(si + finI);
        pout[ntap_stride + tid] = (int)(sq + finQ);
    }
}

template <int DTYPE, int NTAP, int NIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NTAP <= 7 ? 4 : (NTAP <= 13 ? 3 : (NTAP <= 21 ? 2 : 1)), 8))) void trk_corr_ps_kernel(const GcChan *__restrict__ chan,
                                                          const GcTrkUnit *__restrict__ unit,
                                                          const GcUnitSegs *__restrict__ segs,
                                                          const GcRound *__restrict__ rounds,
                                                          int *__restrict__ partial, int nch, int nepoch, int nseg,
                                                          int ntap_stride, int ntap_lo, int max_n, int rpw,
                                                          int ablate, const unsigned short *__restrict__ etab)
{
    using L = PsLayout<DTYPE, NIT>;
    // static, so that every LDS address is a compile-time offset
    __shared__ __attribute__((aligned(16))) char smem[L::bytes(NTAP)];
    // Workgroup order (speed only, never correctness): blocks b and b+8 tend to share an XCD, so every 8th
    // block walks one epoch's channels back to back -- the epoch's IF window is then fetched from HBM once
    // and served to the other channels by that XCD's L2.
    const int tid = threadIdx.x;
    const int per_epoch = nch * nseg;
    const int slot = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int e = (qq / per_epoch) * 8 + slot, rr = qq % per_epoch;
    const int ch = rr / nseg, seg = rr % nseg;
    if (e >= nepoch) return;
    const GcChan &c = chan[ch];
    const int ntap = c.ntap;
    if (c.dtype != DTYPE || ntap > NTAP || ntap <= ntap_lo) return;
    const size_t ui = (size_t)ch * nepoch + e;
    const GcTrkUnit u = unit[ui];
    ps_unit<DTYPE, NTAP, NIT>(c, u, segs + ui, rounds + (ui * nseg + seg) * GC_MAXR, partial + (ui * nseg + seg) * 2 * ntap_stride,
                              ntap_stride, max_n, rpw, seg, ablate, smem, tid, etab ? etab + ui * GC_EDGTAB : nullptr);
}

// Sums the segment partials of every (channel, epoch) into the correlator
// outputs (x CSCALE = 1/32, ref src/sdrcmn.c:716-719) and accumulates them over
// the batch like cumsumcorr() (ref src/sdrtrk.c:64-76; polarity +1: ocode is
// all ones, ref src/sdrinit.c:520-521).  All sums are exact integers.
__global__ __launch_bounds__(256) void trk_finish_kernel(const int *__restrict__ partial,
                                                         double *__restrict__ corrI,
                                                         double *__restrict__ corrQ,
                                                         double *__restrict__ sumI, double *__restrict__ sumQ,
                                                         unsigned long long *__restrict__ scratch,
                                                         int nepoch, int nseg, int ntap)
{
    // block (ch, b) serves epochs [256 b, 256 b + 256) of channel ch; the batch sums meet in
    // scratch[ch] (zero between launches), and the block that arrives last converts and clears them
    __shared__ unsigned long long acc[2 * GNSSCORR_MAXTAPS];
    __shared__ int last;
    constexpr int SS = 2 * GNSSCORR_MAXTAPS + 1;
    const int ch = blockIdx.x, tid = threadIdx.x, e = blockIdx.y * 256 + tid;
    unsigned long long *sc = scratch + (size_t)ch * SS;
    if (tid < 2 * ntap) acc[tid] = 0;
    __syncthreads();
    const int *pp = partial + ((size_t)ch * nepoch + (e < nepoch ? e : 0)) * nseg * 2 * ntap;
    for (int t = 0; t < 2 * ntap; t++) {
        int s = 0;
        if (e < nepoch) {
            for (int g = 0; g < nseg; g++) s += pp[g * 2 * ntap + t];
            const double v = (double)s * (1.0 / 32.0);
            if (t < ntap) corrI[((size_t)ch * nepoch + e) * ntap + t] = v;
            else corrQ[((size_t)ch * nepoch + e) * ntap + (t - ntap)] = v;
        }
        // wavefront sum in two 16-bit halves (64 terms of either fit an int), then one LDS atomic per wave
        const int hi = wave_sum63(s >> 16), lo = wave_sum63(s & 0xFFFF);
        if ((tid & 63) == 63) atomicAdd(&acc[t], (unsigned long long)(((long long)hi << 16) + (long long)lo));
    }
    __syncthreads();
    if (tid < 2 * ntap) __hip_atomic_fetch_add(&sc[tid], acc[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned long long arrived = __hip_atomic_fetch_add(&sc[SS - 1], 1ULL, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = arrived == (unsigned long long)gridDim.y - 1;
    }
    __syncthreads();
    if (!last) return;
    if (tid < 2 * ntap) {
        const long long tot = (long long)__hip_atomic_exchange(&sc[tid], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < ntap) sumI[ch * ntap + tid] = (double)tot * (1.0 / 32.0);
        else sumQ[ch * ntap + (tid - ntap)] = (double)tot * (1.0 / 32.0);
    }
    if (tid == 0) __hip_atomic_store(&sc[SS - 1], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// closed loop: what sdrthread() does around sdrtracking() (ref src/sdrmain.c:264-312)
// ---------------------------------------------------------------------------
// ref src/sdrtrk.c:95-126 (IP = sumI[0], QP = sumQ[0] after the II/QQ swap of :42)
__device__ __forceinline__ void loop_pll(gnsscorr_loop_t *L, GcTrkState &st, int prm, double dt)
{
    GC_FP_STRICT
    const double PI = 3.1415926535897932;
    const double IP = L->sumI[0], QP = L->sumQ[0], oldIP = L->oldsumI[0], oldQP = L->oldsumQ[0];
    double carrErr;
    if (IP > 0) carrErr = atan2(QP, IP) / PI;
    else carrErr = atan2(-QP, -IP) / PI;
    const double f1 = (IP == 0) ? PI / 2 : atan(QP / IP);
    const double f2 = (oldIP == 0) ? PI / 2 : atan(oldQP / oldIP);
    double freqErr = f1 - f2;
    if (freqErr > PI / 2) freqErr = PI - freqErr;
    if (freqErr < -PI / 2) freqErr = -PI - freqErr;
    L->carrNco += L->pllaw[prm] * (carrErr - L->carrErr) + L->pllw2[prm] * dt * carrErr + L->fllw[prm] * dt * freqErr;
    st.carrfreq = L->acqfreq + L->carrNco;
    L->carrErr = carrErr;
    L->freqErr = freqErr;
}

// ref src/sdrtrk.c:135-150
__device__ __forceinline__ void loop_dll(gnsscorr_loop_t *L, GcTrkState &st, int prm, double dt)
{
    GC_FP_STRICT
    const double IE = L->sumI[L->ne], IL = L->sumI[L->nl], QE = L->sumQ[L->ne], QL = L->sumQ[L->nl];
    const double codeErr = (sqrt(IE * IE + QE * QE) - sqrt(IL * IL + QL * QL)) /
                           (sqrt(IE * IE + QE * QE) + sqrt(IL * IL + QL * QL));
    L->codeNco += L->dllaw[prm] * (codeErr - L->codeErr) + L->dllw2[prm] * dt * codeErr;
    st.codefreq = L->crate - L->codeNco + (st.carrfreq - L->f_if - L->foffset) / (L->f_cf / L->crate);
    L->codeErr = codeErr;
}

// emitters that fill the correlator's LDS tables in place (lane-uniform calls from the planning wavefront).
// The table pointers are LDS-typed (address space 3), so every access is a DS instruction: DS instructions of one
// wavefront execute in order.  Through generic pointers the stores become FLAT instructions, which reach the LDS by
// way of the texture path and may be overtaken by a DS read issued after them (CDNA ISA: FLAT completes out of
// order with DS) -- the reader (same wavefront: the rounds of the period; gc_code_chip_at) then sees the previous
// period's pieces.  DESIGN.md section 6.
#ifdef GC_LOOP_FLAT_TABLES          // (tools/debug: the round-2 form, generic pointers)
#define GC_LDS
#else
#define GC_LDS __attribute__((address_space(3)))
#endif
typedef GC_LDS int *gc_lds_int;
typedef GC_LDS GcCarSeg *gc_lds_car;
typedef GC_LDS GcCodeSeg *gc_lds_code;
struct LdsCarTable {
    gc_lds_int k0;
    gc_lds_car seg;
    int n, overflow;
    __device__ void operator()(int k, double x, double d, int)
    {
        const GcCarSeg s = gc_carseg_make(x, d);
        if (n > 0 && s.fx == 0 && s.dfx == 0 && seg[n - 1].fx == 0 && seg[n - 1].dfx == 0) return;
        if (n >= GC_NCAR) { overflow = 1; return; }
        k0[n] = k;
        seg[n].fx = s.fx;
        seg[n].dfx = s.dfx;
        n++;
    }
};
// GcCodeTable (gnsscorr_nco.h) on an LDS-typed table
struct LdsCodeTable {
    gc_lds_code seg;
    int cap, n, overflow;
    __device__ void operator()(int j, double y, double d, int count, int w)
    {
        GC_FP_STRICT
        const double yl = fma((double)(count - 1), d, y);
        if (n > 0 && seg[n - 1].w == w && seg[n - 1].y0 > -1.0 && seg[n - 1].ylast < 1.0 && y > -1.0 && yl < 1.0) {
            seg[n - 1].cnt += count;
            seg[n - 1].ylast = yl;
            seg[n - 1].d = 0.0;
            seg[n - 1].inv = 0.0;
            return;
        }
        if (n >= cap) { overflow = 1; return; }
        seg[n].y0 = y;
        seg[n].d = count > 1 ? d : 0.0;
        seg[n].inv = (count > 1 && d != 0.0) ? 1.0 / d : 0.0;
        seg[n].ylast = yl;
        seg[n].j0 = j;
        seg[n].cnt = count;
        seg[n].w = w;
        seg[n].pad = 0;
        n++;
    }
};

#define GC_LOOP_MAXSEG 4        // workgroup shares of one period (16 rounds each): periods up to 262144 samples

// One workgroup per channel walks its code periods in order: plan (wavefront 0: NCO chain with the piece
// tables written straight into the correlator's LDS image) -> correlate (the whole workgroup, ps_unit) ->
// sums, cumsumcorr and -- when the reference's cadence says so -- pll/dll (wavefront 0), which set the
// frequencies of the next period.  No host round trip between periods.
template <int DTYPE, int NTAP, int NIT>
__global__ __launch_bounds__(256) void trk_loop_kernel(const GcChan *__restrict__ chan, GcTrkState *__restrict__ state,
                                                       gnsscorr_loop_t *__restrict__ loop, const uint64_t *__restrict__ wrpos,
                                                       double *__restrict__ corrI, double *__restrict__ corrQ,
                                                       int *__restrict__ nsamp_out, gnsscorr_trklog_t *__restrict__ log,
                                                       int *__restrict__ ndone, int *__restrict__ nco_overflow, int nch,
                                                       int nper, int nseg, int max_n, int rpw, int ablate)
{
    using L = PsLayout<DTYPE, NIT>;
    __shared__ __attribute__((aligned(16))) char smem[L::bytes(NTAP)];
    __shared__ GcTrkUnit su;
    __shared__ GcRound sr[GC_LOOP_MAXSEG][GC_MAXR];
    __shared__ int spart[GC_LOOP_MAXSEG][2 * NTAP];
    __shared__ int sgo;
    __shared__ gnsscorr_loop_t slp;             // the channel's loop state while the kernel runs
    const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ch >= nch) return;
    const GcChan &c = chan[ch];
    const int ntap = c.ntap;
    if (c.dtype != DTYPE || ntap > NTAP) return;
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(loop + ch);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&slp);
        for (int i = tid; i < (int)(sizeof(gnsscorr_loop_t) / 8); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    gnsscorr_loop_t *lp = &slp;
    int *sk0 = reinterpret_cast<int *>(smem + L::K0_OFF);
    GcCarSeg *scar = reinterpret_cast<GcCarSeg *>(smem + L::CAR_OFF);
    GcCodeSeg *scode = reinterpret_cast<GcCodeSeg *>(smem + L::CODE_OFF);
    const uint64_t wp = wrpos[ch];

    // wavefront 0 keeps the chained state in registers; the step tables of the two NCOs live in LDS (in
    // registers they would push the correlator's accumulators out)
    GcTrkState st = state[ch];
#ifdef GC_LOOP_PLAN_IN_REGS         // (tools/debug: the round-2 form that stalled)
    GcCodePlan PC;
    GcCarPlan PK;
#else
    __shared__ GcCodePlan sPC;
    __shared__ GcCarPlan sPK;
    GcCodePlan &PC = sPC;
    GcCarPlan &PK = sPK;
#endif
    GcFillLanes fill{lane};
    double lastcarr = 0.0, lastcode = 0.0;
    bool have_plan = false;
    if (wave == 0) gc_fast_init(PK.fprem, -GC_NCO_DPI);
    int p = 0;
    for (; p < nper; p++) {
        if (wave == 0) {
            // ---- is the period there yet?  ref src/sdrtrk.c:26-30
            const uint64_t bufflocnow = wp - (uint64_t)c.nsamp;
            const bool go = bufflocnow > st.buffloc;
            if (go) {
                if (!have_plan || st.carrfreq != lastcarr || st.codefreq != lastcode) {     // new frequencies: new step tables
                    const double ci0 = __dmul_rn(c.ti, st.codefreq);
                    gc_code_plan_init(PC, ci0, c.clen, c.smax, false);
                    gc_car_plan_init(PK, gc_carrier_ps(st.carrfreq, c.ti), false, false);
                    lastcarr = st.carrfreq;
                    lastcode = st.codefreq;
                    have_plan = true;
                }
                const double ci = PC.f.s, dlen = (double)c.clen;
                const double q = __ddiv_rn(__dsub_rn(dlen, st.remcode), __ddiv_rn(st.codefreq, c.f_sf));   // ref :31-32
                int n = (q > -2147483648.0 && q < 2147483648.0) ? (int)q : 0;
                const int nt = n + 2 * c.smax;
                GcTrkUnit u;
                const uint64_t a0 = (st.buffloc % c.ringlen) * (uint64_t)DTYPE;
                u.a_al = a0 & ~(uint64_t)15;
                u.head = (int)(a0 - u.a_al);
                u.n = n;
                u.G = (u.head + n * DTYPE + 15) >> 4;
                u.nt = nt;
                u.ncar = u.ncode = 0;
                double remcarr = st.remcarr, remcode = st.remcode;
                const bool valid = n > 0 && n <= max_n && ci > 0.0 && ci < dlen;
                GC_DBG_MARK(0, 100 * p + 1);
                if (valid) {
                    LdsCarTable ct{(gc_lds_int)sk0, (gc_lds_car)scar, 0, 0};
                    LdsCodeTable dt{(gc_lds_code)scode, GC_NCODE, 0, 0};
                    double r;
                    if (gc_carrier_period(PK, st.remcarr, n, fill, &r, ct)) {
                        remcarr = r;
                    } else {            // any other shape: the general walkers
                        ct.n = 0;
                        ct.overflow = 0;
                        gc_fast_init(PK.f, PK.f.s);
                        const double xn = gc_fast_carrier_walk(PK.f, gc_carrier_phis(st.remcarr), n, ct);
                        remcarr = gc_fast_prem(PK.fprem, xn);
                    }
                    GC_DBG_MARK(0, 100 * p + 2);
                    if (gc_code_period(PC, st.remcode, nt, fill, &r, dt)) {
                        remcode = r;
                    } else {
                        GC_DBG_MARK(0, 100 * p + 3);
                        dt.n = 0;
                        dt.overflow = 0;
                        gc_fast_init(PC.f, ci);
                        GC_DBG_MARK(0, 100 * p + 4);
                        const double cend = gc_fast_code_walk(PC.f, gc_code_start(st.remcode, c.smax, ci, c.clen), c.clen, nt, dt);
                        remcode = gc_code_rem(cend, c.smax, ci);
                    }
                    GC_DBG_MARK(0, 100 * p + 5);
                    u.ncar = ct.n;
                    u.ncode = dt.n;
                    // what the correlator's scans rely on: carrier pieces start at sample 0 and at increasing samples,
                    // code pieces are non-empty, contiguous and cover the nt replica positions
                    bool bad = ct.n < 1 || dt.n < 1 || ct.overflow || dt.overflow;
                    if (!bad) {
                        if (lane == 0) bad = sk0[0] != 0;
                        if (lane + 1 < ct.n) bad = bad || !(sk0[lane] < sk0[lane + 1]);
                        if (lane < dt.n) {
                            const int je = scode[lane].j0 + scode[lane].cnt;
                            bad = bad || scode[lane].cnt <= 0 || je != (lane + 1 < dt.n ? scode[lane + 1].j0 : nt);
                        }
                    }
                    if (__any(bad)) {
                        if (lane == 0) atomicAdd(nco_overflow, 1);
                        u.n = 0;
                    }
                    if (lane == 0) sk0[ct.n < GC_NCAR ? ct.n : GC_NCAR] = 0x7fffffff;
                } else {
                    u.n = 0;
                }
                // rounds: lane (seg, r)
                if (u.n > 0) {
                    const int rgrp = L::RGRP, rsamp = L::RSAMP;
                    const int sg = lane / GC_MAXR, r = lane % GC_MAXR;
                    const int g0 = sg * rgrp * rpw;
                    if (sg < nseg && sg < GC_LOOP_MAXSEG && r < rpw && g0 + r * rgrp < u.G) {
                        const unsigned short *rank = (const unsigned short *)(c.code + 1024);
                        const int klo = (g0 * 16 - u.head) / DTYPE;
                        const int kl = klo + r * rsamp;
                        const int kfirst = kl > 0 ? kl : 0;
                        const int kend = (kl + rsamp < n ? kl + rsamp : n);
                        int wa = 0, wb = 0, hint = 0;
                        const int ma = gc_code_chip_at(scode, u.ncode, kfirst, &wa, &hint);
                        const int mb = gc_code_chip_at(scode, u.ncode, kend - 1 + 2 * c.smax, &wb, nullptr);
                        GcRound ro;
                        ro.q0 = wa * c.nedge + (int)rank[ma];
                        ro.q1 = wb * c.nedge + (int)rank[mb];
                        ro.clast = (short)c.code[mb];
                        ro.w0 = (short)wa;
                        ro.hint = hint;
                        sr[sg][r] = ro;
                    }
                }
                if (lane == 0) {
                    su = u;
                    gnsscorr_trklog_t *lg = log + (size_t)ch * nper + p;
                    lg->buffloc = st.buffloc;
                    lg->currnsamp = n;
                    nsamp_out[(size_t)ch * nper + p] = n;
                }
                st.remcarr = remcarr;
                st.remcode = remcode;
                st.buffloc += (uint64_t)(int64_t)n;
            }
            if (lane == 0) sgo = go ? 1 : 0;
        }
        __syncthreads();
        if (!sgo) break;
        GC_DBG_MARK(0, 100 * p + 6);
        // ---- correlate: the whole workgroup, one share of the period after the other
        for (int sg = 0; sg < nseg && sg < GC_LOOP_MAXSEG; sg++) {
            ps_unit<DTYPE, NTAP, NIT>(c, su, (const GcUnitSegs *)nullptr, sr[sg], spart[sg], NTAP, max_n, rpw, sg, ablate, smem, tid);
            __syncthreads();
        }
        GC_DBG_MARK(0, 100 * p + 7);
        // ---- outputs, cumsumcorr, loop filters (ref src/sdrtrk.c:35-36,42,64-86; src/sdrmain.c:269-310)
        if (wave == 0) {
            if (lane < ntap) {
                int sI = 0, sQ = 0;
                for (int sg = 0; sg < nseg && sg < GC_LOOP_MAXSEG; sg++) { sI += spart[sg][lane]; sQ += spart[sg][NTAP + lane]; }
                const double cI = (double)sI * (1.0 / 32.0), cQ = (double)sQ * (1.0 / 32.0);     // correlator's II, QQ (ref src/sdrcmn.c:716-719)
                corrI[((size_t)ch * nper + p) * ntap + lane] = cI;
                corrQ[((size_t)ch * nper + p) * ntap + lane] = cQ;
                // memcpy(oldI, II, 1 + 2*corrn*sizeof(double)): the last tap only gets its lowest byte (ref :35-36)
                const double pII = lp->II[lane], pQQ = lp->QQ[lane];
                double oI = pII, oQ = pQQ;
                if (lane == ntap - 1) {
                    oI = gc_u2d((gc_d2u(lp->oldI[lane]) & ~0xFFull) | (gc_d2u(pII) & 0xFFull));
                    oQ = gc_u2d((gc_d2u(lp->oldQ[lane]) & ~0xFFull) | (gc_d2u(pQQ) & 0xFFull));
                }
                lp->oldI[lane] = oI;
                lp->oldQ[lane] = oQ;
                // correlator(..., trk.QQ, trk.II, ...): trk.II <- sum dataQ*code, trk.QQ <- sum dataI*code (ref :42)
                lp->II[lane] = cQ;
                lp->QQ[lane] = cI;
                // cumsumcorr, polarity +1 (ref :64-76)
                lp->oldsumI[lane] = __dadd_rn(lp->oldsumI[lane], oI);
                lp->oldsumQ[lane] = __dadd_rn(lp->oldsumQ[lane], oQ);
                lp->sumI[lane] = __dadd_rn(lp->sumI[lane], cQ);
                lp->sumQ[lane] = __dadd_rn(lp->sumQ[lane], cI);
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            GC_DBG_MARK(0, 100 * p + 8);
            // loop timing of sdrnavigation()/checkbit() (ref src/sdrnav.c:18,241-262)
            int flag = 0;
            const int flagsync = lp->flagsync;
            int navcnt = lp->navcnt, swloop = lp->swloop;
            const uint64_t cnt = lp->cnt;
            if (flagsync) {
                const int biti = (int)(cnt % (uint64_t)lp->rate);
                const int diffi = biti - lp->synci;
                if (diffi == 1 || diffi == -lp->rate + 1) navcnt = 1;
                swloop = (navcnt % lp->loopms == 0);
                navcnt++;
            }
            if (lane == 0) {
                if (!flagsync) {
                    loop_pll(lp, st, 0, lp->ctime);
                    loop_dll(lp, st, 0, lp->ctime);
                    flag = 1;
                } else if (swloop) {
                    loop_pll(lp, st, 1, (double)lp->loopms / 1000);
                    loop_dll(lp, st, 1, (double)lp->loopms / 1000);
                    flag = 2;
                }
                gnsscorr_trklog_t *lg = log + (size_t)ch * nper + p;
                lg->carrfreq = st.carrfreq;
                lg->codefreq = st.codefreq;
                lg->carrErr = lp->carrErr;
                lg->codeErr = lp->codeErr;
                lg->carrNco = lp->carrNco;
                lg->codeNco = lp->codeNco;
                lg->freqErr = lp->freqErr;
                lg->remcode = st.remcode;
                lg->remcarr = st.remcarr;
                lg->flagloopfilter = flag;
                lp->navcnt = navcnt;
                lp->swloop = swloop;
                lp->cnt = cnt + 1;
            }
            GC_DBG_MARK(0, 100 * p + 9);
            // the new frequencies, for every lane of the planning wavefront
            flag = __builtin_amdgcn_readfirstlane(flag);
            st.carrfreq = gc_u2d(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gc_d2u(st.carrfreq) >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)gc_d2u(st.carrfreq)));
            st.codefreq = gc_u2d(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gc_d2u(st.codefreq) >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)gc_d2u(st.codefreq)));
            if (flag && lane < ntap) {      // clearcumsumcorr (ref src/sdrtrk.c:77-86)
                lp->oldsumI[lane] = 0.0;
                lp->oldsumQ[lane] = 0.0;
                lp->sumI[lane] = 0.0;
                lp->sumQ[lane] = 0.0;
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
        }
    }
    GC_DBG_MARK(10, 1 + (threadIdx.x >> 6));
    if (tid == 0) {
        state[ch] = st;
        ndone[ch] = p;
    }
    __syncthreads();
    GC_DBG_MARK(11, 1 + (threadIdx.x >> 6));
    {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(loop + ch);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&slp);
        for (int i = tid; i < (int)(sizeof(gnsscorr_loop_t) / 8); i += 256) dst[i] = src[i];
    }
}

int g_trk_nit = 0;      // groups per lane per segment workgroup (1, 2, 4 or 8); 0 = not yet chosen
int g_trk_algo = 0;     // 1 = prefix-sum form (default), 2 = replica form (GNSSCORR_TRK_ALGO=replica)

// the edge table of the launch being issued (set by gc_launch_trk_corr around the tap-bucket dispatch below)
static thread_local const unsigned short *t_trk_etab = nullptr;

template <int DTYPE, int NTAP, int NIT>
int launch_corr_ps(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, const GcUnitSegs *segs, const GcRound *rounds, int *partial,
                   int nch, int nepoch, int nseg, int ntap_stride, int ntap_lo, int max_n)
{
    const unsigned short *etab = t_trk_etab;
    static_assert(PsLayout<DTYPE, NIT>::bytes(NTAP) <= 64 * 1024, "static LDS image");
    const int rpw = trk_ps_rounds(DTYPE, max_n, NIT);
    static const int ablate = getenv("GNSSCORR_TRK_ABLATE") ? atoi(getenv("GNSSCORR_TRK_ABLATE")) : 0;
    const long long total = 8LL * ((nepoch + 7) / 8) * nch * nseg;
    if (total > 0x7fffffffLL) return gc_fail(GNSSCORR_EINVAL, "trk_corr: batch too large (%lld workgroups)", total);
    hipLaunchKernelGGL((trk_corr_ps_kernel<DTYPE, NTAP, NIT>), dim3((unsigned)total), dim3(256), 0, st, chan, unit,
                       segs, rounds, partial, nch, nepoch, nseg, ntap_stride, ntap_lo, max_n, rpw, ablate, etab);
    GC_HIP(hipGetLastError());
    return 0;
}

template <int DTYPE, int NTAP, int NIT>
int launch_corr(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, const GcUnitSegs *segs, int *partial, int nch, int nepoch,
                int nseg, int ntap_stride, int ntap_lo, int max_n, int smax_max)
{
    constexpr int RED_BYTES = ((4 * 2 * NTAP * 4) + 15) & ~15;
    constexpr int SEGS = 256 * NIT * (16 / DTYPE);
    constexpr int RS = SEGS / 8 + 64;
    const int lds = (int)((DTYPE == 2 ? 512 : 1024) + RED_BYTES + 8 * RS * 4 + ((GC_NCAR + 4) & ~3) * 4 +
                          GC_NCAR * sizeof(GcCarSeg) + GC_NCODE * sizeof(GcCodeSeg));
    static const int ablate = getenv("GNSSCORR_TRK_ABLATE") ? atoi(getenv("GNSSCORR_TRK_ABLATE")) : 0;
    if (lds > 64 * 1024)
        GC_HIP(hipFuncSetAttribute((const void *)trk_corr_kernel<DTYPE, NTAP, NIT>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const long long total = 8LL * ((nepoch + 7) / 8) * nch * nseg;
    if (total > 0x7fffffffLL) return gc_fail(GNSSCORR_EINVAL, "trk_corr: batch too large (%lld workgroups)", total);
    dim3 grid((unsigned)total), block(256);
    hipLaunchKernelGGL((trk_corr_kernel<DTYPE, NTAP, NIT>), grid, block, lds, st, chan, unit, segs, partial, nch, nepoch,
                       nseg, ntap_stride, ntap_lo, max_n, ablate);
    GC_HIP(hipGetLastError());
    return 0;
}

template <int DTYPE, int NTAP>
int launch_corr_nit(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, const GcUnitSegs *segs, const GcRound *rounds, int *partial, int nch, int nepoch,
                    int nseg, int ntap_stride, int ntap_lo, int max_n, int smax_max)
{
    if (g_trk_algo == 1) {
        if (g_trk_nit == 1 || DTYPE == 1)
            return launch_corr_ps<DTYPE, NTAP, 1>(st, chan, unit, segs, rounds, partial, nch, nepoch, nseg, ntap_stride, ntap_lo, max_n);
        return launch_corr_ps<DTYPE, NTAP, DTYPE == 1 ? 1 : 2>(st, chan, unit, segs, rounds, partial, nch, nepoch, nseg, ntap_stride, ntap_lo, max_n);
    }
    switch (g_trk_nit) {
    default: return launch_corr<DTYPE, NTAP, 2>(st, chan, unit, segs, partial, nch, nepoch, nseg, ntap_stride, ntap_lo, max_n, smax_max);
    case 8: return launch_corr<DTYPE, NTAP, 8>(st, chan, unit, segs, partial, nch, nepoch, nseg, ntap_stride, ntap_lo, max_n, smax_max);
    case 4: return launch_corr<DTYPE, NTAP, 4>(st, chan, unit, segs, partial, nch, nepoch, nseg, ntap_stride, ntap_lo, max_n, smax_max);
    }
}

template <int DTYPE>
int launch_corr_taps(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, const GcUnitSegs *segs, const GcRound *rounds, int *partial, int nch, int nepoch,
                     int nseg, int ntap_stride, int ntap, int max_n, int smax_max)
{
    // smallest instantiation that holds ntap accumulators; it serves (lo, NTAP]
#define GC_LC(N, LO) return launch_corr_nit<DTYPE, N>(st, chan, unit, segs, rounds, partial, nch, nepoch, nseg, ntap_stride, LO, max_n, smax_max)
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
    const char *a = getenv("GNSSCORR_TRK_ALGO");
    g_trk_algo = (a && a[0] == 'r') ? 2 : 1;
    const char *e = getenv("GNSSCORR_TRK_NIT");
    g_trk_nit = e ? atoi(e) : 2;
    if (g_trk_algo == 1) {
        if (g_trk_nit != 1 && g_trk_nit != 2) g_trk_nit = 2;
    } else {
        if (g_trk_nit != 2 && g_trk_nit != 4 && g_trk_nit != 8) g_trk_nit = 2;
    }
}

}  // namespace

int gc_trk_nseg(int dtype, int max_n)
{
    trk_pick_nit();
    const int groups = (15 + max_n * dtype + 15) / 16 + 1;
    // real (1-byte) samples carry 16 running sums per group: one group per lane keeps the LDS image small
    const int nit = g_trk_algo == 1 ? trk_ps_nit(dtype, g_trk_nit) : g_trk_nit;
    const int rounds = (groups + 256 * nit - 1) / (256 * nit);
    return g_trk_algo == 1 ? (rounds + GC_MAXR - 1) / GC_MAXR : rounds;
}

int gc_launch_trk_expand(hipStream_t st, const GcChan *chan, const GcTrkPlan *plan, GcTrkUnit *unit, GcUnitSegs *segs,
                         int *nsamp_out, int nch, int nepoch, GcRound *rounds, int nseg, int max_n, int *nco_overflow)
{
    trk_pick_nit();
    const int total = nch * nepoch;
    hipLaunchKernelGGL(trk_expand_kernel, dim3((total + 63) / 64), dim3(64), 0, st, chan, plan, unit, segs, nsamp_out,
                       nch, nepoch, g_trk_algo == 1 ? rounds : (GcRound *)nullptr, nseg, max_n, g_trk_nit, nco_overflow);
    GC_HIP(hipGetLastError());
    return 0;
}

// One launch serves every channel whose (dtype, tap bucket) matches; callers
// invoke it once per distinct dtype present in the channel set.
// etab: the start samples of the periods' chip edges (gc_launch_trk_edges on the same stream before this), or null
int gc_launch_trk_corr(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, const GcUnitSegs *segs,
                       const GcRound *rounds, int *partial, int nch,
                       int nepoch, int nseg, int ntap_stride, int dtype, int ntap, int max_n, int smax_max,
                       const unsigned short *etab)
{
    trk_pick_nit();
    if (smax_max > 64) return gc_fail(GNSSCORR_EINVAL, "trk_corr: tap offset %d samples (<= 64 supported)", smax_max);
    static const bool noetab = getenv("GNSSCORR_TRK_NOEDGETAB") != nullptr;
    t_trk_etab = (g_trk_algo == 1 && !noetab) ? etab : nullptr;
    int rc = gc_fail(GNSSCORR_EINVAL, "trk_corr: dtype %d not 1 or 2", dtype);
    if (dtype == 2)
        rc = launch_corr_taps<2>(st, chan, unit, segs, rounds, partial, nch, nepoch, nseg, ntap_stride, ntap, max_n, smax_max);
    else if (dtype == 1)
        rc = launch_corr_taps<1>(st, chan, unit, segs, rounds, partial, nch, nepoch, nseg, ntap_stride, ntap, max_n, smax_max);
    t_trk_etab = nullptr;
    return rc;
}

int gc_launch_trk_edges(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, const GcUnitSegs *segs, unsigned short *etab,
                        int nch, int nepoch)
{
    static const bool noetab = getenv("GNSSCORR_TRK_NOEDGETAB") != nullptr;
    trk_pick_nit();
    if (!etab || noetab || g_trk_algo != 1) return 0;
    hipLaunchKernelGGL(trk_edges_kernel, dim3(nch * nepoch), dim3(256), 0, st, chan, unit, segs, etab, nch, nepoch);
    GC_HIP(hipGetLastError());
    return 0;
}

template <int DTYPE>
int launch_loop_taps(hipStream_t st, const GcChan *chan, GcTrkState *state, gnsscorr_loop_t *loop, const uint64_t *wrpos,
                     double *corrI, double *corrQ, int *nsamp_out, gnsscorr_trklog_t *log, int *ndone, int *nco_overflow,
                     int nch, int nper, int nseg, int ntap, int max_n)
{
    constexpr int NIT = DTYPE == 1 ? 1 : 2;
    const int rpw = trk_ps_rounds(DTYPE, max_n, NIT);
    static const int ablate = getenv("GNSSCORR_TRK_ABLATE") ? atoi(getenv("GNSSCORR_TRK_ABLATE")) : 0;
#define GC_LL(N) do { hipLaunchKernelGGL((trk_loop_kernel<DTYPE, N, NIT>), dim3(nch), dim3(256), 0, st, chan, state, loop, wrpos, \
                                         corrI, corrQ, nsamp_out, log, ndone, nco_overflow, nch, nper, nseg, max_n, rpw, ablate); \
                      GC_HIP(hipGetLastError()); return 0; } while (0)
    if (ntap <= 3) GC_LL(3);
    if (ntap <= 5) GC_LL(5);
    if (ntap <= 7) GC_LL(7);
    if (ntap <= 13) GC_LL(13);
    if (ntap <= 21) GC_LL(21);
    GC_LL(33);
#undef GC_LL
}

// closed loop: nper periods of every channel of this dtype (one launch per dtype present)
int gc_launch_trk_loop(hipStream_t st, const GcChan *chan, GcTrkState *state, gnsscorr_loop_t *loop, const uint64_t *wrpos,
                       double *corrI, double *corrQ, int *nsamp_out, gnsscorr_trklog_t *log, int *ndone, int *nco_overflow,
                       int nch, int nper, int nseg, int dtype, int ntap, int max_n, int smax_max)
{
    trk_pick_nit();
    if (smax_max > 64) return gc_fail(GNSSCORR_EINVAL, "trk_loop: tap offset %d samples (<= 64 supported)", smax_max);
    if (nseg > GC_LOOP_MAXSEG) return gc_fail(GNSSCORR_EINVAL, "trk_loop: period of %d samples too long", max_n);
    if (dtype == 2)
        return launch_loop_taps<2>(st, chan, state, loop, wrpos, corrI, corrQ, nsamp_out, log, ndone, nco_overflow, nch, nper, nseg, ntap, max_n);
    if (dtype == 1)
        return launch_loop_taps<1>(st, chan, state, loop, wrpos, corrI, corrQ, nsamp_out, log, ndone, nco_overflow, nch, nper, nseg, ntap, max_n);
    return gc_fail(GNSSCORR_EINVAL, "trk_loop: dtype %d not 1 or 2", dtype);
}

// Every planned period must lie in what its ring holds: written already (the reference waits for
// bufflocnow > buffloc before it tracks a period, ref src/sdrtrk.c:26-30) and not yet overwritten (the
// reference stops on a buffer overrun, ref src/sdrrcv.c:325-349).  Counts the periods that do not.
__global__ void trk_ringcheck_kernel(const GcChan *__restrict__ chan, const GcTrkPlan *__restrict__ plan,
                                     const int8_t *ring0, uint64_t wrpos0, uint64_t wrpos1, int nch, int nepoch,
                                     int *__restrict__ viol)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nch * nepoch) return;
    const int ch = i / nepoch;
    const GcTrkPlan p = plan[i];
    if (p.n <= 0) return;
    const uint64_t wp = chan[ch].ring == ring0 ? wrpos0 : wrpos1, rl = chan[ch].ringlen;     // (front end 1 or 2)
    if (p.buffloc + (uint64_t)p.n > wp || (wp > rl && p.buffloc < wp - rl)) atomicAdd(viol, 1);
}

int gc_launch_trk_ringcheck(hipStream_t st, const GcChan *chan, const GcTrkPlan *plan, const int8_t *ring0, uint64_t wrpos0,
                            uint64_t wrpos1, int nch, int nepoch, int *viol)
{
    const int total = nch * nepoch;
    hipLaunchKernelGGL(trk_ringcheck_kernel, dim3((total + 255) / 256), dim3(256), 0, st, chan, plan, ring0, wrpos0, wrpos1,
                       nch, nepoch, viol);
    GC_HIP(hipGetLastError());
    return 0;
}

int gc_launch_trk_finish(hipStream_t st, const int *partial, double *corrI, double *corrQ, double *sumI,
                         double *sumQ, unsigned long long *scratch, int nch, int nepoch, int nseg, int ntap)
{
    hipLaunchKernelGGL(trk_finish_kernel, dim3(nch, (nepoch + 255) / 256), dim3(256), 0, st, partial, corrI, corrQ,
                       sumI, sumQ, scratch, nepoch, nseg, ntap);
    GC_HIP(hipGetLastError());
    return 0;
}
