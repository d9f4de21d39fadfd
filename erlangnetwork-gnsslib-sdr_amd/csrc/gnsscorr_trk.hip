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


#include "gnsscorr_internal.h"
#include "gnsscorr_ps.h"

namespace {


// ---------------------------------------------------------------------------
// per-unit constants and NCO tables
// ---------------------------------------------------------------------------
// One lane per (channel, epoch): ring offset, the two NCOs of the period as piece tables, and per
// round of the correlator the chip edges its samples can touch.
// trk_expand's emitters.  A lane builds its unit's tables alone, and what bounds it is the latency of memory operations
// that depend on each other, not arithmetic: so the tables go to global memory by stores alone (the piece being built
// stays in registers; the generic emitters of gnsscorr_nco.h read the previous piece back for every new one), their
// invariants are checked as they are emitted, and the five fields the rounds' searches need stay in LDS (lane-strided).
#define GC_EXP_LANES 64
struct ExpCarTable {
    int *k0;
    GcCarSeg *seg;
    int cap, n, overflow, prevk;
    bool lastzero, bad;             // bad: the pieces do not start at sample 0 / at increasing samples
    __device__ void operator()(int k, double x, double d, int /*count*/)
    {
        const GcCarSeg s = gc_carseg_make(x, d);
        const bool zero = s.fx == 0 && s.dfx == 0;
        if (n > 0 && zero && lastzero) return;      // (adjacent all-zero pieces are one piece, as GcCarTable)
        if (n >= cap) { overflow = 1; return; }
        bad = bad || (n == 0 ? k != 0 : !(prevk < k));
        k0[n] = k;
        seg[n] = s;
        prevk = k;
        lastzero = zero;
        n++;
    }
};
struct ExpCodeLds {                 // [piece][lane]
    int j0[GC_NCODE * GC_EXP_LANES], cnt[GC_NCODE * GC_EXP_LANES], w[GC_NCODE * GC_EXP_LANES];
    double y0[GC_NCODE * GC_EXP_LANES], d[GC_NCODE * GC_EXP_LANES];
};
struct ExpCodeTable {
    GcCodeSeg *seg;
    ExpCodeLds *lds;
    int lane, cap, n, overflow, prev_end;
    bool bad;                       // bad: an empty piece, or pieces that do not follow each other
    double py0, pd, pinv, pyl;      // the piece being built (piece n - 1) ...
    int pj0, pcnt, pw;
    __device__ void flush()         // ... is complete
    {
        GcCodeSeg s;
        s.y0 = py0; s.d = pd; s.inv = pinv; s.ylast = pyl; s.j0 = pj0; s.cnt = pcnt; s.w = pw; s.pad = 0;
        seg[n - 1] = s;
        const int x = (n - 1) * GC_EXP_LANES + lane;
        lds->j0[x] = pj0; lds->cnt[x] = pcnt; lds->w[x] = pw; lds->y0[x] = py0; lds->d[x] = pd;
        bad = bad || pcnt <= 0 || (n > 1 && prev_end != pj0);
        prev_end = pj0 + pcnt;
    }
    __device__ void operator()(int j, double y, double d, int count, int w)
    {
        GC_FP_STRICT
        const double yl = fma((double)(count - 1), d, y);
        // (positions inside chip 0 after a wrap join the piece before them, as GcCodeTable)
        if (n > 0 && pw == w && py0 > -1.0 && pyl < 1.0 && y > -1.0 && yl < 1.0) {
            pcnt += count;
            pyl = yl;
            pd = 0.0;
            pinv = 0.0;
            return;
        }
        if (n >= cap) { overflow = 1; return; }
        if (n > 0) flush();
        py0 = y;
        pd = count > 1 ? d : 0.0;
        pinv = (count > 1 && d != 0.0) ? 1.0 / d : 0.0;
        pyl = yl;
        pj0 = j;
        pcnt = count;
        pw = w;
        n++;
    }
    __device__ void finish(int nt)  // the last piece; the pieces cover the nt replica positions
    {
        if (n > 0) {
            flush();
            bad = bad || prev_end != nt;
        }
    }
    // gc_code_chip_at on the LDS copy
    __device__ int chip_at(int j, int *w, int *piece) const
    {
        int lo = 0, hi = n - 1;
        while (lo < hi) {                            // last piece with j0 <= j
            const int mid = (lo + hi + 1) >> 1;
            if (lds->j0[mid * GC_EXP_LANES + lane] <= j) lo = mid; else hi = mid - 1;
        }
        const int x = lo * GC_EXP_LANES + lane;
        if (w) *w = lds->w[x];
        if (piece) *piece = lo;
        int i = j - lds->j0[x];
        const int cnt = lds->cnt[x];
        if (i < 0) i = 0;
        if (i >= cnt) i = cnt - 1;
        const double d = lds->d[x], y0 = lds->y0[x];
        if (d == 0.0) return (int)y0;
        return (int)fma((double)i, d, y0);
    }
};

__global__ __launch_bounds__(GC_EXP_LANES) void trk_expand_kernel(const GcChan *__restrict__ chan, const GcTrkPlan *__restrict__ plan,
                                  GcTrkUnit *__restrict__ unit, GcUnitSegs *__restrict__ segs,
                                  int *__restrict__ nsamp_out, int nch, int nepoch,
                                  GcRound *__restrict__ rounds, int nseg, int max_n, int nit,
                                  int *__restrict__ nco_overflow)
{
    __shared__ ExpCodeLds slds;
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
    ExpCarTable ct{sg->carK0, sg->car, GC_NCAR, 0, 0, 0, false, false};
    ExpCodeTable dt{sg->code, &slds, (int)threadIdx.x, GC_NCODE, 0, 0, 0, false, 0.0, 0.0, 0.0, 0.0, 0, 0, 0};
    {
        GcNcoFast f;
        gc_fast_init(f, gc_carrier_ps(p.carrfreq, c.ti));
        gc_fast_carrier_walk(f, gc_carrier_phis(p.phi0), p.n, ct);
        gc_fast_init(f, ci);
        gc_fast_code_walk(f, gc_code_start(p.coff, c.smax, ci, c.clen), c.clen, u.nt, dt);
        dt.finish(u.nt);
    }
    u.ncar = ct.n;
    u.ncode = dt.n;
    // what the correlator's scans rely on: carrier pieces start at sample 0 and at increasing samples, code pieces
    // are non-empty, contiguous and cover the nt replica positions (anything else is reported, never correlated)
    const bool bad = ct.overflow || dt.overflow || ct.n < 1 || dt.n < 1 || ct.bad || dt.bad;
    if (bad) {
        if (nco_overflow) atomicAdd(nco_overflow, 1);
        u.n = 0;
        unit[i] = u;
        return;
    }

    // rounds of the prefix-sum correlator (same geometry as trk_corr_ps_kernel; a round is one wavefront's share,
    // gnsscorr_ps.h): round r of workgroup
    // seg covers samples [kl, kl + rsamp) of the period and can touch the chips T(first sample) ..
    // T(last sample + 2 smax); rank[] turns those into positions in the code's edge list
    if (!rounds) { unit[i] = u; return; }
    const int nitc = trk_ps_nit(c.dtype, nit), rgrp = GC_PS_WLANES * nitc, rsamp = rgrp * (16 / c.dtype);
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
            const int ma = dt.chip_at(kfirst, &wa, &hint);
            const int mb = dt.chip_at(kend - 1 + 2 * c.smax, &wb, nullptr);
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
        u.eq0 = eq0;
        u.eq1 = eq1;
    }
    unit[i] = u;
}

// ---------------------------------------------------------------------------
// correlator
// ---------------------------------------------------------------------------
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
    if (g_trk_algo == 1) {          // prefix-sum form: rounds of one wavefront each, GC_MAXR per workgroup
        const int rounds = (groups + GC_PS_WLANES * nit - 1) / (GC_PS_WLANES * nit);
        return (rounds + GC_MAXR - 1) / GC_MAXR;
    }
    return (groups + 256 * nit - 1) / (256 * nit);
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
