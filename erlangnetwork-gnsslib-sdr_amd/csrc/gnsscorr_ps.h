// gnsscorr_ps.h -- the prefix-sum E/P/L correlator of one (channel, code period) on one 256-lane workgroup
// (ps_unit), shared by the batch kernel (gnsscorr_trk.hip) and the closed-loop step kernels (gnsscorr_loop.hip).
// gfx950 only.  Replaces correlator() = mixcarr + rescode + dot_22/dot_23 (ref src/sdrcmn.c:608-722).
#pragma once

#include <type_traits>
#include "gnsscorr_internal.h"

#ifdef GC_PS_TRACE      // (tools/debug: shader-clock stamps of sampled correlator workgroups, lane 0 of wave 0)
#define GC_PS_TRACE_N 2048
__device__ unsigned long long gc_ps_trace[GC_PS_TRACE_N * 16];
#define GC_PSTAMP(i) do { if (ptr_) ptr_[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int gnsscorr_debug_ps_trace(unsigned long long *dst)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(gc_ps_trace), sizeof(unsigned long long) * GC_PS_TRACE_N * 16) == hipSuccess ? 0 : -1;
}
#else
#define GC_PSTAMP(i) do { } while (0)
#endif

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

// a . b over four int8 lanes + c into a NEW register (the three-operand form, by name: through the builtin the compiler
// picks the accumulate-in-place form and copies the accumulator first -- every running sum of phase A is stored, so
// that is two extra moves per sample)
__device__ __forceinline__ int dot4_run(int a, int b, int c)
{
    int d;
    asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

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
#ifdef GC_PS_TRACE
    unsigned long long *ptr_ = nullptr;
    if (tid == 0 && (blockIdx.x % 15) == 0 && blockIdx.x / 15 < GC_PS_TRACE_N) ptr_ = gc_ps_trace + (blockIdx.x / 15) * 16;
    GC_PSTAMP(0);
#endif
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
    GC_PSTAMP(1);                                       // unit and channel constants are here
    load_round(0, vA);                                  // in flight while the tables are set up
    GcRound ronext = myrounds[0];                       // (likewise; every round asks for the next one's record)

    // ---- chip edges (ref src/sdrcmn.c:608-621 in closed form) --------------------------------
    // The replica position of chip M's first sample is B_M = min{j : T(j) >= M},
    // T(j) = trunc(fma(j, ci, cs)).  Only chips at which the code changes value matter; they are
    // numbered q = period * nedge + list index, and rank[] converts a chip number into that
    // numbering; trk_expand prepared, per round, the edges [q0, q1) its samples can touch.
    const gc_gptr_i8 code = (gc_gptr_i8)c.code;
    const int __attribute__((address_space(1))) *edges = (const int __attribute__((address_space(1))) *)(code + 3072);
    const int nedge = c.nedge;
    const int ncar = u.ncar, ncode = u.ncode;
    if (gs) {
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
    GC_PSTAMP(2);                                       // tables staged

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
        // (the round's record was asked for a round ago: nothing here waits for memory it has just requested)
        const GcRound ro = ronext;
        if (r + 1 < nround) ronext = myrounds[r + 1];
        const int rq0 = ro.q0, rq1 = ro.q1, rlast = ro.clast, rw0 = ro.w0, rhint = ro.hint;
        int q = rq0 + tl, ew = 0, ed = 0, jsraw = 0;
        const int q1 = (ablate & 1) ? 0 : rq1;
        if (q < q1) {                                   // in flight during the mixing phase
            ed = edge_load(q, rw0, &ew);
            if (have_etab) jsraw = (int)etab_u[q - u.eq0];
        }
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
                    aI = dot4_run((int)wd, (int)l[i].x, aI);
                    aQ = dot4_run((int)wd, (int)l[i].y, aQ);
                    const int p = it * SPG + i + 1;
                    if (p < LSP) loc[tl * LPAD + p] = make_int2(aI, aQ);
                }
            }
        };
        if (r == 0) GC_PSTAMP(3);                       // round 0: edge record + piece scan done, samples needed now
        if (!(ablate & 2)) { if (onepiece) run(std::false_type{}); else run(std::true_type{}); }
        if (r == 0) GC_PSTAMP(4);                       // round 0: mixing done
        // the start sample of this lane's chip edge (no LDS involved: overlaps the image writes)
        if (q < q1) js = (have_etab ? jsraw : edge_js(q, ed, ew, rhint)) - roff;
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
        if (r == 0) GC_PSTAMP(5);                       // round 0: scan + atomics
        __syncthreads();
        if (r == 0) GC_PSTAMP(6);                       // round 0: barrier
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
        if (r == 0) GC_PSTAMP(7);                       // round 0: look-ups
        if (r + 1 < nround) __syncthreads();            // look-ups done before the image is rewritten
        if (r == 0) GC_PSTAMP(8);                       // round 0: barrier
    };
    for (int r = 0; r < nround; r += 2) {
        round(r, vA, vB);
        if (r + 1 < nround) round(r + 1, vB, vA);
    }

    GC_PSTAMP(9);                                       // all rounds
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
    GC_PSTAMP(10);                                      // reduced and stored
}


}  // namespace
