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

// A round of the prefix-sum correlator is what ONE wavefront mixes, scans and looks up on its own: 64 * nit sample
// groups.  Rounds per workgroup: a whole period when it fits GC_MAXR rounds (the workgroup's four wavefronts share them).
#define GC_PS_WLANES 64
__host__ __device__ inline int trk_ps_rounds(int dtype, int max_n, int nit)
{
    const int groups = (15 + max_n * dtype + 15) / 16 + 1;
    const int rounds = (groups + GC_PS_WLANES * nit - 1) / (GC_PS_WLANES * nit);
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
// in rounds of 64*NIT sample groups; a round is ONE WAVEFRONT's work, on the
// wavefront's own LDS image, and the workgroup's wavefronts take the rounds in turn:
//   phase A  lane L mixes its NIT consecutive groups (two chained dot4 per
//            sample) and stores the running sums loc[L][p] (p samples into
//            the lane's span; entry 0 is constant zero); a DPP scan over the
//            wavefront turns the lane totals into lbase[L], the sum in front
//            of the lane's span, and the round's total.
//   phase B  one chip edge per lane: B_m from the unit's edge table, or from
//            the code piece that holds it (a reciprocal estimate, corrected
//            against T itself), then per tap P = loc + lbase at the clamped
//            sample position.
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
    static constexpr int RGRP = GC_PS_WLANES * NIT;             // groups per round (one wavefront)
    static constexpr int RSAMP = GC_PS_WLANES * LSP;            // samples per round
    static constexpr int LPAD = LSP + 1;                        // image stride per lane: odd in 8-byte units
    static constexpr int MAXR = GC_MAXR;                        // rounds per workgroup, at most
    static constexpr int LUTPOS = DTYPE == 2 ? 2 : 4;
    static constexpr int LUT_BYTES = 32 * 8 * LUTPOS;
    // per wavefront: lbase[64 + 1] and the prefix image loc[64 lanes][LPAD] + closing entry
    static constexpr int LB_WAVE = (GC_PS_WLANES + 1) * 8;
    static constexpr int LOC_WAVE = (GC_PS_WLANES * LPAD + 1) * 8;
    static constexpr int LB_OFF = LUT_BYTES;
    static constexpr int LOC_OFF = LB_OFF + ((4 * LB_WAVE + 15) & ~15);
    // the unit's NCO tables: carrier piece starts (+ closing sentinel), carrier pieces, code pieces
    static constexpr int K0_OFF = LOC_OFF + ((4 * LOC_WAVE + 15) & ~15);
    static constexpr int CAR_OFF = K0_OFF + (((GC_NCAR + 1) * 4 + 15) & ~15);
    static constexpr int CODE_OFF = CAR_OFF + GC_NCAR * (int)sizeof(GcCarSeg);
    static constexpr int RED_OFF = CODE_OFF + GC_NCODE * (int)sizeof(GcCodeSeg);
    static constexpr int bytes(int ntap) { return RED_OFF + 4 * (2 * ntap + 2) * 4 + 16; }
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
// trk_corr_ps_kernel, also called from the closed-loop step kernel.  smem: PsLayout bytes.
// Every lane of the workgroup must call it (it synchronises the workgroup twice: tables in, sums out).
//
// The workgroup's rounds are independent pieces of work -- each has its own prefix image, its own edge range
// [q0, q1) and its own closing term -- so each WAVEFRONT takes rounds of its own (wave w: rounds w, w + 4, ...) and
// runs them from the samples to the tap accumulators without meeting the other wavefronts: no barrier and no
// cross-wave sums inside the loop, and a wavefront that waits for its samples does not hold up the other three.
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
    // this thread's element of the unit's NCO tables, requested before anything waits for the unit's record (the piece
    // counts are in it: elements past them are stale and never copied to LDS) -- the two round trips overlap
    int k0r = 0;
    GcCarSeg carr;
    GcCodeSeg coder;
    carr.fx = carr.dfx = 0;
    coder.y0 = coder.d = coder.inv = coder.ylast = 0.0; coder.j0 = coder.cnt = coder.w = coder.pad = 0;
    if (gs) {
        if (tid < GC_NCAR) { k0r = gs->carK0[tid]; carr = gs->car[tid]; }
        else if (tid >= 64 && tid - 64 < GC_NCODE) coder = gs->code[tid - 64];
    }
    // the channel's constants, all requested here: a field first read behind the barrier below would cost the workgroup
    // one more round trip to memory after it (the compiler does not move loads across a barrier)
    const int ntap = c.ntap;
    const int pm1_raw = c.pm1;
    int tapv[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; t++) tapv[t] = c.tapoff[t < GNSSCORR_MAXTAPS ? t : 0];
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
    // (the wavefront number as a scalar: everything derived from it -- round index, sample window, table scans -- stays
    // on the scalar unit instead of being computed per lane under execution masks)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    uint2 *lut = reinterpret_cast<uint2 *>(smem);
    int2 *lbase = reinterpret_cast<int2 *>(smem + L::LB_OFF + wv * L::LB_WAVE);       // [64 + 1]: sums in front of the lanes' spans, total
    int2 *loc = reinterpret_cast<int2 *>(smem + L::LOC_OFF + wv * L::LOC_WAVE);       // [64 lanes][LPAD] + closing entry
    int *red = reinterpret_cast<int *>(smem + L::RED_OFF);            // 4 x (2*NTAP + 2)
    int *sk0 = reinterpret_cast<int *>(smem + L::K0_OFF);             // [ncar] + INT_MAX
    GcCarSeg *scar = reinterpret_cast<GcCarSeg *>(smem + L::CAR_OFF);
    GcCodeSeg *scode = reinterpret_cast<GcCodeSeg *>(smem + L::CODE_OFF);

    const gc_gptr_i8 ring = (gc_gptr_i8)c.ring;
    const uint64_t ringbytes = c.ringlen * (uint64_t)DTYPE;
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
                const gc_u4v t4 = *(gc_gptr_u4)(base + (unsigned)(lane * NIT + it) * 16u);
                dst[it] = make_uint4(t4.x, t4.y, t4.z, t4.w);
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int g = g0 + r * RGRP + lane * NIT + it;
                uint64_t addr = u.a_al + (uint64_t)(g < G ? g : g0) * 16;
                if (addr >= ringbytes) addr -= ringbytes;
                const gc_u4v t4 = *(gc_gptr_u4)(ring + addr);
                dst[it] = make_uint4(t4.x, t4.y, t4.z, t4.w);
            }
        }
    };
    uint4 vA[NIT], vB[NIT];
    GC_PSTAMP(1);                                       // unit and channel constants are here
    GcRound ronext;
    ronext.q0 = ronext.q1 = 0; ronext.clast = 0; ronext.w0 = 0; ronext.hint = 0;
    if (wv < nround) {
        load_round(wv, vA);                             // in flight while the tables are set up
        ronext = myrounds[wv];                          // (likewise; every round asks for the next one's record)
    }

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
        if (tid < ncar) { sk0[tid] = k0r; scar[tid] = carr; }
        if (tid == ncar) sk0[tid] = 0x7fffffff;
        if (tid >= 64 && tid - 64 < ncode) scode[tid - 64] = coder;
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
    // constant part of the wavefront's prefix image: entry 0 of every lane (nothing summed yet) and of the closing lane
    loc[lane * LPAD] = make_int2(0, 0);
    if (lane == 0) loc[GC_PS_WLANES * LPAD] = make_int2(0, 0);       // there P = lbase[64] = the round's total
    for (int x = tid; x < 4 * (2 * NTAP + 2); x += 256) red[x] = 0;  // wavefronts without a round leave their slots at zero
    // (the constants requested at the top are here by now: pinned, so that their loads cannot sink behind the barrier)
    asm volatile("" :: "s"(pm1_raw));
#pragma unroll
    for (int t = 0; t < NTAP; t++) asm volatile("" :: "s"(tapv[t]));
    __syncthreads();
    GC_PSTAMP(2);                                       // tables staged

    const bool pm1 = pm1_raw != 0;
    unsigned accI[NTAP], accQ[NTAP], finI = 0, finQ = 0;
    int toff[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; t++) {
        accI[t] = 0;
        accQ[t] = 0;
        toff[t] = smax + (t < ntap ? tapv[t] : 0) + klo;
    }
    int wseg = 0;                                       // wave-uniform: carrier piece of the round's first sample
    bool busy = false;                                  // wave-uniform: this wave ran a round
    // start sample of edge q: from the unit's edge table (trk_edges) when there is one, else searched here
    const bool have_etab = etab_u != nullptr && u.eq0 >= 0;
    // (a readable address in either case: the entry is requested unconditionally, see the rounds)
    const unsigned short __attribute__((address_space(1))) *etab_any =
        have_etab ? (const unsigned short __attribute__((address_space(1))) *)etab_u
                  : (const unsigned short __attribute__((address_space(1))) *)c.code;
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
        return edges[q < nedge ? q : nedge - 1];        // (q < nedge for every edge of a round; the clamp serves lanes without one)
    };
    // the wavefront's own order between its LDS writes and the reads of other lanes' entries (one wavefront's LDS
    // operations execute in order; this keeps the compiler from moving them across)
    auto wave_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    auto round = [&](int r, uint4 *vdata, uint4 *vnext) {
        // opaque copy of the lane id: per-lane address arithmetic stays inside the round instead of being
        // hoisted out of the loop over rounds into registers that would then spill
        int tl = lane;
        asm volatile("" : "+v"(tl));
        if (r == wv) GC_PSTAMP(11);                     // first round entered
        // (the round's record was asked for a round ago: nothing here waits for memory it has just requested)
        const GcRound ro = ronext;
        // Every pass issues the SAME vector-memory loads in the same order -- this lane's edge entry and its start
        // sample (index clamped into the range where the lane has no edge), then the samples of the wavefront's next
        // round (its last round asks for its own again) -- with no branch around any of them: the memory counter is
        // in-order, and only then can the compiler wait for exactly the loads that are older than the ones it still
        // wants in flight during the mixing phase (with a skipped load on any path it waits for all of them, the
        // samples just requested included).
        const int rn = r + 4 < nround ? r + 4 : r;
        ronext = myrounds[rn];
        const int rq0 = ro.q0, rq1 = ro.q1, rlast = ro.clast, rw0 = ro.w0, rhint = ro.hint;
        int q = rq0 + tl, ew = 0, ed = 0, jsraw = 0;
        const int q1 = (ablate & 1) ? 0 : rq1;
        {
            const int qc = q < q1 ? q : rq0;
            ed = edge_load(qc, rw0, &ew);
            int xe = have_etab ? qc - u.eq0 : 0;
            xe = xe < 0 ? 0 : (xe > GC_EDGTAB - 1 ? GC_EDGTAB - 1 : xe);
            jsraw = (int)etab_any[xe];
        }
        load_round(rn, vnext);
        if (r == wv) GC_PSTAMP(12);                     // first round: loads issued
        busy = true;
        const int kl = klo + r * RSAMP;                 // the round's first sample

        // ---- phase A: carrier mixing (ref src/sdrcmn.c:643-662) and running sums ------------
        int aI = 0, aQ = 0, js = 0;
        const int roff = r * RSAMP;
        // only the rounds that hold the period's first or last sample see samples outside [0, n): the others skip
        // the blanking test altogether
        const bool ragged = !(ablate & 4) && (kl < 0 || kl + RSAMP > n || g0 + (r + 1) * RGRP > G);
        // carrier pieces: the round's samples start in piece wseg; when no other piece starts
        // inside them (the common case -- a piece is a whole binade of the running phase) every lane
        // steps the same piece, otherwise each lane finds its own and switches where the next one starts
        // (every scan over the piece starts is bounded by the piece count: it never depends on the closing
        // sentinel alone -- an LDS read past the table returns 0 and would keep an unbounded scan going for ever)
        while (wseg + 1 < ncar && sk0[wseg + 1] <= kl) wseg++;
        const bool onepiece = (ablate & 8) || wseg + 1 >= ncar || sk0[wseg + 1] >= kl + RSAMP;
        if (r == wv) GC_PSTAMP(13);                     // first round: piece scan done
        auto run = [&](auto multi_tag) {
            constexpr bool MULTI = decltype(multi_tag)::value;
            int sp = wseg, knext = 0x7fffffff;
            const int kb0 = kl + tl * LSP;
            if (MULTI) {
                while (sp + 1 < ncar && sk0[sp + 1] <= kb0) sp++;
                knext = sp + 1 < ncar ? sk0[sp + 1] : 0x7fffffff;
            }
            unsigned long long dfx = scar[sp].dfx;
            unsigned long long phi = scar[sp].fx + (unsigned long long)(long long)(kb0 - sk0[sp]) * dfx;
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int gl = tl * NIT + it, g = g0 + r * RGRP + gl;
                uint4 &v = vdata[it];               // (blanked in place: the register set is dead after this round)
                const int kb = kl + gl * SPG;
                if (ragged) {
                    const bool edge = kb < 0 || kb + SPG > n || g >= G;
                    if (__ballot(edge) != 0ULL) {
                        if (edge) {                     // blank the samples outside [0, n): the group's valid bytes are [blo, bhi)
                            const int blo = kb < 0 ? -kb * DTYPE : 0;
                            int bhi = (n - kb) * DTYPE;
                            bhi = (g >= G || bhi < 0) ? 0 : (bhi > 16 ? 16 : bhi);
                            unsigned m[4];
#pragma unroll
                            for (int d = 0; d < 4; d++) {
                                int lo = blo - 4 * d, hi = bhi - 4 * d;
                                lo = lo < 0 ? 0 : (lo > 4 ? 4 : lo);
                                hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
                                // (hi > lo: 1 <= hi <= 4 and lo <= 3, both shifts stay below 32)
                                m[d] = hi > lo ? ((0xFFFFFFFFu >> (8 * (4 - hi))) & (0xFFFFFFFFu << (8 * lo))) : 0u;
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
        if (r == wv) GC_PSTAMP(3);                      // first round: edge record + piece scan done, samples needed now
        if (!(ablate & 2)) { if (onepiece) run(std::false_type{}); else run(std::true_type{}); }
        if (r == wv) GC_PSTAMP(4);                      // first round: mixing done
        // the start sample of this lane's chip edge (no LDS involved: overlaps the image writes)
        if (q < q1) js = (have_etab ? jsraw : edge_js(q, ed, ew, rhint)) - roff;
        const int sI = wave_scan(aI), sQ = wave_scan(aQ);
        lbase[tl] = make_int2(sI - aI, sQ - aQ);       // sum in front of this lane's span
        if (tl == 63) lbase[GC_PS_WLANES] = make_int2(sI, sQ);      // the round's total, where the closing lane looks it up
        {
            const int ti = __builtin_amdgcn_readlane(sI, 63), tq = __builtin_amdgcn_readlane(sQ, 63);
            finI += (unsigned)rlast * (unsigned)ti;         // c_b P(S), the term of the round's last chip
            finQ += (unsigned)rlast * (unsigned)tq;
        }
        wave_fence();
        if (r == wv) GC_PSTAMP(5);                      // first round: scan

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
                    // running sum inside the lane + lanes in front
                    const int2 a = loc[ee + col], b = lbase[col];
                    const unsigned pI = (unsigned)(a.x + b.x), pQ = (unsigned)(a.y + b.y);
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
                q += GC_PS_WLANES;
                if (q < q1) { ed = edge_load(q, rw0, &ew); js = edge_js(q, ed, ew, rhint) - roff; }
            }
        };
        if (pm1) lookups(std::true_type{}); else lookups(std::false_type{});
        wave_fence();                                   // look-ups done before the wavefront rewrites its image
        if (r == wv) GC_PSTAMP(7);                      // first round: look-ups
    };
    for (int r = wv; r < nround; r += 8) {
        round(r, vA, vB);
        if (r + 4 < nround) round(r + 4, vB, vA);
    }

    GC_PSTAMP(9);                                       // all rounds
    // wavefront then workgroup reduction (wavefronts without a round leave red[] at its initial zero)
    if (busy) {
        int *rw = red + wv * (2 * NTAP + 2);
#pragma unroll
        for (int t = 0; t < NTAP; t++) {
            const int si = wave_sum63((int)accI[t]), sq = wave_sum63((int)accQ[t]);
            if (lane == 63) {
                rw[t] = si;
                rw[NTAP + t] = sq;
            }
        }
        if (lane == 63) { rw[2 * NTAP] = (int)finI; rw[2 * NTAP + 1] = (int)finQ; }
    }
    __syncthreads();
    if (tid < ntap) {
        unsigned si = 0, sq = 0, fI = 0, fQ = 0;
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) {
            const int *rw = red + w4 * (2 * NTAP + 2);
            si += (unsigned)rw[tid];
            sq += (unsigned)rw[NTAP + tid];
            fI += (unsigned)rw[2 * NTAP];
            fQ += (unsigned)rw[2 * NTAP + 1];
        }
        if (pm1) { si *= 2u; sq *= 2u; }
        pout[tid] = (int)(si + fI);
        pout[ntap_stride + tid] = (int)(sq + fQ);
    }
    GC_PSTAMP(10);                                      // reduced and stored
}


}  // namespace
