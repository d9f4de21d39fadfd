// gnsscorr_plan.hip -- the tracking planner: the NCO chain of sdrtracking() from period to period
// (ref src/sdrtrk.c:31-43), bit for bit.  Discovery pass (trk_spec_kernel), the evaluating chain
// (trk_plan2_kernel) and the chain that certifies its own crossings (trk_plan_kernel); DESIGN.md 3.1,
// gnsscorr_nco.h.  Compiled on its own: the step instances make it the longest translation unit.
#include <cstdlib>
#include <cstddef>
#include <hip/hip_runtime.h>

#include "gnsscorr_internal.h"

namespace {

// ---------------------------------------------------------------------------
// NCO chain of sdrtracking() (ref src/sdrtrk.c:31-43): the reference's running
// fp64 sums walked piece by piece (gnsscorr_nco.h), bit for bit
// ---------------------------------------------------------------------------
// One channel per wavefront: the chain of a channel is sequential in its periods, the channels run side
// by side on different compute units.  Per period the lanes compute, one binade boundary each, where the
// running sums cross it (certified against the accumulated rounding, gnsscorr_nco.h); what is left
// to the sequential chain is one fma and one addition per binade.

// Discovery pass of the batch planner: one lane per (channel, period).  From the batch's start state and the
// closed-form period starts (gc_spec_start) it runs the period steps in their discovering form
// (gnsscorr_nco.h: "period steps on claims") and keeps the structure they find: GC_CLAIM_ROW ints per NCO
// and period.  What is sequential in a batch -- trk_plan2_kernel -- then only evaluates and checks.
__global__ __launch_bounds__(64) void trk_spec_kernel(const GcChan *__restrict__ chan, const GcTrkState *__restrict__ state_in,
                                                       int *__restrict__ claims_code, int *__restrict__ claims_car,
                                                       int nch, int nepoch, int e_off)
{
    // e_off: periods between the state handed in and the batch's first period (0: the batch starts at that
    // state; nepoch: the state is the start of the batch BEFORE this one, whose chain is still running -- the
    // closed forms reach over it just as well, and the claims are checked either way)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nch * nepoch) return;
    const int ch = i / nepoch, e = i - ch * nepoch;
    const GcChan &c = chan[ch];
    const GcTrkState s = state_in[ch];
    GcCodeClaims cc;
    GcCarClaims ck;
    cc.tag = 0;
    ck.tag = 0;
    const double ci = __dmul_rn(c.ti, s.codefreq), spc = __ddiv_rn(s.codefreq, c.f_sf), dlen = (double)c.clen;
    if (ci > 0.0 && ci < dlen && spc > 1e-300 && spc < 1e300) {
        double remcode, remcarr, dummy;
        int n;
        const double ps = gc_carrier_ps(s.carrfreq, c.ti);
        gc_spec_start(s.remcode, s.remcarr, ci, spc, ps, dlen, e + e_off, &remcode, &remcarr, &n);
        if (n > 0 && n <= (1 << 24)) {
            {
                GcCodePlan PC;
                gc_code_plan_init(PC, ci, c.clen, c.smax);
                gc_code_claims<true>(PC, remcode, n + 2 * c.smax, cc, &dummy);
            }
            {
                GcCarPlan PK;
                gc_car_plan_init(PK, ps, false, false);
                GcCarStepC CK;
                gc_car_stepc_init(CK, PK, c.nsamp + 16);
                gc_carrier_claims_step<true>(PK, CK, remcarr, n, ck, &dummy);
            }
        }
    }
    int4 *rc = reinterpret_cast<int4 *>(claims_code) + (size_t)i * (GC_CLAIM_ROW / 4);
    int4 *rk = reinterpret_cast<int4 *>(claims_car) + (size_t)i * (GC_CLAIM_ROW / 4);
    const int *pc = reinterpret_cast<const int *>(&cc), *pk = reinterpret_cast<const int *>(&ck);
#pragma unroll
    for (int q = 0; q < GC_CLAIM_ROW / 4; q++) {
        rc[q] = make_int4(pc[4 * q], pc[4 * q + 1], pc[4 * q + 2], pc[4 * q + 3]);
        rk[q] = make_int4(pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]);
    }
}

// which path served the periods of the batches planned so far: [0] code on claims, [1] code certified,
// [2] code walkers, [3..5] the same for the carrier (tools/debug, tests)
__device__ unsigned long long gc_plan_stats[8];

// Which instance of the batch chain serves a channel: the table binade that holds the code length (the shape
// of the code step: 2..64 samples per chip give 7..12) and whether the tail fits 8, 15 or 32 positions.  -1: none
// (the chain below serves it).  Host and device use the same function.
__host__ __device__ inline int plan2_class(double ti, double codefreq, int clen, int smax)
{
    const double ci = ti * codefreq;
    const uint64_t us = gc_d2u(ci);
    const int es = (int)((us >> 52) & 0x7FF);
    if (!(ci > 0.0) || es <= 60 || es >= 0x7FF - GC_NB - 4) return -1;
    const int itop = gc_expo(gc_u2d(gc_d2u((double)clen) - 1)) - (es + 2);
    if (itop < 7 || itop > 12) return -1;
    if (smax + 1 > GC_CLAIM_TAIL2) return -1;      // (tail longer than the widest instance: the certifying chain)
    return (itop - 7) * 3 + (smax + 1 > 8 ? (smax + 1 > GC_CLAIM_TAIL ? 2 : 1) : 0);
}

// Two wavefronts per channel: wavefront 0 chains the code NCO (and with it the samples per period and the
// buffer positions), wavefront 1 follows one step behind with the carrier NCO, which needs only the
// period lengths -- the two chains are independent otherwise and each is latency bound.
#define GC_PLAN_MAXE 4096          // periods per batch handed from wave to wave through LDS (longer: one wave does both)
__device__ __attribute__((noinline)) void trk_plan_body(const GcChan *__restrict__ chan,
                                                        const GcTrkState *__restrict__ state_in,
                                                        GcTrkState *__restrict__ state_out,
                                                        GcTrkPlan *__restrict__ plan, int nch, int nepoch)
{
    __shared__ int Ks2[2][GC_NB + 2];
    __shared__ int nsh[GC_PLAN_MAXE];
    __shared__ int prog;
    const int ch = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (ch >= nch) return;
    // the chain is latency bound and shares its SIMD with correlator wavefronts of the batch before:
    // let it issue first
    __builtin_amdgcn_s_setprio(3);
    const bool split = nepoch <= GC_PLAN_MAXE;          // else wavefront 0 does both chains
    if (threadIdx.x == 0) prog = 0;
    __syncthreads();
    if (!split && wave == 1) return;
    int *Ks = Ks2[wave];
    const GcChan c = chan[ch];
    GcTrkState s = state_in[ch];
    const double ci = __dmul_rn(c.ti, s.codefreq);          // ti*crate, ref src/sdrcmn.c:709
    const double spc = __ddiv_rn(s.codefreq, c.f_sf);       // chips per sample
    const double ps = gc_carrier_ps(s.carrfreq, c.ti);
    const double dlen = (double)c.clen;
    const bool code_ok = ci > 0.0 && ci < dlen;             // the reference's one-subtraction wrap (:617) needs it
    GcTrkPlan *out = plan + (size_t)ch * nepoch;
    GcNoEmit ne;
    GcFillLanes fill{lane};
    const bool do_code = wave == 0, do_car = wave == 1 || !split;
    // per-binade constants of the addends (the frequencies are held over the batch) and the
    // shape-specialised period steps built on them
    GcCodePlan PC;
    GcCarPlan PK;
    if (do_code) gc_code_plan_init(PC, ci, c.clen, c.smax);
    if (do_car) gc_car_plan_init(PK, ps);
    const GcNcoFast &fcode = PC.f, &fcar = PK.f, &fprem = PK.fprem;
    const double yspc = __ddiv_rn(1.0, spc), ydpi = __ddiv_rn(1.0, GC_NCO_DPI);
    const double smaxci = __dmul_rn((double)c.smax, ci);
    const bool fastdiv = spc > 1e-300 && spc < 1e300 && yspc < 1e300;
    unsigned tally[6] = {0, 0, 0, 0, 0, 0};
    for (int e = 0; e < nepoch; e++) {
        int n;
        if (do_code) {
            const double num = __dsub_rn(dlen, s.remcode);                      // ref src/sdrtrk.c:31-32
            const double q = fastdiv ? gc_div_y(num, spc, yspc) : __ddiv_rn(num, spc);
            n = (q > -2147483648.0 && q < 2147483648.0) ? (int)q : 0;
            if (lane == 0) {
                out[e].buffloc = s.buffloc;
                out[e].coff = s.remcode;
                out[e].carrfreq = s.carrfreq;
                out[e].codefreq = s.codefreq;
                out[e].n = n;
                out[e].pad = 0;
            }
            if (split) {
                if (lane == 0) nsh[e] = n;
                __threadfence_block();
                if (lane == 0) __hip_atomic_store(&prog, e + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            while (__hip_atomic_load(&prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= e) __builtin_amdgcn_s_sleep(1);
            n = nsh[e];
        }
        const bool walk = n > 0 && n <= (1 << 24);
        if (do_car) {
            if (lane == 0) out[e].phi0 = s.remcarr;
            double rp;
            if (walk && gc_carrier_period(PK, s.remcarr, n, fill, &rp)) {
                s.remcarr = rp;
                tally[4]++;
            } else if (walk) {      // any other shape: the general walkers
                tally[5]++;
                const double phis = gc_div_y(__dmul_rn(s.remcarr, GC_NCO_CDIV), GC_NCO_DPI, ydpi);     // ref src/sdrcmn.c:649
                double xn;
                if (!plan_carrier_dev(fcar, phis, n, Ks, lane, &xn)) xn = gc_fast_carrier_walk(fcar, phis, n, ne);
                s.remcarr = gc_fast_prem(fprem, xn);
            }
        }
        if (do_code) {
            double rc;
            if (walk && code_ok && gc_code_period(PC, s.remcode, n + 2 * c.smax, fill, &rc)) {
                s.remcode = rc;
                tally[1]++;
            } else if (walk && code_ok) {
                tally[2]++;
                const double c0 = gc_code_start_fast(s.remcode, smaxci, c.clen);
                double cend;
                if (!plan_code_dev(fcode, c0, c.clen, n + 2 * c.smax, Ks, lane, &cend))
                    cend = gc_fast_code_walk(fcode, c0, c.clen, n + 2 * c.smax, ne);
                s.remcode = __dsub_rn(cend, smaxci);
            }
            s.buffloc += (uint64_t)(int64_t)n;
        }
    }
    if (lane == 0) {
        if (do_code) {
            state_out[ch].carrfreq = s.carrfreq;
            state_out[ch].codefreq = s.codefreq;
            state_out[ch].remcode = s.remcode;
            state_out[ch].buffloc = s.buffloc;
        }
        if (do_car) state_out[ch].remcarr = s.remcarr;
#pragma unroll
        for (int t = 0; t < 6; t++)
            if (tally[t]) atomicAdd(&gc_plan_stats[t], (unsigned long long)tally[t]);
    }
}

__global__ __launch_bounds__(128) void trk_plan_kernel(const GcChan *__restrict__ chan, const GcTrkState *__restrict__ state_in,
                                                       GcTrkState *__restrict__ state_out, GcTrkPlan *__restrict__ plan,
                                                       int nch, int nepoch)
{
    trk_plan_body(chan, state_in, state_out, plan, nch, nepoch);
}

// ---- the batch planner's chain: evaluate and check (gnsscorr_nco.h: "period steps on claims") ----
// Same two wavefronts per channel.  Each stages the claims of GC_PLAN_BLK periods at a time in LDS (the next
// block's rows are in flight while this one is evaluated), keeps the periods' results in its lanes (lane l:
// period l of the block) and writes them out once per block; the carrier wavefront starts a block when the
// code wavefront has finished it, so the period lengths it needs are all there.
#define GC_PLAN_BLK 32
struct Plan2Shared {
    int Ks2[2][GC_NB + 2];
    int nsh[GC_PLAN_MAXE];
    int prog;
    int pad[3];
    int rows[2][2][(GC_PLAN_BLK + 1) * GC_CLAIM_ROW];     // (+1 row: the row after a block's last is read, never used)
};

#ifdef GC_PLAN_PROF
__device__ unsigned long long gc_plan_prof[16];
#define GC_PP(i) do { const unsigned long long t1_ = __builtin_readcyclecounter(); pp[i] += t1_ - pp_t; pp_t = t1_; } while (0)
#else
#define GC_PP(i) do { } while (0)
#endif
// (file scope: the two wave functions below are called, not inlined -- each instance of the code step gets its
// own register allocation -- and reach the workgroup's LDS by name)
__shared__ __attribute__((aligned(16))) Plan2Shared g_plan2;

// (arguments of a called function arrive in vector registers: what is the same in every lane is said so)
__device__ __forceinline__ int plan2_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T *plan2_uni(T *p)
{
    const uint64_t u = (uint64_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return (T *)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double plan2_uni(double x)
{
    const uint64_t u = gc_d2u(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return gc_u2d(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void plan2_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// claims rows as the kernels move them (five 16-byte quads) -> fields
__device__ __forceinline__ GcCodeClaims plan2_code_row(const int4 (&v)[GC_CLAIM_ROW / 4])
{
    GcCodeClaims c;
    c.tag = v[0].x; c.i0 = v[0].y; c.q = v[0].z; c.nl = v[0].w;
    c.jsum = v[1].x; c.dm[0] = v[1].y; c.dm[1] = v[1].z; c.dm[2] = v[1].w;
    c.dm[3] = v[2].x; c.dm[4] = v[2].y; c.dm[5] = v[2].z; c.dm[6] = v[2].w;
    c.dm[7] = v[3].x; c.dm[8] = v[3].y; c.dm[9] = v[3].z; c.dm[10] = v[3].w;
    c.dm[11] = v[4].x; c.dm[12] = v[4].y; c.pad[0] = 0; c.pad[1] = 0;
    return c;
}
__device__ __forceinline__ GcCarClaims plan2_car_row(const int4 (&v)[GC_CLAIM_ROW / 4])
{
    GcCarClaims c;
    c.tag = v[0].x; c.nl = v[0].y; c.i0 = v[0].z; c.nseg = v[0].w;
    c.kprem = v[1].x; c.dm[0] = v[1].y; c.dm[1] = v[1].z; c.dm[2] = v[1].w;
    c.dm[3] = v[2].x; c.dm[4] = v[2].y; c.dm[5] = v[2].z; c.dm[6] = v[2].w;
    c.dm[7] = v[3].x; c.dm[8] = v[3].y; c.dm[9] = v[3].z; c.dm[10] = v[3].w;
    c.dm[11] = v[4].x; c.dm[12] = v[4].y; c.pad[0] = 0; c.pad[1] = 0;
    return c;
}
// "these claims are here": the LDS reads that brought them were issued a period ago.  Said before the next
// period's reads are issued, it keeps the compiler from waiting for THOSE at the first use of these.
template <class T>
__device__ __forceinline__ void plan2_touch(T &c)
{
    int *p = reinterpret_cast<int *>(&c);
#pragma unroll
    for (int i = 0; i < GC_CLAIM_ROW; i++) asm volatile("" : "+v"(p[i]));
}
static_assert(sizeof(GcCodeClaims) == GC_CLAIM_ROW * 4 && sizeof(GcCarClaims) == GC_CLAIM_ROW * 4, "claims rows are GC_CLAIM_ROW ints");
static_assert(offsetof(GcCodeClaims, dm) == 20 && offsetof(GcCarClaims, dm) == 20, "claims layout");

// A period whose claims did not hold (a few in ten thousand): the certified step, then the walkers, with the
// tables they need built here -- out of line, so that the chain's loop carries none of it.
// returns 1: certified step, 2: walkers
__device__ __attribute__((noinline)) int plan2_code_slow(double ci_, int clen_, int smax_, double remcode_, int n_, int lane, double *out)
{
    const double ci = plan2_uni(ci_), remcode = plan2_uni(remcode_);
    const int clen = plan2_uni(clen_), smax = plan2_uni(smax_), n = plan2_uni(n_);
    GcCodePlan PC;
    gc_code_plan_init(PC, ci, clen, smax);
    GcFillLanes fill{lane};
    GcNoEmit ne;
    double rc;
    if (gc_code_period(PC, remcode, n + 2 * smax, fill, &rc)) { *out = rc; return 1; }
    const double smaxci = __dmul_rn((double)smax, ci);
    const double c0 = gc_code_start_fast(remcode, smaxci, clen);
    double cend;
    if (!plan_code_dev(PC.f, c0, clen, n + 2 * smax, g_plan2.Ks2[0], lane, &cend))
        cend = gc_fast_code_walk(PC.f, c0, clen, n + 2 * smax, ne);
    *out = __dsub_rn(cend, smaxci);
    return 2;
}

__device__ __attribute__((noinline)) int plan2_car_slow(double ps_, double remcarr_, int n_, int lane, double *out)
{
    const double ps = plan2_uni(ps_), remcarr = plan2_uni(remcarr_);
    const int n = plan2_uni(n_);
    GcCarPlan PK;
    gc_car_plan_init(PK, ps);
    GcFillLanes fill{lane};
    GcNoEmit ne;
    double rp;
    if (gc_carrier_period(PK, remcarr, n, fill, &rp)) { *out = rp; return 1; }
    const double phis = gc_div_y(__dmul_rn(remcarr, GC_NCO_CDIV), GC_NCO_DPI, __ddiv_rn(1.0, GC_NCO_DPI));     // ref src/sdrcmn.c:649
    double xn;
    if (!plan_carrier_dev(PK.f, phis, n, g_plan2.Ks2[1], lane, &xn)) xn = gc_fast_carrier_walk(PK.f, phis, n, ne);
    *out = gc_fast_prem(PK.fprem, xn);
    return 2;
}

template <int ITOP, int TMAX>
__device__ __attribute__((noinline)) void plan2_code_wave(const GcChan &c_, GcTrkState s_, GcTrkState *__restrict__ state_out_,
                                                GcTrkPlan *__restrict__ out_, int nepoch_, const int4 *__restrict__ src_, int lane, int dbg_)
{
    // the channel constants the loop needs, as values (the reference points into the caller's frame)
    struct { double ti, f_sf; int clen, smax; } c;
    c.ti = plan2_uni(c_.ti);
    c.f_sf = plan2_uni(c_.f_sf);
    c.clen = plan2_uni(c_.clen);
    c.smax = plan2_uni(c_.smax);
    GcTrkState s;
    s.carrfreq = plan2_uni(s_.carrfreq);
    s.codefreq = plan2_uni(s_.codefreq);
    s.remcode = plan2_uni(s_.remcode);
    s.remcarr = plan2_uni(s_.remcarr);
    s.buffloc = gc_d2u(plan2_uni(gc_u2d(s_.buffloc)));
    GcTrkState *__restrict__ state_out = plan2_uni(state_out_);
    GcTrkPlan *__restrict__ out = plan2_uni(out_);
    const int4 *__restrict__ src = plan2_uni(src_);
    const int nepoch = plan2_uni(nepoch_), dbg = plan2_uni(dbg_);
    const double ci = __dmul_rn(c.ti, s.codefreq);
    const double spc = __ddiv_rn(s.codefreq, c.f_sf);
    const double dlen = (double)c.clen;
    const bool code_ok = ci > 0.0 && ci < dlen;
    GcCodePlan PC;
    gc_code_plan_init(PC, ci, c.clen, c.smax);
    GcCodeStepC<ITOP> SC;
    gc_code_stepc_init(SC, PC);
    const double yspc = __ddiv_rn(1.0, spc);
    const bool fastdiv = spc > 1e-300 && spc < 1e300 && yspc < 1e300;
    int (*rows)[(GC_PLAN_BLK + 1) * GC_CLAIM_ROW] = g_plan2.rows[0];
    constexpr int RQ = GC_CLAIM_ROW / 4;
    // (a zero the compiler takes for a per-lane value: the claims read through it stay in vector registers
    // until they are used, so the next period's row really is in flight during this period's step)
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    unsigned tally[3] = {0, 0, 0};
#ifdef GC_PLAN_PROF
    unsigned long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pp_t = __builtin_readcyclecounter();
#endif
    const int nblk = (nepoch + GC_PLAN_BLK - 1) / GC_PLAN_BLK;
    int4 pf[RQ];
    auto fetch_block = [&](int b) {
        const int e = b * GC_PLAN_BLK + lane;
        if (lane < GC_PLAN_BLK && e < nepoch) {
#pragma unroll
            for (int q = 0; q < RQ; q++) pf[q] = src[(size_t)e * RQ + q];
        }
    };
    auto store_block = [&](int buf) {
        if (lane < GC_PLAN_BLK) {
#pragma unroll
            for (int q = 0; q < RQ; q++) reinterpret_cast<int4 *>(rows[buf])[lane * RQ + q] = pf[q];
        }
        plan2_wave_sync();
    };
#pragma unroll
    for (int q = 0; q < RQ; q++) pf[q] = make_int4(0, 0, 0, 0);
    fetch_block(0);
    store_block(0);
    for (int b = 0; b < nblk; b++) {
        const int buf = b & 1, e0 = b * GC_PLAN_BLK;
        const int e1 = e0 + GC_PLAN_BLK < nepoch ? e0 + GC_PLAN_BLK : nepoch;
        if (b + 1 < nblk) fetch_block(b + 1);
        uint64_t k_buff = 0;
        double k_coff = 0.0;
        int k_n = 0;
        GcCodeClaims nx;
        {
            const int4 *r = reinterpret_cast<const int4 *>(rows[buf]) + vzero;
            int4 v[RQ];
#pragma unroll
            for (int q = 0; q < RQ; q++) v[q] = r[q];
            nx = plan2_code_row(v);
        }
        for (int e = e0; e < e1; e++) {
            GC_PP(0);
            GcCodeClaims cl = nx;
            plan2_touch(cl);
            {                                       // the next period's claims, in flight during this one
                const int4 *r = reinterpret_cast<const int4 *>(rows[buf]) + (e + 1 - e0) * RQ + vzero;
                int4 v[RQ];
#pragma unroll
                for (int q = 0; q < RQ; q++) v[q] = r[q];
                nx = plan2_code_row(v);
            }
            const double num = __dsub_rn(dlen, s.remcode);                      // ref src/sdrtrk.c:31-32
            double qn = gc_div_y(num, spc, yspc);
            if (__builtin_expect(!fastdiv, 0)) qn = __ddiv_rn(num, spc);
            const int n = plan2_uni((qn > -2147483648.0 && qn < 2147483648.0) ? (int)qn : 0);     // (the same in every lane)
            const bool mine = lane == e - e0;
            k_buff = mine ? s.buffloc : k_buff;
            k_coff = mine ? s.remcode : k_coff;
            k_n = mine ? n : k_n;
            g_plan2.nsh[e] = n;                          // (every lane, the same value)
            GC_PP(1);
            const bool walk = n > 0 && n <= (1 << 24) && code_ok && !(dbg & 1);
            double rc, rcf;
            if (__builtin_expect(walk && gc_code_claims_step<ITOP, TMAX, false>(PC, SC, s.remcode, n + 2 * c.smax, cl, &rcf), 1)) {
                s.remcode = rcf;
                tally[0]++;
            } else if (walk) {
                tally[plan2_code_slow(ci, c.clen, c.smax, s.remcode, n, lane, &rc)]++;
                s.remcode = rc;
            }
            s.buffloc += (uint64_t)(int64_t)n;
            GC_PP(2);
        }
        if (lane < e1 - e0) {
            GcTrkPlan &o = out[e0 + lane];
            o.buffloc = k_buff;
            o.coff = k_coff;
            o.carrfreq = s.carrfreq;
            o.codefreq = s.codefreq;
            o.n = k_n;
            o.pad = 0;
        }
        __threadfence_block();
        if (lane == 0) __hip_atomic_store(&g_plan2.prog, e1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (b + 1 < nblk) store_block(buf ^ 1);
        GC_PP(3);
    }
#ifdef GC_PLAN_PROF
    if (lane == 0 && blockIdx.x == 0)
        for (int i = 0; i < 4; i++) atomicAdd(&gc_plan_prof[i], pp[i]);
#endif
    if (lane == 0) {
        state_out->carrfreq = s.carrfreq;
        state_out->codefreq = s.codefreq;
        state_out->remcode = s.remcode;
        state_out->buffloc = s.buffloc;
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (tally[t]) atomicAdd(&gc_plan_stats[t], (unsigned long long)tally[t]);
    }
}

__device__ __attribute__((noinline)) void plan2_car_wave(const GcChan &c_, GcTrkState s_, GcTrkState *__restrict__ state_out_,
                                               GcTrkPlan *__restrict__ out_, int nepoch_, const int4 *__restrict__ src_, int lane, int dbg_)
{
    struct { double ti; int nsamp; } c;
    c.ti = plan2_uni(c_.ti);
    c.nsamp = plan2_uni(c_.nsamp);
    GcTrkState s;
    s.carrfreq = plan2_uni(s_.carrfreq);
    s.codefreq = plan2_uni(s_.codefreq);
    s.remcode = plan2_uni(s_.remcode);
    s.remcarr = plan2_uni(s_.remcarr);
    s.buffloc = s_.buffloc;
    GcTrkState *__restrict__ state_out = plan2_uni(state_out_);
    GcTrkPlan *__restrict__ out = plan2_uni(out_);
    const int4 *__restrict__ src = plan2_uni(src_);
    const int nepoch = plan2_uni(nepoch_), dbg = plan2_uni(dbg_);
    const double ps = gc_carrier_ps(s.carrfreq, c.ti);
    GcCarPlan PK;
    gc_car_plan_init(PK, ps, false, false);
    GcCarStepC CK;
    gc_car_stepc_init(CK, PK, c.nsamp + 16);
    int (*rows)[(GC_PLAN_BLK + 1) * GC_CLAIM_ROW] = g_plan2.rows[1];
    constexpr int RQ = GC_CLAIM_ROW / 4;
    // (a zero the compiler takes for a per-lane value: the claims read through it stay in vector registers
    // until they are used, so the next period's row really is in flight during this period's step)
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    unsigned tally[3] = {0, 0, 0};
#ifdef GC_PLAN_PROF
    unsigned long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pp_t = __builtin_readcyclecounter();
#endif
    const int nblk = (nepoch + GC_PLAN_BLK - 1) / GC_PLAN_BLK;
    int4 pf[RQ];
    auto fetch_block = [&](int b) {
        const int e = b * GC_PLAN_BLK + lane;
        if (lane < GC_PLAN_BLK && e < nepoch) {
#pragma unroll
            for (int q = 0; q < RQ; q++) pf[q] = src[(size_t)e * RQ + q];
        }
    };
    auto store_block = [&](int buf) {
        if (lane < GC_PLAN_BLK) {
#pragma unroll
            for (int q = 0; q < RQ; q++) reinterpret_cast<int4 *>(rows[buf])[lane * RQ + q] = pf[q];
        }
        plan2_wave_sync();
    };
#pragma unroll
    for (int q = 0; q < RQ; q++) pf[q] = make_int4(0, 0, 0, 0);
    fetch_block(0);
    store_block(0);
    for (int b = 0; b < nblk; b++) {
        const int buf = b & 1, e0 = b * GC_PLAN_BLK;
        const int e1 = e0 + GC_PLAN_BLK < nepoch ? e0 + GC_PLAN_BLK : nepoch;
        if (b + 1 < nblk) fetch_block(b + 1);
        // the code wavefront has finished this block: its period lengths are in nsh[]
        while (__hip_atomic_load(&g_plan2.prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < e1) __builtin_amdgcn_s_sleep(1);
        double k_phi = 0.0;
        GcCarClaims nx;
        int nn;
        {
            const int4 *r = reinterpret_cast<const int4 *>(rows[buf]) + vzero;
            int4 v[RQ];
#pragma unroll
            for (int q = 0; q < RQ; q++) v[q] = r[q];
            nx = plan2_car_row(v);
            nn = g_plan2.nsh[e0];
        }
        GC_PP(4);
        for (int e = e0; e < e1; e++) {
            GC_PP(5);
            GcCarClaims cl = nx;
            plan2_touch(cl);
            const int n = plan2_uni(nn);
            {
                const int4 *r = reinterpret_cast<const int4 *>(rows[buf]) + (e + 1 - e0) * RQ + vzero;
                int4 v[RQ];
#pragma unroll
                for (int q = 0; q < RQ; q++) v[q] = r[q];
                nx = plan2_car_row(v);
                nn = g_plan2.nsh[e + 1 < GC_PLAN_MAXE ? e + 1 : e];
            }
            const bool mine = lane == e - e0;
            k_phi = mine ? s.remcarr : k_phi;
            GC_PP(6);
            const bool walk = n > 0 && n <= (1 << 24) && !(dbg & 2);
            double rp, rpf;
            if (__builtin_expect(walk && gc_carrier_claims_step<false>(PK, CK, s.remcarr, n, cl, &rpf), 1)) {
                s.remcarr = rpf;
                tally[0]++;
            } else if (walk) {
                tally[plan2_car_slow(ps, s.remcarr, n, lane, &rp)]++;
                s.remcarr = rp;
            }
            GC_PP(7);
        }
        GC_PP(5);
        if (lane < e1 - e0) out[e0 + lane].phi0 = k_phi;
        if (b + 1 < nblk) store_block(buf ^ 1);
    }
#ifdef GC_PLAN_PROF
    if (lane == 0 && blockIdx.x == 0)
        for (int i = 4; i < 8; i++) atomicAdd(&gc_plan_prof[i], pp[i]);
#endif
    if (lane == 0) {
        state_out->remcarr = s.remcarr;
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (tally[t]) atomicAdd(&gc_plan_stats[3 + t], (unsigned long long)tally[t]);
    }
}

__global__ __launch_bounds__(128) void trk_plan2_kernel(const GcChan *__restrict__ chan,
                                                        const GcTrkState *__restrict__ state_in,
                                                        GcTrkState *__restrict__ state_out,
                                                        GcTrkPlan *__restrict__ plan, int nch, int nepoch,
                                                        const int *__restrict__ claims_code, const int *__restrict__ claims_car, int dbg)
{
    const int ch = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (ch >= nch) return;
    const int cls = plan2_class(chan[ch].ti, state_in[ch].codefreq, chan[ch].clen, chan[ch].smax);
    if (cls < 0) {                  // no instance of the batch chain for this channel: the chain that certifies its own crossings
        trk_plan_body(chan, state_in, state_out, plan, nch, nepoch);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) g_plan2.prog = 0;
    __syncthreads();
    const GcChan c = chan[ch];
    const GcTrkState s = state_in[ch];
    GcTrkPlan *out = plan + (size_t)ch * nepoch;
    constexpr int RQ = GC_CLAIM_ROW / 4;
    if (wave == 1) {
        plan2_car_wave(c, s, state_out + ch, out, nepoch, reinterpret_cast<const int4 *>(claims_car) + (size_t)ch * nepoch * RQ, lane, dbg);
        return;
    }
    const int4 *src = reinterpret_cast<const int4 *>(claims_code) + (size_t)ch * nepoch * RQ;
    switch (cls) {
    case 0:  plan2_code_wave<7, 8>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 1:  plan2_code_wave<7, GC_CLAIM_TAIL>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 2:  plan2_code_wave<7, GC_CLAIM_TAIL2>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 3:  plan2_code_wave<8, 8>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 4:  plan2_code_wave<8, GC_CLAIM_TAIL>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 5:  plan2_code_wave<8, GC_CLAIM_TAIL2>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 6:  plan2_code_wave<9, 8>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 7:  plan2_code_wave<9, GC_CLAIM_TAIL>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 8:  plan2_code_wave<9, GC_CLAIM_TAIL2>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 9:  plan2_code_wave<10, 8>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 10: plan2_code_wave<10, GC_CLAIM_TAIL>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 11: plan2_code_wave<10, GC_CLAIM_TAIL2>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 12: plan2_code_wave<11, 8>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 13: plan2_code_wave<11, GC_CLAIM_TAIL>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 14: plan2_code_wave<11, GC_CLAIM_TAIL2>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 15: plan2_code_wave<12, 8>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    case 16: plan2_code_wave<12, GC_CLAIM_TAIL>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    default: plan2_code_wave<12, GC_CLAIM_TAIL2>(c, s, state_out + ch, out, nepoch, src, lane, dbg); break;
    }
}


}  // namespace

#ifdef GC_PLAN_PROF
extern "C" int gnsscorr_debug_plan_prof(unsigned long long *dst)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(gc_plan_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int gnsscorr_debug_plan_stats(unsigned long long *dst, int reset)
{
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(gc_plan_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gc_plan_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}

static bool trk_nospec()
{
    static const bool nospec = getenv("GNSSCORR_TRK_NOSPEC") != nullptr;
    return nospec;
}

// claims: 2 * nch * nepoch * GC_CLAIM_ROW ints of scratch (code rows, then carrier rows)
int gc_launch_trk_spec(hipStream_t st, const GcChan *chan, const GcTrkState *state_in, int nch, int nepoch, int *claims,
                       int e_off)
{
    if (!claims || trk_nospec() || nepoch > GC_PLAN_MAXE) return 0;
    const int total = nch * nepoch;
    hipLaunchKernelGGL(trk_spec_kernel, dim3((total + 63) / 64), dim3(64), 0, st, chan, state_in, claims,
                       claims + (size_t)nch * nepoch * GC_CLAIM_ROW, nch, nepoch, e_off);
    GC_HIP(hipGetLastError());
    return 0;
}

// claims: filled by gc_launch_trk_spec for the same state and batch -> the batch form of the chain (evaluate and
// check) for every channel it has an instance for (plan2_class); inside the same launch the chain that certifies
// its crossings itself, period by period, serves the rest -- and everything when claims is null or the batch is
// longer than GC_PLAN_MAXE periods.
int gc_launch_trk_plan(hipStream_t st, const GcChan *chan, const GcTrkState *state_in, GcTrkState *state_out,
                       GcTrkPlan *plan, int nch, int nepoch, int *claims)
{
    static const int dbg = getenv("GNSSCORR_PLAN_DBG") ? atoi(getenv("GNSSCORR_PLAN_DBG")) : 0;
    const bool batch = claims && !trk_nospec() && nepoch <= GC_PLAN_MAXE;
    if (batch)
        hipLaunchKernelGGL(trk_plan2_kernel, dim3(nch), dim3(128), 0, st, chan, state_in, state_out, plan, nch, nepoch,
                           claims, claims + (size_t)nch * nepoch * GC_CLAIM_ROW, dbg);
    else
        hipLaunchKernelGGL(trk_plan_kernel, dim3(nch), dim3(128), 0, st, chan, state_in, state_out, plan, nch, nepoch);
    GC_HIP(hipGetLastError());
    return 0;
}

