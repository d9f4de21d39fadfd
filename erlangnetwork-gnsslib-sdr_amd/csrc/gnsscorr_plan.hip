// gnsscorr_plan.hip -- the tracking planner: the NCO chain of sdrtracking() from period to period
// (ref src/sdrtrk.c:31-43), bit for bit.  Discovery pass (trk_spec_kernel), the evaluating chain
// (trk_plan3_kernel) and the chain that certifies its own crossings (trk_plan_kernel); DESIGN.md 3.1,
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
// and period.  What is sequential in a batch -- trk_plan3_kernel -- then only evaluates (and checks beside it).
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

// ---- the batch planner's chain (gnsscorr_nco.h: "period steps on claims") ----
// What the step on claims computes falls in two parts of very different cost: the VALUES -- the period's start
// carried through ~60 dependent fp64 operations to the next period's start: 0.28 us (code) / 0.39 us (carrier) per
// period on a lone wavefront -- and the CHECKS that make the claims the definitions they are: another 0.4 us of
// compares and mask arithmetic in the same instruction stream (tools/ubench/claims_chain.hip).  Only the values are
// sequential.  So one wavefront per NCO chains the values (the same step function, its verdict unused: the compiler
// drops the checks) and publishes every period's start in LDS; three more per NCO take the periods in turn, run the
// complete step from the published start with the same claims, and report the first period whose claims do not hold
// or whose value differs.  The batch goes by in blocks of GC_P3_BLK periods: at the end of a block the workgroup
// meets; if a period failed, the block is redone from that period on -- its own step by the certified path (exact
// whatever the claims say), the rest on claims again -- until no check fails (a few periods in ten thousand fail).
// Every value that leaves the kernel has therefore been produced by a step that passed its checks, or by the
// certified step: the same guarantee as the chain that evaluated and checked in one wavefront (round 2: 0.94 us per
// period; this: ~0.45).
#define GC_P3_BLK 64            // periods per block
#define GC_P3_NW 8              // wavefronts: code chain, carrier chain, 3 code checkers, 3 carrier checkers
#define GC_P3_NCHK 3
#define GC_P3_NONE 0x7fffffff
struct Plan3Shared {
    int rows[3][2][GC_P3_BLK * GC_CLAIM_ROW];      // [block mod 3][code | carrier][period of the block][GC_CLAIM_ROW]
    double vcode[2][GC_P3_BLK + 1];                 // [block parity] remcode at the start of each period of the block (+ the next block's first)
    unsigned long long vbuff[2][GC_P3_BLK + 1];     // buffloc likewise
    int vn[2][GC_P3_BLK];                           // samples per period
    double vcar[GC_P3_BLK + 1];                     // remcarr at the start of each period of the carrier's block
    int prog_code, prog_car, fail_code, fail_car;
    int Ks2[2][GC_NB + 2];
};
__shared__ __attribute__((aligned(16))) Plan3Shared g_plan3;

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

// a claims row in LDS -> fields
__device__ __forceinline__ GcCodeClaims plan3_code_row(const int *row)
{
    const int4 *r = reinterpret_cast<const int4 *>(row);
    const int4 v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3], v4 = r[4];
    GcCodeClaims c;
    c.tag = v0.x; c.i0 = v0.y; c.q = v0.z; c.nl = v0.w;
    c.jsum = v1.x; c.dm[0] = v1.y; c.dm[1] = v1.z; c.dm[2] = v1.w;
    c.dm[3] = v2.x; c.dm[4] = v2.y; c.dm[5] = v2.z; c.dm[6] = v2.w;
    c.dm[7] = v3.x; c.dm[8] = v3.y; c.dm[9] = v3.z; c.dm[10] = v3.w;
    c.dm[11] = v4.x; c.dm[12] = v4.y; c.pad[0] = 0; c.pad[1] = 0;
    return c;
}
__device__ __forceinline__ GcCarClaims plan3_car_row(const int *row)
{
    const int4 *r = reinterpret_cast<const int4 *>(row);
    const int4 v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3], v4 = r[4];
    GcCarClaims c;
    c.tag = v0.x; c.nl = v0.y; c.i0 = v0.z; c.nseg = v0.w;
    c.kprem = v1.x; c.dm[0] = v1.y; c.dm[1] = v1.z; c.dm[2] = v1.w;
    c.dm[3] = v2.x; c.dm[4] = v2.y; c.dm[5] = v2.z; c.dm[6] = v2.w;
    c.dm[7] = v3.x; c.dm[8] = v3.y; c.dm[9] = v3.z; c.dm[10] = v3.w;
    c.dm[11] = v4.x; c.dm[12] = v4.y; c.pad[0] = 0; c.pad[1] = 0;
    return c;
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
    if (!plan_code_dev(PC.f, c0, clen, n + 2 * smax, g_plan3.Ks2[0], lane, &cend))
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
    if (!plan_carrier_dev(PK.f, phis, n, g_plan3.Ks2[1], lane, &xn)) xn = gc_fast_carrier_walk(PK.f, phis, n, ne);
    *out = gc_fast_prem(PK.fprem, xn);
    return 2;
}

__device__ __forceinline__ int plan3_load(int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void plan3_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

#ifdef GC_PLAN_PROF     // (tools/debug: per channel and wavefront, clocks inside the role's work and inside the whole protocol)
__device__ unsigned long long gc_plan_prof[64 * 16];
#endif
struct Plan3Job {               // what every wavefront of the workgroup knows about the batch
    const int4 *claims_code, *claims_car;           // the channel's rows
    GcTrkPlan *out;
    GcTrkState s;
    int nepoch, nblk, tid;
    bool code_ok;
};

// The block protocol, run by every wavefront of the workgroup in step (the barriers pair up whatever the role).
// Super-round r: the code roles work on block r, the carrier roles on block r - 1 (whose period lengths are final by
// then: the carrier never waits for the code chain inside a block).  Within a super-round each NCO repeats its block
// from the first period whose check failed until none fails.
// work_code(block, start, nb, redo), work_car(block, start, nb, redo): the calling wavefront's share (empty for the other NCO's roles).
template <class WorkCode, class WorkCar>
__device__ __forceinline__ void plan3_protocol(const Plan3Job &J, WorkCode work_code, WorkCar work_car)
{
    constexpr int RW4 = GC_CLAIM_ROW / 4;
    const int tid = J.tid;
    auto nb_of = [&](int b) { const int left = J.nepoch - b * GC_P3_BLK; return left < GC_P3_BLK ? left : GC_P3_BLK; };
    auto stage = [&](int b) {   // the claims of block b: [code | carrier][period][row]
        const int e0 = b * GC_P3_BLK, nb = nb_of(b);
        for (int x = tid; x < 2 * nb * RW4; x += 64 * GC_P3_NW) {
            const int which = x / (nb * RW4), r = x - which * nb * RW4;
            reinterpret_cast<int4 *>(g_plan3.rows[b % 3][which])[r] = (which ? J.claims_car : J.claims_code)[(size_t)e0 * RW4 + r];
        }
    };
#ifdef GC_PLAN_PROF
    unsigned long long pw_ = 0, pt0_ = __builtin_readcyclecounter();
#endif
    if (tid == 0) {
        g_plan3.vcode[0][0] = J.s.remcode;
        g_plan3.vbuff[0][0] = J.s.buffloc;
        g_plan3.vcar[0] = J.s.remcarr;
    }
    stage(0);
    __syncthreads();
    for (int r = 0; r <= J.nblk; r++) {
        if (r + 1 < J.nblk) stage(r + 1);               // (lands while this super-round runs; its closing barrier orders it)
        int cstart = 0, credo = 0, kstart = 0, kredo = 0;
        bool cdone = r >= J.nblk || !J.code_ok, kdone = r < 1 || !J.code_ok;
        const int cnb = r < J.nblk ? nb_of(r) : 0, knb = r >= 1 ? nb_of(r - 1) : 0;
        while (!(cdone && kdone)) {
            if (tid == 0) {
                if (!cdone) { g_plan3.prog_code = cstart; g_plan3.fail_code = GC_P3_NONE; }
                if (!kdone) { g_plan3.prog_car = kstart; g_plan3.fail_car = GC_P3_NONE; }
            }
            __syncthreads();
#ifdef GC_PLAN_PROF
            const unsigned long long pa_ = __builtin_readcyclecounter();
#endif
            if (!cdone) work_code(r, cstart, cnb, credo);
            if (!kdone) work_car(r - 1, kstart, knb, kredo);
#ifdef GC_PLAN_PROF
            pw_ += __builtin_readcyclecounter() - pa_;
#endif
            __syncthreads();
            if (!cdone) {
                const int f = g_plan3.fail_code;
                if (f == GC_P3_NONE) cdone = true; else { cstart = f; credo = 1; }      // (every period before f passed: its start value is exact)
            }
            if (!kdone) {
                const int f = g_plan3.fail_car;
                if (f == GC_P3_NONE) kdone = true; else { kstart = f; kredo = 1; }
            }
            __syncthreads();                            // (the fail words are reset at the top)
        }
        // block r - 1 is complete: its plan entries, one lane per period
        if (r >= 1) {
            const int bp = (r - 1) & 1, e0 = (r - 1) * GC_P3_BLK;
            if (tid < knb) {
                GcTrkPlan &o = J.out[e0 + tid];
                if (J.code_ok) {
                    o.buffloc = g_plan3.vbuff[bp][tid];
                    o.coff = g_plan3.vcode[bp][tid];
                    o.phi0 = g_plan3.vcar[tid];
                    o.n = g_plan3.vn[bp][tid];
                } else {                                // (nothing is stepped: every period starts where the batch did)
                    o.buffloc = J.s.buffloc;
                    o.coff = J.s.remcode;
                    o.phi0 = J.s.remcarr;
                    o.n = 0;
                }
                o.carrfreq = J.s.carrfreq;
                o.codefreq = J.s.codefreq;
                o.pad = 0;
            }
        }
        __syncthreads();
        if (tid == 0 && J.code_ok) {
            if (r < J.nblk) {                           // the next code block starts where this one ended
                g_plan3.vcode[(r + 1) & 1][0] = g_plan3.vcode[r & 1][cnb];
                g_plan3.vbuff[(r + 1) & 1][0] = g_plan3.vbuff[r & 1][cnb];
            }
            if (r >= 1) g_plan3.vcar[0] = g_plan3.vcar[knb];
        }
        __syncthreads();
    }
#ifdef GC_PLAN_PROF
    if ((tid & 63) == 0 && blockIdx.x < 64) {
        atomicAdd(&gc_plan_prof[blockIdx.x * 16 + (tid >> 6) * 2], pw_);
        atomicAdd(&gc_plan_prof[blockIdx.x * 16 + (tid >> 6) * 2 + 1], __builtin_readcyclecounter() - pt0_);
    }
#endif
}

// The code NCO's wavefronts.  role 0 chains the values of periods [start, nb) of its block (period `start` by the
// certified path when `redo`; it stops as soon as a check fails); roles 1..GC_P3_NCHK check periods start + (role - 1),
// + GC_P3_NCHK, ...  (Out of line, one instance per shape of the code step: its constants sit in registers for the batch.)
template <int ITOP, int TMAX>
__device__ __attribute__((noinline)) void plan3_code_wave(int role_, double ci_, double spc_, int clen_, int smax_, const Plan3Job *J_, int lane)
{
    const int role = plan2_uni(role_), clen = plan2_uni(clen_), smax = plan2_uni(smax_);
    const double ci = plan2_uni(ci_), spc = plan2_uni(spc_);
    const Plan3Job J = *plan2_uni(J_);
    const double dlen = (double)clen;
    GcCodePlan PC;
    gc_code_plan_init(PC, ci, clen, smax);
    GcCodeStepC<ITOP> SC;
    gc_code_stepc_init(SC, PC);
    const double yspc = __ddiv_rn(1.0, spc);
    const bool fastdiv = spc > 1e-300 && spc < 1e300 && yspc < 1e300;
    unsigned tally[3] = {0, 0, 0};
    plan3_protocol(J,
        [&](int b, int start, int nb, int redo) {
            const int *rows = g_plan3.rows[b % 3][0];
            double *vcode = g_plan3.vcode[b & 1];
            unsigned long long *vbuff = g_plan3.vbuff[b & 1];
            int *vn = g_plan3.vn[b & 1];
            if (role == 0) {
                double remcode = vcode[start];
                unsigned long long buffloc = vbuff[start];
                GcCodeClaims nx = plan3_code_row(rows + start * GC_CLAIM_ROW);
                for (int e = start; e < nb; e++) {
                    if (((e - start) & 3) == 3 && *(volatile int *)&g_plan3.fail_code != GC_P3_NONE) break;    // a check failed behind us: the block is redone from there
                    const GcCodeClaims cl = nx;
                    if (e + 1 < nb) nx = plan3_code_row(rows + (e + 1) * GC_CLAIM_ROW);     // (in flight during this period's step)
                    const double num = __dsub_rn(dlen, remcode);                    // ref src/sdrtrk.c:31-32
                    double qn = gc_div_y(num, spc, yspc);
                    if (__builtin_expect(!fastdiv, 0)) qn = __ddiv_rn(num, spc);
                    const int n = plan2_uni((qn > -2147483648.0 && qn < 2147483648.0) ? (int)qn : 0);
                    if (lane == 0) vn[e] = n;
                    const bool walk = n > 0 && n <= (1 << 24);
                    if (walk) {
                        double rc;
                        if ((redo && e == start) || cl.tag != 1) {  // claims that failed, or none (discovery declined the period): the certified step, exact whatever they say
                            tally[plan2_code_slow(ci, clen, smax, remcode, n, lane, &rc)]++;
                        } else {
                            GcCodeClaims c2 = cl;
                            (void)gc_code_claims_step<ITOP, TMAX, false>(PC, SC, remcode, n + 2 * smax, c2, &rc);     // the value; the checkers judge
                        }
                        remcode = rc;
                    }
                    buffloc += (unsigned long long)(long long)n;
                    // (value, then progress: LDS stores of one lane, executed in order -- a checker that sees the count sees the value)
                    if (lane == 0) { vcode[e + 1] = remcode; vbuff[e + 1] = buffloc; *(volatile int *)&g_plan3.prog_code = e + 1; }
                }
            } else {
                for (int e = start + role - 1; e < nb; e += GC_P3_NCHK) {
                    bool stop = false;
                    while (plan3_load(&g_plan3.prog_code) <= e) {
                        if (plan3_load(&g_plan3.fail_code) <= e) { stop = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (stop || plan3_load(&g_plan3.fail_code) <= e) break;         // (periods from the failed one on are redone)
                    if (redo && e == start) continue;   // (computed by the certified path)
                    const int n = vn[e];
                    if (!(n > 0 && n <= (1 << 24))) continue;                       // nothing was stepped
                    const double y0 = vcode[e], y1 = vcode[e + 1];
                    GcCodeClaims cl = plan3_code_row(rows + e * GC_CLAIM_ROW);
                    if (cl.tag != 1) continue;          // (no claims: the chain took the certified step)
                    double rc;
                    const bool ok = gc_code_claims_step<ITOP, TMAX, false>(PC, SC, y0, n + 2 * smax, cl, &rc);
                    if (!(ok && rc == y1)) { if (lane == 0) atomicMin(&g_plan3.fail_code, e); }
                    else tally[0]++;
                }
            }
        },
        [](int, int, int, int) {});
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (tally[t]) atomicAdd(&gc_plan_stats[t], (unsigned long long)tally[t]);
    }
}

// The carrier NCO's wavefronts, likewise, one block behind the code's (it needs the period lengths, final by then)
__device__ __attribute__((noinline)) void plan3_car_wave(int role_, double ps_, int nsamp_, const Plan3Job *J_, int lane)
{
    const int role = plan2_uni(role_), nsamp = plan2_uni(nsamp_);
    const double ps = plan2_uni(ps_);
    const Plan3Job J = *plan2_uni(J_);
    GcCarPlan PK;
    gc_car_plan_init(PK, ps, false, false);
    GcCarStepC CK;
    gc_car_stepc_init(CK, PK, nsamp + 16);
    unsigned tally[3] = {0, 0, 0};
    plan3_protocol(J,
        [](int, int, int, int) {},
        [&](int b, int start, int nb, int redo) {
            const int *rows = g_plan3.rows[b % 3][1];
            const int *vn = g_plan3.vn[b & 1];
            double *vcar = g_plan3.vcar;
            if (role == 0) {
                double remcarr = vcar[start];
                GcCarClaims nx = plan3_car_row(rows + start * GC_CLAIM_ROW);
                for (int e = start; e < nb; e++) {
                    if (((e - start) & 3) == 3 && *(volatile int *)&g_plan3.fail_car != GC_P3_NONE) break;
                    const GcCarClaims cl = nx;
                    if (e + 1 < nb) nx = plan3_car_row(rows + (e + 1) * GC_CLAIM_ROW);
                    const int n = plan2_uni(vn[e]);
                    const bool walk = n > 0 && n <= (1 << 24);
                    if (walk) {
                        double rp;
                        if ((redo && e == start) || cl.tag == 0) {
                            tally[plan2_car_slow(ps, remcarr, n, lane, &rp)]++;
                        } else {
                            GcCarClaims c2 = cl;
                            (void)gc_carrier_claims_step<false>(PK, CK, remcarr, n, c2, &rp);
                        }
                        remcarr = rp;
                    }
                    if (lane == 0) { vcar[e + 1] = remcarr; *(volatile int *)&g_plan3.prog_car = e + 1; }
                }
            } else {
                for (int e = start + role - 1; e < nb; e += GC_P3_NCHK) {
                    bool stop = false;
                    while (plan3_load(&g_plan3.prog_car) <= e) {
                        if (plan3_load(&g_plan3.fail_car) <= e) { stop = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (stop || plan3_load(&g_plan3.fail_car) <= e) break;
                    if (redo && e == start) continue;
                    const int n = vn[e];
                    if (!(n > 0 && n <= (1 << 24))) continue;
                    const double x0 = vcar[e], x1 = vcar[e + 1];
                    GcCarClaims cl = plan3_car_row(rows + e * GC_CLAIM_ROW);
                    if (cl.tag == 0) continue;
                    double rp;
                    const bool ok = gc_carrier_claims_step<false>(PK, CK, x0, n, cl, &rp);
                    if (!(ok && rp == x1)) { if (lane == 0) atomicMin(&g_plan3.fail_car, e); }
                    else tally[0]++;
                }
            }
        });
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (tally[t]) atomicAdd(&gc_plan_stats[3 + t], (unsigned long long)tally[t]);
    }
}

template <int ITOP>
__device__ __forceinline__ void plan3_code_dispatch(int tcls, int role, double ci, double spc, int clen, int smax, const Plan3Job *J, int lane)
{
    if (tcls == 0) plan3_code_wave<ITOP, 8>(role, ci, spc, clen, smax, J, lane);
    else if (tcls == 1) plan3_code_wave<ITOP, GC_CLAIM_TAIL>(role, ci, spc, clen, smax, J, lane);
    else plan3_code_wave<ITOP, GC_CLAIM_TAIL2>(role, ci, spc, clen, smax, J, lane);
}

__global__ __launch_bounds__(64 * GC_P3_NW) void trk_plan3_kernel(const GcChan *__restrict__ chan, const GcTrkState *__restrict__ state_in,
                                                                   GcTrkState *__restrict__ state_out, GcTrkPlan *__restrict__ plan,
                                                                   int nch, int nepoch, const int *__restrict__ claims_code,
                                                                   const int *__restrict__ claims_car)
{
    const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ch >= nch) return;
    const int cls = plan2_class(chan[ch].ti, state_in[ch].codefreq, chan[ch].clen, chan[ch].smax);
    if (cls < 0) {                  // no instance of the batch chain for this channel: the chain that certifies its own crossings
        if (wave >= 2) return;
        trk_plan_body(chan, state_in, state_out, plan, nch, nepoch);
        return;
    }
    const GcChan c = chan[ch];
    const GcTrkState s = state_in[ch];
    const double ci = __dmul_rn(c.ti, s.codefreq), spc = __ddiv_rn(s.codefreq, c.f_sf), ps = gc_carrier_ps(s.carrfreq, c.ti);
    const int itop = cls / 3 + 7, tcls = cls % 3;
    constexpr int RW4 = GC_CLAIM_ROW / 4;
    Plan3Job J;
    J.claims_code = reinterpret_cast<const int4 *>(claims_code) + (size_t)ch * nepoch * RW4;
    J.claims_car = reinterpret_cast<const int4 *>(claims_car) + (size_t)ch * nepoch * RW4;
    J.out = plan + (size_t)ch * nepoch;
    J.s = s;
    J.nepoch = nepoch;
    J.nblk = (nepoch + GC_P3_BLK - 1) / GC_P3_BLK;
    J.tid = tid;
    J.code_ok = ci > 0.0 && ci < (double)c.clen;
    // the value chains issue first on their SIMDs (each shares one with a checker)
    if (__builtin_amdgcn_readfirstlane(wave) < 2) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    if (wave == 0 || (wave >= 2 && wave < 2 + GC_P3_NCHK)) {
        const int role = wave == 0 ? 0 : wave - 1;
        switch (itop) {
        case 7:  plan3_code_dispatch<7>(tcls, role, ci, spc, c.clen, c.smax, &J, lane); break;
        case 8:  plan3_code_dispatch<8>(tcls, role, ci, spc, c.clen, c.smax, &J, lane); break;
        case 9:  plan3_code_dispatch<9>(tcls, role, ci, spc, c.clen, c.smax, &J, lane); break;
        case 10: plan3_code_dispatch<10>(tcls, role, ci, spc, c.clen, c.smax, &J, lane); break;
        case 11: plan3_code_dispatch<11>(tcls, role, ci, spc, c.clen, c.smax, &J, lane); break;
        default: plan3_code_dispatch<12>(tcls, role, ci, spc, c.clen, c.smax, &J, lane); break;
        }
    } else {
        plan3_car_wave(wave == 1 ? 0 : wave - 1 - GC_P3_NCHK, ps, c.nsamp, &J, lane);
    }
    // the state the batch leaves behind (the protocol carried the ends of the last blocks into the first slots)
    if (tid == 0) {
        GcTrkState so;
        so.carrfreq = s.carrfreq;
        so.codefreq = s.codefreq;
        so.remcode = J.code_ok ? g_plan3.vcode[J.nblk & 1][0] : s.remcode;
        so.buffloc = J.code_ok ? g_plan3.vbuff[J.nblk & 1][0] : s.buffloc;
        so.remcarr = J.code_ok ? g_plan3.vcar[0] : s.remcarr;
        state_out[ch] = so;
    }
}


}  // namespace


#ifdef GC_PLAN_PROF
extern "C" int gnsscorr_debug_plan_prof(unsigned long long *dst)
{
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(gc_plan_prof), sizeof(unsigned long long) * 64 * 16) != hipSuccess) return -1;
    return 0;
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
    (void)dbg;
    if (batch)
        hipLaunchKernelGGL(trk_plan3_kernel, dim3(nch), dim3(64 * GC_P3_NW), 0, st, chan, state_in, state_out, plan, nch, nepoch,
                           claims, claims + (size_t)nch * nepoch * GC_CLAIM_ROW);
    else
        hipLaunchKernelGGL(trk_plan_kernel, dim3(nch), dim3(128), 0, st, chan, state_in, state_out, plan, nch, nepoch);
    GC_HIP(hipGetLastError());
    return 0;
}

