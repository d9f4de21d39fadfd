// gnsscorr_plan.hip -- the tracking planner: the NCO chain of sdrtracking() from period to period
// (ref src/sdrtrk.c:31-43), bit for bit.  Discovery pass (trk_spec_kernel), the evaluating chain
// (trk_plan4_kernel) and the chain that certifies its own crossings (trk_plan_kernel); DESIGN.md 3.1,
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

// Discovery of the batch planner, in two passes.
//
// The chain of a channel is sequential because every period starts where the one before ended, to the last bit;
// but the STRUCTURE of a period's step (gnsscorr_nco.h: "period steps on claims") is the same for every start in
// an interval around it.  Each operation of the step is a rounded addition, multiplication or fma of the running
// value with constants: non-decreasing in the period's start.  Each check of the step compares such a value with a
// constant (below the top of a binade, at or above it, ...), or compares integers that do not depend on the start
// once the period's sample count is fixed.  Hence: if the complete step passes its checks from `lo` and from `hi`
// with the same claims and the same sample count (and takes the same side of the one branch of the start value,
// ref src/sdrcmn.c:614), it passes them from every start in [lo, hi] -- the values in between are bracketed by the
// two ends operation by operation.  The chain then only has to see that its exact start lies in the bracket and
// evaluate (the values alone: 0.28 us per period for the code, 0.39 for the carrier, against 0.70 / 0.83 with the checks).
// (The one step whose validity depends on a parity -- a whole period inside one binade, tag 2, in which the addend
// may be a tie -- carries its own conditions and is evaluated WITH them by the chain: it costs one fma.)
//
// pass 1 (trk_specdev_kernel, one lane per period): the closed-form period starts (gc_spec_start) ignore the
//         rounding the reference's sums collect, which is systematic: ~5e-10 chips per period, the same sign every
//         period.  From every closed-form start the exact step is taken, and what it lands away from the next
//         closed-form start is that period's deviation.
// pass 2 (trk_spec_kernel): closed form + the deviations summed over the periods before = the start to first
//         order (what is left is the rounding of a rounding: < 1e-12 per period); a bracket of +-2^-30 around it;
//         claims discovered at the lower end, checked at the upper.
// Nothing here is ever used for a result: a bracket that misses the exact start, or claims that fail at an end,
// send the period down the chain's certified path.
#define GC_SPEC_WC  9.313225746154785e-10       // 2^-30 chips
#define GC_SPEC_WK  9.313225746154785e-10       // 2^-30 rad, at least; 2^-36 of the phase beyond that

__device__ __forceinline__ int spec_nsamp(double dlen, double remcode, double spc)      // ref src/sdrtrk.c:31-32
{
    const double qn = __ddiv_rn(__dsub_rn(dlen, remcode), spc);
    return (qn > -2147483648.0 && qn < 2147483648.0) ? (int)qn : 0;
}

struct SpecChan {               // a channel's constants for the discovery passes
    GcCodePlan PC;
    GcCarPlan PK;
    GcCarStepC CK;
    double ci, spc, ps, dlen;
    int clen, smax;
    int tmax;                   // tail positions of the chain's instance for this channel (plan2_class)
    bool ok;
};

__device__ __forceinline__ void spec_chan_init(SpecChan &C, const GcChan &c, const GcTrkState &s, bool with_prem)
{
    C.ci = __dmul_rn(c.ti, s.codefreq);
    C.spc = __ddiv_rn(s.codefreq, c.f_sf);
    C.dlen = (double)c.clen;
    C.clen = c.clen;
    C.smax = c.smax;
    C.ps = gc_carrier_ps(s.carrfreq, c.ti);
    C.tmax = c.smax + 1 > 8 ? (c.smax + 1 > GC_CLAIM_TAIL ? GC_CLAIM_TAIL2 : GC_CLAIM_TAIL) : 8;
    C.ok = C.ci > 0.0 && C.ci < C.dlen && C.spc > 1e-300 && C.spc < 1e300;
    if (!C.ok) return;
    gc_code_plan_init(C.PC, C.ci, c.clen, c.smax);
    gc_car_plan_init(C.PK, C.ps, true, with_prem);
    gc_car_stepc_init(C.CK, C.PK, c.nsamp + 16);
}

// pass 1: devc/devk[ch * T + e] = (exact step from the closed-form start of period e) - (closed-form start of e + 1)
__global__ __launch_bounds__(64) void trk_specdev_kernel(const GcChan *__restrict__ chan, const GcTrkState *__restrict__ state_in,
                                                          double *__restrict__ devc, double *__restrict__ devk, int nch, int T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nch * T) return;
    const int ch = i / T, e = i - ch * T;
    const GcTrkState s = state_in[ch];
    SpecChan C;
    spec_chan_init(C, chan[ch], s, true);
    double dc = 0.0, dk = 0.0;
    if (C.ok) {
        double r0, g0, r1, g1;
        int nhat;
        gc_spec_start(s.remcode, s.remcarr, C.ci, C.spc, C.ps, C.dlen, e, &r0, &g0, &nhat);
        gc_spec_start(s.remcode, s.remcarr, C.ci, C.spc, C.ps, C.dlen, e + 1, &r1, &g1, &nhat);
        const int n = spec_nsamp(C.dlen, r0, C.spc);
        if (n > 0 && n <= (1 << 24)) {
            GcFillLoop fill;
            GcNoEmit ne;
            const int nt = n + 2 * C.smax;
            GcCodeClaims cc;
            double F = r1;
            if (!gc_code_claims<true>(C.PC, r0, nt, cc, &F) && !gc_code_period(C.PC, r0, nt, fill, &F)) {
                const double c0 = gc_code_start_fast(r0, C.PC.smaxci, C.clen);
                F = __dsub_rn(gc_fast_code_walk(C.PC.f, c0, C.clen, nt, ne), C.PC.smaxci);
            }
            dc = F - r1;
            GcCarClaims ck;
            double G = g1;
            if (!gc_carrier_claims_step<true>(C.PK, C.CK, g0, n, ck, &G) && !gc_carrier_period(C.PK, g0, n, fill, &G)) {
                const double phis = gc_div_y(__dmul_rn(g0, GC_NCO_CDIV), GC_NCO_DPI, C.PK.ydpi);
                G = gc_fast_prem(C.PK.fprem, gc_fast_carrier_walk(C.PK.f, phis, n, ne));
            }
            dk = G - g1;
            // (the closed form and the step may sit on different sides of a whole turn)
            dk = dk > 0.5 * GC_NCO_DPI ? dk - GC_NCO_DPI : (dk < -0.5 * GC_NCO_DPI ? dk + GC_NCO_DPI : dk);
            if (!(fabs(dc) < 1.0e300)) dc = 0.0;
            if (!(fabs(dk) < 1.0e300)) dk = 0.0;
        }
    }
    devc[i] = dc;
    devk[i] = dk;
}

// pass 2: one workgroup per channel and GC_SPEC_CHUNK periods of the batch; T = e_off + nepoch periods lie between
// the state handed in and the end of the batch
#define GC_SPEC_CHUNK 256
__global__ __launch_bounds__(GC_SPEC_CHUNK) void trk_spec_kernel(const GcChan *__restrict__ chan, const GcTrkState *__restrict__ state_in,
                                                                  const double *__restrict__ devc, const double *__restrict__ devk,
                                                                  int *__restrict__ claims_code, int *__restrict__ claims_car,
                                                                  int nch, int nepoch, int e_off)
{
    // e_off: periods between the state handed in and the batch's first period (0: the batch starts at that
    // state; nepoch: the state is the start of the batch BEFORE this one, whose chain is still running -- the
    // closed forms and the deviations reach over it just as well)
    __shared__ double shc[GC_SPEC_CHUNK], shk[GC_SPEC_CHUNK];
    const int nchunk = (nepoch + GC_SPEC_CHUNK - 1) / GC_SPEC_CHUNK;
    const int ch = blockIdx.x / nchunk, chunk = blockIdx.x - ch * nchunk, tid = threadIdx.x;
    if (ch >= nch) return;
    const int T = e_off + nepoch;
    const int eb = chunk * GC_SPEC_CHUNK + tid;         // period of the batch
    const int e = e_off + eb;                           // ... counted from the state
    const int base = e_off + chunk * GC_SPEC_CHUNK;
    // deviations of the periods before the chunk, then those before each of its periods
    double pc = 0.0, pk = 0.0;
    for (int j = tid; j < base; j += GC_SPEC_CHUNK) {
        pc += devc[(size_t)ch * T + j];
        pk += devk[(size_t)ch * T + j];
    }
    shc[tid] = pc;
    shk[tid] = pk;
    __syncthreads();
    for (int st = GC_SPEC_CHUNK / 2; st > 0; st >>= 1) {
        if (tid < st) { shc[tid] += shc[tid + st]; shk[tid] += shk[tid + st]; }
        __syncthreads();
    }
    const double basec = shc[0], basek = shk[0];
    __syncthreads();
    shc[tid] = e < T ? devc[(size_t)ch * T + e] : 0.0;
    shk[tid] = e < T ? devk[(size_t)ch * T + e] : 0.0;
    __syncthreads();
    for (int st = 1; st < GC_SPEC_CHUNK; st <<= 1) {    // inclusive scan
        const double ac = tid >= st ? shc[tid - st] : 0.0, ak = tid >= st ? shk[tid - st] : 0.0;
        __syncthreads();
        shc[tid] += ac;
        shk[tid] += ak;
        __syncthreads();
    }
    if (eb >= nepoch) return;
    const double sumc = basec + (tid ? shc[tid - 1] : 0.0), sumk = basek + (tid ? shk[tid - 1] : 0.0);
    const GcTrkState s = state_in[ch];
    SpecChan C;
    spec_chan_init(C, chan[ch], s, false);
    GcCodeClaims cc;
    GcCarClaims ck;
    cc.tag = 0;
    ck.tag = 0;
    cc.n = cc.pad = 0;
    cc.lo = cc.hi = ck.lo = ck.hi = 0.0;
    if (C.ok) {
        double r0, g0;
        int nhat;
        gc_spec_start(s.remcode, s.remcarr, C.ci, C.spc, C.ps, C.dlen, e, &r0, &g0, &nhat);
        const double rt = r0 + sumc, gt = g0 + sumk;
        // code: a bracket of +-2^-30, else (a comparison of the step falls inside it: one period in a million -- or every
        // period of a channel whose sums hit their thresholds exactly, e.g. fresh out of acquisition with a chip step of
        // 1/16) one of +-2^-38, else the claims of the estimate itself, for the chain to evaluate WITH the checks (tag 3)
        int ncode = 0;
        bool bracketed = false;
#pragma unroll 1
        for (int att = 0; att < 2 && !bracketed; att++) {
            const double w = att == 0 ? GC_SPEC_WC : GC_SPEC_WC * 0.00390625;
            const double lo = rt - w, hi = rt + w;
            const int nlo = spec_nsamp(C.dlen, lo, C.spc), nhi = spec_nsamp(C.dlen, hi, C.spc);
            if (!(nlo == nhi && nlo > 0 && nlo <= (1 << 24))) continue;
            double dummy;
            const bool side = (lo - C.PC.smaxci < 0.0) == (hi - C.PC.smaxci < 0.0);     // (ref src/sdrcmn.c:614: one branch for the whole bracket)
            const bool oklo = gc_code_claims<true>(C.PC, lo, nlo + 2 * C.smax, cc, &dummy);
            const bool okhi = oklo && gc_code_claims<false>(C.PC, hi, nlo + 2 * C.smax, cc, &dummy);
            // (the discovering step allows the widest tail; the chain's instance for this channel may be narrower, and
            // in its value form nothing would notice a tail it cannot hold)
            const bool tailfits = nlo + 2 * C.smax - cc.jsum <= C.tmax;
            if (side && oklo && okhi && tailfits) {
                bracketed = true;
                ncode = nlo;
                cc.tag = 1;
                cc.n = nlo;
                cc.lo = lo;
                cc.hi = hi;
            }
        }
        if (!bracketed) {
            ncode = spec_nsamp(C.dlen, rt, C.spc);
            cc.tag = 0;
            if (ncode > 0 && ncode <= (1 << 24)) {
                double dummy;
                cc.tag = (gc_code_claims<true>(C.PC, rt, ncode + 2 * C.smax, cc, &dummy) && ncode + 2 * C.smax - cc.jsum <= C.tmax) ? 3 : 0;
            }
            cc.n = ncode;
            cc.lo = cc.hi = rt;
        }
        // carrier, likewise, for the period length the code side settled on
        const int nk = ncode;
        if (nk > 0 && nk <= (1 << 24)) {
            bool kbr = false;
            int tag = 0;
#pragma unroll 1
            for (int att = 0; att < 2 && !kbr; att++) {
                const double w = fmax(GC_SPEC_WK, fabs(gt) * 1.4551915228366852e-11) * (att == 0 ? 1.0 : 0.00390625);    // 2^-30, 2^-36 of the phase; / 256
                const double klo = gt - w, khi = gt + w;
                double dummy;
                bool ok = gc_carrier_claims_step<true>(C.PK, C.CK, klo, nk, ck, &dummy);
                tag = ck.tag;
                ok = ok && gc_carrier_claims_step<false, true>(C.PK, C.CK, khi, nk, ck, &dummy);
                if (ok) {
                    kbr = true;
                    ck.tag = tag;
                    ck.lo = klo;
                    ck.hi = khi;
                }
            }
            if (!kbr) {
                double dummy;
                const bool ok = gc_carrier_claims_step<true>(C.PK, C.CK, gt, nk, ck, &dummy);
                ck.tag = ok ? (ck.tag == 2 ? 2 : 3) : 0;
                ck.lo = ck.hi = gt;
            }
            ck.nl = nk;
        }
    }
    const size_t row = (size_t)ch * nepoch + eb;
    int4 *rc = reinterpret_cast<int4 *>(claims_code) + row * (GC_CLAIM_ROW / 4);
    int4 *rk = reinterpret_cast<int4 *>(claims_car) + row * (GC_CLAIM_ROW / 4);
    const int4 *pc4 = reinterpret_cast<const int4 *>(&cc), *pk4 = reinterpret_cast<const int4 *>(&ck);
#pragma unroll
    for (int q = 0; q < GC_CLAIM_ROW / 4; q++) {
        rc[q] = pc4[q];
        rk[q] = pk4[q];
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
// Two wavefronts per channel -- the code NCO's chain and, behind it, the carrier NCO's, which needs the periods'
// sample counts and nothing else from the code side.  Per period a chain sees that its exact start lies in the
// bracket the discovery proved the claims for (two compares; the sample count matches as well) and evaluates the
// step: the VALUES alone, ~60 dependent fp64 operations from one period's start to the next -- the step function's
// verdict is unused, so the compiler drops the checks (tools/ubench/claims_chain.hip: 0.28 us per period for the
// code, 0.39 for the carrier, on a lone wavefront; with the checks 0.70 / 0.83).  A start outside its bracket, or a
// period without claims, takes the certified step, then the walkers: exact whatever the discovery said.
// (Rounds 2-3 checked every step in the chain -- 0.94 us per period -- or on checker wavefronts beside it -- the
// compute unit's four SIMDs then carried 5300 clocks of fp64 work per period: 0.82 us.)
// GNSSCORR_PLAN_VERIFY=1 (tests): the chain evaluates WITH the checks; a bracketed start whose step fails them is
// counted in gc_plan_stats[6] (and redone by the certified path) -- the proof above says the count stays zero.
#define GC_P4_BLK 64            // periods whose plan entries are written at a time
struct Plan4Shared {
    int nsh[GC_PLAN_MAXE];                          // samples per period, code chain -> carrier chain
    double vstart[2][GC_P4_BLK];                    // the block's period starts (remcode | remcarr), for the plan entries
    unsigned long long vbuff[GC_P4_BLK];
    int prog;                                       // periods the code chain has finished
    GcCarPlan pkfull;                               // the carrier's tables as the certified step wants them, built once per batch
    int Ks2[2][GC_NB + 2];
};
__shared__ __attribute__((aligned(16))) Plan4Shared g_plan4;

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

// A claims row -> fields.  The rows were written by the discovery kernels before this one started and the address
// is the same in every lane: read through the constant address space they arrive by scalar loads, in scalar
// registers -- the chain's vector unit, which is what it is short of, never touches them.
#define GC_CONST __attribute__((address_space(4)))
typedef int gc_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ GcCodeClaims plan4_code_row(const GC_CONST gc_v4i *r)
{
    const gc_v4i v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3], v4 = r[4], v5 = r[5];
    GcCodeClaims c;
    c.tag = v0.x; c.i0 = v0.y; c.q = v0.z; c.nl = v0.w;
    c.jsum = v1.x; c.dm[0] = v1.y; c.dm[1] = v1.z; c.dm[2] = v1.w;
    c.dm[3] = v2.x; c.dm[4] = v2.y; c.dm[5] = v2.z; c.dm[6] = v2.w;
    c.dm[7] = v3.x; c.dm[8] = v3.y; c.dm[9] = v3.z; c.dm[10] = v3.w;
    c.dm[11] = v4.x; c.dm[12] = v4.y; c.n = v4.z; c.pad = 0;
    c.lo = gc_u2d(((uint64_t)(unsigned)v5.y << 32) | (unsigned)v5.x);
    c.hi = gc_u2d(((uint64_t)(unsigned)v5.w << 32) | (unsigned)v5.z);
    return c;
}
__device__ __forceinline__ GcCarClaims plan4_car_row(const GC_CONST gc_v4i *r)
{
    const gc_v4i v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3], v5 = r[5];
    GcCarClaims c;
    c.tag = v0.x; c.nl = v0.y; c.i0 = v0.z; c.nseg = v0.w;
    c.kprem = v1.x; c.dm[0] = v1.y; c.dm[1] = v1.z; c.dm[2] = v1.w;
    c.dm[3] = v2.x; c.dm[4] = v2.y; c.dm[5] = v2.z; c.dm[6] = v2.w;
    c.dm[7] = v3.x; c.dm[8] = v3.y; c.dm[9] = v3.z; c.dm[10] = v3.w;
    c.dm[11] = 0; c.dm[12] = 0; c.pad[0] = 0; c.pad[1] = 0;      // (the window has GC_CLAIM_CWIN positions: the rest is never read)
    c.lo = gc_u2d(((uint64_t)(unsigned)v5.y << 32) | (unsigned)v5.x);
    c.hi = gc_u2d(((uint64_t)(unsigned)v5.w << 32) | (unsigned)v5.z);
    return c;
}
static_assert(GC_CLAIM_CWIN <= 11, "plan4_car_row reads dm[0..10]");
static_assert(sizeof(GcCodeClaims) == GC_CLAIM_ROW * 4 && sizeof(GcCarClaims) == GC_CLAIM_ROW * 4, "claims rows are GC_CLAIM_ROW ints");
static_assert(offsetof(GcCodeClaims, dm) == 20 && offsetof(GcCarClaims, dm) == 20, "claims layout");
static_assert(offsetof(GcCodeClaims, n) == 72 && offsetof(GcCodeClaims, lo) == 80 && offsetof(GcCarClaims, lo) == 80, "claims layout");
static_assert(GC_CLAIM_ROW == 24, "plan4_*_row read six int4 per row");

// A period without a bracket around its start (a few in a million): the certified step, then the walkers, with the
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
    if (!plan_code_dev(PC.f, c0, clen, n + 2 * smax, g_plan4.Ks2[0], lane, &cend))
        cend = gc_fast_code_walk(PC.f, c0, clen, n + 2 * smax, ne);
    *out = __dsub_rn(cend, smaxci);
    return 2;
}

__device__ __attribute__((noinline)) int plan2_car_slow(double ps_, double remcarr_, int n_, int lane, double *out)
{
    const double remcarr = plan2_uni(remcarr_);
    const int n = plan2_uni(n_);
    (void)ps_;
    const GcCarPlan &PK = g_plan4.pkfull;           // (plan4_car_wave built it for this channel's ps)
    GcFillLanes fill{lane};
    GcNoEmit ne;
    double rp;
    if (gc_carrier_period(PK, remcarr, n, fill, &rp)) { *out = rp; return 1; }
    const double phis = gc_div_y(__dmul_rn(remcarr, GC_NCO_CDIV), GC_NCO_DPI, __ddiv_rn(1.0, GC_NCO_DPI));     // ref src/sdrcmn.c:649
    double xn;
    if (!plan_carrier_dev(PK.f, phis, n, g_plan4.Ks2[1], lane, &xn)) xn = gc_fast_carrier_walk(PK.f, phis, n, ne);
    *out = gc_fast_prem(PK.fprem, xn);
    return 2;
}

#ifdef GC_PLAN_PROF     // (tools/debug) per channel and chain: clocks in the loop, in the slow path, waiting for rows, waiting for n; slow periods
__device__ unsigned long long gc_plan_prof[64 * 16];
#define GC_PP_DECL unsigned long long pp_[7] = {0, 0, 0, 0, 0, 0, 0}
#define GC_PP_T0(v) const unsigned long long v = __builtin_readcyclecounter()
#define GC_PP_ADD(k, v) pp_[k] += __builtin_readcyclecounter() - (v)
#define GC_PP_INC(k) pp_[k] += 1
#define GC_PP_OUT(which) do { if (lane == 0 && blockIdx.x < 64) for (int k_ = 0; k_ < 7; k_++) atomicAdd(&gc_plan_prof[blockIdx.x * 16 + (which) * 8 + k_], pp_[k_]); } while (0)
#else
#define GC_PP_DECL do { } while (0)
#define GC_PP_T0(v) do { } while (0)
#define GC_PP_ADD(k, v) do { } while (0)
#define GC_PP_INC(k) do { } while (0)
#define GC_PP_OUT(which) do { } while (0)
#endif
struct Plan4Job {               // what both wavefronts know about the batch
    const int4 *claims_code, *claims_car;           // the channel's rows
    GcTrkPlan *out;
    GcTrkState *state_out;
    GcTrkState s;
    int nepoch, nblk, verify;
};

// (a job read back from the caller's frame arrives through vector loads: what is the same in every lane is said so)
__device__ __forceinline__ Plan4Job plan4_job(const Plan4Job *J_)
{
    Plan4Job J = *plan2_uni(J_);
    J.claims_code = plan2_uni(J.claims_code);
    J.claims_car = plan2_uni(J.claims_car);
    J.out = plan2_uni(J.out);
    J.state_out = plan2_uni(J.state_out);
    J.nepoch = plan2_uni(J.nepoch);
    J.nblk = plan2_uni(J.nblk);
    J.verify = plan2_uni(J.verify);
    return J;
}

__device__ __forceinline__ int plan4_nb(const Plan4Job &J, int b)
{
    const int left = J.nepoch - b * GC_P4_BLK;
    return left < GC_P4_BLK ? left : GC_P4_BLK;
}

#define GC_GLOBAL __attribute__((address_space(1)))
// progress of the code chain: LDS operations of one wavefront execute in order, so the count stored after a period's
// sample count is seen after it (the barriers are for the compiler)
__device__ __forceinline__ void plan4_publish(int v)
{
    asm volatile("" ::: "memory");
    __hip_atomic_store(&g_plan4.prog, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int plan4_progress()
{
    const int v = __hip_atomic_load(&g_plan4.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}
// What the chain's loop does not carry: the step WITH its checks for a period whose claims have no bracket around this
// start (tag 3; a start outside its bracket; verify mode), then the certified step, then the walkers.
// returns 0: the claims held (checked), 1: certified step, 2: walkers
template <int ITOP, int TMAX>
__device__ __attribute__((noinline)) int plan4_code_other(double ci_, int clen_, int smax_, double remcode_, int n_, int lane, const GcCodeClaims *cl_, double *out)
{
    const GcCodeClaims *clp = plan2_uni(cl_);       // (the row where the discovery wrote it)
    if (plan2_uni(clp->tag) == 1 || plan2_uni(clp->tag) == 3) {
        const double ci = plan2_uni(ci_), remcode = plan2_uni(remcode_);
        const int clen = plan2_uni(clen_), smax = plan2_uni(smax_), n = plan2_uni(n_);
        GcCodePlan PC;
        gc_code_plan_init(PC, ci, clen, smax, false);
        GcCodeStepC<ITOP> SC;
        gc_code_stepc_init(SC, PC);
        GcCodeClaims c2 = *clp;
        c2.tag = 1;
        double rc;
        if (gc_code_claims_step<ITOP, TMAX, false, true>(PC, SC, remcode, n + 2 * smax, c2, &rc)) { *out = rc; return 0; }
    }
    return plan2_code_slow(ci_, clen_, smax_, remcode_, n_, lane, out);
}

// a scalar that outlives the registers it came in (a real copy: the row's registers are about to be loaded again)
__device__ __forceinline__ int plan4_scopy(int x)
{
    int c;
    asm volatile("s_mov_b32 %0, %1" : "=s"(c) : "s"(x));
    return c;
}
__device__ __forceinline__ bool plan4_bcopy(bool x) { return __builtin_amdgcn_readfirstlane(x ? 1 : 0) != 0; }

// The code NCO's chain (one instance per shape of the code step: its constants sit in registers for the batch)
template <int ITOP, int TMAX>
__device__ __attribute__((noinline)) void plan4_code_wave(double ci_, double spc_, int clen_, int smax_, const Plan4Job *J_, int lane)
{
    const int clen = plan2_uni(clen_), smax = plan2_uni(smax_);
    const double ci = plan2_uni(ci_), spc = plan2_uni(spc_);
    const Plan4Job J = plan4_job(J_);
    const double dlen = (double)clen;
    GcCodePlan PC;
    gc_code_plan_init(PC, ci, clen, smax);
    GcCodeStepC<ITOP> SC;
    gc_code_stepc_init(SC, PC);
    const double yspc = __ddiv_rn(1.0, spc);
    const bool fastdiv = spc > 1e-300 && spc < 1e300 && yspc < 1e300;
    unsigned tally0 = 0, tally1 = 0, tally2 = 0, miss = 0, mism = 0;    // (scalars: an array indexed by a variable lives in memory)
    GC_GLOBAL GcTrkPlan *out = (GC_GLOBAL GcTrkPlan *)J.out;
    double remcode = J.s.remcode;
    unsigned long long buffloc = J.s.buffloc;
    GC_PP_DECL;
    GC_PP_T0(ptot_);
    for (int b = 0; b < J.nblk; b++) {
        const int nb = plan4_nb(J, b), e0 = b * GC_P4_BLK;
        const GC_CONST gc_v4i *rows = (const GC_CONST gc_v4i *)J.claims_code + (size_t)e0 * (GC_CLAIM_ROW / 4);
        // One set of scalar registers holds the row: a period first turns what it needs of it into vector values and
        // scalar copies (the chain pays ~4 clocks for every instruction it issues, whatever the instruction: no copy
        // of the whole row from "next" to "this"), then asks for the next row into the same registers, and evaluates
        // while that is in flight.
        GcCodeClaims row = plan4_code_row(rows);
        for (int i = 0; i < nb; i++) {
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): the row has landed
            const double num = __dsub_rn(dlen, remcode);                            // ref src/sdrtrk.c:31-32
            double qn = gc_div_y(num, spc, yspc);
            if (__builtin_expect(!fastdiv, 0)) qn = __ddiv_rn(num, spc);
            const int n = plan2_uni((qn > -2147483648.0 && qn < 2147483648.0) ? (int)qn : 0);
            const int tag = row.tag;
            const bool inside = tag == 1 && remcode >= row.lo && remcode <= row.hi && n == row.n;
            GcCodeClaims c2;
            double dmd[ITOP + 1];
            c2.tag = 1;
            c2.i0 = 0;
            c2.q = plan4_scopy(row.q);
            c2.nl = plan4_scopy(row.nl);
            c2.jsum = plan4_scopy(row.jsum);
#pragma unroll
            for (int k = 0; k <= ITOP; k++) {
                dmd[k] = (double)row.dm[k];
                GC_PIN_V(dmd[k]);
            }
#pragma unroll
            for (int k = 0; k < 13; k++) c2.dm[k] = 0;
            const bool inside_u = plan4_bcopy(inside);
            const bool claims_u = plan4_bcopy(tag == 1);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nb) row = plan4_code_row(rows + (i + 1) * (GC_CLAIM_ROW / 4));      // (in flight during this period's step)
            __builtin_amdgcn_sched_barrier(0);
            if (lane == 0) {
                g_plan4.vstart[0][i] = remcode;
                g_plan4.vbuff[i] = buffloc;
                g_plan4.nsh[e0 + i] = n;
                plan4_publish(e0 + i + 1);
            }
            if (n > 0 && n <= (1 << 24)) {
                double rc = remcode;
                if (inside_u && !J.verify) {
                    (void)gc_code_claims_step<ITOP, TMAX, false>(PC, SC, remcode, n + 2 * smax, c2, &rc, dmd);     // the value; the bracket is the proof
                    tally0++;
                } else {                        // (its own result variable: what a called function gets the address of lives in memory)
                    double rs;
                    GC_PP_T0(pslow_);
                    const int how = plan4_code_other<ITOP, TMAX>(ci, clen, smax, remcode, n, lane,
                                                                 reinterpret_cast<const GcCodeClaims *>(J.claims_code) + e0 + i, &rs);
                    GC_PP_ADD(1, pslow_);
                    GC_PP_INC(4);
                    tally0 += how == 0 ? 1 : 0;
                    tally1 += how == 1 ? 1 : 0;
                    tally2 += how == 2 ? 1 : 0;
                    mism += (inside_u && how != 0) ? 1 : 0;
                    miss += (!inside_u && claims_u) ? 1 : 0;
                    rc = rs;
                }
                remcode = rc;
            }
            buffloc += (unsigned long long)(long long)n;
        }
        // the block's plan entries, one lane per period (the carrier chain adds phi0)
        if (lane < nb) {
            GC_GLOBAL GcTrkPlan *o = out + e0 + lane;
            o->buffloc = g_plan4.vbuff[lane];
            o->coff = g_plan4.vstart[0][lane];
            o->carrfreq = J.s.carrfreq;
            o->codefreq = J.s.codefreq;
            o->n = g_plan4.nsh[e0 + lane];
            o->pad = 0;
        }
    }
    GC_PP_ADD(0, ptot_);
    GC_PP_OUT(0);
    if (lane == 0) {
        GC_GLOBAL GcTrkState *so = (GC_GLOBAL GcTrkState *)J.state_out;
        so->carrfreq = J.s.carrfreq;
        so->codefreq = J.s.codefreq;
        so->remcode = remcode;
        so->buffloc = buffloc;
        if (tally0) atomicAdd(&gc_plan_stats[0], (unsigned long long)tally0);
        if (tally1) atomicAdd(&gc_plan_stats[1], (unsigned long long)tally1);
        if (tally2) atomicAdd(&gc_plan_stats[2], (unsigned long long)tally2);
        if (mism) atomicAdd(&gc_plan_stats[6], (unsigned long long)mism);
        if (miss) atomicAdd(&gc_plan_stats[7], (unsigned long long)miss);
    }
}

__device__ __attribute__((noinline)) int plan4_car_other(double ps_, int nsamp_, double remcarr_, int n_, int lane, const GcCarClaims *cl_, double *out)
{
    const GcCarClaims *clp = plan2_uni(cl_);
    const int tag = plan2_uni(clp->tag);
    if (tag != 0) {
        const double ps = plan2_uni(ps_), remcarr = plan2_uni(remcarr_);
        const int n = plan2_uni(n_), nsamp = plan2_uni(nsamp_);
        GcCarPlan PK;
        gc_car_plan_init(PK, ps, false, false);
        GcCarStepC CK;
        gc_car_stepc_init(CK, PK, nsamp + 16);
        GcCarClaims c2 = *clp;
        c2.tag = tag == 2 ? 2 : 1;
        double rp;
        if (gc_carrier_claims_step<false, true>(PK, CK, remcarr, n, c2, &rp)) { *out = rp; return 0; }
    }
    return plan2_car_slow(ps_, remcarr_, n_, lane, out);
}

// The carrier chain of a channel whose periods the discovery has no claims for (a real front end at 4 MHz IF: the
// phase climbs thirteen binades per period from below the table): the certified step for every period, its tables
// in registers for the batch -- what the chain of round 2 did for such channels (3.4 us per period), instead of the
// off-path call per period.
__device__ __attribute__((noinline)) void plan4_car_cert_wave(double ps_, const Plan4Job *J_, int lane)
{
    const double ps = plan2_uni(ps_);
    const Plan4Job J = plan4_job(J_);
    GcCarPlan PK;
    gc_car_plan_init(PK, ps);
    GcFillLanes fill{lane};
    GcNoEmit ne;
    const double ydpi = __ddiv_rn(1.0, GC_NCO_DPI);
    unsigned tally1 = 0, tally2 = 0;
    GC_GLOBAL GcTrkPlan *out = (GC_GLOBAL GcTrkPlan *)J.out;
    double remcarr = J.s.remcarr;
    int seen = 0;
    for (int b = 0; b < J.nblk; b++) {
        const int nb = plan4_nb(J, b), e0 = b * GC_P4_BLK;
        for (int i = 0; i < nb; i++) {
            while (seen <= e0 + i) {
                seen = plan2_uni(plan4_progress());
                if (seen <= e0 + i) __builtin_amdgcn_s_sleep(1);
            }
            const int n = plan2_uni(g_plan4.nsh[e0 + i]);
            if (lane == 0) g_plan4.vstart[1][i] = remcarr;
            if (n > 0 && n <= (1 << 24)) {
                double rp;
                if (gc_carrier_period(PK, remcarr, n, fill, &rp)) {
                    tally1++;
                } else {
                    const double phis = gc_div_y(__dmul_rn(remcarr, GC_NCO_CDIV), GC_NCO_DPI, ydpi);      // ref src/sdrcmn.c:649
                    double xn;
                    if (!plan_carrier_dev(PK.f, phis, n, g_plan4.Ks2[1], lane, &xn)) xn = gc_fast_carrier_walk(PK.f, phis, n, ne);
                    rp = gc_fast_prem(PK.fprem, xn);
                    tally2++;
                }
                remcarr = rp;
            }
        }
        if (lane < nb) out[e0 + lane].phi0 = g_plan4.vstart[1][lane];
    }
    if (lane == 0) {
        ((GC_GLOBAL GcTrkState *)J.state_out)->remcarr = remcarr;
        if (tally1) atomicAdd(&gc_plan_stats[4], (unsigned long long)tally1);
        if (tally2) atomicAdd(&gc_plan_stats[5], (unsigned long long)tally2);
    }
}

// The carrier NCO's chain, behind the code's
__device__ __attribute__((noinline)) void plan4_car_wave(double ps_, int nsamp_, const Plan4Job *J_, int lane)
{
    const int nsamp = plan2_uni(nsamp_);
    const double ps = plan2_uni(ps_);
    const Plan4Job J = plan4_job(J_);
    {
        // a channel the discovery has no carrier claims for at all (its first periods say so): the certifying chain
        const int nlook = J.nepoch < 16 ? J.nepoch : 16;
        int none = 0;
        for (int e = 0; e < nlook; e++)
            none += ((const GC_CONST int *)J.claims_car)[(size_t)e * GC_CLAIM_ROW] == 0 ? 1 : 0;
        if (none == nlook && nlook >= 8) {
            plan4_car_cert_wave(ps, J_, lane);
            return;
        }
    }
    {
        GcCarPlan full;
        gc_car_plan_init(full, ps);
        if (lane == 0) g_plan4.pkfull = full;       // (this wavefront's own later reads: LDS operations of a wavefront execute in order)
    }
    GcCarPlan PK;
    gc_car_plan_init(PK, ps, false, false);
    GcCarStepC CK;
    gc_car_stepc_init(CK, PK, nsamp + 16);
    unsigned tally0 = 0, tally1 = 0, tally2 = 0, miss = 0, mism = 0;    // (scalars: an array indexed by a variable lives in memory)
    GC_GLOBAL GcTrkPlan *out = (GC_GLOBAL GcTrkPlan *)J.out;
    double remcarr = J.s.remcarr;
    int seen = 0;
    GC_PP_DECL;
    GC_PP_T0(ptot_);
    for (int b = 0; b < J.nblk; b++) {
        const int nb = plan4_nb(J, b), e0 = b * GC_P4_BLK;
        const GC_CONST gc_v4i *rows = (const GC_CONST gc_v4i *)J.claims_car + (size_t)e0 * (GC_CLAIM_ROW / 4);
        GcCarClaims row = plan4_car_row(rows);
        for (int i = 0; i < nb; i++) {
            GC_PP_T0(ptop_);
            if (seen <= e0 + i) {
                GC_PP_T0(pn_);
                while (seen <= e0 + i) {
                    seen = plan2_uni(plan4_progress());
                    if (seen <= e0 + i) __builtin_amdgcn_s_sleep(1);
                }
                GC_PP_ADD(3, pn_);
            }
            const int n = plan2_uni(g_plan4.nsh[e0 + i]);
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): n and the row are here
            const int tag = row.tag;
            const bool inside = tag == 1 && remcarr >= row.lo && remcarr <= row.hi && n == row.nl;
            GcCarClaims c2;
            double dmd[GC_CLAIM_CWIN];
            c2.tag = tag == 2 ? 2 : 1;
            c2.nl = n;
            c2.i0 = plan4_scopy(row.i0);
            c2.nseg = plan4_scopy(row.nseg);
            c2.kprem = plan4_scopy(row.kprem);
#pragma unroll
            for (int k = 0; k < GC_CLAIM_CWIN; k++) {
                dmd[k] = (double)row.dm[k];
                GC_PIN_V(dmd[k]);
            }
#pragma unroll
            for (int k = 0; k < GC_CLAIM_CSEG; k++) c2.dm[k] = 0;
            const bool inside_u = plan4_bcopy(inside);
            const int tag_u = plan4_scopy(tag);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nb) row = plan4_car_row(rows + (i + 1) * (GC_CLAIM_ROW / 4));       // (in flight during this period's step)
            __builtin_amdgcn_sched_barrier(0);
            if (lane == 0) g_plan4.vstart[1][i] = remcarr;
            GC_PP_ADD(5, ptop_);
            GC_PP_T0(pstep_);
            if (n > 0 && n <= (1 << 24)) {
                double rp = remcarr;
                bool done = false;
                if (inside_u && !J.verify) {
                    (void)gc_carrier_claims_step<false, false, 1>(PK, CK, remcarr, n, c2, &rp, dmd);
                    done = true;
                } else if (tag_u == 2) {
                    // a period inside one binade: the step carries its own conditions (gc_one_binade_walk) and is judged by
                    // them; the remainder's one claim is checked
                    double r2 = remcarr;
                    done = gc_carrier_claims_step<false, false, 2>(PK, CK, remcarr, n, c2, &r2);
                    rp = done ? r2 : rp;
                }
                if (done) {
                    tally0++;
                } else {
                    // claims without a bracket around this start, none at all, or verify mode
                    double rs;
                    GC_PP_T0(pslow_);
                    const int how = plan4_car_other(ps, nsamp, remcarr, n, lane, reinterpret_cast<const GcCarClaims *>(J.claims_car) + e0 + i, &rs);
                    GC_PP_ADD(1, pslow_);
                    GC_PP_INC(4);
                    tally0 += how == 0 ? 1 : 0;
                    tally1 += how == 1 ? 1 : 0;
                    tally2 += how == 2 ? 1 : 0;
                    mism += (inside_u && how != 0) ? 1 : 0;
                    miss += (!inside_u && tag_u == 1) ? 1 : 0;
                    rp = rs;
                }
                remcarr = rp;
            }
            GC_PP_ADD(6, pstep_);
        }
        if (lane < nb) out[e0 + lane].phi0 = g_plan4.vstart[1][lane];
    }
    GC_PP_ADD(0, ptot_);
    GC_PP_OUT(1);
    if (lane == 0) {
        ((GC_GLOBAL GcTrkState *)J.state_out)->remcarr = remcarr;
        if (tally0) atomicAdd(&gc_plan_stats[3], (unsigned long long)tally0);
        if (tally1) atomicAdd(&gc_plan_stats[4], (unsigned long long)tally1);
        if (tally2) atomicAdd(&gc_plan_stats[5], (unsigned long long)tally2);
        if (mism) atomicAdd(&gc_plan_stats[6], (unsigned long long)mism);
        if (miss) atomicAdd(&gc_plan_stats[7], (unsigned long long)miss);
    }
}

template <int ITOP>
__device__ __forceinline__ void plan4_code_dispatch(int tcls, double ci, double spc, int clen, int smax, const Plan4Job *J, int lane)
{
    if (tcls == 0) plan4_code_wave<ITOP, 8>(ci, spc, clen, smax, J, lane);
    else if (tcls == 1) plan4_code_wave<ITOP, GC_CLAIM_TAIL>(ci, spc, clen, smax, J, lane);
    else plan4_code_wave<ITOP, GC_CLAIM_TAIL2>(ci, spc, clen, smax, J, lane);
}

__global__ __launch_bounds__(128) void trk_plan4_kernel(const GcChan *__restrict__ chan, const GcTrkState *__restrict__ state_in,
                                                        GcTrkState *__restrict__ state_out, GcTrkPlan *__restrict__ plan,
                                                        int nch, int nepoch, const int *__restrict__ claims_code,
                                                        const int *__restrict__ claims_car, int verify)
{
    const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ch >= nch) return;
    const GcChan c = chan[ch];
    const GcTrkState s = state_in[ch];
    const double ci = __dmul_rn(c.ti, s.codefreq), spc = __ddiv_rn(s.codefreq, c.f_sf), ps = gc_carrier_ps(s.carrfreq, c.ti);
    const int cls = plan2_class(c.ti, s.codefreq, c.clen, c.smax);
    if (cls < 0 || !(ci > 0.0 && ci < (double)c.clen)) {     // no instance of the batch chain for this channel: the chain that certifies its own crossings
        trk_plan_body(chan, state_in, state_out, plan, nch, nepoch);
        return;
    }
    const int itop = cls / 3 + 7, tcls = cls % 3;
    constexpr int RW4 = GC_CLAIM_ROW / 4;
    Plan4Job J;
    J.claims_code = reinterpret_cast<const int4 *>(claims_code) + (size_t)ch * nepoch * RW4;
    J.claims_car = reinterpret_cast<const int4 *>(claims_car) + (size_t)ch * nepoch * RW4;
    J.out = plan + (size_t)ch * nepoch;
    J.state_out = state_out + ch;
    J.s = s;
    J.nepoch = nepoch;
    J.nblk = (nepoch + GC_P4_BLK - 1) / GC_P4_BLK;
    J.verify = verify;
    if (tid == 0) g_plan4.prog = 0;
    __syncthreads();
    // the chains are latency bound and share their SIMDs with correlator wavefronts of the batch before: let them issue first
    __builtin_amdgcn_s_setprio(3);
    if (wave == 0) {
        switch (itop) {
        case 7:  plan4_code_dispatch<7>(tcls, ci, spc, c.clen, c.smax, &J, lane); break;
        case 8:  plan4_code_dispatch<8>(tcls, ci, spc, c.clen, c.smax, &J, lane); break;
        case 9:  plan4_code_dispatch<9>(tcls, ci, spc, c.clen, c.smax, &J, lane); break;
        case 10: plan4_code_dispatch<10>(tcls, ci, spc, c.clen, c.smax, &J, lane); break;
        case 11: plan4_code_dispatch<11>(tcls, ci, spc, c.clen, c.smax, &J, lane); break;
        default: plan4_code_dispatch<12>(tcls, ci, spc, c.clen, c.smax, &J, lane); break;
        }
    } else {
        plan4_car_wave(ps, c.nsamp, &J, lane);
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

// claims: gc_trk_spec_ints(nch * nepoch) ints of scratch: code rows, carrier rows, then the deviations of the
// e_off + nepoch <= 2 * nepoch periods between the state and the batch's end (two doubles per period)
size_t gc_trk_spec_ints(size_t units) { return units * (2 * GC_CLAIM_ROW + 8); }

int gc_launch_trk_spec(hipStream_t st, const GcChan *chan, const GcTrkState *state_in, int nch, int nepoch, int *claims,
                       int e_off)
{
    if (!claims || trk_nospec() || nepoch > GC_PLAN_MAXE || e_off < 0 || e_off > nepoch) return 0;
    const size_t units = (size_t)nch * nepoch;
    const int T = e_off + nepoch;
    double *devc = reinterpret_cast<double *>(claims + 2 * units * GC_CLAIM_ROW), *devk = devc + 2 * units;
    hipLaunchKernelGGL(trk_specdev_kernel, dim3((nch * T + 63) / 64), dim3(64), 0, st, chan, state_in, devc, devk, nch, T);
    GC_HIP(hipGetLastError());
    const int nchunk = (nepoch + GC_SPEC_CHUNK - 1) / GC_SPEC_CHUNK;
    hipLaunchKernelGGL(trk_spec_kernel, dim3(nch * nchunk), dim3(GC_SPEC_CHUNK), 0, st, chan, state_in, devc, devk, claims,
                       claims + units * GC_CLAIM_ROW, nch, nepoch, e_off);
    GC_HIP(hipGetLastError());
    return 0;
}

// claims: filled by gc_launch_trk_spec for the same batch -> the batch form of the chain for every channel it has an
// instance for (plan2_class); inside the same launch the chain that certifies its crossings itself, period by
// period, serves the rest -- and everything when claims is null or the batch is longer than GC_PLAN_MAXE periods.
int gc_launch_trk_plan(hipStream_t st, const GcChan *chan, const GcTrkState *state_in, GcTrkState *state_out,
                       GcTrkPlan *plan, int nch, int nepoch, int *claims)
{
    static const int verify = getenv("GNSSCORR_PLAN_VERIFY") ? atoi(getenv("GNSSCORR_PLAN_VERIFY")) : 0;
    const bool batch = claims && !trk_nospec() && nepoch <= GC_PLAN_MAXE;
    if (batch)
        hipLaunchKernelGGL(trk_plan4_kernel, dim3(nch), dim3(128), 0, st, chan, state_in, state_out, plan, nch, nepoch,
                           claims, claims + (size_t)nch * nepoch * GC_CLAIM_ROW, verify);
    else
        hipLaunchKernelGGL(trk_plan_kernel, dim3(nch), dim3(128), 0, st, chan, state_in, state_out, plan, nch, nepoch);
    GC_HIP(hipGetLastError());
    return 0;
}
