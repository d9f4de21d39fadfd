// gnsscorr_loop.hip -- closed-loop tracking on the device, in steps.  gfx950 (MI355X).
//
// What sdrthread() does around sdrtracking() (ref src/sdrmain.c:264-312): correlate one code period, let
// sdrnavigation() look at the prompt sum (bit synchronisation and bit decision, ref src/sdrnav.c:15-36,198-282),
// accumulate (cumsumcorr, ref src/sdrtrk.c:64-76), run pll()/dll() (ref src/sdrtrk.c:95-150) -- every period until
// the nav bit is synchronised, afterwards whenever checkbit() raises swloop (every loopms periods counted from the
// bit edge) -- and clear the sums after a filter update.
//
// The frequencies only change at a filter update, so the periods between two updates of a channel (its "filter
// interval": 1 period before bit sync, up to loopms after) are open-loop work: they are planned in one go and
// correlated side by side.  A run is a chain of launch pairs on one stream, no host round trip:
//
//   trk_step_tail  one wavefront per channel: closes the interval the previous correlator launch produced
//                  (partial sums -> II/QQ, sdrnavigation's bit sync / bit decision, cumsumcorr, pll/dll where due,
//                  log rows) and plans the next one (the exact NCO chain of gnsscorr_nco.h with the piece tables of
//                  every period, unit constants and rounds, written for the correlator).
//   trk_step_corr  (channel, period of the interval, quarter) -> one 256-lane workgroup running ps_unit
//                  (gnsscorr_ps.h) on four rounds of 1024 IQ samples, one per wavefront; int32 partial sums per workgroup.
//
// Round 2 ran all of this in ONE workgroup per channel, period by period (32 of 256 CUs busy, 35 us per period).
#include <cstdlib>
#include <type_traits>

// the shaped code step inline: its emitter then lives in registers (handed by reference to an out-of-line function it
// sits in scratch memory, and every piece costs a few scratch round trips: 20 000 clocks per period instead of 8 000)
#define GC_CODE_PERIOD_INLINE
#include "gnsscorr_internal.h"
#include "gnsscorr_ps.h"

#ifdef GC_TAIL_PROF     // (tools/debug: shader-clock stamps of channel 0's tail wavefront, summed over the launches)
__device__ unsigned long long gc_tail_prof[16];
#define GC_TSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
        gc_tail_prof[i] += t_ - tprev_; tprev_ = t_; } } while (0)
extern "C" int gnsscorr_debug_tail_prof(unsigned long long *dst, int reset)
{
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(gc_tail_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(gc_tail_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define GC_TSTAMP(i) do { } while (0)
#endif

namespace {

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

#define GC_NAVSYNCTH 50         // ref src/sdr.h:157

// checksync(), ref src/sdrnav.c:198-233.  nav->sdreph.prn is the channel's PRN (ref src/sdrinit.c:506), so every
// PRN above 5 takes the first branch (written for BeiDou's NH20 overlay): a shift register of the last `rate` prompt
// signs, correlated with the overlay code -- all ones for L1CA / SBAS / G1 (ref src/sdrinit.c:520-521,542-543,557-558)
// -- i.e. synchronised once `rate` consecutive prompts have one sign.  PRN 1-5: votes for the position of sign
// changes, synchronised once a position has more than NAVSYNCTH votes.
__device__ __forceinline__ int nav_checksync(gnsscorr_loop_t *L, double IP, double IPold)
{
    GC_FP_STRICT
    const int rate = L->rate;
    if (L->prn > 5) {
        for (int i = 0; i + 1 < rate; i++) L->bitsync[i] = L->bitsync[i + 1];      // shiftdata(&bitsync[0], &bitsync[1], ., rate-1)
        L->bitsync[rate - 1] = IP < 0 ? -1 : 1;
        int corr = 0;
        for (int i = 0; i < rate; i++) corr += L->bitsync[i];                      // ocode[i] = 1
        if ((corr < 0 ? -corr : corr) == rate) {
            L->synci = L->biti;
            return 1;
        }
    } else {
        if (IPold * IP < 0) {
            L->bitsync[L->biti] += 1;
            int maxi = L->bitsync[0], ind = 0;                                      // maxvi(bitsync, rate, -1, -1, &synci)
            for (int i = 1; i < rate; i++)
                if (maxi < L->bitsync[i]) { maxi = L->bitsync[i]; ind = i; }
            L->synci = ind;
            if (maxi > GC_NAVSYNCTH) {
                L->synci--;
                if (L->synci < 0) L->synci = rate - 1;
                return 1;
            }
        }
    }
    return 0;
}

// checkbit(), ref src/sdrnav.c:241-282 (the frame bit buffer fbits[] is the nav decoder's: the decided bits leave
// through the log rows instead).  Returns the reference's syncflag.
__device__ __forceinline__ int nav_checkbit(gnsscorr_loop_t *L, double IP)
{
    GC_FP_STRICT
    const int diffi = L->biti - L->synci;
    int syncflag = 1;
    L->swreset = 0;
    L->swsync = 0;
    if (diffi == 1 || diffi == -L->rate + 1) {
        L->bitIP = IP;
        L->swreset = 1;
        L->navcnt = 1;
    } else {
        L->bitIP += IP;
        if (L->bitIP * IP < 0) syncflag = 0;
    }
    L->swloop = (L->navcnt % L->loopms == 0) ? 1 : 0;
    if (diffi == 0) {
        const int polarity = L->flagpol ? -1 : 1;
        L->bit = (L->bitIP < 0) ? -polarity : polarity;
        L->swsync = 1;
    }
    L->navcnt++;
    return syncflag;
}

// the part of sdrnavigation() in front of the frame decoder, ref src/sdrnav.c:18-36
// cnt % rate (ref src/sdrnav.c:18) without the 64-bit division while cnt fits 32 bits
__device__ __forceinline__ int nav_biti(uint64_t cnt, int rate)
{
    if ((cnt >> 32) == 0) return (int)((unsigned)cnt % (unsigned)rate);
    return (int)(cnt % (uint64_t)rate);
}

// late_after = 2000 / (ctime * 1000), the threshold of ref src/sdrnav.c:26,30 (the same for every period of a launch);
// II0 / oldI0: sdr->trk.II[0] and sdr->trk.oldI[0] as sdrtracking() has left them when it calls sdrnavigation()
__device__ __forceinline__ void nav_step_io(gnsscorr_loop_t *L, double late_after, double II0, double oldI0)
{
    GC_FP_STRICT
    const uint64_t cnt = L->cnt;
    L->biti = nav_biti(cnt, L->rate);
    const bool late = (double)cnt > late_after;
    if (L->rate == 1 && late) {
        L->synci = 0;
        L->flagsync = 1;
    }
    if (!L->flagsync && late) L->flagsync = nav_checksync(L, II0, oldI0);
    if (L->flagsync) nav_checkbit(L, II0);
}

// emitters that fill LDS tables (lane-uniform calls from one wavefront).  The table pointers are LDS-typed (address
// space 3), so every access is a DS instruction, which one wavefront executes in order.  Through generic pointers the
// stores become FLAT instructions; those reach the LDS by way of the texture path and can be overtaken by a DS read
// issued after them -- the round-2 closed-loop kernel stalled on exactly that.
// Both emitters keep what they need of the previous piece in registers (no LDS read on the chain's path) and leave
// what can be done for all pieces at once -- the fixed-point form of the carrier pieces, the reciprocal steps of the
// code pieces -- to a lane-parallel pass afterwards (lds_car_finish / lds_code_finish).
typedef __attribute__((address_space(3))) int *gc_lds_int;
typedef __attribute__((address_space(3))) GcCarSeg *gc_lds_car;
typedef __attribute__((address_space(3))) GcCodeSeg *gc_lds_code;
typedef __attribute__((address_space(3))) double *gc_lds_f64;
struct LdsCarTable {        // GcCarTable (gnsscorr_nco.h): pieces as (x, d) doubles in the slots, converted by lds_car_finish
    gc_lds_int k0;
    gc_lds_car seg;
    int n, overflow;
    bool lastzero;
    __device__ void operator()(int k, double x, double d, int)
    {
        const int e = (int)((gc_d2u(x) >> 52) & 0x7FF) - 1023;
        const bool zero = e < 0 || e >= 31;             // gc_carseg_make gives the all-zero piece exactly then
        if (n > 0 && zero && lastzero) return;
        if (n >= GC_NCAR) { overflow = 1; return; }
        k0[n] = k;
        gc_lds_f64 raw = (gc_lds_f64)(seg + n);
        raw[0] = x;
        raw[1] = d;
        lastzero = zero;
        n++;
    }
};
__device__ __forceinline__ void lds_car_finish(gc_lds_car seg, int n, int lane)
{
    if (lane < n) {
        gc_lds_f64 raw = (gc_lds_f64)(seg + lane);
        const GcCarSeg s = gc_carseg_make(raw[0], raw[1]);
        seg[lane].fx = s.fx;
        seg[lane].dfx = s.dfx;
    }
}
struct LdsCodeTable {       // GcCodeTable (gnsscorr_nco.h); inv is filled in by lds_code_finish
    gc_lds_code seg;
    int cap, n, overflow;
    int lw, lcnt;           // the last piece: wrap count, positions
    double ly0, lyl;        //                 first and last value
    __device__ void operator()(int j, double y, double d, int count, int w)
    {
        GC_FP_STRICT
        const double yl = fma((double)(count - 1), d, y);
        if (n > 0 && lw == w && ly0 > -1.0 && lyl < 1.0 && y > -1.0 && yl < 1.0) {
            lcnt += count;
            lyl = yl;
            seg[n - 1].cnt = lcnt;
            seg[n - 1].ylast = yl;
            seg[n - 1].d = 0.0;                  // d = 0 with y0 in (-1, 1): every position is chip 0
            return;
        }
        if (n >= cap) { overflow = 1; return; }
        seg[n].y0 = y;
        seg[n].d = count > 1 ? d : 0.0;
        seg[n].inv = 0.0;
        seg[n].ylast = yl;
        seg[n].j0 = j;
        seg[n].cnt = count;
        seg[n].w = w;
        seg[n].pad = 0;
        lw = w; lcnt = count; ly0 = y; lyl = yl;
        n++;
    }
};
__device__ __forceinline__ void lds_code_finish(gc_lds_code seg, int n, int lane)
{
    if (lane < n) {
        const double d = seg[lane].d;
        seg[lane].inv = (seg[lane].cnt > 1 && d != 0.0) ? __ddiv_rn(1.0, d) : 0.0;
    }
}

// chip under replica position j from an LDS-typed code table (gc_code_chip_at of gnsscorr_nco.h)
__device__ __forceinline__ int lds_code_chip_at(gc_lds_code seg, int nseg, int j, int *w, int *piece)
{
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid].j0 <= j) lo = mid; else hi = mid - 1;
    }
    if (w) *w = seg[lo].w;
    if (piece) *piece = lo;
    int i = j - seg[lo].j0;
    if (i < 0) i = 0;
    if (i >= seg[lo].cnt) i = seg[lo].cnt - 1;
    const double d = seg[lo].d, y0 = seg[lo].y0;
    if (d == 0.0) return (int)y0;
    return (int)__fma_rn((double)i, d, y0);
}

// periods up to and including the next one after which the filters run, for a synchronised channel: checkbit()'s
// counter (ref src/sdrnav.c:249-262) run ahead.  At most kmax.
__device__ __forceinline__ int nav_interval(const gnsscorr_loop_t *L, int kmax)
{
    if (!L->flagsync) return 1;                     // prm1 after every period (ref src/sdrmain.c:272-276)
    int navcnt = L->navcnt;
    for (int j = 1; j <= kmax; j++) {
        const int biti = nav_biti(L->cnt + (uint64_t)(j - 1), L->rate);
        const int diffi = biti - L->synci;
        if (diffi == 1 || diffi == -L->rate + 1) navcnt = 1;
        if (navcnt % L->loopms == 0) return j;
        navcnt++;
    }
    return kmax;
}

// gc_fast_init (gnsscorr_nco.h) with one lane per binade: the table entries of an addend, written straight into
// an LDS-resident table (DS stores, by name).  Every lane of the calling wavefront takes part.
__device__ __forceinline__ void fast_init_lanes(GcNcoFast &f, double s, bool with_inv, int lane)
{
    GC_FP_STRICT
    const uint64_t us = gc_d2u(s);
    const int es = (int)((us >> 52) & 0x7FF);
    const bool ok = es > 60 && es < 0x7FF - GC_NB - 4;
    double d = 0.0, inv = 0.0;
    bool tie = false;
    if (ok && lane < GC_NB) {
        const int ex = es + 2 + lane;
        const int et = es + 1075 - ex;
        double b = 0.0;
        if (et >= 1023 - 1) {
            const double t = gc_u2d((us & 0x800FFFFFFFFFFFFFull) | ((uint64_t)et << 52));
            b = rint(t);
            tie = fabs(t - b) == 0.5;
        }
        d = ldexp(b, ex - 1075);
        if (b != 0.0 && with_inv) inv = __ddiv_rn(1.0, fabs(d));
    }
    const unsigned long long tm = __ballot(tie);
    if (lane < GC_NB) {
        f.d[lane] = d;
        f.inv[lane] = inv;
    }
    if (lane == 0) {
        f.s = s;
        f.inv_s = __ddiv_rn(1.0, fabs(s));
        f.tie = (unsigned)tm & ((1u << GC_NB) - 1u);
        f.ex0 = ok ? es + 2 : 0x7FFFFFF;
    }
}
// gc_code_plan_init(P, ci, len, smax) around fast_init_lanes
__device__ __forceinline__ void code_plan_init_lanes(GcCodePlan &P, double ci, int len, int smax, int lane)
{
    GC_FP_STRICT
    fast_init_lanes(P.f, ci, true, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double dlen = (double)len;
    const double limtop = gc_u2d(gc_d2u(dlen) - 1);
    const int ex0 = P.f.ex0;
    const unsigned tie = P.f.tie;
    const int itop = (int)((gc_d2u(limtop) >> 52) & 0x7FF) - ex0;
    const bool ok = ex0 != 0x7FFFFFF && ci > 0.0 && itop >= 1 && itop < GC_NB &&
                    dlen + ci < gc_u2d((uint64_t)(ex0 + itop + 1) << 52) && !((tie >> itop) & 1);
    int it = -1;
#pragma unroll
    for (int i = 0; i < GC_NB; i++)
        if (((tie >> i) & 1) && i <= itop) it = i;
    if (lane == 0) {
        P.dlen = dlen;
        P.smaxci = (double)smax * ci;
        P.limtop = limtop;
        P.itop = itop;
        P.ok = ok;
        P.exact = ok && gc_cert_exact(limtop, ci, dlen);
        P.it = it;
    }
}

// the general walkers for a period the shaped steps decline (a tie in the top binade, a phase next to zero, ...):
// out of line, with tables of their own (the workgroup's shared tables stay read-only)
template <class Emit>
__device__ __attribute__((noinline)) double tail_carrier_slow(double ps, double remcarr, int n, Emit &emit)
{
    GcNcoFast f, fp;
    gc_fast_init(f, ps);
    gc_fast_init(fp, -GC_NCO_DPI);
    const double xn = gc_fast_carrier_walk(f, gc_carrier_phis(remcarr), n, emit);
    return gc_fast_prem(fp, xn);
}
template <class Emit>
__device__ __attribute__((noinline)) double tail_code_slow(double ci, double remcode, int clen, int smax, int nt, Emit &emit)
{
    GcNcoFast f;
    gc_fast_init(f, ci);
    const double cend = gc_fast_code_walk(f, gc_code_start(remcode, smax, ci, clen), clen, nt, emit);
    return gc_code_rem(cend, smax, ci);
}

// the step WITH its checks, for a period whose start lies outside the bracket its claims were proved for (rare): out
// of line, so that the chains' loops carry the value form only.  PC / PK / cl: the workgroup's tables and rows (LDS).
template <int IT>
__device__ __attribute__((noinline)) bool tail_code_checked(const GcCodePlan *PC, double remcode, int nt, int tcls, const GcCodeClaims *cl, double *out)
{
    GcCodeStepC<IT> SC;
    gc_code_stepc_init(SC, *PC);
    GcCodeClaims c2 = *cl;
    double r;
    bool ok;
    if (tcls == 0) ok = gc_code_claims_step<IT, 8, false, true>(*PC, SC, remcode, nt, c2, &r);
    else if (tcls == 1) ok = gc_code_claims_step<IT, GC_CLAIM_TAIL, false, true>(*PC, SC, remcode, nt, c2, &r);
    else ok = gc_code_claims_step<IT, GC_CLAIM_TAIL2, false, true>(*PC, SC, remcode, nt, c2, &r);
    if (ok) *out = r;
    return ok;
}
__device__ __attribute__((noinline)) bool tail_carrier_checked(const GcCarPlan *PK, int nsamp, double remcarr, int n, const GcCarClaims *cl, double *out)
{
    GcCarStepC CK;
    gc_car_stepc_init(CK, *PK, nsamp + 16);
    GcCarClaims c2 = *cl;
    double r;
    const bool ok = gc_carrier_claims_step<false, true>(*PK, CK, remcarr, n, c2, &r);
    if (ok) *out = r;
    return ok;
}

#define GC_TAIL_NW 8            // wavefronts of the tail workgroup
struct TailShared {
    gnsscorr_loop_t lp;
    GcTrkState st;                                  // the channel's state while the kernel runs (frequencies: wavefront 0)
    GcCodePlan PC;
    GcCarPlan PK;
    // the interval's periods: start values of the chain (entry k = the state the interval leaves behind), ...
    double remcode[GC_STEP_KMAX + 1], remcarr[GC_STEP_KMAX + 1];
    uint64_t buffloc[GC_STEP_KMAX + 1];
    int n[GC_STEP_KMAX], valid[GC_STEP_KMAX], ncar[GC_STEP_KMAX], ncode[GC_STEP_KMAX], bad[GC_STEP_KMAX];
    int want, k, prog, progc, nexttask, starved, navdone;
    int nflagsync[GC_STEP_KMAX], nswloop[GC_STEP_KMAX], nbit[GC_STEP_KMAX];     // sdrnavigation()'s verdict per closing period
    int psum[GC_STEP_KMAX][2 * GNSSCORR_MAXTAPS];   // the closing interval's correlator sums: [period][tap | ntap + tap]
    GcCodeClaims ccl[GC_STEP_KMAX];                 // the periods' claims (gnsscorr_nco.h: period steps on claims), discovered
    GcCarClaims kcl[GC_STEP_KMAX];                  // side by side, one lane per period, then evaluated and checked by the chain
    // ... and their NCO tables
    int k0[GC_STEP_KMAX][GC_NCAR + 4];
    GcCarSeg car[GC_STEP_KMAX][GC_NCAR];
    GcCodeSeg code[GC_STEP_KMAX][GC_NCODE];
};

__device__ __forceinline__ void tail_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One workgroup (GC_TAIL_NW wavefronts) per channel.
//   close     wavefront 0: the interval the previous correlator launch produced -- per period the sums, sdrnavigation's
//             bit sync / bit decision, cumsumcorr, the filters where due, the log row;
//   plan      how many periods the next interval has (nav_interval, the run's end, what the ring holds); the step tables
//             of its frequencies (one lane per binade); for more than one period the chain of period starts -- code on
//             wavefront 1, carrier on wavefront 0 one step behind (it needs the period lengths), nothing emitted;
//   tables    every (period, NCO) is a task of its own: the period step again, from its known start, with the piece
//             table emitted into LDS -- all wavefronts side by side (a one-period interval skips the chain: these steps
//             are the chain);
//   out       per period: fixed-point / reciprocal forms (lane-parallel), the tables' invariants, the rounds, and the
//             copy to the correlator's buffers.
__global__ __launch_bounds__(64 * GC_TAIL_NW) void trk_step_tail_kernel(
    const GcChan *__restrict__ chan, GcTrkState *__restrict__ state, gnsscorr_loop_t *__restrict__ loop,
    GcStepMeta *__restrict__ meta, const uint64_t *__restrict__ wrpos, const int *__restrict__ partial,
    GcTrkUnit *__restrict__ unit, GcUnitSegs *__restrict__ segs, GcRound *__restrict__ rounds, double *__restrict__ corrI,
    double *__restrict__ corrQ, int *__restrict__ nsamp_out, gnsscorr_trklog_t *__restrict__ log, int *__restrict__ ndone,
    int *__restrict__ nco_overflow, unsigned *__restrict__ hostflags, int nch, int nper, int nseg, int ntap_stride, int max_n,
    int kcap, int plan)
{
    __shared__ __attribute__((aligned(16))) TailShared S;
    const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ch >= nch) return;
#ifdef GC_TAIL_PROF
    unsigned long long tprev_ = __builtin_readcyclecounter();
#endif
    const GcChan &c = chan[ch];
    const int ntap = c.ntap, dtype = c.dtype;
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(loop + ch);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&S.lp);
        for (int i = tid; i < (int)(sizeof(gnsscorr_loop_t) / 8); i += 64 * GC_TAIL_NW) dst[i] = src[i];
    }
    GcStepMeta m = meta[ch];
    if (tid == 0) { S.st = state[ch]; S.navdone = 0; }
    __syncthreads();
    gnsscorr_loop_t *lp = &S.lp;
    const double II0_before = S.lp.II[0];               // trk.II[0] of the last period closed (wavefront 0 overwrites it below)
    const bool was_finished = m.finished != 0;
    GC_TSTAMP(0);       // state in

    // ---- close the interval the correlator launch before this one produced (wavefront 0) -----------------------
    // the interval's correlator sums: every (period, tap, rail) of it at once
    for (int x = tid; x < m.k * 2 * ntap; x += 64 * GC_TAIL_NW) {
        const int e = x / (2 * ntap), r = x - e * 2 * ntap;            // r: tap (I rail), ntap + tap (Q rail)
        const int col = r < ntap ? r : ntap_stride + (r - ntap);
        const int *pp = partial + ((size_t)ch * m.kcap + e) * nseg * 2 * ntap_stride + col;
        int sum = 0;
        for (int sg = 0; sg < nseg; sg++) sum += pp[(size_t)sg * 2 * ntap_stride];
        S.psum[e][r] = sum;
    }
    __syncthreads();
    GC_TSTAMP(1);       // sums
    // sdrnavigation()'s bit synchronisation / bit decision for every period of the interval (ref src/sdrnav.c:18-36): it
    // only looks at the prompt sums trk.II[0] / trk.oldI[0] and its own state, so wavefront 1 runs it for all periods
    // while wavefront 0 accumulates; the filters (wavefront 0, below) then find each period's flags in LDS.
    if (wave == 1 && lane == 0 && m.k > 0) {
        const double late_after = __ddiv_rn(2000.0, __dmul_rn(lp->ctime, 1000.0));
        double prevII0 = II0_before;                    // trk.II[0] of the period before (what memcpy leaves in oldI[0]: ntap >= 3)
        for (int e = 0; e < m.k; e++) {
            const double II0 = (double)S.psum[e][ntap] * (1.0 / 32.0);      // trk.II[0] = the correlator's QQ[0] (ref src/sdrtrk.c:42)
            nav_step_io(lp, late_after, II0, prevII0);
            S.nflagsync[e] = lp->flagsync;
            S.nswloop[e] = lp->swloop;
            S.nbit[e] = (lp->flagsync && lp->swsync) ? lp->bit : 0;
            lp->cnt = lp->cnt + 1;
            prevII0 = II0;
            __hip_atomic_store(&S.navdone, e + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (wave == 0) {
        GcTrkState st = S.st;
        for (int e = 0; e < m.k; e++) {
            const int p = m.pbase + e;
            if (lane < ntap) {
                const int sI = S.psum[e][lane], sQ = S.psum[e][ntap + lane];
                const double cI = (double)sI * (1.0 / 32.0), cQ = (double)sQ * (1.0 / 32.0);     // correlator's II, QQ (ref src/sdrcmn.c:716-719)
                corrI[((size_t)ch * nper + p) * ntap + lane] = cI;
                corrQ[((size_t)ch * nper + p) * ntap + lane] = cQ;
                // memcpy(oldI, II, 1 + 2*corrn*sizeof(double)): the last tap only gets its lowest byte (ref src/sdrtrk.c:35-36)
                const double pII = lp->II[lane], pQQ = lp->QQ[lane];
                double oI = pII, oQ = pQQ;
                if (lane == ntap - 1) {
                    oI = gc_u2d((gc_d2u(lp->oldI[lane]) & ~0xFFull) | (gc_d2u(pII) & 0xFFull));
                    oQ = gc_u2d((gc_d2u(lp->oldQ[lane]) & ~0xFFull) | (gc_d2u(pQQ) & 0xFFull));
                }
                lp->oldI[lane] = oI;
                lp->oldQ[lane] = oQ;
                // correlator(..., trk.QQ, trk.II, ...): trk.II <- sum dataQ*code, trk.QQ <- sum dataI*code (ref src/sdrtrk.c:42)
                lp->II[lane] = cQ;
                lp->QQ[lane] = cI;
                // cumsumcorr, polarity +1: the overlay code is all ones (ref src/sdrtrk.c:64-76, src/sdrmain.c:269)
                lp->oldsumI[lane] = __dadd_rn(lp->oldsumI[lane], oI);
                lp->oldsumQ[lane] = __dadd_rn(lp->oldsumQ[lane], oQ);
                lp->sumI[lane] = __dadd_rn(lp->sumI[lane], cQ);
                lp->sumQ[lane] = __dadd_rn(lp->sumQ[lane], cI);
            }
            tail_wave_sync();
            // (this period's nav flags: wavefront 1 is usually ahead)
            while (__hip_atomic_load(&S.navdone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= e) __builtin_amdgcn_s_sleep(1);
            int flag = 0;
            if (lane == 0) {
                if (!S.nflagsync[e]) {
                    loop_pll(lp, st, 0, lp->ctime);
                    loop_dll(lp, st, 0, lp->ctime);
                    flag = 1;
                } else if (S.nswloop[e]) {
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
                lg->flagloopfilter = flag;
                lg->flagsync = S.nflagsync[e];
                lg->navbit = S.nbit[e];
                if (flag && e + 1 < m.k) m.early = 1;   // the plan held the frequencies beyond a filter update: must not happen
            }
            flag = __builtin_amdgcn_readfirstlane(flag);
            st.carrfreq = gc_u2d(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gc_d2u(st.carrfreq) >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)gc_d2u(st.carrfreq)));
            st.codefreq = gc_u2d(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gc_d2u(st.codefreq) >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)gc_d2u(st.codefreq)));
            m.early = __builtin_amdgcn_readfirstlane(m.early);
            if (flag && lane < ntap) {      // clearcumsumcorr (ref src/sdrtrk.c:77-86)
                lp->oldsumI[lane] = 0.0;
                lp->oldsumQ[lane] = 0.0;
                lp->sumI[lane] = 0.0;
                lp->sumQ[lane] = 0.0;
            }
            tail_wave_sync();
            GC_TSTAMP(2);   // nav + filters + log
        }
        // the next interval: up to the next filter update, the end of the run, the step's capacity
        int want = 0;
        if (plan && m.done < nper) {
            want = nav_interval(lp, kcap);
            if (want > nper - m.done) want = nper - m.done;
        }
        if (lane == 0) {
            S.st.carrfreq = st.carrfreq;
            S.st.codefreq = st.codefreq;
            S.want = want;
            S.k = 0;
            S.prog = 0;
            S.progc = 0;
            S.nexttask = 0;
            S.starved = 0;
            S.remcode[0] = st.remcode;
            S.remcarr[0] = st.remcarr;
            S.buffloc[0] = st.buffloc;
        }
    }
    __syncthreads();
    m.consumed += m.k;
    m.k = 0;

    // ---- plan the next interval --------------------------------------------------------------------------------
    const int want = S.want;
    const double carrfreq = S.st.carrfreq, codefreq = S.st.codefreq;
    const double dlen = (double)c.clen;
    const double ci = __dmul_rn(c.ti, codefreq), ps = gc_carrier_ps(carrfreq, c.ti);
    const double spc = __ddiv_rn(codefreq, c.f_sf);
    const bool shape_ok = ci > 0.0 && ci < dlen;
    GcFillLanes fill{lane};
    if (want > 0) {
        // step tables of the interval's two frequencies, one lane per binade
        if (wave == 0) {
            fast_init_lanes(S.PK.f, ps, false, lane);
            if (lane == 0) S.PK.ydpi = __ddiv_rn(1.0, GC_NCO_DPI);
        } else if (wave == 1) {
            code_plan_init_lanes(S.PC, ci, c.clen, c.smax, lane);
        } else if (wave == 2) {
            fast_init_lanes(S.PK.fprem, -GC_NCO_DPI, true, lane);
        }
    }
    __syncthreads();
    GC_TSTAMP(3);       // step tables
    if (want > 1) {
        // The structure of the interval's periods from their closed-form starts (gc_spec_start), proved for a bracket
        // around each start as the batch planner's discovery does (gnsscorr_plan.hip): lane e discovers the claims of
        // period e from the bracket's lower end, lane 32 + e from its upper end; where both ends pass their checks with
        // the same claims and the same sample count, every start in between does (every operation of the step is
        // monotone in the start), and the chain below evaluates such a period without the checks.  The closed form
        // reaches at most GC_STEP_KMAX periods from the interval's exact start: +-2^-25 holds what its missing rounding
        // adds up to (< 1e-8).  A period without a bracket takes the step with its checks, then the certified step.
        const double W = 2.98023223876953125e-8;         // 2^-25
        const bool mine = lane < want || (lane >= 32 && lane - 32 < want);
        const int pe = lane & 31;
        const bool hiend = lane >= 32;
        if (wave == 2 && mine) {
            GcCodeClaims cc;
            cc.tag = 0;
            cc.n = cc.pad = 0;
            cc.lo = cc.hi = 0.0;
            bool ok = false;
            int n = 0, side = 0;
            double end = 0.0;
            if (shape_ok && spc > 1e-300 && spc < 1e300) {
                double rc, rk, dummy;
                int nhat;
                gc_spec_start(S.remcode[0], S.remcarr[0], ci, spc, ps, dlen, pe, &rc, &rk, &nhat);
                end = hiend ? rc + W : rc - W;
                const double q = __ddiv_rn(__dsub_rn(dlen, end), spc);           // ref src/sdrtrk.c:31-32
                n = (q > -2147483648.0 && q < 2147483648.0) ? (int)q : 0;
                side = end - S.PC.smaxci < 0.0 ? 1 : 0;                          // (ref src/sdrcmn.c:614: one branch for the whole bracket)
                if (n > 0 && n <= (1 << 24)) ok = gc_code_claims<true>(S.PC, end, n + 2 * c.smax, cc, &dummy);
            }
            bool same = ok && __shfl(ok ? 1 : 0, lane ^ 32, 64) != 0;
            same = same && n == __shfl(n, lane ^ 32, 64) && side == __shfl(side, lane ^ 32, 64);
            same = same && cc.i0 == __shfl(cc.i0, lane ^ 32, 64) && cc.q == __shfl(cc.q, lane ^ 32, 64) &&
                   cc.nl == __shfl(cc.nl, lane ^ 32, 64) && cc.jsum == __shfl(cc.jsum, lane ^ 32, 64);
#pragma unroll
            for (int i = 0; i < 13; i++) same = same && cc.dm[i] == __shfl(cc.dm[i], lane ^ 32, 64);
            const double other = __shfl(end, lane ^ 32, 64);
            if (!hiend) {
                // (the discovering step allows the widest tail; the chain's instance may be narrower)
                const int tmax = c.smax + 1 > 8 ? (c.smax + 1 > GC_CLAIM_TAIL ? GC_CLAIM_TAIL2 : GC_CLAIM_TAIL) : 8;
                cc.tag = (same && n + 2 * c.smax - cc.jsum <= tmax) ? 1 : 0;
                cc.n = n;
                cc.lo = end;
                cc.hi = other;
                S.ccl[pe] = cc;
            }
        }
        if (wave == 3 && mine) {
            GcCarClaims ck;
            ck.tag = 0;
            ck.nl = ck.i0 = ck.nseg = ck.kprem = 0;
            ck.pad[0] = ck.pad[1] = 0;
            ck.lo = ck.hi = 0.0;
#pragma unroll
            for (int i = 0; i < GC_CLAIM_CSEG; i++) ck.dm[i] = 0;
            bool ok = false;
            int n = 0;
            double end = 0.0;
            if (shape_ok && spc > 1e-300 && spc < 1e300) {
                double rc, rk, dummy;
                gc_spec_start(S.remcode[0], S.remcarr[0], ci, spc, ps, dlen, pe, &rc, &rk, &n);
                end = hiend ? rk + W : rk - W;
                if (n > 0 && n <= (1 << 24)) {
                    GcCarStepC CK;
                    gc_car_stepc_init(CK, S.PK, c.nsamp + 16);
                    ok = gc_carrier_claims_step<true>(S.PK, CK, end, n, ck, &dummy);
                }
            }
            bool same = ok && __shfl(ok ? 1 : 0, lane ^ 32, 64) != 0;
            same = same && ck.tag == __shfl(ck.tag, lane ^ 32, 64) && ck.nl == __shfl(ck.nl, lane ^ 32, 64) &&
                   ck.i0 == __shfl(ck.i0, lane ^ 32, 64) && ck.nseg == __shfl(ck.nseg, lane ^ 32, 64) &&
                   ck.kprem == __shfl(ck.kprem, lane ^ 32, 64);
#pragma unroll
            for (int i = 0; i < GC_CLAIM_CSEG; i++) same = same && ck.dm[i] == __shfl(ck.dm[i], lane ^ 32, 64);
            const double other = __shfl(end, lane ^ 32, 64);
            if (!hiend) {
                // (tag 2, a period inside one binade, carries its own conditions and is judged by them: it needs no bracket)
                ck.tag = same ? ck.tag : (ok && ck.tag == 2 ? 2 : 0);
                ck.nl = n;
                ck.lo = end;
                ck.hi = other;
                S.kcl[pe] = ck;
            }
        }
        __syncthreads();
    }
    GC_TSTAMP(8);       // claims
    if (want > 0) {
        const uint64_t wp = wrpos[ch];
        const bool have_data = wp >= (uint64_t)c.nsamp;
        const uint64_t bufflocnow = wp - (uint64_t)c.nsamp;
        if (wave == 1) {
            // period starts: is the period there yet (ref src/sdrtrk.c:26-30), its length (:31-32), and -- for an interval of
            // several periods -- the code chain (nothing emitted).  One instance per shape of the code step (the table binade
            // that holds the code length), so that the step's constants stay in registers over the interval.
            auto code_chain = [&](auto itop_tag) {
                constexpr int ITOP = decltype(itop_tag)::value;       // (0: no instance: the certified step serves the channel)
                constexpr int IT = ITOP ? ITOP : 7;
                GcCodeStepC<IT> SC;
                const bool inst = ITOP != 0 && S.PC.ok;
                if (inst) gc_code_stepc_init(SC, S.PC);
                const int tcls = c.smax + 1 > 8 ? (c.smax + 1 > GC_CLAIM_TAIL ? 2 : 1) : 0;       // tail positions: 8, 15 or 32
                double remcode = S.remcode[0];
                uint64_t buffloc = S.buffloc[0];
                GcNoEmit ne;
                int k = 0;
                const double yspc = __ddiv_rn(1.0, spc);
                const bool fastdiv = spc > 1e-300 && spc < 1e300 && yspc < 1e300;
                for (int e = 0; e < want; e++) {
                    if (!(have_data && bufflocnow > buffloc)) { if (lane == 0) S.starved = 1; break; }
                    // (dlen - remcode) / (codefreq / f_sf), ref src/sdrtrk.c:31-32: correctly rounded through the reciprocal
                    // (gc_div_y) where that is safe, as the batch planner divides
                    const double num = __dsub_rn(dlen, remcode);
                    const double q = fastdiv ? gc_div_y(num, spc, yspc) : __ddiv_rn(num, spc);
                    const int n = (q > -2147483648.0 && q < 2147483648.0) ? (int)q : 0;
                    const bool valid = n > 0 && n <= max_n && shape_ok;
                    if (lane == 0) {
                        S.n[e] = n;
                        S.valid[e] = valid ? 1 : 0;
                        S.remcode[e] = remcode;
                        S.buffloc[e] = buffloc;
                    }
                    k = e + 1;
                    if (want > 1) {
                        tail_wave_sync();
                        if (lane == 0) __hip_atomic_store(&S.prog, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (valid) {
                            double r = remcode;
                            GcCodeClaims cl = S.ccl[e];
                            const int nt = n + 2 * c.smax;
                            // the exact start inside the bracket the claims were proved for: the values alone (the step's
                            // verdict is unused, the compiler drops the checks); else the step with its checks
                            const bool inside = __builtin_amdgcn_readfirstlane((inst && cl.tag == 1 && remcode >= cl.lo && remcode <= cl.hi && n == cl.n) ? 1 : 0) != 0;
                            bool ok = inside;
                            if (inside) {
                                if (tcls == 0) (void)gc_code_claims_step<IT, 8, false>(S.PC, SC, remcode, nt, cl, &r);
                                else if (tcls == 1) (void)gc_code_claims_step<IT, GC_CLAIM_TAIL, false>(S.PC, SC, remcode, nt, cl, &r);
                                else (void)gc_code_claims_step<IT, GC_CLAIM_TAIL2, false>(S.PC, SC, remcode, nt, cl, &r);
                            } else if (inst && cl.tag == 1) {
                                double r2;
                                ok = tail_code_checked<IT>(&S.PC, remcode, nt, tcls, &S.ccl[e], &r2);
                                if (ok) r = r2;
                            }
                            if (ok) remcode = r;
                            else if (gc_code_period(S.PC, remcode, nt, fill, &r, ne)) remcode = r;
                            else remcode = tail_code_slow(ci, remcode, c.clen, c.smax, nt, ne);
                        }
                    }
                    buffloc += (uint64_t)(int64_t)n;
                    if (lane == 0) { S.remcode[e + 1] = remcode; S.buffloc[e + 1] = buffloc; }
                }
                tail_wave_sync();
                if (lane == 0) {
                    S.k = k;
                    __hip_atomic_store(&S.prog, 0x7fffffff, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            };
            switch ((want > 1 && S.PC.ok) ? S.PC.itop : 0) {
            case 7:  code_chain(std::integral_constant<int, 7>{}); break;
            case 8:  code_chain(std::integral_constant<int, 8>{}); break;
            case 9:  code_chain(std::integral_constant<int, 9>{}); break;
            case 10: code_chain(std::integral_constant<int, 10>{}); break;
            case 11: code_chain(std::integral_constant<int, 11>{}); break;
            case 12: code_chain(std::integral_constant<int, 12>{}); break;
            default: code_chain(std::integral_constant<int, 0>{}); break;
            }
        } else if (wave == 0 && want > 1) {
            // the carrier chain, one step behind
            double remcarr = S.remcarr[0];
            GcNoEmit ne;
            GcCarStepC CK;
            gc_car_stepc_init(CK, S.PK, c.nsamp + 16);
            for (int e = 0; e < want; e++) {
                int pg;
                while ((pg = __hip_atomic_load(&S.prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) <= e) __builtin_amdgcn_s_sleep(1);
                if (pg == 0x7fffffff && e >= S.k) break;
                const int n = S.n[e];
                if (S.valid[e]) {
                    double r = remcarr;
                    GcCarClaims cl = S.kcl[e];
                    const int tagu = __builtin_amdgcn_readfirstlane(cl.tag);
                    const bool inside = __builtin_amdgcn_readfirstlane((tagu == 1 && remcarr >= cl.lo && remcarr <= cl.hi && n == cl.nl) ? 1 : 0) != 0;
                    bool ok = inside;
                    if (inside) {
                        (void)gc_carrier_claims_step<false, false, 1>(S.PK, CK, remcarr, n, cl, &r);      // the values alone: the bracket is the proof
                    } else if (tagu == 2) {
                        double r2 = remcarr;
                        ok = gc_carrier_claims_step<false, false, 2>(S.PK, CK, remcarr, n, cl, &r2);       // (its own conditions, checked)
                        if (ok) r = r2;
                    } else if (tagu == 1) {
                        double r2;
                        ok = tail_carrier_checked(&S.PK, c.nsamp, remcarr, n, &S.kcl[e], &r2);
                        if (ok) r = r2;
                    }
                    if (ok) remcarr = r;
                    else if (gc_carrier_period(S.PK, remcarr, n, fill, &r, ne)) remcarr = r;
                    else remcarr = tail_carrier_slow(ps, remcarr, n, ne);
                }
                if (lane == 0) S.remcarr[e + 1] = remcarr;
                tail_wave_sync();
                if (lane == 0) __hip_atomic_store(&S.progc, e + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (lane == 0) __hip_atomic_store(&S.progc, 0x7fffffff, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    GC_TSTAMP(4);       // chain
    // ---- tables: task t = (period t/2, carrier | code), taken from a queue in order -- a wavefront that has no chain
    // to run starts on period 0 as soon as the chain has published its start, the chain's wavefronts join when done ----
    for (;;) {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&S.nexttask, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= 2 * want) break;
        const int e = t >> 1;
        // the period's start: code side from the code chain (prog), carrier side from the carrier chain (progc)
        int pg;
        while ((pg = __hip_atomic_load(&S.prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) <= e) __builtin_amdgcn_s_sleep(1);
        if (pg == 0x7fffffff && e >= S.k) break;            // the ring holds fewer periods than wanted
        if (!(t & 1) && want > 1)
            while (__hip_atomic_load(&S.progc, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < e) __builtin_amdgcn_s_sleep(1);
        const int k = want;                                 // (k == 1 below: the one-period interval, whose steps are the chain)
        if (!S.valid[e]) {
            if (lane == 0) { if (t & 1) S.ncode[e] = 0; else S.ncar[e] = 0; S.bad[e] = 0; }
            if (k == 1 && lane == 0) { if (t & 1) S.remcode[1] = S.remcode[0]; else S.remcarr[1] = S.remcarr[0]; }
            continue;
        }
        const int n = S.n[e];
        if (t & 1) {
            LdsCodeTable dt{(gc_lds_code)S.code[e], GC_NCODE, 0, 0, 0, 0, 0.0, 0.0};
            const double remcode = S.remcode[e];
            double r;
            if (!gc_code_period(S.PC, remcode, n + 2 * c.smax, fill, &r, dt)) {
                dt.n = 0;
                dt.overflow = 0;
                r = tail_code_slow(ci, remcode, c.clen, c.smax, n + 2 * c.smax, dt);
            }
            tail_wave_sync();
            lds_code_finish((gc_lds_code)S.code[e], dt.n, lane);
            if (lane == 0) {
                S.ncode[e] = dt.overflow ? -1 : dt.n;
                if (k == 1) S.remcode[1] = r;           // (one period: this step is the chain)
            }
        } else {
            LdsCarTable ct{(gc_lds_int)S.k0[e], (gc_lds_car)S.car[e], 0, 0, false};
            const double remcarr = S.remcarr[e];
            double r;
            if (!gc_carrier_period(S.PK, remcarr, n, fill, &r, ct)) {
                ct.n = 0;
                ct.overflow = 0;
                ct.lastzero = false;
                r = tail_carrier_slow(ps, remcarr, n, ct);
            }
            tail_wave_sync();
            lds_car_finish((gc_lds_car)S.car[e], ct.n, lane);
            if (lane == 0) {
                S.ncar[e] = ct.overflow ? -1 : ct.n;
                if (k == 1) S.remcarr[1] = r;
            }
        }
    }
    __syncthreads();
    GC_TSTAMP(5);       // tables
    // ---- out: per period the invariants, the rounds, the copy to the correlator's buffers ---------------------------
    const int k = S.k;
    const int nit = trk_ps_nit(dtype, 2), rgrp = GC_PS_WLANES * nit, rsamp = rgrp * (16 / dtype);     // a round: one wavefront's share (gnsscorr_ps.h)
    for (int e = wave; e < k; e += GC_TAIL_NW) {
        const int n = S.n[e], nt = n + 2 * c.smax;
        const int p = m.done + e;
        const size_t ui = (size_t)ch * kcap + e;
        const uint64_t buffloc = S.buffloc[e];
        gc_lds_int sk0 = (gc_lds_int)S.k0[e];
        gc_lds_car scar = (gc_lds_car)S.car[e];
        gc_lds_code scode = (gc_lds_code)S.code[e];
        GcTrkUnit u;
        const uint64_t a0 = (buffloc % c.ringlen) * (uint64_t)dtype;
        u.a_al = a0 & ~(uint64_t)15;
        u.head = (int)(a0 - u.a_al);
        u.n = S.valid[e] ? n : 0;
        u.G = (u.head + n * dtype + 15) >> 4;
        u.nt = nt;
        u.ncar = S.ncar[e];
        u.ncode = S.ncode[e];
        u.eq0 = u.eq1 = -1;                             // no edge table: the correlator finds the edges' start samples itself
        if (u.n > 0) {
            // what the correlator's scans rely on: carrier pieces start at sample 0 and at increasing samples, code pieces
            // are non-empty, contiguous and cover the nt replica positions
            bool bad = u.ncar < 1 || u.ncode < 1;
            if (!bad) {
                if (lane == 0) bad = sk0[0] != 0;
                if (lane + 1 < u.ncar) bad = bad || !(sk0[lane] < sk0[lane + 1]);
                if (lane < u.ncode) {
                    const int je = scode[lane].j0 + scode[lane].cnt;
                    bad = bad || scode[lane].cnt <= 0 || je != (lane + 1 < u.ncode ? scode[lane + 1].j0 : nt);
                }
            }
            if (__any(bad)) {
                if (lane == 0) atomicAdd(nco_overflow, 1);
                u.n = 0;
                u.ncar = u.ncode = 0;
            }
        } else {
            u.ncar = u.ncode = 0;
        }
        if (u.n > 0) {
            GcUnitSegs *gs = segs + ui;
            if (lane < u.ncar) {
                gs->carK0[lane] = sk0[lane];
                GcCarSeg cs;
                cs.fx = scar[lane].fx;
                cs.dfx = scar[lane].dfx;
                gs->car[lane] = cs;
            }
            if (lane < u.ncode) {
                GcCodeSeg sg;
                sg.y0 = scode[lane].y0; sg.d = scode[lane].d; sg.inv = scode[lane].inv; sg.ylast = scode[lane].ylast;
                sg.j0 = scode[lane].j0; sg.cnt = scode[lane].cnt; sg.w = scode[lane].w; sg.pad = 0;
                gs->code[lane] = sg;
            }
            // rounds: four per correlator workgroup, one per wavefront (lane = round)
            const int g0 = lane * rgrp;
            if (lane < 4 * nseg && g0 < u.G) {
                const unsigned short *rank = (const unsigned short *)(c.code + 1024);
                const int kl = (g0 * 16 - u.head) / dtype;
                const int kfirst = kl > 0 ? kl : 0;
                const int kend = (kl + rsamp < n ? kl + rsamp : n);
                int wa = 0, wb = 0, hint = 0;
                const int ma = lds_code_chip_at(scode, u.ncode, kfirst, &wa, &hint);
                const int mb = lds_code_chip_at(scode, u.ncode, kend - 1 + 2 * c.smax, &wb, nullptr);
                GcRound ro;
                ro.q0 = wa * c.nedge + (int)rank[ma];
                ro.q1 = wb * c.nedge + (int)rank[mb];
                ro.clast = (short)c.code[mb];
                ro.w0 = (short)wa;
                ro.hint = hint;
                rounds[ui * nseg * 4 + lane] = ro;
            }
        }
        if (lane == 0) {
            unit[ui] = u;
            gnsscorr_trklog_t *lg = log + (size_t)ch * nper + p;
            lg->buffloc = buffloc;
            lg->currnsamp = n;
            lg->remcode = S.remcode[e + 1];
            lg->remcarr = S.remcarr[e + 1];
            nsamp_out[(size_t)ch * nper + p] = n;
        }
    }
    GC_TSTAMP(6);       // checks, tables out, rounds
    if (want > 0) {
        m.k = k;
        m.kcap = kcap;
        m.pbase = m.done;
        m.done += k;
    }
    // the run is over for this channel once nothing is planned and nothing waits to be closed
    if (m.k == 0 && (m.done >= nper || S.starved || !plan)) m.finished = 1;
    if (tid == 0) {
        GcTrkState st = S.st;
        st.remcode = S.remcode[k];
        st.remcarr = S.remcarr[k];
        st.buffloc = S.buffloc[k];
        state[ch] = st;
        meta[ch] = m;
        ndone[ch] = m.consumed;
        if (hostflags) {
            if (m.finished && !was_finished) __hip_atomic_fetch_add(&hostflags[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (lp->flagsync) __hip_atomic_store(&hostflags[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(loop + ch);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&S.lp);
        for (int i = tid; i < (int)(sizeof(gnsscorr_loop_t) / 8); i += 64 * GC_TAIL_NW) dst[i] = src[i];
    }
    GC_TSTAMP(7);       // state out
}

// (channel, period of its interval, quarter) -> one workgroup: ps_unit on four rounds of the period, one per wavefront
template <int DTYPE, int NTAP, int NIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NTAP <= 7 ? 4 : (NTAP <= 13 ? 3 : (NTAP <= 21 ? 2 : 1)), 8)))
void trk_step_corr_kernel(const GcChan *__restrict__ chan, const GcStepMeta *__restrict__ meta, const GcTrkUnit *__restrict__ unit,
                          const GcUnitSegs *__restrict__ segs, const GcRound *__restrict__ rounds, int *__restrict__ partial,
                          int nch, int kcap, int nseg, int ntap_stride, int max_n)
{
    using L = PsLayout<DTYPE, NIT>;
    __shared__ __attribute__((aligned(16))) char smem[L::bytes(NTAP)];
    // block -> (period, round, channel), channel fastest: the channels of one period read the same IF window
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int ch = b % nch, seg = (b / nch) % nseg, e = b / (nch * nseg);
    if (e >= kcap) return;
    const GcChan &c = chan[ch];
    if (c.dtype != DTYPE || c.ntap > NTAP) return;
    if (e >= meta[ch].k) return;
    const size_t ui = (size_t)ch * kcap + e;
    const GcTrkUnit u = unit[ui];
    ps_unit<DTYPE, NTAP, NIT>(c, u, segs + ui, rounds + (ui * nseg + seg) * 4, partial + (ui * nseg + seg) * 2 * ntap_stride,
                              ntap_stride, max_n, 4, seg, 0, smem, tid, nullptr);
}

template <int DTYPE>
int launch_step_corr(hipStream_t st, const GcChan *chan, const GcStepMeta *meta, const GcTrkUnit *unit, const GcUnitSegs *segs,
                     const GcRound *rounds, int *partial, int nch, int kcap, int nseg, int ntap, int max_n)
{
    constexpr int NIT = DTYPE == 1 ? 1 : 2;
    const unsigned grid = (unsigned)(nch * nseg * kcap);
#define GC_SC(N) do { hipLaunchKernelGGL((trk_step_corr_kernel<DTYPE, N, NIT>), dim3(grid), dim3(256), 0, st, chan, meta, unit, segs, \
                                         rounds, partial, nch, kcap, nseg, ntap, max_n); \
                      GC_HIP(hipGetLastError()); return 0; } while (0)
    if (ntap <= 3) GC_SC(3);
    if (ntap <= 5) GC_SC(5);
    if (ntap <= 7) GC_SC(7);
    if (ntap <= 13) GC_SC(13);
    if (ntap <= 21) GC_SC(21);
    GC_SC(33);
#undef GC_SC
}

}  // namespace

// workgroups per period in step mode: four rounds (one per wavefront) each
int gc_step_nseg(int dtype, int max_n)
{
    const int nit = trk_ps_nit(dtype, 2);
    const int groups = (15 + max_n * dtype + 15) / 16 + 1;
    return (groups + 256 * nit - 1) / (256 * nit);
}

int gc_launch_step_tail(hipStream_t st, const GcChan *chan, GcTrkState *state, gnsscorr_loop_t *loop, GcStepMeta *meta,
                        const uint64_t *wrpos, const int *partial, GcTrkUnit *unit, GcUnitSegs *segs, GcRound *rounds,
                        double *corrI, double *corrQ, int *nsamp_out, gnsscorr_trklog_t *log, int *ndone, int *nco_overflow,
                        unsigned *hostflags, int nch, int nper, int nseg, int ntap, int max_n, int kcap, int plan)
{
    if (kcap < 1 || kcap > GC_STEP_KMAX) return gc_fail(GNSSCORR_EINVAL, "trk_step: %d periods per step (1..%d)", kcap, GC_STEP_KMAX);
    hipLaunchKernelGGL(trk_step_tail_kernel, dim3(nch), dim3(64 * GC_TAIL_NW), 0, st, chan, state, loop, meta, wrpos, partial, unit, segs, rounds,
                       corrI, corrQ, nsamp_out, log, ndone, nco_overflow, hostflags, nch, nper, nseg, ntap, max_n, kcap, plan);
    GC_HIP(hipGetLastError());
    return 0;
}

// one launch per dtype present among the channels
int gc_launch_step_corr(hipStream_t st, const GcChan *chan, const GcStepMeta *meta, const GcTrkUnit *unit, const GcUnitSegs *segs,
                        const GcRound *rounds, int *partial, int nch, int kcap, int nseg, int dtype, int ntap, int max_n,
                        int smax_max)
{
    if (smax_max > 64) return gc_fail(GNSSCORR_EINVAL, "trk_step: tap offset %d samples (<= 64 supported)", smax_max);
    if (dtype == 2) return launch_step_corr<2>(st, chan, meta, unit, segs, rounds, partial, nch, kcap, nseg, ntap, max_n);
    if (dtype == 1) return launch_step_corr<1>(st, chan, meta, unit, segs, rounds, partial, nch, kcap, nseg, ntap, max_n);
    return gc_fail(GNSSCORR_EINVAL, "trk_step: dtype %d not 1 or 2", dtype);
}
