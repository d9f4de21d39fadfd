// gnsscorr_internal.h -- device-side data layout shared by the tracking and
// acquisition kernels and the C-ABI layer.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gnsscorr.h"
#include "gnsscorr_nco.h"

#define GC_DPI      (2.0*3.1415926535897932)  // DPI with the reference's PI literal (ref src/sdr.h:103-104)
#define GC_CDIV     32
#define GC_RCPAD    16                   // guard chips either side of the resampled code in LDS

// Pointers read out of a struct are generic to the compiler, which then emits flat_load: those
// count on the LDS counter too, so every LDS wait would also wait for HBM.  Ring and code-pool
// pointers are always device global memory: say so.
typedef const __attribute__((address_space(1))) int8_t *gc_gptr_i8;
typedef unsigned gc_u4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) gc_u4v *gc_gptr_u4;

// Per-channel constants (HBM, one entry per channel).
struct GcChan {
    const int8_t *ring;      // IF ring of the channel's front end
    uint64_t ringlen;        // samples
    const int8_t *code;      // code block in the code pool: chips, edge ranks, edge list (gc_build_codeblock)
    int    dtype, clen, nsamp, nsampchip;
    int    ntap, smax;
    int    tapoff[GNSSCORR_MAXTAPS];   // tap offsets in samples: 0,-s0,+s0,-s1,+s1,...
    double ti, f_sf, crate, ctime;
    // acquisition
    int    nfreq, intg, nfft, grid;    // grid = index of the (ring, freq grid) group
    int    freq_off;                   // offset into the frequency pool
    int    nedge;                      // chip edges per code period (entries of the edge list)
    int    pm1, pad1;                  // pm1: every edge steps by +-2 (a +-1 code)
};

// Code block = GC_CODEBLOCK bytes per channel:
//   [0, 1024)     int8 chips (clen of them, zero padded)
//   [1024, 3072)  uint16 rank[x], x < clen: number of chip edges at chip indices <= x
//   [3072, 7168)  int16 pairs (m, d): edge list sorted by m; d = code[m-1] - code[m] != 0, m-1 cyclic
// A "chip edge" is a chip index at which the periodic code changes value; the prefix-sum
// correlator (gnsscorr_trk.hip) visits only these.
#define GC_CODEBLOCK 7168
static inline void gc_build_codeblock(const short *code, int clen, int8_t *block, int *nedge, int *pm1)
{
    uint16_t *rank = (uint16_t *)(block + 1024);
    int16_t *edge = (int16_t *)(block + 3072);
    for (int i = 0; i < GC_CODEBLOCK; i++) block[i] = 0;
    int ne = 0, all2 = 1;
    for (int m = 0; m < clen; m++) {
        block[m] = (int8_t)code[m];
        const int d = (int)(int8_t)code[(m + clen - 1) % clen] - (int)(int8_t)code[m];
        if (d != 0) {
            edge[2 * ne] = (int16_t)m;
            edge[2 * ne + 1] = (int16_t)d;
            ne++;
            if (d != 2 && d != -2) all2 = 0;
        }
        rank[m] = (uint16_t)ne;
    }
    *nedge = ne;
    *pm1 = all2;
}

// Tracking state carried from epoch to epoch (ref sdrtrk_t, src/sdr.h:371-381).
struct GcTrkState {
    double   carrfreq, codefreq, remcode, remcarr;
    uint64_t buffloc;
};

// One (channel, epoch) work unit produced by the planner.
struct GcTrkPlan {
    uint64_t buffloc;   // first sample of the period
    double   coff;      // code phase at buffloc (chips)   = oldremcode
    double   phi0;      // carrier phase at buffloc (rad)  = oldremcarr
    double   carrfreq;  // Hz, held over the batch
    double   codefreq;  // chip/s, held over the batch
    int      n;         // currnsamp
    int      pad;
};

// Per-unit constants derived from the plan entry (one per channel and epoch).
struct GcTrkUnit {
    uint64_t a_al;      // 16-byte aligned ring byte offset of the period's first sample group
    int      head;      // bytes between a_al and the first sample
    int      n;         // currnsamp; 0 = nothing to correlate (outside the reference's scratch, an
                        // undefined chip step, or an NCO table that overflowed -- counted in nco_overflow)
    int      G;         // 16-byte groups covering the period
    int      nt;        // replica length n + 2*smax
    int      ncar;      // pieces of the carrier table
    int      ncode;     // pieces of the code table
    int      eq0, eq1;  // chip edges [eq0, eq1) the period's rounds can touch (their start samples: trk_edges); eq0 < 0: no table
};
#define GC_EDGTAB 704   // start samples (uint16) per unit in the edge table: one code period of edges and some

// The unit's two NCOs as piecewise-linear tables (gnsscorr_nco.h): every LUT index and every chip
// the kernels derive from them equals what the reference's sample-by-sample fp64 sums give.
#define GC_NCAR  40     // carrier pieces per code period (one per binade visited: <= ~37 for any phase / frequency below Nyquist)
#define GC_NCODE 24     // code pieces per call (<= ~2.5 code periods: sdrtracking() asks for one)
struct GcUnitSegs {
    int       carK0[GC_NCAR];      // first sample of carrier piece i
    GcCarSeg  car[GC_NCAR];
    GcCodeSeg code[GC_NCODE];
};

// Row statistics of the accumulated power after one acquisition iteration.
struct GcAcqRow {
    double rowmax;      // max over lags (first index on ties)
    double sum_out;     // sum outside the +-2 chip window around argmax
    double max_out;     // maxvd() outside the window (element 0 always a candidate)
    int    argmax;
    int    pad;
};

#define GC_HIP(call)                                                           \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) return gc_fail_hip(e_, #call, __FILE__, __LINE__); \
    } while (0)

int gc_fail_hip(hipError_t e, const char *what, const char *file, int line);
int gc_fail(int code, const char *fmt, ...);

// One round of the prefix-sum correlator (gnsscorr_trk.hip): the chip edges [q0, q1) its samples can
// touch, in the numbering period * nedge + list index, and the value of its last chip.
#define GC_MAXR 24
struct GcRound {
    int q0, q1;         // edges [q0, q1)
    short clast, w0;    // value of the round's last chip; whole code periods in front of edge q0
    int hint;           // code piece that holds the round's first replica position
};

// kernel launchers (definitions in gnsscorr_trk.hip / gnsscorr_plan.hip / gnsscorr_acq.hip)
size_t gc_trk_spec_ints(size_t units);
int gc_launch_trk_spec(hipStream_t st, const GcChan *chan, const GcTrkState *state_in, int nch, int nepoch, int *claims,
                       int e_off);
int gc_launch_trk_plan(hipStream_t st, const GcChan *chan, const GcTrkState *state_in, GcTrkState *state_out,
                       GcTrkPlan *plan, int nch, int nepoch, int *claims);
// nco_overflow: device counter of units whose NCO tables did not fit (their outputs are zero)
int gc_launch_trk_expand(hipStream_t st, const GcChan *chan, const GcTrkPlan *plan, GcTrkUnit *unit, GcUnitSegs *segs,
                         int *nsamp_out, int nch, int nepoch, GcRound *rounds, int nseg, int max_n, int *nco_overflow);
int gc_launch_trk_edges(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, const GcUnitSegs *segs, unsigned short *etab,
                        int nch, int nepoch);
int gc_launch_trk_corr(hipStream_t st, const GcChan *chan, const GcTrkUnit *unit, const GcUnitSegs *segs,
                       const GcRound *rounds, int *partial, int nch,
                       int nepoch, int nseg, int ntap_stride, int dtype, int ntap, int max_n, int smax_max,
                       const unsigned short *etab);
// scratch: GC_FINISH_SCRATCH 64-bit words per channel, zero before the first launch (the kernel leaves them zero)
#define GC_FINISH_SCRATCH (2 * GNSSCORR_MAXTAPS + 1)
int gc_launch_trk_finish(hipStream_t st, const int *partial, double *corrI, double *corrQ, double *sumI,
                         double *sumQ, unsigned long long *scratch, int nch, int nepoch, int nseg, int ntap);
int gc_trk_nseg(int dtype, int max_n);
int gc_launch_trk_ringcheck(hipStream_t st, const GcChan *chan, const GcTrkPlan *plan, const int8_t *ring0, uint64_t wrpos0,
                            uint64_t wrpos1, int nch, int nepoch, int *viol);

// ---- closed loop in steps (gnsscorr_loop.hip) ---------------------------------------------------------------
// One step = one filter interval per channel: the periods up to and including the next one after which pll()/dll()
// run (1 before the nav bit is synchronised; afterwards up to loopms, counted from the bit edge).
#define GC_STEP_KMAX 20         // periods per step and channel, at most
struct GcStepMeta {             // per channel, device resident
    int k;                      // periods planned for the correlator launch that follows (0: none)
    int kcap;                   // unit stride of that plan (the kcap of the launch that planned it)
    int pbase;                  // index, within the run, of the interval's first period
    int done;                   // periods planned so far in this run
    int consumed;               // periods closed (sums, nav bit, filters, log) so far in this run
    int finished;               // the run is over for this channel (all periods closed, or the ring has no more data)
    int early;                  // (diagnostic) a filter update fell inside an interval: must stay 0
    int pad;
};
int gc_step_nseg(int dtype, int max_n);          // workgroups (one round each) per period
// closes the intervals the previous correlator launch produced and plans the next ones (plan = 0: closes only)
int gc_launch_step_tail(hipStream_t st, const GcChan *chan, GcTrkState *state, gnsscorr_loop_t *loop, GcStepMeta *meta,
                        const uint64_t *wrpos, const int *partial, GcTrkUnit *unit, GcUnitSegs *segs, GcRound *rounds,
                        double *corrI, double *corrQ, int *nsamp_out, gnsscorr_trklog_t *log, int *ndone, int *nco_overflow,
                        unsigned *hostflags, int nch, int nper, int nseg, int ntap, int max_n, int kcap, int plan);
int gc_launch_step_corr(hipStream_t st, const GcChan *chan, const GcStepMeta *meta, const GcTrkUnit *unit, const GcUnitSegs *segs,
                        const GcRound *rounds, int *partial, int nch, int kcap, int nseg, int dtype, int ntap, int max_n,
                        int smax_max);
