/*
 * gnsscorr.h -- C ABI of libgnsscorr.so, the MI355X (gfx950) correlation
 * engine behind GNSS-SDRLIB's acquisition / tracking entry points.
 *
 * Two layers are exported:
 *
 *  1. The reference's own symbols (sdracquisition, sdrtracking, correlator,
 *     pcorrelator, checkacquisition, ... : see sdr_compat.h).  They keep the
 *     reference signatures and operate on the reference's sdrch_t, one channel
 *     and one code period per call, exactly like src/sdracq.c / src/sdrtrk.c.
 *
 *  2. The batched, device-resident interface below.  It is what the per-call
 *     symbols are built on and what a scheduler that wants throughput calls:
 *     the IF sample ring lives in HBM, every channel of an epoch batch runs in
 *     one launch, and results stay on the device until fetched.
 *
 * Plain C types only: no HIP or torch types appear in any signature; device
 * pointers and streams cross as void*.
 *
 * Each entry point cites the reference interface it replaces as
 * "ref <file>:<line>" (paths relative to the reference root).
 *
 * All functions return 0 on success and a negative GNSSCORR_E* code on
 * failure; gnsscorr_last_error() gives the message.  Like the reference
 * (src/sdrcmn.c:697-702) a failing call leaves its outputs untouched.
 */
#ifndef GNSSCORR_H
#define GNSSCORR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNSSCORR_OK         0
#define GNSSCORR_EINVAL    -1   /* bad argument / unsupported shape          */
#define GNSSCORR_EHIP      -2   /* HIP runtime error (no device, OOM, ...)   */
#define GNSSCORR_ESTATE    -3   /* call order (no ring, no channels, ...)    */

#define GNSSCORR_MAXTAPS   33   /* 1 + 2*corrn, corrn <= 16                  */
#define GNSSCORR_MAXFREQ   256  /* Doppler bins per channel                  */

typedef struct gnsscorr_ctx gnsscorr_ctx;

const char *gnsscorr_last_error(void);
int  gnsscorr_device_count(void);

/* One context = one GPU + one HIP stream.  stream == NULL creates a private
 * stream; otherwise `stream` is a hipStream_t owned by the caller. */
int  gnsscorr_create(gnsscorr_ctx **ctx, int device, void *stream);
void gnsscorr_destroy(gnsscorr_ctx *ctx);
void *gnsscorr_stream(gnsscorr_ctx *ctx);
int  gnsscorr_sync(gnsscorr_ctx *ctx);

/* ---- IF sample ring in HBM -------------------------------------------------
 * Replaces sdrstat.buff / buff2 and file_getbuff() (ref src/sdrrcv.c:505-532,
 * ring size ref src/sdr.h:134,137).  Sample index s of front end `ftype`
 * lives at byte dtype*(s % ringlen); the write position is the reference's
 * sdrstat.fendbuffsize*sdrstat.buffcnt.
 * devmem == NULL allocates dtype*ringlen bytes with hipMalloc; otherwise the
 * caller's device buffer is used (e.g. the tensor an RCCL broadcast lands
 * in).  dtype*ringlen must be a multiple of 16. */
int  gnsscorr_ring_create(gnsscorr_ctx *ctx, int ftype, int dtype,
                          uint64_t ringlen, void *devmem);
/* append nsamp samples from host memory (what file_pushtomembuf's fread
 * delivers, ref src/sdrrcv.c:469-495) and advance the write position.  A chunk
 * may not exceed the ring; chunks of any size are staged piecewise. */
int  gnsscorr_ring_push(gnsscorr_ctx *ctx, int ftype, const void *host,
                        uint64_t nsamp);
/* The same for a front end's packed byte stream, expanded on the device on its way into the ring(s):
 *   GNSSCORR_FMT_STEREO  NSL Stereo: one byte per sample instant, bits 7-6 = front end 1 (real, {-3,-1,+1,+3}),
 *                        bits 5-3 / 2-0 = front end 2 I / Q ({+1,+3,+5,+7,-7,-5,-3,-1}) -- ref
 *                        src/rcv/stereo/stereo.c:160-205.  Feeds ring 1 (dtype 1) and ring 2 (dtype 2), whichever
 *                        exist, nsamp samples each.
 *   GNSSCORR_FMT_RTLSDR  RTL-SDR: unsigned 8-bit I, Q pairs, (char)(value - 127.5) -- ref
 *                        src/rcv/rtlsdr/rtlsdr.c:136-143.  Feeds ring 1 (dtype 2) from 2*nsamp bytes.
 * Both push calls return as soon as the host buffer may be reused; the transfer runs on a copy stream of
 * the context's own behind pinned staging buffers and never waits for (or stalls) the compute stream --
 * tracking / acquisition calls issued afterwards are ordered behind it. */
#define GNSSCORR_FMT_STEREO 1
#define GNSSCORR_FMT_RTLSDR 2
int  gnsscorr_ring_push_packed(gnsscorr_ctx *ctx, int format, const void *host,
                               uint64_t nsamp);
/* rcvgetbuff() on the HBM ring (ref src/sdrrcv.c:406-463,505-532): n samples from sample index buffloc,
 * wrapped like file_getbuff(); synchronises */
int  gnsscorr_ring_read(gnsscorr_ctx *ctx, int ftype, uint64_t buffloc, int n,
                        void *host);
/* the ring memory was filled by someone else (RCCL, a kernel): advance only */
int  gnsscorr_ring_commit(gnsscorr_ctx *ctx, int ftype, uint64_t nsamp);
uint64_t gnsscorr_ring_wrpos(gnsscorr_ctx *ctx, int ftype);
void *gnsscorr_ring_devptr(gnsscorr_ctx *ctx, int ftype);

/* ---- channels ---------------------------------------------------------------
 * The constants initsdrch() derives (ref src/sdrinit.c:583-657). */
typedef struct {
    int    prn, ctype;
    int    dtype, ftype;        /* ref sdrch_t.dtype / .ftype                */
    int    clen, nsamp, nsampchip;
    double f_sf, f_if, foffset; /* Hz                                        */
    double crate, ctime, ti;
    const short *code;          /* clen chips, +-1 (ref sdrch_t.code)        */
    int    intg;                /* ref sdracq_t.intg                         */
    int    nfreq;               /* ref sdracq_t.nfreq (<= GNSSCORR_MAXFREQ)  */
    const double *freq;         /* ref sdracq_t.freq                         */
    int    nfft;                /* ref sdracq_t.nfft (= 2*nsamp)             */
    int    corrn;               /* ref sdrtrk_t.corrn                        */
    const int *corrp;           /* ref sdrtrk_t.corrp                        */
} gnsscorr_chan_t;

int  gnsscorr_set_channels(gnsscorr_ctx *ctx, int nch,
                           const gnsscorr_chan_t *ch);
int  gnsscorr_num_channels(gnsscorr_ctx *ctx);

/* ---- tracking: E/P/L correlators + carrier wipe-off ------------------------
 * One (channel, epoch) unit is one call of the reference's correlator()
 * (ref src/sdrcmn.c:687-722) as driven by sdrtracking() (ref
 * src/sdrtrk.c:31-43): currnsamp from remcode/codefreq, carrier phase and
 * code phase continued from the previous epoch. */
typedef struct {
    double   carrfreq, codefreq;    /* ref sdrtrk_t.carrfreq / .codefreq     */
    double   remcode, remcarr;      /* ref sdrtrk_t.remcode / .remcarr       */
    uint64_t buffloc;               /* sample index of the next code period  */
} gnsscorr_trkstate_t;

int  gnsscorr_trk_set_state(gnsscorr_ctx *ctx, int ch0, int nch,
                            const gnsscorr_trkstate_t *st);
int  gnsscorr_trk_get_state(gnsscorr_ctx *ctx, int ch0, int nch,
                            gnsscorr_trkstate_t *st);
/* Correlate `nepoch` consecutive code periods of every channel with the
 * frequencies held (as between two loop-filter updates, ref
 * src/sdrmain.c:272-302).  Asynchronous; advances the device-resident state.
 * The correlator launches go to the context's stream back to back; the
 * planner (next batch) and the conversion of the partial sums into the
 * result arrays run on streams of the context's own.  gnsscorr_sync,
 * gnsscorr_trk_fetch*, gnsscorr_trk_get_state and gnsscorr_trk_devptrs order
 * the caller behind them. */
int  gnsscorr_trk_run(gnsscorr_ctx *ctx, int nepoch);
/* Results of the last gnsscorr_trk_run, [nch][nepoch][1+2*corrn] each, in the
 * reference's tap order {P,E1,L1,E2,L2,...}.  trkII / trkQQ are what
 * sdrtracking() leaves in sdr->trk.II / sdr->trk.QQ (II = sum dataQ*code/32,
 * QQ = sum dataI*code/32: the reference's swapped hand-over, ref
 * src/sdrtrk.c:42).  nsamp_out[nch][nepoch] = currnsamp of each period.
 * Any pointer may be NULL.  Synchronises the stream. */
int  gnsscorr_trk_fetch(gnsscorr_ctx *ctx, double *trkII, double *trkQQ,
                        int *nsamp_out);
/* cumsumcorr() over the epochs of the last run (ref src/sdrtrk.c:64-76):
 * sumI/sumQ [nch][1+2*corrn] */
int  gnsscorr_trk_fetch_sums(gnsscorr_ctx *ctx, double *sumI, double *sumQ);
/* device pointers to the result arrays of the last run (layout as fetch);
 * work queued on the context's stream after this call sees the results of
 * every gnsscorr_trk_run issued before it */
int  gnsscorr_trk_devptrs(gnsscorr_ctx *ctx, void **trkII, void **trkQQ);

/* ---- tracking, closed loop: cumsumcorr() + pll() + dll() on the device -------
 * What sdrthread() does around sdrtracking() every code period (ref
 * src/sdrmain.c:264-312): accumulate the correlator outputs (ref
 * src/sdrtrk.c:64-76), run the loop filters -- every period with prm1 until the
 * nav bit is synchronised, then whenever checkbit() raises swloop (every loopms
 * periods counted from the bit edge, ref src/sdrnav.c:241-262) with prm2 -- and
 * clear the sums after each filter update.  Bit synchronisation (checksync(),
 * ref src/sdrnav.c:198-233) and the bit decisions of checkbit() run on the
 * device too, so a channel goes acquisition -> loop every period -> flagsync ->
 * loop every loopms periods without the host.  The loop state lives on the
 * device next to the NCO state; a run of N periods needs no host round trip:
 * it is a chain of launches -- per filter interval one that closes the interval
 * (sums, nav bit, filters) and plans the next, and one that correlates it. */
typedef struct {
    double acqfreq;                         /* ref sdracq_t.acqfreq                   */
    double f_if, foffset, f_cf, crate, ctime;   /* ref sdrch_t                        */
    double pllaw[2], pllw2[2], fllw[2];     /* ref sdrtrkprm_t of prm1 [0], prm2 [1]  */
    double dllaw[2], dllw2[2];
    int    ne, nl;                          /* ref sdrtrk_t.ne / .nl                  */
    int    loopms;                          /* ref sdrtrk_t.loopms                    */
    int    rate;                            /* ref sdrnav_t.rate                      */
    int    flagsync, synci;                 /* ref sdrnav_t.flagsync / .synci         */
    int    navcnt, swloop;                  /* ref sdrnav_t.cnt / .swloop (checkbit)  */
    uint64_t cnt;                           /* ref sdrthread's cnt: periods tracked   */
    double carrNco, codeNco, carrErr, codeErr, freqErr;     /* ref sdrtrk_t           */
    double II[GNSSCORR_MAXTAPS], QQ[GNSSCORR_MAXTAPS];      /* ref sdrtrk_t (all 8)   */
    double oldI[GNSSCORR_MAXTAPS], oldQ[GNSSCORR_MAXTAPS];
    double sumI[GNSSCORR_MAXTAPS], sumQ[GNSSCORR_MAXTAPS];
    double oldsumI[GNSSCORR_MAXTAPS], oldsumQ[GNSSCORR_MAXTAPS];
    /* navigation bit synchronisation: the part of sdrnavigation() that schedules the loops (ref
     * src/sdrnav.c:18-36: biti, checksync() :198-233, checkbit() :241-282), run on the device after
     * every period's correlator like the reference does from sdrtracking() (ref src/sdrtrk.c:46) */
    int    prn;                             /* ref sdrnav_t.sdreph.prn (= the channel's PRN, src/sdrinit.c:506):
                                               checksync() takes its sign shift-register branch for prn > 5 (:203) */
    int    biti;                            /* ref sdrnav_t.biti                      */
    int    bit;                             /* ref sdrnav_t.bit: last decided bit, +-1 */
    int    swsync, swreset;                 /* ref sdrnav_t.swsync / .swreset         */
    int    flagpol;                         /* ref sdrnav_t.flagpol (the frame decoder's; 0 unless the caller sets it) */
    double bitIP;                           /* ref sdrnav_t.bitIP                     */
    int    bitsync[20];                     /* ref sdrnav_t.bitsync[rate], rate <= 20 */
} gnsscorr_loop_t;

int  gnsscorr_loop_set(gnsscorr_ctx *ctx, int ch0, int nch, const gnsscorr_loop_t *lp);
int  gnsscorr_loop_get(gnsscorr_ctx *ctx, int ch0, int nch, gnsscorr_loop_t *lp);

/* One row per code period, the columns writelog() prints (ref src/sdrout.c:412-437) */
typedef struct {
    double carrfreq, codefreq;              /* after this period's filter update      */
    double carrErr, codeErr, carrNco, codeNco, freqErr;
    double remcode, remcarr;                /* after this period                      */
    uint64_t buffloc;                       /* first sample of this period            */
    int    currnsamp;
    int    flagloopfilter;                  /* 0 none, 1 prm1, 2 prm2                 */
    int    flagsync;                        /* ref sdrnav_t.flagsync after this period */
    int    navbit;                          /* +-1: checkbit() decided a bit in this period (swsync), else 0 */
} gnsscorr_trklog_t;

/* Observables on the batched outputs: setobsdata() (ref src/sdrtrk.c:160-209) replayed over the log of a closed-loop
 * run.  The reference calls it from sdrthread() after every prm2 filter update (ref src/sdrmain.c:279-288) with
 * snrflag every SNSMOOTHMS = 100 ms; everything it reads is in the log row of that period, the row before it
 * (oldremcode) and the prompt sums of the interval (sumI[0] before clearcumsumcorr).  Plain host code, no device.
 * State carried from call to call (a channel's log may be replayed in pieces): */
typedef struct {
    /* constants of the channel: ref sdrch_t.f_sf / .f_if / .foffset / .ctime, sdrtrk_t.loopms */
    double f_sf, f_if, foffset, ctime;
    int    loopms;
    /* from the frame decoder (ref sdrnav_t.flagsyncf / .polarity / .firstsftow / .firstsfcnt); zero until it sets them */
    int    flagsyncf, polarity;
    double firstsftow;
    uint64_t firstsfcnt;
    /* running state: ref sdrtrk_t.L[0] / .Isum / .flagremcarradd / .flagpolarityadd, sdrthread's loopcnt, the
     * interval's sumI[0] so far, the remcode of the period before the next one (sdrtrk_t.oldremcode) */
    double L, Isum, sumI0, oldremcode;
    int    flagremcarradd, flagpolarityadd;
    uint64_t loopcnt;
} gnsscorr_obs_t;
/* one row per call of setobsdata(): element [0] of sdrtrk_t.tow / codei / cntout / remcout / L / D after it,
 * and S / codeisum when the call computed them (snr != 0) */
typedef struct {
    double tow, remcout, L, D, S;
    uint64_t codei, cntout;
    int    snr, pad;
} gnsscorr_obsrow_t;
/* log[nper]: rows of one channel (gnsscorr_trk_fetch_log); II0[nper]: that channel's sdrtrk_t.II[0] per period
 * (gnsscorr_trk_fetch's II, tap 0); cnt0: sdrthread's cnt at log[0].  Writes at most max_out rows, returns their
 * number (or GNSSCORR_EINVAL). */
int  gnsscorr_obs_replay(gnsscorr_obs_t *st, const gnsscorr_trklog_t *log, const double *II0, int nper, uint64_t cnt0,
                         gnsscorr_obsrow_t *out, int max_out);

/* Frame synchronisation on the batched nav bits (GPS / QZSS L1 C/A): what sdrnavigation() does behind checkbit()
 * (ref src/sdrnav.c:41-82) -- the last 302 decided bits, the preamble search with its parity check over the ten words
 * (ref :373-411, :325-346, src/sdrnav_gps.c:141-164), and of the subframe decoder the subframe number and the time of
 * week in the hand-over word (ref src/sdrnav_gps.c:123-135,170-190) -- replayed over the `navbit` column of a
 * closed-loop log.  It yields what setobsdata() needs from the frame decoder: flagsyncf, polarity, firstsfcnt,
 * firstsftow.  Ephemeris decoding is not part of this library.  Plain host code. */
typedef struct {
    int    fbits[302];                      /* ref sdrnav_t.fbits (flen 300 + addflen 2), newest last     */
    int    polarity, flagsyncf, flagtow, flagdec;   /* ref sdrnav_t                                        */
    int    sfid;                            /* subframe number decoded last (1..5; other: not a subframe) */
    int    pad;
    uint64_t firstsf, firstsfcnt;           /* ref sdrnav_t: sample / period counter of the frame's end   */
    double firstsftow, tow_gpst;            /* ref sdrnav_t.firstsftow, sdreph_t.tow_gpst                 */
} gnsscorr_frame_t;
/* log[nper]: one channel's rows; cnt0: sdrthread's cnt at log[0].  Returns 0, or GNSSCORR_EINVAL. */
int  gnsscorr_frame_replay(gnsscorr_frame_t *st, const gnsscorr_trklog_t *log, int nper, uint64_t cnt0);

/* Track `nperiod` code periods of every channel closed loop.  A channel stops early
 * where sdrtracking() would find no data yet (ref src/sdrtrk.c:26-30: bufflocnow
 * <= buffloc).  Returns when the last launches are queued (it keeps at most a few
 * filter intervals of launches ahead of the device); results by gnsscorr_trk_fetch
 * (II/QQ per period, periods not run are zero with nsamp_out 0) and
 * gnsscorr_trk_fetch_log. */
int  gnsscorr_trk_run_loop(gnsscorr_ctx *ctx, int nperiod);
/* log[nch][nperiod] of the last gnsscorr_trk_run_loop; ndone[nch] = periods run */
int  gnsscorr_trk_fetch_log(gnsscorr_ctx *ctx, gnsscorr_trklog_t *log, int *ndone);

/* ---- acquisition: parallel code phase search --------------------------------
 * For every channel: up to `intg` iterations of pcorrelator() (ref
 * src/sdrcmn.c:738-773) over the channel's Doppler grid, accumulated
 * non-coherently, with checkacquisition() (ref src/sdracq.c:71-95) evaluated
 * after each iteration, as sdracquisition() does (ref src/sdracq.c:24-43). */
typedef struct {
    int      acqcodei, freqi;       /* ref sdracq_t                           */
    double   acqfreq, cn0, peakr;
    int      flagacq;               /* ref sdrch_t.flagacq                    */
    int      iters;                 /* iterations the reference would run     */
    uint64_t buffloc;               /* return value of sdracquisition()       */
} gnsscorr_acqres_t;

/* wrpos == 0: use each ring's current write position.  Asynchronous. */
int  gnsscorr_acq_run(gnsscorr_ctx *ctx, uint64_t wrpos);
int  gnsscorr_acq_fetch(gnsscorr_ctx *ctx, gnsscorr_acqres_t *res);
/* Device-side hand-over of the last gnsscorr_acq_run to tracking: every acquired
 * channel gets the state sdracquisition() leaves behind (ref src/sdracq.c:51-55:
 * carrfreq = acqfreq, codefreq = crate, remcode = remcarr = 0, buffloc = the
 * returned sample index); channels not acquired keep theirs.  Asynchronous. */
int  gnsscorr_trk_start_from_acq(gnsscorr_ctx *ctx);
/* The reference's `power` array for one channel: nfreq*nsamp doubles,
 * accumulated over res.iters iterations (re-runs the search for that channel
 * with the iteration count of the last gnsscorr_acq_run). */
int  gnsscorr_acq_power(gnsscorr_ctx *ctx, int ch, double *power);

/* ---- op-level device entry points (used by the per-call symbols and tests) --
 * 16384-point complex FFT batches on device memory, unnormalised, sign -1
 * forward / +1 backward; in/out are device pointers to float2[batch][16384] */
int  gnsscorr_fft16k(gnsscorr_ctx *ctx, const void *in, void *out, int sign,
                     int batch);
/* cpxpspec (ref src/sdrcmn.c:261-276) for n = 16384 or 32768 on host data */
int  gnsscorr_pspec(gnsscorr_ctx *ctx, const float *cpx, int n, int flagsum,
                    double *pspec);

/* per-kernel launch timing: enable, run, then read the accumulated HIP-event
 * time of the named kernel ("trk_corr", "trk_plan", "trk_spec", "trk_expand",
 * "trk_finish", "acq_fwd", "acq_corr", "acq_code", "acq_final").
 * on = 1: every kernel; on = 2: only the two correlator kernels ("trk_corr",
 * "acq_corr"), leaving the planner and finish streams free of events; 0: off */
int  gnsscorr_timing_enable(gnsscorr_ctx *ctx, int on);
int  gnsscorr_timing_read(gnsscorr_ctx *ctx, const char *kernel,
                          double *total_ms, int *launches);
int  gnsscorr_timing_reset(gnsscorr_ctx *ctx);

/* the process-wide context the per-call reference symbols use (device 0 or
 * $GNSSCORR_DEVICE); created on first use */
gnsscorr_ctx *gnsscorr_default_ctx(void);

#ifdef __cplusplus
}
#endif
#endif
