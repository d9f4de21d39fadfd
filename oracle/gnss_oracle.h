/*
 * gnss_oracle.h -- CPU restatement of the GNSS-SDRLIB acquisition + tracking
 * correlation path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 * The product (libgnsscorr.so) never links, loads or calls anything in here.
 *
 * PARITY UNPINNED: the reference (mfkiwl/erlangnetwork-gnsslib-sdr) ships no
 * tests, golden vectors or result fixtures for this path, and it cannot be
 * compiled in this image (every .c file under src/ includes src/sdr.h, which includes
 * fftw3.h, fec.h and libusb-1.0/libusb.h -- none of them installed, and the
 * lib/fftw3 + lib/ka9q-fec submodule directories are empty).  This file is a
 * line-cited restatement read off the reference sources; the only external
 * known answers it is checked against are the IS-GPS-200 C/A "first 10 chips"
 * octal table (tests/golden/ca_first10_octal.json) and numpy's FFT.
 *
 * The two NCOs are restated literally (*_seq: phase / code offset advanced by
 * repeated += in fp64, exactly as the reference loops do); GPU results are
 * compared with them bit for bit.
 */
#ifndef GNSS_ORACLE_H
#define GNSS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_CDIV    32
#define ORC_CSCALE  (1.0/32.0)
#define ORC_PI      3.1415926535897932
#define ORC_DPI     (2.0*ORC_PI)

#define ORC_CTYPE_L1CA   1
#define ORC_CTYPE_G1     20
#define ORC_CTYPE_L1SBAS 27

/* ref src/sdrcode.c:101-154, :426-444, :523-539 */
int orc_gencode(int prn, int ctype, short *code, int *len, double *crate);

/* ref src/sdrcmn.c:633-669 */
void   orc_carrier_lut(short *cost, short *sint);
double orc_mixcarr_seq(const signed char *data, int dtype, double ti, int n,
                       double freq, double phi0, short *I, short *Q);

/* ref src/sdrcmn.c:608-621 */
double orc_rescode_seq(const short *code, int len, double coff, int smax,
                       double ci, int n, short *rcode);

/* ref src/sdrcmn.c:687-722 */
void orc_correlator(const signed char *data, int dtype, double ti, int n,
                    double freq, double phi0, double crate, double coff,
                    const int *s, int ns, double *II, double *QQ,
                    double *remc, double *remp, const short *code, int clen);

/* ref src/sdrcmn.c:185-195 */
void orc_cpxcpx(const short *I, const short *Q, double scale, int n,
                float *cpx);
/* unnormalised DFT of any length, sign -1 forward / +1 backward, float in/out
 * (stands in for FFTW3f: ref src/sdrcmn.c:134-175) */
void orc_fft(float *cpx, int n, int sign);
/* ref src/sdrcmn.c:228-251 */
void orc_cpxconv(float *cpxa, const float *cpxb, int m, int n, int flagsum,
                 double *conv);
/* ref src/sdrcmn.c:261-276 */
void orc_cpxpspec(float *cpx, int n, int flagsum, double *pspec);
/* ref src/sdrinit.c:645-655 : xcode = FFT_nfft(zero-padded resampled code) */
void orc_codespectrum(const short *code, int clen, double ci, int nsamp,
                      int nfft, float *xcode);
/* ref src/sdrcmn.c:738-773 */
void orc_pcorrelator(const signed char *data, int dtype, double ti, int n,
                     const double *freq, int nfreq, double crate, int m,
                     const float *codex, double *P);
/* time-domain evaluation of the same quantity (SURVEY 8a normative form):
 * P[b][k] += |sum_j w[k+j] r[j]|^2/(32 m)^2, lags k in [k0,k1) only */
void orc_pcorrelator_td(const signed char *data, int dtype, double ti, int n,
                        const double *freq, int nfreq, int m,
                        const short *code, int clen, double ci,
                        int k0, int k1, double *P);

/* ref src/sdrcmn.c:461-497, :574-578 */
double orc_maxvd(const double *d, int n, int exinds, int exinde, int *ind);
double orc_meanvd(const double *d, int n, int exinds, int exinde);

typedef struct {
    int    acqcodei;
    int    freqi;
    double acqfreq;
    double cn0;
    double peakr;
    int    acquired;
} orc_acqres_t;
/* ref src/sdracq.c:71-95 */
int orc_checkacquisition(const double *P, int nsamp, int nfreq, int nsampchip,
                         double ctime, const double *freq, orc_acqres_t *res);

/* IF sample ring as in ref src/sdrrcv.c:505-532 (file front end) */
typedef struct {
    const signed char *buff;   /* dtype*ringlen bytes                        */
    uint64_t ringlen;          /* samples in the ring (MEMBUFFLEN*FILE_BUFFSIZE) */
    uint64_t wrpos;            /* fendbuffsize*buffcnt: samples written so far */
} orc_ring_t;
void orc_getbuff(const orc_ring_t *ring, uint64_t buffloc, int n, int dtype,
                 signed char *out);

/* channel description + tracking state used by the driver restatements */
#define ORC_MAXTAPS 33
typedef struct {
    /* constants (ref src/sdrinit.c:583-657) */
    int    dtype, clen, nsamp, nsampchip;
    double f_sf, f_if, foffset, f_cf, crate, ctime, ti, ci;
    short  code[1023];
    /* acquisition (ref src/sdr.h:344-356) */
    int    intg, nfreq, nfft;
    double freq[256];
    const float *xcode;        /* nfft complex floats                        */
    orc_acqres_t acq;
    int    flagacq, flagtrk;
    /* tracking (ref src/sdr.h:371-412) */
    int    corrn, corrp[16], ne, nl, loopms;
    double codefreq, carrfreq, remcode, remcarr, oldremcode, oldremcarr;
    double codeNco, codeErr, carrNco, carrErr, freqErr;
    double II[ORC_MAXTAPS], QQ[ORC_MAXTAPS], oldI[ORC_MAXTAPS],
           oldQ[ORC_MAXTAPS], sumI[ORC_MAXTAPS], sumQ[ORC_MAXTAPS],
           oldsumI[ORC_MAXTAPS], oldsumQ[ORC_MAXTAPS];
    int    currnsamp;
    /* loop parameters (ref src/sdrinit.c:402-425) */
    double dllw2[2], dllaw[2], pllw2[2], pllaw[2], fllw[2];
    /* nav timing that schedules the loop filters (ref src/sdrnav.c:18-31,241-262) + thread counter */
    int    rate, flagsync, synci, navcnt, swloop, flagloopfilter;
    uint64_t cnt;
    /* navigation bit synchronisation and decision (ref sdrnav_t src/sdr.h:441-480; src/sdrnav.c:18-36,198-282);
     * navcnt above is sdrnav_t.cnt */
    int    prn;             /* sdrnav_t.sdreph.prn = the channel's PRN (ref src/sdrinit.c:506) */
    int    biti, bit, swsync, swreset, flagpol;
    double bitIP;
    int    bitsync[20];     /* sdrnav_t.bitsync[rate] */
    /* observables (ref sdrtrk_t tow/codei/cntout/remcout/L/D/S [0], Isum, the two once-only flags; src/sdr.h:384-411)
     * and what the frame decoder hands setobsdata() (ref sdrnav_t firstsftow/firstsfcnt/flagsyncf/polarity) */
    double obs_tow, obs_remcout, obs_L, obs_D, obs_S, obs_Isum;
    uint64_t obs_codei, obs_cntout, obs_codeisum, firstsfcnt, loopcnt;
    double firstsftow;
    int    flagremcarradd, flagpolarityadd, flagsyncf, polarity, obs_n, obs_nsnr;
} orc_chan_t;

/* ref src/sdrinit.c:583-657 (+ :385-394, :402-480); xcode left NULL */
int orc_initchan(orc_chan_t *ch, int prn, int ctype, int dtype, double f_cf,
                 double f_sf, double f_if, int corrn, int corrd, int corrp,
                 const double *dllb, const double *pllb, const double *fllb);
/* ref src/sdracq.c:14-62 (no sleep, no printf); power = nsamp*nfreq zeroed
 * doubles; returns buffloc */
uint64_t orc_sdracquisition(orc_chan_t *ch, const orc_ring_t *ring,
                            double *power, int *iters_done);
/* ref src/sdrtrk.c:15-54 (without sdrnavigation); returns bufflocnow */
uint64_t orc_sdrtracking(orc_chan_t *ch, const orc_ring_t *ring,
                         uint64_t buffloc);
/* ref src/sdrtrk.c:64-86 */
void orc_cumsumcorr(orc_chan_t *ch, int polarity);
void orc_clearcumsumcorr(orc_chan_t *ch);
/* ref src/sdrnav.c:198-233, :241-282 and the part of sdrnavigation() in front of the frame decoder (:18-36) */
int  orc_checksync(double IP, double IPold, orc_chan_t *ch);
int  orc_checkbit(double IP, int loopms, orc_chan_t *ch);
void orc_sdrnavigation_sync(orc_chan_t *ch, uint64_t cnt);
/* ref src/sdrtrk.c:95-150; prm = 0 (before nav sync) or 1 (after) */
void orc_pll(orc_chan_t *ch, int prm, double dt);
void orc_dll(orc_chan_t *ch, int prm, double dt);
/* One pass of sdrthread()'s tracking branch (ref src/sdrmain.c:264-312): sdrtracking incl. the loop
 * timing part of sdrnavigation/checkbit (ref src/sdrnav.c:18-20,241-262; bit sync itself is an input:
 * ch->flagsync/synci), cumsumcorr, pll/dll per the flagsync/swloop cadence, clearcumsumcorr, cnt++,
 * *buffloc += currnsamp.  Returns ch->flagtrk. */
int orc_sdrthread_step(orc_chan_t *ch, const orc_ring_t *ring, uint64_t *buffloc);
/* Frame synchronisation of GPS L1 C/A on the decided nav bits: the part of sdrnavigation() behind checkbit()
 * (ref src/sdrnav.c:41-82) with predecodefec (:288-297: a copy for L1CA), findpreamble (:373-411), paritycheck
 * (:325-346) + paritycheck_l1ca (src/sdrnav_gps.c:141-164) and, of decodenav, the subframe number and the time of
 * week of the hand-over word (decode_l1ca :170-190, decode_frame_l1ca :123-135, tow_gpst = getbitu(buff,30,17)*6). */
typedef struct {
    int fbits[302], fbitsdec[302];          /* ref sdrnav_t.fbits / .fbitsdec, flen 300 + addflen 2 */
    int polarity, flagsyncf, flagtow, flagdec, sfid;
    uint64_t firstsf, firstsfcnt;
    double firstsftow, tow_gpst;
} orc_frame_t;
/* bit: what checkbit() appended to fbits in this period (+-1), or 0 when it decided none (swsync off) */
void orc_navframe_l1ca(orc_frame_t *f, int bit, uint64_t buffloc, uint64_t cnt);

/* ref src/sdrtrk.c:160-209: element [0] of the observable histories after the call (the reference shifts the
 * histories down by one first; element [0] keeps its value across the shift: L accumulates) */
void orc_setobsdata(orc_chan_t *ch, uint64_t buffloc, uint64_t cnt, int snrflag);

/* front-end sample expansion: ref src/rcv/stereo/stereo.c:160-205 (dtype 1: front end 1 from bits 7-6,
 * dtype 2: front end 2 I/Q from bits 5-3 / 2-0) and src/rcv/rtlsdr/rtlsdr.c:136-143 */
void orc_stereo_exp(const unsigned char *buf, int n, int dtype, signed char *expbuf);
void orc_rtlsdr_exp(const unsigned char *buf, int n, signed char *expbuf);

#ifdef __cplusplus
}
#endif
#endif
