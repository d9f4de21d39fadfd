/*
 * sdr_compat.h -- the slice of GNSS-SDRLIB's src/sdr.h that crosses the
 * acquisition / tracking boundary, restated so that libgnsscorr can be linked
 * under the reference's own channel thread without the reference's headers
 * (which need fftw3.h, fec.h, libusb.h).
 *
 * Everything here mirrors a declaration in the reference, cited as
 * "ref <file>:<line>" (paths relative to the reference root).  Layout matters:
 * sdrch_t is shared memory between the reference's sdrthread() and this
 * library, so every struct is restated field for field, in order, with the
 * reference's types (Linux build, no -DENAGLO/-DENAGAL/... as in
 * bin/Makefile:16-30: MAXSAT = 32 GPS + 23 SBAS = 55).
 *
 * The RTKLIB types embedded in sdrnav_t (gtime_t, eph_t, geph_t; RTKLIB 2.4.2
 * p13, BSD-2, ref lib/RTKLIB/src/rtklib.h:475-570) are restated for layout
 * only; this library never reads them.
 */
#ifndef SDR_COMPAT_H
#define SDR_COMPAT_H

#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include <pthread.h>

#ifdef __cplusplus
extern "C" {
#endif

/* constants: ref src/sdr.h:101-207 ------------------------------------------*/
#define PI            3.1415926535897932
#define DPI           (2.0*PI)
#define ON            1
#define OFF           0
#define FEND_FILE     10
#define FTYPE1        1
#define FTYPE2        2
#define DTYPEI        1
#define DTYPEIQ       2
#define MEMBUFFLEN    5000      /* ref src/sdr.h:134 */
#define FILE_BUFFSIZE 65536     /* ref src/sdr.h:137 */
#define ACQINTG_L1CA  10        /* ref src/sdr.h:141-148 */
#define ACQINTG_G1    10
#define ACQINTG_SBAS  10
#define ACQHBAND      7000
#define ACQSTEP       200
#define ACQTH         3.0
#define ACQSLEEP      2000
#define LOOP_L1CA     10        /* ref src/sdr.h:152-154 */
#define LOOP_G1       10
#define LOOP_SBAS     2
#define OBSINTERPN    80        /* ref src/sdr.h:197 */
#define LENSBASMSG    32        /* ref src/sdr.h:238-239 */
#define LENSBASNOV    80
#define CTYPE_L1CA    1         /* ref src/sdr.h:205-212 */
#define CTYPE_L1SBAS  27
#define CTYPE_G1      20
#define SYS_GPS       0x01      /* ref lib/RTKLIB/src/rtklib.h:96-98 */
#define SYS_SBS       0x02
#define SYS_GLO       0x04
#define MAXSAT        55        /* ref lib/RTKLIB/src/rtklib.h:122-192 */
#define FREQ1_GLO     1.60200E9 /* ref lib/RTKLIB/src/rtklib.h:79-80 */
#define DFRQ1_GLO     0.56250E6
#define CDIV          32        /* ref src/sdrcmn.c:9-11 */
#define CMASK         0x1F
#define CSCALE        (1.0/32.0)

/* ref src/sdr.h:255-269 (pthread flavour) */
#define thread_t      pthread_t
#define mlock_t       pthread_mutex_t
#define mlock(f)      pthread_mutex_lock(&f)
#define unmlock(f)    pthread_mutex_unlock(&f)

/* ref src/sdr.h:275: typedef fftwf_complex cpx_t (= float[2]) */
typedef float cpx_t[2];

/* RTKLIB layout stand-ins: ref lib/RTKLIB/src/rtklib.h:475-478, :536-570 */
typedef struct { time_t time; double sec; } gtime_t;
typedef struct {
    int sat, iode, iodc, sva, svh, week, code, flag;
    gtime_t toe, toc, ttr;
    double A, e, i0, OMG0, omg, M0, deln, OMGd, idot;
    double crc, crs, cuc, cus, cic, cis;
    double toes, fit, f0, f1, f2, tgd[4], Adot, ndot;
} eph_t;
typedef struct {
    int sat, iode, frq, svh, sva, age;
    gtime_t toe, tof;
    double pos[3], vel[3], acc[3], taun, gamn, dtaun;
} geph_t;

/* ref src/sdr.h:278-317 */
typedef struct {
    int fend;
    double f_cf[2], f_sf[2], f_if[2];
    int dtype[2];
    FILE *fp1, *fp2;
    char file1[1024], file2[1024];
    int useif1, useif2;
    int nch, nchL1, nchL2, nchL5, nchL6;
    int prn[MAXSAT], sys[MAXSAT], ctype[MAXSAT], ftype[MAXSAT];
    int pltacq, plttrk, pltspec, outms, rinex, rtcm, sbas, log;
    char rinexpath[1024];
    int rtcmport, sbasport;
    int trkcorrn, trkcorrd, trkcorrp;
    double trkdllb[2], trkpllb[2], trkfllb[2];
    int rtlsdrppmerr;
} sdrini_t;

/* ref src/sdr.h:320-329 */
typedef struct {
    int stopflag, specflag, buffsize, fendbuffsize;
    unsigned char *buff, *buff2, *tmpbuff;
    uint64_t buffcnt;
} sdrstat_t;

/* ref src/sdr.h:344-356 */
typedef struct {
    int intg;
    double hband, step;
    int nfreq;
    double *freq;
    int acqcodei, freqi;
    double acqfreq;
    int nfft;
    double cn0, peakr;
} sdracq_t;

/* ref src/sdr.h:359-368 */
typedef struct {
    double pllb, dllb, fllb, dllw2, dllaw, pllw2, pllaw, fllw;
} sdrtrkprm_t;

/* ref src/sdr.h:371-412 */
typedef struct {
    double codefreq, carrfreq, remcode, remcarr, oldremcode, oldremcarr;
    double codeNco, codeErr, carrNco, carrErr, freqErr;
    uint64_t buffloc;
    double tow[OBSINTERPN];
    uint64_t codei[OBSINTERPN], codeisum[OBSINTERPN], cntout[OBSINTERPN];
    double remcout[OBSINTERPN], L[OBSINTERPN], D[OBSINTERPN], S[OBSINTERPN];
    double *II, *QQ, *oldI, *oldQ, *sumI, *sumQ, *oldsumI, *oldsumQ;
    double Isum;
    int loop, loopms, flagpolarityadd, flagremcarradd, flagloopfilter;
    int corrn;
    int *corrp;
    double *corrx;
    int ne, nl;
    sdrtrkprm_t prm1, prm2;
} sdrtrk_t;

/* ref src/sdr.h:415-433 */
typedef struct {
    eph_t eph;
    geph_t geph;
    int ctype;
    double tow_gpst;
    int week_gpst, cnt, cntth, update, prn;
    int tk[3], nt, n4, s1cnt;
    double toc_gst;
    int week_gst;
    unsigned int toe1, toe2;
    int toe_bds, f1p3, cucp4, ep5, cicp6, i0p7, OMGdp8, omgp9;
    unsigned int f1p4, cucp5, ep6, cicp7, i0p8, OMGdp9, omgp10;
} sdreph_t;

/* ref src/sdr.h:436-442 */
typedef struct {
    unsigned char msg[LENSBASMSG], novatelmsg[LENSBASNOV];
    int id, week;
    double tow;
} sdrsbas_t;

/* ref src/sdr.h:445-479 */
typedef struct {
    FILE *fpnav;
    int ctype, rate, flen, addflen;
    int prebits[32];
    int prelen, bit, biti, cnt;
    double bitIP;
    int *fbits, *fbitsdec;
    int update;
    int *bitsync;
    int synci;
    uint64_t firstsf, firstsfcnt;
    double firstsftow;
    int polarity, flagpol;
    void *fec;
    short *ocode;
    int ocodei, swsync, swreset, swloop;
    int flagsync, flagsyncf, flagtow, flagdec;
    sdreph_t sdreph;
    sdrsbas_t sbas;
} sdrnav_t;

/* ref src/sdr.h:482-511 */
typedef struct {
    thread_t hsdr;
    int no, sat, sys, prn;
    char satstr[5];
    int ctype, dtype, ftype;
    double f_cf, f_sf, f_if, foffset;
    short *code;
    cpx_t *xcode;
    int clen;
    double crate, ctime, ti, ci;
    int nsamp, currnsamp, nsampchip;
    sdracq_t acq;
    sdrtrk_t trk;
    sdrnav_t nav;
    int flagacq, flagtrk;
} sdrch_t;

/* globals the hot path reads: ref src/sdr.h:571-585, defined by the
 * reference in src/sdrmain.c:14-29.  libgnsscorr carries weak definitions so
 * that it also works stand-alone; when linked into the reference, the
 * reference's strong definitions win. */
extern mlock_t hbuffmtx, hreadmtx, hfftmtx, hobsmtx;
extern sdrini_t  sdrini;
extern sdrstat_t sdrstat;

/* ---- primary entry points: ref src/sdr.h:610-620 ------------------------- */
extern uint64_t sdracquisition(sdrch_t *sdr, double *power);
extern int checkacquisition(double *P, sdrch_t *sdr);
extern uint64_t sdrtracking(sdrch_t *sdr, uint64_t buffloc, uint64_t cnt);
extern void cumsumcorr(sdrtrk_t *trk, int polarity);
extern void setobsdata(sdrch_t *sdr, uint64_t buffloc, uint64_t cnt, sdrtrk_t *trk, int snrflag);      /* ref src/sdr.h: setobsdata, src/sdrtrk.c:160-209 */
extern void clearcumsumcorr(sdrtrk_t *trk);
extern void pll(sdrch_t *sdr, sdrtrkprm_t *prm, double dt);
extern void dll(sdrch_t *sdr, sdrtrkprm_t *prm, double dt);

/* ---- channel set-up: ref src/sdr.h:624-638 -------------------------------- */
extern int readinifile(sdrini_t *ini);
extern int chk_initvalue(sdrini_t *ini);
extern void initacqstruct(int sys, int ctype, int prn, sdracq_t *acq);
extern void inittrkprmstruct(sdrtrk_t *trk);
extern int inittrkstruct(int sat, int ctype, double ctime, sdrtrk_t *trk);
extern int initsdrch(int chno, int sys, int prn, int ctype, int dtype,
                     int ftype, double f_cf, double f_sf, double f_if,
                     sdrch_t *sdr);
extern void freesdrch(sdrch_t *sdr);

/* ---- op-level seam: ref src/sdr.h:652-688, :691 --------------------------- */
extern void cpxcpx(const short *II, const short *QQ, double scale, int n,
                   cpx_t *cpx);
/* the fftwf_plan arguments of the reference are opaque here and ignored */
extern void cpxfft(void *plan, cpx_t *cpx, int n);
extern void cpxifft(void *plan, cpx_t *cpx, int n);
extern void cpxconv(void *plan, void *iplan, cpx_t *cpxa, cpx_t *cpxb, int m,
                    int n, int flagsum, double *conv);
extern void cpxpspec(void *plan, cpx_t *cpx, int n, int flagsum, double *pspec);
extern double mixcarr(const char *data, int dtype, double ti, int n,
                      double freq, double phi0, short *II, short *QQ);
extern double rescode(const short *code, int len, double coff, int smax,
                      double ci, int n, short *rcode);
extern void pcorrelator(const char *data, int dtype, double ti, int n,
                        double *freq, int nfreq, double crate, int m,
                        cpx_t *codex, double *P);
extern void correlator(const char *data, int dtype, double ti, int n,
                       double freq, double phi0, double crate, double coff,
                       int *s, int ns, double *II, double *QQ, double *remc,
                       double *remp, short *codein, int coden);
extern double maxvd(const double *data, int n, int exinds, int exinde, int *ind);
extern double meanvd(const double *data, int n, int exinds, int exinde);
extern void ind2sub(int ind, int nx, int ny, int *subx, int *suby);
extern short *gencode(int prn, int ctype, int *len, double *crate);

/* ---- sample ring: ref src/sdr.h (sdrrcv.c section), src/sdrrcv.c:406-532 -- */
extern int rcvgetbuff(sdrini_t *ini, uint64_t buffloc, int n, int ftype,
                      int dtype, char *expbuf);
extern void file_pushtomembuf(void);
extern void file_getbuff(uint64_t buffloc, int n, int ftype, int dtype,
                         char *expbuf);

/* called by sdrtracking() after each correlation (ref src/sdrtrk.c:46);
 * navigation decoding is outside this library: the weak default is a no-op
 * and the reference's src/sdrnav.c:15 overrides it when linked in. */
extern void sdrnavigation(sdrch_t *sdr, uint64_t buffloc, uint64_t cnt);

#ifdef __cplusplus
}
#endif
#endif
