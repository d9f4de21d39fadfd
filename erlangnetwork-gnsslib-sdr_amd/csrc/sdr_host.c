/*
 * sdr_host.c -- the C host side of the acquisition / tracking boundary:
 * configuration (gnss-sdrcli.ini + front-end INI), PRN code generation,
 * channel set-up, loop filters and the host mirror of the IF sample ring.
 *
 * These are the scalar, once-per-channel / once-per-loop-update parts of the
 * reference that surround the correlation kernels; the correlation itself is
 * in the HIP files.  Each function cites the reference routine whose
 * behaviour it reproduces ("ref <file>:<line>", reference root relative).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sdr_compat.h"
#include "../../include/gnsscorr.h"

#define SDRPRINTF printf

/* ---- globals (weak: the reference's src/sdrmain.c:14-29 wins when linked) -- */
__attribute__((weak)) mlock_t hbuffmtx = PTHREAD_MUTEX_INITIALIZER;
__attribute__((weak)) mlock_t hreadmtx = PTHREAD_MUTEX_INITIALIZER;
__attribute__((weak)) mlock_t hfftmtx  = PTHREAD_MUTEX_INITIALIZER;
__attribute__((weak)) mlock_t hobsmtx  = PTHREAD_MUTEX_INITIALIZER;
__attribute__((weak)) sdrini_t  sdrini;
__attribute__((weak)) sdrstat_t sdrstat;

/* navigation decoding is outside this library (ref src/sdrnav.c:15) */
__attribute__((weak)) void sdrnavigation(sdrch_t *sdr, uint64_t buffloc, uint64_t cnt)
{
    (void)sdr; (void)buffloc; (void)cnt;
}

/* ------------------------------------------------------------------------- */
/* INI files: ref src/sdrinit.c:17-46 (value lookup), :106-211 (keys)        */
/* ------------------------------------------------------------------------- */

/* First "key = value" of [sec] wins; ';' starts a comment; the key is
 * right-trimmed of blanks and tabs; the value keeps its leading blanks and
 * loses only trailing CR/LF.  Missing file, section or key yields `def`. */
static void ini_get(const char *file, const char *sec, const char *key, const char *def, char *out,
                    int len)
{
    FILE *fp;
    char line[1024];
    int inside = 0;
    strncpy(out, def, (size_t)len - 1);
    out[len - 1] = '\0';
    if (!(fp = fopen(file, "r"))) {
        fprintf(stderr, "ini file open error [%s]\n", file);
        return;
    }
    while (fgets(line, sizeof(line), fp)) {
        char *semi = strchr(line, ';'), *lb, *rb, *eq;
        if (semi) *semi = '\0';
        lb = strchr(line, '[');
        rb = lb ? strchr(lb + 1, ']') : NULL;
        if (lb && rb) {
            *rb = '\0';
            inside = strcmp(lb + 1, sec) == 0;
            continue;
        }
        if (!inside || !(eq = strchr(line, '='))) continue;
        *eq = '\0';
        for (char *q = eq - 1; q >= line && (*q == ' ' || *q == '\t'); q--) *q = '\0';
        if (strcmp(line, key)) continue;
        for (char *q = eq + strlen(eq + 1); q > eq && (*q == '\r' || *q == '\n'); q--) *q = '\0';
        strncpy(out, eq + 1, (size_t)len - 1);
        out[len - 1] = '\0';
        break;
    }
    fclose(fp);
}

static int ini_int(const char *f, const char *s, const char *k)
{
    char v[256];
    ini_get(f, s, k, "", v, sizeof(v));
    return atoi(v);
}

static double ini_double(const char *f, const char *s, const char *k)
{
    char v[256];
    ini_get(f, s, k, "", v, sizeof(v));
    return atof(v);
}

static int ini_ints(const char *f, const char *s, const char *k, int *out, int n)
{
    char v[256], *tok, *save = NULL;
    int i;
    ini_get(f, s, k, "", v, sizeof(v));
    for (i = 0; i < n; i++) {
        if (!(tok = strtok_r(i ? NULL : v, ",", &save))) return -1;
        out[i] = atoi(tok);
    }
    return 0;
}

static int file_exists(const char *f)
{
    FILE *fp = fopen(f, "r");
    if (!fp) return 0;
    fclose(fp);
    return 1;
}

/* Same keys and error behaviour as readinifile() (ref src/sdrinit.c:106-211)
 * for an explicit path of the receiver INI.  Only the FILE front end maps to
 * a front-end code this library drives; the hardware front ends are parsed
 * and reported as in the reference but cannot be started from here. */
int readinifile_at(sdrini_t *ini, const char *inifile)
{
    static const struct { const char *name; int id; } fends[] = {
        {"STEREO", 0}, {"GN3SV2", 1}, {"GN3SV3", 2}, {"RTLSDR", 3}, {"BLADERF", 4},
        {"FILESTEREO", 5}, {"FILEGN3SV2", 6}, {"FILEGN3SV3", 7}, {"FILERTLSDR", 8},
        {"FILEBLADERF", 9}, {"FILE", FEND_FILE}};
    char fendfile[256], str[256];
    int i, found = 0;

    if (!file_exists(inifile)) {
        SDRPRINTF("error: gnss-sdrcli.ini doesn't exist\n");
        return -1;
    }
    ini_get(inifile, "RCV", "FENDCONF", "", fendfile, sizeof(fendfile));
    if (!file_exists(fendfile)) {
        SDRPRINTF("error: %s doesn't exist\n", fendfile);
        return -1;
    }
    ini_get(fendfile, "FEND", "TYPE", "", str, sizeof(str));
    for (i = 0; i < (int)(sizeof(fends) / sizeof(fends[0])); i++)
        if (!strcmp(str, fends[i].name)) { ini->fend = fends[i].id; found = 1; }
    if (!found) {
        SDRPRINTF("error: wrong frontend type: %s\n", str);
        return -1;
    }
    if (ini->fend == FEND_FILE || (ini->fend >= 5 && ini->fend <= 9)) {
        ini_get(fendfile, "FEND", "FILE1", "", ini->file1, 256);
        if (strcmp(ini->file1, "")) ini->useif1 = ON;
    }
    if (ini->fend == FEND_FILE) {
        ini_get(fendfile, "FEND", "FILE2", "", ini->file2, 256);
        if (strcmp(ini->file2, "")) ini->useif2 = ON;
    }
    ini->f_cf[0] = ini_double(fendfile, "FEND", "CF1");
    ini->f_sf[0] = ini_double(fendfile, "FEND", "SF1");
    ini->f_if[0] = ini_double(fendfile, "FEND", "IF1");
    ini->dtype[0] = ini_int(fendfile, "FEND", "DTYPE1");
    ini->f_cf[1] = ini_double(fendfile, "FEND", "CF2");
    ini->f_sf[1] = ini_double(fendfile, "FEND", "SF2");
    ini->f_if[1] = ini_double(fendfile, "FEND", "IF2");
    ini->dtype[1] = ini_int(fendfile, "FEND", "DTYPE2");
    ini->rtlsdrppmerr = ini_int(fendfile, "FEND", "PPMERR");

    ini->trkcorrn = ini_int(fendfile, "TRACK", "CORRN");
    ini->trkcorrd = ini_int(fendfile, "TRACK", "CORRD");
    ini->trkcorrp = ini_int(fendfile, "TRACK", "CORRP");
    ini->trkdllb[0] = ini_double(fendfile, "TRACK", "DLLB1");
    ini->trkpllb[0] = ini_double(fendfile, "TRACK", "PLLB1");
    ini->trkfllb[0] = ini_double(fendfile, "TRACK", "FLLB1");
    ini->trkdllb[1] = ini_double(fendfile, "TRACK", "DLLB2");
    ini->trkpllb[1] = ini_double(fendfile, "TRACK", "PLLB2");
    ini->trkfllb[1] = ini_double(fendfile, "TRACK", "FLLB2");

    ini->nch = ini_int(inifile, "CHANNEL", "NCH");
    if (ini->nch < 1 || ini->nch > MAXSAT) {
        SDRPRINTF("error: wrong inifile value NCH=%d\n", ini->nch);
        return -1;
    }
    if (ini_ints(inifile, "CHANNEL", "PRN", ini->prn, ini->nch) < 0 ||
        ini_ints(inifile, "CHANNEL", "SYS", ini->sys, ini->nch) < 0 ||
        ini_ints(inifile, "CHANNEL", "CTYPE", ini->ctype, ini->nch) < 0 ||
        ini_ints(inifile, "CHANNEL", "FTYPE", ini->ftype, ini->nch) < 0) {
        SDRPRINTF("error: wrong inifile value NCH=%d\n", ini->nch);
        return -1;
    }
    ini->pltacq = ini_int(inifile, "PLOT", "ACQ");
    ini->plttrk = ini_int(inifile, "PLOT", "TRK");
    ini->outms = ini_int(inifile, "OUTPUT", "OUTMS");
    ini->rinex = ini_int(inifile, "OUTPUT", "RINEX");
    ini->rtcm = ini_int(inifile, "OUTPUT", "RTCM");
    ini->sbas = ini_int(inifile, "OUTPUT", "SBAS");
    ini->log = ini_int(inifile, "OUTPUT", "LOG");
    ini_get(inifile, "OUTPUT", "RINEXPATH", "", ini->rinexpath, 256);
    ini->rtcmport = ini_int(inifile, "OUTPUT", "RTCMPORT");
    ini->sbasport = ini_int(inifile, "OUTPUT", "SBASPORT");
    ini->pltspec = ini_int(inifile, "SPECTRUM", "SPEC");

    ini->nchL1 = 0;
    for (i = 0; i < ini->nch; i++)
        if (ini->ctype[i] == CTYPE_L1CA || ini->ctype[i] == CTYPE_G1) ini->nchL1++;
    return 0;
}

/* ref src/sdrinit.c:106: the receiver INI is ./gnss-sdrcli.ini of the cwd */
int readinifile(sdrini_t *ini) { return readinifile_at(ini, "./gnss-sdrcli.ini"); }

/* ref src/sdrinit.c:217-271: same checks, same order, same messages */
int chk_initvalue(sdrini_t *ini)
{
    if (ini->f_sf[0] <= 0 || ini->f_sf[0] > 100e6 || ini->f_if[0] < 0 || ini->f_if[0] > 100e6) {
        SDRPRINTF("error: wrong freq. input sf1: %.0f if1: %.0f\n", ini->f_sf[0], ini->f_if[0]);
        return -1;
    }
    if (ini->useif2 || ini->fend == 0 /* FEND_STEREO */) {
        if (ini->f_sf[1] <= 0 || ini->f_sf[1] > 100e6 || ini->f_if[1] < 0 || ini->f_if[1] > 100e6) {
            SDRPRINTF("error: wrong freq. input sf2: %.0f if2: %.0f\n", ini->f_sf[1], ini->f_if[1]);
            return -1;
        }
    }
    if (ini->rtcmport < 0 || ini->rtcmport > 32767) {
        SDRPRINTF("error: wrong rtcm port rtcm:%d\n", ini->rtcmport);
        return -1;
    }
    if (ini->fend == FEND_FILE || (ini->fend >= 5 && ini->fend <= 9)) {
        if (ini->useif1 && !file_exists(ini->file1)) {
            SDRPRINTF("error: file1 doesn't exist: %s\n", ini->file1);
            return -1;
        }
        if (ini->useif2 && !file_exists(ini->file2)) {
            SDRPRINTF("error: file2 doesn't exist: %s\n", ini->file2);
            return -1;
        }
        if (!ini->useif1 && !ini->useif2) {
            SDRPRINTF("error: file1 or file2 are not selected\n");
            return -1;
        }
    }
    if (ini->rinex && !file_exists(ini->rinexpath)) {
        SDRPRINTF("error: rinex output directory doesn't exist: %s\n", ini->rinexpath);
        return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* PRN codes: ref src/sdrcode.c:101-154 (C/A), :426-444 (GLONASS), :523-539  */
/* ------------------------------------------------------------------------- */

/* G2 phase selection as a delay in chips, PRN 1..210 (ref src/sdrcode.c:103-125) */
static const unsigned short g2_delay[210] = {
    5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471, 472,
    473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862, 863, 950, 947, 948, 950, 67, 103, 91,
    19, 679, 225, 625, 946, 638, 161, 1001, 554, 280, 710, 709, 775, 864, 558, 220, 397, 55, 898, 759,
    367, 299, 1018, 729, 695, 780, 801, 788, 732, 34, 320, 327, 389, 407, 525, 405, 221, 761, 260, 326,
    955, 653, 699, 422, 188, 438, 959, 539, 879, 677, 586, 153, 792, 814, 446, 264, 1015, 278, 536, 819,
    156, 957, 159, 712, 885, 461, 248, 713, 126, 807, 279, 122, 197, 693, 632, 771, 467, 647, 203, 145,
    175, 52, 21, 237, 235, 886, 657, 634, 762, 355, 1012, 176, 603, 130, 359, 595, 68, 386, 797, 456,
    499, 883, 307, 127, 211, 121, 118, 163, 628, 853, 484, 289, 811, 202, 1021, 463, 568, 904, 670, 230,
    911, 684, 309, 644, 932, 12, 314, 891, 212, 185, 675, 503, 150, 395, 345, 846, 798, 992, 357, 995,
    877, 112, 144, 476, 193, 109, 445, 291, 87, 399, 292, 901, 339, 208, 711, 189, 263, 537, 663, 942,
    173, 900, 30, 500, 935, 556, 373, 85, 652, 310};

/* Fibonacci LFSR stepped `n` times: out[i] = bit of stage `outstage`
 * (1-based) before the shift, feedback = XOR of the stages in `taps`. */
static void lfsr_run(int nstage, unsigned taps, int outstage, int n, unsigned char *out)
{
    unsigned reg = (1u << nstage) - 1, mask = reg;
    for (int i = 0; i < n; i++) {
        out[i] = (reg >> (outstage - 1)) & 1;
        reg = ((reg << 1) | (unsigned)__builtin_parity(reg & taps)) & mask;
    }
}

static short *gen_ca(int prn, int *len, double *crate)
{
    unsigned char g1[1023], g2[1023];
    short *code;
    if (prn < 1 || prn > 210 || !(code = (short *)malloc(sizeof(short) * 1023))) return NULL;
    lfsr_run(10, (1u << 2) | (1u << 9), 10, 1023, g1);                                         /* 1+x^3+x^10 */
    lfsr_run(10, (1u << 1) | (1u << 2) | (1u << 5) | (1u << 7) | (1u << 8) | (1u << 9), 10, 1023, g2);
    for (int i = 0; i < 1023; i++)
        code[i] = (g1[i] ^ g2[(i + 1023 - g2_delay[prn - 1]) % 1023]) ? 1 : -1;
    *len = 1023;
    *crate = 1.023e6;
    return code;
}

static short *gen_glo(int *len, double *crate)
{
    unsigned char b[511];
    short *code = (short *)malloc(sizeof(short) * 511);
    if (!code) return NULL;
    lfsr_run(9, (1u << 4) | (1u << 8), 7, 511, b);                                             /* 1+x^5+x^9, 7th stage */
    for (int i = 0; i < 511; i++) code[i] = b[i] ? 1 : -1;
    *len = 511;
    *crate = 0.511e6;
    return code;
}

/* ref src/sdrcode.c:523-539.  CTYPE_G1 is accepted here although the
 * reference's switch lacks the case (its generator at :426-444 is
 * unreachable, so initsdrch() fails for every GLONASS channel there). */
short *gencode(int prn, int ctype, int *len, double *crate)
{
    switch (ctype) {
    case CTYPE_L1CA:
    case CTYPE_L1SBAS: return gen_ca(prn, len, crate);
    case CTYPE_G1:     return gen_glo(len, crate);
    default:
        SDRPRINTF("error: gencode prn:%d ctype:%d", prn, ctype);
        return NULL;
    }
}

/* ------------------------------------------------------------------------- */
/* channel set-up: ref src/sdrinit.c:385-480, :583-686                       */
/* ------------------------------------------------------------------------- */

/* RTKLIB satno()/satno2id() for the systems the reference build enables
 * (GPS + SBAS; ref lib/RTKLIB/src/rtkcmn.c satno, satno2id) */
static int sat_number(int sys, int prn)
{
    if (sys == SYS_GPS && prn >= 1 && prn <= 32) return prn;
    if (sys == SYS_SBS && prn >= 120 && prn <= 142) return 32 + prn - 120 + 1;
    return 0;
}

static void sat_id(int sys, int prn, char *id)
{
    if (sys == SYS_GPS && prn >= 1 && prn <= 32) snprintf(id, 5, "G%02d", prn);
    else if (sys == SYS_SBS && prn >= 120 && prn <= 142) snprintf(id, 5, "%03d", prn);
    else id[0] = '\0';
}

void initacqstruct(int sys, int ctype, int prn, sdracq_t *acq)
{
    (void)sys; (void)prn;
    if (ctype == CTYPE_L1CA) acq->intg = ACQINTG_L1CA;
    if (ctype == CTYPE_G1) acq->intg = ACQINTG_G1;
    if (ctype == CTYPE_L1SBAS) acq->intg = ACQINTG_SBAS;
    acq->hband = ACQHBAND;
    acq->step = ACQSTEP;
    acq->nfreq = 2 * (ACQHBAND / ACQSTEP) + 1;
}

static void loop_constants(sdrtrkprm_t *p)
{
    p->dllw2 = (p->dllb / 0.53) * (p->dllb / 0.53);
    p->dllaw = 1.414 * (p->dllb / 0.53);
    p->pllw2 = (p->pllb / 0.53) * (p->pllb / 0.53);
    p->pllaw = 1.414 * (p->pllb / 0.53);
    p->fllw = p->fllb / 0.25;
}

void inittrkprmstruct(sdrtrk_t *trk)
{
    trk->prm1.dllb = sdrini.trkdllb[0]; trk->prm1.pllb = sdrini.trkpllb[0]; trk->prm1.fllb = sdrini.trkfllb[0];
    trk->prm2.dllb = sdrini.trkdllb[1]; trk->prm2.pllb = sdrini.trkpllb[1]; trk->prm2.fllb = sdrini.trkfllb[1];
    loop_constants(&trk->prm1);
    loop_constants(&trk->prm2);
}

int inittrkstruct(int sat, int ctype, double ctime, sdrtrk_t *trk)
{
    const int ntap_cap = 1;   /* arrays are 1+2*corrn long */
    int i, ctimems = (int)(ctime * 1000);
    (void)sat;
    inittrkprmstruct(trk);
    trk->corrn = sdrini.trkcorrn;
    trk->corrp = (int *)malloc(sizeof(int) * (size_t)trk->corrn);
    for (i = 0; i < trk->corrn; i++) {
        trk->corrp[i] = sdrini.trkcorrd * (i + 1);
        if (trk->corrp[i] == sdrini.trkcorrp) {
            trk->ne = 2 * (i + 1) - 1;
            trk->nl = 2 * (i + 1);
        }
    }
    trk->corrx = (double *)calloc((size_t)(2 * trk->corrn + ntap_cap), sizeof(double));
    for (i = 1; i <= trk->corrn; i++) {
        trk->corrx[2 * i - 1] = -sdrini.trkcorrd * i;
        trk->corrx[2 * i] = sdrini.trkcorrd * i;
    }
    double **arr[8] = {&trk->II, &trk->QQ, &trk->oldI, &trk->oldQ,
                       &trk->sumI, &trk->sumQ, &trk->oldsumI, &trk->oldsumQ};
    for (i = 0; i < 8; i++) {
        *arr[i] = (double *)calloc((size_t)(1 + 2 * trk->corrn), sizeof(double));
        if (!*arr[i]) {
            SDRPRINTF("error: inittrkstruct memory allocation\n");
            return -1;
        }
    }
    if (ctype == CTYPE_L1CA) trk->loop = LOOP_L1CA;
    if (ctype == CTYPE_G1) trk->loop = LOOP_G1;
    if (ctype == CTYPE_L1SBAS) trk->loop = LOOP_SBAS;
    trk->loopms = trk->loop * ctimems;
    return 0;
}

/* ref src/sdrinit.c:583-657.  sdr->xcode (the reference's FFT of the padded
 * replica at length 2*nsamp) is left NULL: the GPU path derives its own
 * length-32768 spectrum from sdr->code on the device.  Navigation state is
 * reduced to what the channel loop reads (ocode all ones, :520-521). */
int initsdrch(int chno, int sys, int prn, int ctype, int dtype, int ftype, double f_cf, double f_sf,
              double f_if, sdrch_t *sdr)
{
    int i;
    sdr->no = chno; sdr->sys = sys; sdr->prn = prn;
    sdr->sat = sat_number(sys, prn);
    sdr->ctype = ctype; sdr->dtype = dtype; sdr->ftype = ftype;
    sdr->f_sf = f_sf; sdr->f_if = f_if;
    sdr->ti = 1 / f_sf;
    if (!(sdr->code = gencode(prn, ctype, &sdr->clen, &sdr->crate))) {
        SDRPRINTF("error: gencode\n");
        return -1;
    }
    sdr->ci = sdr->ti * sdr->crate;
    sdr->ctime = sdr->clen / sdr->crate;
    sdr->nsamp = (int)(f_sf * sdr->ctime);
    sdr->nsampchip = (int)(sdr->nsamp / sdr->clen);
    sat_id(sys, prn, sdr->satstr);
    if (ctype == CTYPE_G1) {
        snprintf(sdr->satstr, 5, "R%d", prn);
        sdr->f_cf = FREQ1_GLO + DFRQ1_GLO * prn;
        sdr->foffset = DFRQ1_GLO * prn;
    } else {
        sdr->f_cf = f_cf;
        sdr->foffset = 0.0;
    }
    initacqstruct(sys, ctype, prn, &sdr->acq);
    sdr->acq.nfft = 2 * sdr->nsamp;
    if (!(sdr->acq.freq = (double *)malloc(sizeof(double) * (size_t)sdr->acq.nfreq))) {
        SDRPRINTF("error: initsdrch memory alocation\n");
        return -1;
    }
    for (i = 0; i < sdr->acq.nfreq; i++)
        sdr->acq.freq[i] = sdr->f_if + ((i - (sdr->acq.nfreq - 1) / 2) * sdr->acq.step) + sdr->foffset;
    if (inittrkstruct(sdr->sat, ctype, sdr->ctime, &sdr->trk) < 0) return -1;

    sdr->nav.ctype = ctype;
    sdr->nav.rate = ctype == CTYPE_L1CA ? 20 : ctype == CTYPE_G1 ? 10 : 2;
    sdr->nav.ocode = (short *)calloc((size_t)sdr->nav.rate, sizeof(short));
    if (!sdr->nav.ocode) return -1;
    for (i = 0; i < sdr->nav.rate; i++) sdr->nav.ocode[i] = 1;
    sdr->xcode = NULL;
    return 0;
}

/* ref src/sdrinit.c:663-686 */
void freesdrch(sdrch_t *sdr)
{
    free(sdr->code);
    free(sdr->trk.II); free(sdr->trk.QQ); free(sdr->trk.oldI); free(sdr->trk.oldQ);
    free(sdr->trk.sumI); free(sdr->trk.sumQ); free(sdr->trk.oldsumI); free(sdr->trk.oldsumQ);
    free(sdr->trk.corrp); free(sdr->trk.corrx);
    free(sdr->acq.freq);
    free(sdr->nav.ocode);
    memset(sdr, 0, sizeof(*sdr));
}

/* ------------------------------------------------------------------------- */
/* loop side of tracking: ref src/sdrtrk.c:64-150                            */
/* ------------------------------------------------------------------------- */
void cumsumcorr(sdrtrk_t *trk, int polarity)
{
    for (int i = 0; i < 1 + 2 * trk->corrn; i++) {
        trk->II[i] *= polarity;
        trk->QQ[i] *= polarity;
        trk->oldsumI[i] += trk->oldI[i];
        trk->oldsumQ[i] += trk->oldQ[i];
        trk->sumI[i] += trk->II[i];
        trk->sumQ[i] += trk->QQ[i];
    }
}

void clearcumsumcorr(sdrtrk_t *trk)
{
    for (int i = 0; i < 1 + 2 * trk->corrn; i++)
        trk->oldsumI[i] = trk->oldsumQ[i] = trk->sumI[i] = trk->sumQ[i] = 0;
}

/* ------------------------------------------------------------------------- */
/* observables: ref src/sdrtrk.c:160-209                                     */
/* ------------------------------------------------------------------------- */
/* the histories move down by one; entry 0 keeps its value (what the reference's shiftdata() through a temporary
 * amounts to, ref src/sdrcmn.c:587-596): L[0] therefore accumulates from call to call */
static void obs_age_d(double *h) { memmove(h + 1, h, sizeof(double) * (OBSINTERPN - 1)); }
static void obs_age_u(uint64_t *h) { memmove(h + 1, h, sizeof(uint64_t) * (OBSINTERPN - 1)); }

void setobsdata(sdrch_t *sdr, uint64_t buffloc, uint64_t cnt, sdrtrk_t *trk, int snrflag)
{
    obs_age_d(trk->tow); obs_age_d(trk->L); obs_age_d(trk->D);
    obs_age_u(trk->codei); obs_age_u(trk->cntout); obs_age_d(trk->remcout);
    trk->tow[0] = sdr->nav.firstsftow + (double)(cnt - sdr->nav.firstsfcnt) * sdr->ctime;
    trk->codei[0] = buffloc;
    trk->cntout[0] = cnt;
    trk->remcout[0] = trk->oldremcode * sdr->f_sf / trk->codefreq;
    trk->D[0] = -(trk->carrfreq - sdr->f_if - sdr->foffset);                /* Doppler */
    if (!trk->flagremcarradd) {                                             /* carrier phase: the start phase, once */
        trk->L[0] -= trk->remcarr / DPI;
        trk->flagremcarradd = ON;
    }
    if (sdr->nav.flagsyncf && !trk->flagpolarityadd) {                      /* half a cycle for an inverted frame, once */
        if (sdr->nav.polarity == 1) trk->L[0] += 0.5;
        trk->flagpolarityadd = ON;
    }
    trk->L[0] += trk->D[0] * (trk->loopms * sdr->currnsamp / sdr->f_sf);
    trk->Isum += fabs(trk->sumI[0]);
    if (snrflag) {
        obs_age_d(trk->S); obs_age_u(trk->codeisum);
        trk->S[0] = 10 * log(trk->Isum / 100.0 / 100.0) + log(500.0) + 5;
        trk->codeisum[0] = buffloc;
        trk->Isum = 0;
    }
}

/* setobsdata() over the log of a device-resident closed-loop run (include/gnsscorr.h): the same statements in the
 * same order on the same doubles, fed from the log rows instead of sdrtrk_t */
int gnsscorr_obs_replay(gnsscorr_obs_t *st, const gnsscorr_trklog_t *rows, const double *II0, int nper, uint64_t cnt0,
                        gnsscorr_obsrow_t *out, int max_out)
{
    if (!st || !rows || !II0 || nper < 0 || (max_out > 0 && !out) || st->loopms < 1) return GNSSCORR_EINVAL;
    int nout = 0;
    for (int p = 0; p < nper; p++) {
        const gnsscorr_trklog_t *r = rows + p;
        const uint64_t cnt = cnt0 + (uint64_t)p;
        st->sumI0 += II0[p];                                                /* cumsumcorr(): sumI += II (polarity +1) */
        if (r->flagloopfilter == 2) {                                       /* ref src/sdrmain.c:277-288 */
            const int snrflag = st->loopcnt % (uint64_t)(100 / st->loopms) == 0;        /* SNSMOOTHMS = 100, ref src/sdr.h:198 */
            gnsscorr_obsrow_t o;
            memset(&o, 0, sizeof(o));
            o.tow = st->firstsftow + (double)(cnt - st->firstsfcnt) * st->ctime;
            o.codei = r->buffloc;
            o.cntout = cnt;
            o.remcout = st->oldremcode * st->f_sf / r->codefreq;
            o.D = -(r->carrfreq - st->f_if - st->foffset);
            if (!st->flagremcarradd) {
                st->L -= r->remcarr / DPI;
                st->flagremcarradd = 1;
            }
            if (st->flagsyncf && !st->flagpolarityadd) {
                if (st->polarity == 1) st->L += 0.5;
                st->flagpolarityadd = 1;
            }
            st->L += o.D * (st->loopms * r->currnsamp / st->f_sf);
            o.L = st->L;
            st->Isum += fabs(st->sumI0);
            if (snrflag) {
                o.S = 10 * log(st->Isum / 100.0 / 100.0) + log(500.0) + 5;
                o.snr = 1;
                st->Isum = 0;
            }
            st->loopcnt++;
            if (nout < max_out) out[nout] = o;
            nout++;
        }
        if (r->flagloopfilter) st->sumI0 = 0.0;                             /* clearcumsumcorr() */
        st->oldremcode = r->remcode;                                        /* the next period's oldremcode */
    }
    return nout;
}

/* ------------------------------------------------------------------------- */
/* frame synchronisation of L1 C/A on the decided bits: ref src/sdrnav.c:41-82 */
/* ------------------------------------------------------------------------- */
/* One (32,26) word in the reference's +-1 arithmetic (ref src/sdrnav_gps.c:141-164): w[0], w[1] are the last two
 * bits of the word before, w[2..25] the data bits as sent or complemented back, w[26..31] the parity bits.  Each
 * parity bit is the product of the bits its row of the code names. */
static int l1ca_word_ok(const int *w)
{
    static const signed char rows[6][17] = {
        {0, 2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24, -1},
        {1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22, 25, -1},
        {0, 2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23, -1},
        {1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24, -1},
        {1, 2, 4, 6, 7, 8, 10, 11, 15, 16, 17, 18, 19, 22, 23, 25, -1},
        {0, 4, 6, 7, 9, 10, 11, 12, 14, 16, 20, 23, 24, 25, -1}};
    int stat = 0;
    for (int r = 0; r < 6; r++) {
        int prod = 1;
        for (int k = 0; rows[r][k] >= 0; k++) prod *= w[rows[r][k]];
        stat += prod - w[26 + r];               /* (summed, as the reference does: not a per-bit comparison) */
    }
    return stat == 0;
}

/* the ten words of the frame in `raw` under the polarity the preamble gave (ref src/sdrnav.c:325-346) */
static int l1ca_frame_parity(const int *raw, int polarity)
{
    int b[302], good = 0;
    for (int i = 0; i < 302; i++) b[i] = polarity * raw[i];
    for (int wd = 0; wd < 10; wd++) {
        int *w = b + 30 * wd;
        if (w[1] == -1)
            for (int j = 2; j < 26; j++) w[j] = -w[j];
        good += l1ca_word_ok(w);
    }
    return good == 10;
}

int gnsscorr_frame_replay(gnsscorr_frame_t *st, const gnsscorr_trklog_t *rows, int nper, uint64_t cnt0)
{
    static const int preamble[8] = {1, -1, -1, -1, 1, -1, 1, 1};           /* ref src/sdrinit.c:492 */
    if (!st || !rows || nper < 0) return GNSSCORR_EINVAL;
    for (int p = 0; p < nper; p++) {
        const int bit = rows[p].navbit;
        if (!bit) continue;                     /* checkbit() decided no bit in this period: swsync is off */
        const uint64_t cnt = cnt0 + (uint64_t)p;
        memmove(st->fbits, st->fbits + 1, sizeof(int) * 301);
        st->fbits[301] = bit;
        if (!st->flagtow) {                     /* preamble search (no FEC on this signal: the bits as they are) */
            int corr = 0;
            for (int i = 0; i < 8; i++) corr += st->fbits[2 + i] * preamble[i];
            st->flagsyncf = 0;
            if (corr == 8 || corr == -8) {
                st->polarity = corr > 0 ? 1 : -1;
                st->flagsyncf = l1ca_frame_parity(st->fbits, st->polarity);
            }
            if (st->flagsyncf) {
                st->firstsf = rows[p].buffloc;
                st->firstsfcnt = cnt;
                st->flagtow = 1;
            }
        }
        if (st->flagtow && (int)(cnt - st->firstsfcnt) % 6000 == 0) {      /* a whole subframe: 300 bits x 20 periods */
            /* decode_l1ca(): data bits complemented back where the word before ended in a one (-1), then packed with
             * -1 as one (ref src/sdrnav.c:154-171); subframe number = bits 49..51, time of week = bits 30..46 x 6 s */
            int d[302];
            memcpy(d, st->fbits, sizeof(d));
            for (int wd = 0; wd < 10; wd++)
                if (d[30 * wd + 1] == -1)
                    for (int j = 2; j < 26; j++) d[30 * wd + j] = -d[30 * wd + j];
            unsigned id = 0, tow = 0;
            for (int i = 0; i < 3; i++) id = id * 2 + (d[2 + 49 + i] < 0);
            for (int i = 0; i < 17; i++) tow = tow * 2 + (d[2 + 30 + i] < 0);
            st->sfid = (int)id;
            if (id >= 1 && id <= 5) st->tow_gpst = tow * 6.0;
            if (st->tow_gpst == 0) {            /* no time of week: start over (ref src/sdrnav.c:68-71) */
                st->flagsyncf = 0;
                st->flagtow = 0;
            } else if (cnt == st->firstsfcnt) {
                st->flagdec = 1;
                st->firstsftow = st->tow_gpst;
            }
        }
    }
    return GNSSCORR_OK;
}

/* 2nd order PLL assisted by a 1st order FLL */
void pll(sdrch_t *sdr, sdrtrkprm_t *prm, double dt)
{
    sdrtrk_t *t = &sdr->trk;
    const double ip = t->sumI[0], qp = t->sumQ[0], oip = t->oldsumI[0], oqp = t->oldsumQ[0];
    double cerr = ip > 0 ? atan2(qp, ip) / PI : atan2(-qp, -ip) / PI;
    double ferr = (ip == 0 ? PI / 2 : atan(qp / ip)) - (oip == 0 ? PI / 2 : atan(oqp / oip));
    if (ferr > PI / 2) ferr = PI - ferr;
    if (ferr < -PI / 2) ferr = -PI - ferr;
    t->carrNco += prm->pllaw * (cerr - t->carrErr) + prm->pllw2 * dt * cerr + prm->fllw * dt * ferr;
    t->carrfreq = sdr->acq.acqfreq + t->carrNco;
    t->carrErr = cerr;
    t->freqErr = ferr;
}

/* 2nd order DLL with carrier aiding */
void dll(sdrch_t *sdr, sdrtrkprm_t *prm, double dt)
{
    sdrtrk_t *t = &sdr->trk;
    const double ie = t->sumI[t->ne], il = t->sumI[t->nl], qe = t->sumQ[t->ne], ql = t->sumQ[t->nl];
    const double cerr = (sqrt(ie * ie + qe * qe) - sqrt(il * il + ql * ql)) /
                        (sqrt(ie * ie + qe * qe) + sqrt(il * il + ql * ql));
    t->codeNco += prm->dllaw * (cerr - t->codeErr) + prm->dllw2 * dt * cerr;
    t->codefreq = sdr->crate - t->codeNco + (t->carrfreq - sdr->f_if - sdr->foffset) / (sdr->f_cf / sdr->crate);
    t->codeErr = cerr;
}

/* ------------------------------------------------------------------------- */
/* IF sample ring, file front end: ref src/sdrrcv.c:208-218, :406-532        */
/* ------------------------------------------------------------------------- */

/* Allocate the host rings like rcvinit() does for FEND_FILE (ref
 * src/sdrrcv.c:196-218) and mirror them in HBM on the default context. */
int rcvinit_file(sdrini_t *ini)
{
    gnsscorr_ctx *ctx = gnsscorr_default_ctx();
    if (!ctx) return -1;
    sdrstat.fendbuffsize = FILE_BUFFSIZE;
    sdrstat.buffsize = FILE_BUFFSIZE * MEMBUFFLEN;
    sdrstat.buffcnt = 0;
    if (ini->useif1) {
        if (!ini->fp1 && !(ini->fp1 = fopen(ini->file1, "rb"))) {
            SDRPRINTF("error: failed to open file(FILE1): %s\n", ini->file1);
            return -1;
        }
        sdrstat.buff = (unsigned char *)malloc((size_t)ini->dtype[0] * FILE_BUFFSIZE * MEMBUFFLEN);
        if (!sdrstat.buff) { SDRPRINTF("error: failed to allocate memory for the buffer\n"); return -1; }
        if (gnsscorr_ring_create(ctx, FTYPE1, ini->dtype[0], (uint64_t)FILE_BUFFSIZE * MEMBUFFLEN, NULL)) {
            SDRPRINTF("error: %s\n", gnsscorr_last_error());
            return -1;
        }
    }
    if (ini->useif2) {
        if (!ini->fp2 && !(ini->fp2 = fopen(ini->file2, "rb"))) {
            SDRPRINTF("error: failed to open file(FILE2): %s\n", ini->file2);
            return -1;
        }
        sdrstat.buff2 = (unsigned char *)malloc((size_t)ini->dtype[1] * FILE_BUFFSIZE * MEMBUFFLEN);
        if (!sdrstat.buff2) { SDRPRINTF("error: failed to allocate memory for the buffer\n"); return -1; }
        if (gnsscorr_ring_create(ctx, FTYPE2, ini->dtype[1], (uint64_t)FILE_BUFFSIZE * MEMBUFFLEN, NULL)) {
            SDRPRINTF("error: %s\n", gnsscorr_last_error());
            return -1;
        }
    }
    return 0;
}

/* ref src/sdrrcv.c:469-495, plus the H2D append of the same block */
void file_pushtomembuf(void)
{
    gnsscorr_ctx *ctx = gnsscorr_default_ctx();
    size_t nread1 = 0, nread2 = 0;
    unsigned char *b1 = NULL, *b2 = NULL;

    mlock(hbuffmtx);
    if (sdrini.fp1 != NULL) {
        b1 = &sdrstat.buff[(sdrstat.buffcnt % MEMBUFFLEN) * sdrini.dtype[0] * FILE_BUFFSIZE];
        nread1 = fread(b1, 1, (size_t)sdrini.dtype[0] * FILE_BUFFSIZE, sdrini.fp1);
    }
    if (sdrini.fp2 != NULL) {
        b2 = &sdrstat.buff2[(sdrstat.buffcnt % MEMBUFFLEN) * sdrini.dtype[1] * FILE_BUFFSIZE];
        nread2 = fread(b2, 1, (size_t)sdrini.dtype[1] * FILE_BUFFSIZE, sdrini.fp2);
    }
    if (ctx && b1) gnsscorr_ring_push(ctx, FTYPE1, b1, FILE_BUFFSIZE);
    if (ctx && b2) gnsscorr_ring_push(ctx, FTYPE2, b2, FILE_BUFFSIZE);
    unmlock(hbuffmtx);

    if ((sdrini.fp1 != NULL && (int)nread1 < sdrini.dtype[0] * FILE_BUFFSIZE) ||
        (sdrini.fp2 != NULL && (int)nread2 < sdrini.dtype[1] * FILE_BUFFSIZE)) {
        sdrstat.stopflag = ON;
        SDRPRINTF("end of file!\n");
    }
    mlock(hreadmtx);
    sdrstat.buffcnt++;
    unmlock(hreadmtx);
}

/* ref src/sdrrcv.c:505-532 (host mirror; the kernels read the HBM ring) */
void file_getbuff(uint64_t buffloc, int n, int ftype, int dtype, char *expbuf)
{
    const uint64_t ringbytes = (uint64_t)MEMBUFFLEN * dtype * FILE_BUFFSIZE;
    const uint64_t loc = (uint64_t)dtype * buffloc % ringbytes;
    const unsigned char *src = ftype == FTYPE1 ? sdrstat.buff : sdrstat.buff2;
    int nb = dtype * n, nout = (int)((int64_t)(loc + (uint64_t)nb) - (int64_t)ringbytes);
    mlock(hbuffmtx);
    if (nout > 0) {
        memcpy(expbuf, src + loc, (size_t)(nb - nout));
        memcpy(expbuf + (nb - nout), src, (size_t)nout);
    } else {
        memcpy(expbuf, src + loc, (size_t)nb);
    }
    unmlock(hbuffmtx);
}

/* ref src/sdrrcv.c:406-463: only the file front end is served here */
int rcvgetbuff(sdrini_t *ini, uint64_t buffloc, int n, int ftype, int dtype, char *expbuf)
{
    if (ini->fend != FEND_FILE) return -1;
    file_getbuff(buffloc, n, ftype, dtype, expbuf);
    return 0;
}

/* ref src/sdrcmn.c:574-578 */
void ind2sub(int ind, int nx, int ny, int *subx, int *suby)
{
    *subx = ind % nx;
    *suby = ny * ind / (nx * ny);
}
