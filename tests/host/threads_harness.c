/* 32 pthreads calling the drop-in sdrtracking() + cumsumcorr() + pll() + dll() the way the reference's
 * sdrthread() does (ref src/sdrmain.c:144-149,264-312): measures the call rate without an interpreter in the
 * way (tests/test_gpu_symbols.py builds and runs it on the GPU box and checks the printed sums).
 * usage: threads_harness <if file (int8 real, 16.368 Msps, IF 4.092 MHz)> <nblocks> <nch> <nper> */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/sdr_compat.h"

extern int rcvinit_file(sdrini_t *ini);

static sdrch_t ch[64];
static uint64_t start[64];
static int nper;
static double chk[64];

static void *worker(void *arg)
{
    const int i = (int)(size_t)arg;
    sdrch_t *sdr = &ch[i];
    uint64_t b = start[i];
    double acc = 0;
    for (int cnt = 0; cnt < nper; cnt++) {
        sdrtracking(sdr, b, (uint64_t)cnt);
        if (!sdr->flagtrk) { fprintf(stderr, "channel %d: no data at period %d\n", i, cnt); break; }
        for (int t = 0; t < 5; t++) acc += sdr->trk.II[t] * (t + 1) + sdr->trk.QQ[t] * (t + 7);
        cumsumcorr(&sdr->trk, 1);
        pll(sdr, &sdr->trk.prm1, sdr->ctime);
        dll(sdr, &sdr->trk.prm1, sdr->ctime);
        clearcumsumcorr(&sdr->trk);
        b += (uint64_t)sdr->currnsamp;
    }
    chk[i] = acc;
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int nblocks = atoi(argv[2]), nch = atoi(argv[3]);
    nper = atoi(argv[4]);
    sdrini.fend = FEND_FILE; sdrini.useif1 = ON; sdrini.useif2 = OFF;
    strncpy(sdrini.file1, argv[1], sizeof(sdrini.file1) - 1);
    sdrini.dtype[0] = DTYPEI; sdrini.f_sf[0] = 16.368e6; sdrini.f_if[0] = 4.092e6; sdrini.f_cf[0] = 1575.42e6;
    sdrini.trkcorrn = 2; sdrini.trkcorrd = 3; sdrini.trkcorrp = 3;
    sdrini.trkdllb[0] = 5.0; sdrini.trkdllb[1] = 1.0;
    sdrini.trkpllb[0] = 30.0; sdrini.trkpllb[1] = 10.0;
    sdrini.trkfllb[0] = 200.0; sdrini.trkfllb[1] = 50.0;
    if (rcvinit_file(&sdrini)) return 3;
    for (int i = 0; i < nblocks; i++) file_pushtomembuf();
    unsigned seed = 12345;
    for (int i = 0; i < nch; i++) {
        if (initsdrch(i + 1, SYS_GPS, i + 1, CTYPE_L1CA, DTYPEI, FTYPE1, 1575.42e6, 16.368e6, 4.092e6, &ch[i])) return 4;
        seed = seed * 1103515245u + 12345u;
        ch[i].flagacq = ON;
        ch[i].acq.acqfreq = 4.092e6 + 200.0 * (double)((int)((seed >> 16) % 41) - 20);
        ch[i].trk.carrfreq = ch[i].acq.acqfreq;
        ch[i].trk.codefreq = ch[i].crate;
        seed = seed * 1103515245u + 12345u;
        start[i] = (seed >> 8) % 16368;
    }
    sdrch_t dummy;
    memset(&dummy, 0, sizeof(dummy));
    dummy.nsamp = 16368;
    sdrtracking(&dummy, (uint64_t)1 << 60, 0);      /* context creation outside the timed region */
    pthread_t th[64];
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int i = 0; i < nch; i++) pthread_create(&th[i], NULL, worker, (void *)(size_t)i);
    for (int i = 0; i < nch; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    const double dt = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
    printf("calls_per_s %.1f\n", nch * (double)nper / dt);
    for (int i = 0; i < nch; i++) printf("chk %d %.17g %.17g %.17g\n", i, chk[i], ch[i].trk.carrfreq, ch[i].trk.codefreq);
    return 0;
}
