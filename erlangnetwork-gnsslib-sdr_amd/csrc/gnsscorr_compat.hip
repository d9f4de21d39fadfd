// gnsscorr_compat.hip -- the reference's own per-call symbols (one channel,
// one code period / one acquisition attempt per call), implemented on the HIP
// kernels through the process-wide default context.  Signatures and side
// effects follow the reference (cited per function); see INTEGRATION.md for
// how they replace src/sdracq.c, src/sdrtrk.c and the helpers in
// src/sdrcmn.c when linking the reference's channel thread.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

#include "../../include/sdr_compat.h"
#include "gnsscorr_ctx.h"

#define SDRPRINTF printf

// per-call scratch on the default context (guarded by ctx->mtx)
struct GcOnce {
    int8_t *data = nullptr;  size_t data_cap = 0;    // scratch "ring" for host-supplied samples
    int8_t *code = nullptr;                          // one code block
    GcChan *chan = nullptr;
    GcTrkPlan *plan = nullptr;
    GcTrkUnit *unit = nullptr;
    GcUnitSegs *segs = nullptr;
    int *overflow = nullptr;
    double *out = nullptr;                           // 4*GNSSCORR_MAXTAPS: corrI, corrQ, sumI, sumQ
    int *partial = nullptr;  int partial_cap = 0;    // nseg*2*ntap
    GcRound *rounds = nullptr;  int rounds_cap = 0;  // nseg*GC_MAXR
    unsigned long long *finish = nullptr;            // GC_FINISH_SCRATCH words, zero between launches
};
static GcOnce g_once;

static int once_init(gnsscorr_ctx *ctx)
{
    if (g_once.chan) return 0;
    GC_HIP(hipSetDevice(ctx->device));
    GC_HIP(hipMalloc((void **)&g_once.code, GC_CODEBLOCK));
    GC_HIP(hipMalloc((void **)&g_once.chan, sizeof(GcChan)));
    GC_HIP(hipMalloc((void **)&g_once.plan, sizeof(GcTrkPlan)));
    GC_HIP(hipMalloc((void **)&g_once.unit, sizeof(GcTrkUnit)));
    GC_HIP(hipMalloc((void **)&g_once.segs, sizeof(GcUnitSegs)));
    GC_HIP(hipMalloc((void **)&g_once.overflow, sizeof(int)));
    GC_HIP(hipMemsetAsync(g_once.overflow, 0, sizeof(int), ctx->stream));
    GC_HIP(hipMalloc((void **)&g_once.out, sizeof(double) * 4 * GNSSCORR_MAXTAPS));
    GC_HIP(hipMalloc((void **)&g_once.finish, sizeof(unsigned long long) * GC_FINISH_SCRATCH));
    GC_HIP(hipMemsetAsync(g_once.finish, 0, sizeof(unsigned long long) * GC_FINISH_SCRATCH, ctx->stream));
    return 0;
}

// NCO remainders returned by the reference (ref src/sdrcmn.c:620,666-668): its running sums walked
// piece by piece on the host -- the same code as the device planner (gnsscorr_nco.h)
static void host_rems(double phi0, double freq, double ti, int n, double coff, int smax, double ci,
                      int len, double *remc, double *remp)
{
    GcNoEmit ne;
    *remp = gc_carrier_prem(gc_carrier_walk(gc_carrier_phis(phi0), gc_carrier_ps(freq, ti), n, ne));
    if (ci > 0.0 && ci < (double)len)
        *remc = gc_code_rem(gc_code_walk(gc_code_start(coff, smax, ci, len), ci, len, n + 2 * smax, ne), smax, ci);
}

// One (channel, period) unit on samples that already sit in a device ring.
static int corr_unit(gnsscorr_ctx *ctx, const int8_t *ring, uint64_t ringlen, int dtype, double ti,
                     int n, double freq, double phi0, double crate, double coff, const int *s, int ns,
                     const short *codein, int coden, uint64_t buffloc, double *cI, double *cQ)
{
    if (ns < 1 || 1 + 2 * ns > GNSSCORR_MAXTAPS) return gc_fail(GNSSCORR_EINVAL, "correlator: ns %d", ns);
    if (coden < 1 || coden > 1023) return gc_fail(GNSSCORR_EINVAL, "correlator: code length %d", coden);
    if (n < 1) return gc_fail(GNSSCORR_EINVAL, "correlator: n %d", n);
    int rc = once_init(ctx);
    if (rc) return rc;
    GcChan c;
    memset(&c, 0, sizeof(c));
    c.ring = ring; c.ringlen = ringlen; c.code = g_once.code;
    c.dtype = dtype; c.clen = coden; c.nsamp = n; c.ntap = 1 + 2 * ns; c.smax = s[ns - 1];
    c.tapoff[0] = 0;
    for (int k = 0; k < ns; k++) { c.tapoff[1 + 2 * k] = -s[k]; c.tapoff[2 + 2 * k] = s[k]; }
    c.ti = ti;
    GcTrkPlan p;
    memset(&p, 0, sizeof(p));
    p.buffloc = buffloc; p.coff = coff; p.phi0 = phi0; p.carrfreq = freq; p.codefreq = crate; p.n = n;
    static int8_t block[GC_CODEBLOCK];     // guarded by ctx->mtx; every call ends with a stream sync
    gc_build_codeblock(codein, coden, block, &c.nedge, &c.pm1);
    GC_HIP(hipMemcpyAsync(g_once.code, block, GC_CODEBLOCK, hipMemcpyHostToDevice, ctx->stream));
    GC_HIP(hipMemcpyAsync(g_once.chan, &c, sizeof(c), hipMemcpyHostToDevice, ctx->stream));
    GC_HIP(hipMemcpyAsync(g_once.plan, &p, sizeof(p), hipMemcpyHostToDevice, ctx->stream));
    const int nseg = gc_trk_nseg(dtype, n);
    if (nseg * 2 * c.ntap > g_once.partial_cap) {
        if (g_once.partial) hipFree(g_once.partial);
        g_once.partial = nullptr; g_once.partial_cap = 0;
        GC_HIP(hipMalloc((void **)&g_once.partial, sizeof(int) * nseg * 2 * c.ntap));
        g_once.partial_cap = nseg * 2 * c.ntap;
    }
    if (nseg > g_once.rounds_cap) {
        if (g_once.rounds) hipFree(g_once.rounds);
        g_once.rounds = nullptr; g_once.rounds_cap = 0;
        GC_HIP(hipMalloc((void **)&g_once.rounds, sizeof(GcRound) * nseg * GC_MAXR));
        g_once.rounds_cap = nseg;
    }
    rc = gc_launch_trk_expand(ctx->stream, g_once.chan, g_once.plan, g_once.unit, g_once.segs, nullptr, 1, 1, g_once.rounds,
                              nseg, n, g_once.overflow);
    if (rc) return rc;
    rc = gc_launch_trk_corr(ctx->stream, g_once.chan, g_once.unit, g_once.segs, g_once.rounds, g_once.partial, 1, 1, nseg,
                            c.ntap, dtype, c.ntap, n, c.smax);
    if (rc) return rc;
    rc = gc_launch_trk_finish(ctx->stream, g_once.partial, g_once.out, g_once.out + GNSSCORR_MAXTAPS,
                              g_once.out + 2 * GNSSCORR_MAXTAPS, g_once.out + 3 * GNSSCORR_MAXTAPS, g_once.finish, 1, 1, nseg,
                              c.ntap);
    if (rc) return rc;
    double host[2 * GNSSCORR_MAXTAPS];
    int over = 0;
    GC_HIP(hipMemcpyAsync(host, g_once.out, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipMemcpyAsync(&over, g_once.overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    if (over) {
        GC_HIP(hipMemsetAsync(g_once.overflow, 0, sizeof(int), ctx->stream));
        return gc_fail(GNSSCORR_EINVAL, "correlator: the call spans more code periods (chip step %g x %d samples over %d chips) "
                       "or carrier binades than the NCO tables hold", ti * crate, n, coden);
    }
    memcpy(cI, host, sizeof(double) * c.ntap);
    memcpy(cQ, host + GNSSCORR_MAXTAPS, sizeof(double) * c.ntap);
    return 0;
}

extern "C" {

// ref src/sdrcmn.c:687-722.  On failure prints an error and leaves the
// outputs untouched, like the reference's allocation-failure path (:697-702).
void correlator(const char *data, int dtype, double ti, int n, double freq, double phi0, double crate,
                double coff, int *s, int ns, double *II, double *QQ, double *remc, double *remp,
                short *codein, int coden)
{
    gnsscorr_ctx *ctx = gnsscorr_default_ctx();
    if (!ctx) { SDRPRINTF("error: correlator: no GPU context (%s)\n", gnsscorr_last_error()); return; }
    std::lock_guard<std::mutex> lk(ctx->mtx);
    if (hipSetDevice(ctx->device) != hipSuccess) { SDRPRINTF("error: correlator: hipSetDevice\n"); return; }
    const size_t bytes = (((size_t)n * dtype + 15) & ~(size_t)15) + 32;
    if (bytes > g_once.data_cap) {
        if (g_once.data) hipFree(g_once.data);
        g_once.data = nullptr; g_once.data_cap = 0;
        if (hipMalloc((void **)&g_once.data, bytes) != hipSuccess) {
            SDRPRINTF("error: correlator memory allocation\n");
            return;
        }
        g_once.data_cap = bytes;
    }
    if (hipMemcpyAsync(g_once.data, data, (size_t)n * dtype, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        SDRPRINTF("error: correlator: sample upload failed\n");
        return;
    }
    double cI[GNSSCORR_MAXTAPS], cQ[GNSSCORR_MAXTAPS];
    const uint64_t ringlen = g_once.data_cap / dtype;      // multiple of 16 bytes by construction
    if (corr_unit(ctx, g_once.data, ringlen, dtype, ti, n, freq, phi0, crate, coff, s, ns, codein, coden, 0,
                  cI, cQ)) {
        SDRPRINTF("error: correlator: %s\n", gnsscorr_last_error());
        return;
    }
    memcpy(II, cI, sizeof(double) * (1 + 2 * ns));
    memcpy(QQ, cQ, sizeof(double) * (1 + 2 * ns));
    host_rems(phi0, freq, ti, n, coff, s[ns - 1], ti * crate, coden, remc, remp);
}

// ref src/sdrtrk.c:15-54.  Samples come from the HBM mirror of the ring
// (front end sdr->ftype) instead of rcvgetbuff().
uint64_t sdrtracking(sdrch_t *sdr, uint64_t buffloc, uint64_t cnt)
{
    uint64_t bufflocnow;
    sdr->flagtrk = OFF;

    mlock(hreadmtx);
    bufflocnow = sdrstat.fendbuffsize * sdrstat.buffcnt - sdr->nsamp;
    unmlock(hreadmtx);

    if (bufflocnow > buffloc) {
        gnsscorr_ctx *ctx = gnsscorr_default_ctx();
        if (!ctx) { SDRPRINTF("error: sdrtracking: no GPU context (%s)\n", gnsscorr_last_error()); return bufflocnow; }
        std::lock_guard<std::mutex> lk(ctx->mtx);
        const GcRing &r = ctx->ring[sdr->ftype == FTYPE2 ? 1 : 0];
        if (!r.mem || r.dtype != sdr->dtype) {
            SDRPRINTF("error: sdrtracking: IF ring %d is not mirrored on the GPU\n", sdr->ftype);
            return bufflocnow;
        }
        sdr->currnsamp = (int)((sdr->clen - sdr->trk.remcode) / (sdr->trk.codefreq / sdr->f_sf));

        // the reference copies 1+2*corrn*sizeof(double) bytes here (:35-36)
        memcpy(sdr->trk.oldI, sdr->trk.II, 1 + 2 * sdr->trk.corrn * sizeof(double));
        memcpy(sdr->trk.oldQ, sdr->trk.QQ, 1 + 2 * sdr->trk.corrn * sizeof(double));
        sdr->trk.oldremcode = sdr->trk.remcode;
        sdr->trk.oldremcarr = sdr->trk.remcarr;

        double cI[GNSSCORR_MAXTAPS], cQ[GNSSCORR_MAXTAPS];
        if (hipSetDevice(ctx->device) != hipSuccess ||
            corr_unit(ctx, r.mem, r.ringlen, sdr->dtype, sdr->ti, sdr->currnsamp, sdr->trk.carrfreq,
                      sdr->trk.oldremcarr, sdr->trk.codefreq, sdr->trk.oldremcode, sdr->trk.corrp,
                      sdr->trk.corrn, sdr->code, sdr->clen, buffloc, cI, cQ)) {
            SDRPRINTF("error: sdrtracking: %s\n", gnsscorr_last_error());
            return bufflocnow;
        }
        // correlator(..., sdr->trk.QQ, sdr->trk.II, ...): the swapped hand-over of :40-43
        const int ntap = 1 + 2 * sdr->trk.corrn;
        memcpy(sdr->trk.QQ, cI, sizeof(double) * ntap);
        memcpy(sdr->trk.II, cQ, sizeof(double) * ntap);
        host_rems(sdr->trk.oldremcarr, sdr->trk.carrfreq, sdr->ti, sdr->currnsamp, sdr->trk.oldremcode,
                  sdr->trk.corrp[sdr->trk.corrn - 1], sdr->ti * sdr->trk.codefreq, sdr->clen,
                  &sdr->trk.remcode, &sdr->trk.remcarr);

        sdrnavigation(sdr, buffloc, cnt);
        sdr->flagtrk = ON;
    } else {
        usleep(1000);   // sleepms(1)
    }
    return bufflocnow;
}

}  // extern "C"
