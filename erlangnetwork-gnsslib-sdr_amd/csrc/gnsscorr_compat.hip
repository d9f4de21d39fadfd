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
#include <ctime>
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
    if (ci > 0.0 && ci < (double)len) {
        *remc = gc_code_rem(gc_code_walk(gc_code_start(coff, smax, ci, len), ci, len, n + 2 * smax, ne), smax, ci);
    } else {
        // a chip step outside (0, len): the reference's table reads are out of bounds there (nothing is correlated), but
        // rescode() still returns a remainder -- its own loop, literally (ref src/sdrcmn.c:613-620)
        GC_FP_STRICT
        const double dlen = (double)len;
        double c = gc_code_start(coff, smax, ci, len);
        for (int i = 0; i < n + 2 * smax; i++) {
            if (c >= dlen) c = c - dlen;
            c = c + ci;
        }
        *remc = gc_code_rem(c, smax, ci);
    }
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
                            c.ntap, dtype, c.ntap, n, c.smax, nullptr);
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

// ---- sdrtracking(): concurrent callers share one launch chain ------------------------------------------
// The reference runs one pthread per channel and every one of them calls sdrtracking() once per code period
// (ref src/sdrmain.c:144-149,264-265).  One launch chain per call would serialise 32 threads on the GPU
// queue at ~100 us each; instead the callers that arrive while a chain is in flight are collected and the
// next chain serves all of them at once (flat combining: whoever finds no leader becomes the leader, takes
// every queued request, runs expand -> correlate -> finish for the whole set, hands the results back and
// wakes the others).  Per-channel code blocks stay on the device between calls.
}   // extern "C" (helpers below are C++)

#include <condition_variable>
#include <map>
#include <vector>

namespace {

struct TrkReq {
    sdrch_t *sdr;
    uint64_t buffloc;
    int n;
    double cI[GNSSCORR_MAXTAPS], cQ[GNSSCORR_MAXTAPS];
    unsigned long codesum;          // hash of sdr->code[0..clen), computed by the calling thread (cmb_code)
    int rc;
    bool done;
    char err[200];
};

struct CodeSlot { int8_t *dcode = nullptr; unsigned long sum = ~0ul; int clen = 0, nedge = 0, pm1 = 0; };

struct TrkCombiner {
    std::mutex qm;
    std::condition_variable cv;
    std::vector<TrkReq *> queue;
    bool leader = false;
    std::map<sdrch_t *, CodeSlot> codes;
    // device / pinned staging for `cap` requests.  One pinned block goes down per launch chain (the requests' GcChan
    // and GcTrkPlan records, side by side); the results come back without a copy: trk_finish writes the sums into
    // pinned host memory (hres), and the piece-table overflow counter lives there too (hover).
    int cap = 0, nseg_cap = 0, ntap_cap = 0;
    char *dstage = nullptr, *hstage = nullptr;
    GcTrkUnit *dunit = nullptr;
    GcUnitSegs *dsegs = nullptr;
    GcRound *drounds = nullptr;
    int *dpartial = nullptr;
    double *dsum = nullptr;                             // sumI[cap][ntap], sumQ[cap][ntap] (device; unused by sdrtracking)
    double *hres = nullptr, *hres_dev = nullptr;        // corrI[cap][ntap], corrQ[cap][ntap] (pinned, written by the device)
    int *hover = nullptr, *hover_dev = nullptr;         // NCO piece-table overflow (pinned)
    unsigned long long *dfinish = nullptr;
};
TrkCombiner g_cmb;

// GNSSCORR_CMB_PROF=1: where a combined launch chain's host time goes, printed when the process ends
struct CmbProf {
    bool on = getenv("GNSSCORR_CMB_PROF") != nullptr;
    unsigned long long batches = 0, reqs = 0;
    double t[5] = {0, 0, 0, 0, 0};
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
    ~CmbProf()
    {
        if (on && batches)
            fprintf(stderr, "gnsscorr combiner: %llu launch chains, %.1f requests each; us per chain: prepare %.1f, enqueue %.1f, "
                            "wait %.1f, hand out %.1f, whole cmb_run %.1f\n", batches, (double)reqs / batches, t[0] / batches,
                    t[1] / batches, t[2] / batches, t[3] / batches, t[4] / batches);
    }
};
CmbProf g_cprof;

int cmb_reserve(gnsscorr_ctx *ctx, int k, int nseg, int ntap)
{
    TrkCombiner &q = g_cmb;
    if (k <= q.cap && nseg <= q.nseg_cap && ntap <= q.ntap_cap) return 0;
    GC_HIP(hipStreamSynchronize(ctx->stream));
    // (every pointer is cleared as it is freed: an allocation that fails below leaves nothing dangling, and the next
    // call -- capacities are zero -- starts over)
    auto dfree = [](auto *&p) { if (p) hipFree(p); p = nullptr; };
    auto hfree = [](auto *&p) { if (p) hipHostFree(p); p = nullptr; };
    dfree(q.dstage); dfree(q.dunit); dfree(q.dsegs); dfree(q.drounds); dfree(q.dpartial);
    dfree(q.dsum); dfree(q.dfinish);
    hfree(q.hstage); hfree(q.hres); hfree(q.hover);
    q.hres_dev = nullptr; q.hover_dev = nullptr;
    const int cap = k > 64 ? k : 64, ns = nseg > 1 ? nseg : 1, nt = GNSSCORR_MAXTAPS;
    q.cap = q.nseg_cap = q.ntap_cap = 0;
    const size_t stage = (sizeof(GcChan) + sizeof(GcTrkPlan)) * cap + 64;
    GC_HIP(hipMalloc((void **)&q.dstage, stage));
    GC_HIP(hipMalloc((void **)&q.dunit, sizeof(GcTrkUnit) * cap));
    GC_HIP(hipMalloc((void **)&q.dsegs, sizeof(GcUnitSegs) * cap));
    GC_HIP(hipMalloc((void **)&q.drounds, sizeof(GcRound) * cap * ns * GC_MAXR));
    GC_HIP(hipMalloc((void **)&q.dpartial, sizeof(int) * cap * ns * 2 * nt));
    GC_HIP(hipMalloc((void **)&q.dsum, sizeof(double) * cap * 2 * nt));
    GC_HIP(hipMalloc((void **)&q.dfinish, sizeof(unsigned long long) * cap * GC_FINISH_SCRATCH));
    GC_HIP(hipMemsetAsync(q.dfinish, 0, sizeof(unsigned long long) * cap * GC_FINISH_SCRATCH, ctx->stream));
    GC_HIP(hipHostMalloc((void **)&q.hstage, stage));
    GC_HIP(hipHostMalloc((void **)&q.hres, sizeof(double) * cap * 2 * nt, hipHostMallocMapped));
    GC_HIP(hipHostGetDevicePointer((void **)&q.hres_dev, q.hres, 0));
    GC_HIP(hipHostMalloc((void **)&q.hover, 64, hipHostMallocMapped));
    GC_HIP(hipHostGetDevicePointer((void **)&q.hover_dev, q.hover, 0));
    *q.hover = 0;
    q.cap = cap; q.nseg_cap = ns; q.ntap_cap = nt;
    return 0;
}

// the channel's code block on the device (uploaded when the code of this sdrch_t is first seen or changes)
int cmb_code(gnsscorr_ctx *ctx, sdrch_t *sdr, unsigned long sum, CodeSlot **out)
{
    // (keyed by the caller's sdrch_t: a receiver has at most MAXSAT of them; a caller that keeps handing in new structs
    // gets the table emptied instead of growing without bound -- the stream is idle here, every call ends synchronised)
    if (g_cmb.codes.size() > 256 && !g_cmb.codes.count(sdr)) {
        for (auto &kv : g_cmb.codes) if (kv.second.dcode) hipFree(kv.second.dcode);
        g_cmb.codes.clear();
    }
    CodeSlot &cs = g_cmb.codes[sdr];
    if (!cs.dcode || cs.sum != sum || cs.clen != sdr->clen) {
        if (!cs.dcode) GC_HIP(hipMalloc((void **)&cs.dcode, GC_CODEBLOCK));
        int8_t block[GC_CODEBLOCK];
        gc_build_codeblock(sdr->code, sdr->clen, block, &cs.nedge, &cs.pm1);
        GC_HIP(hipMemcpy(cs.dcode, block, GC_CODEBLOCK, hipMemcpyHostToDevice));
        cs.sum = sum;
        cs.clen = sdr->clen;
    }
    *out = &cs;
    return 0;
}

// one launch chain for a set of requests that share dtype and tap count
int cmb_run_group(gnsscorr_ctx *ctx, std::vector<TrkReq *> &grp)
{
    TrkCombiner &q = g_cmb;
    const int k = (int)grp.size();
    const int dtype = grp[0]->sdr->dtype, ntap = 1 + 2 * grp[0]->sdr->trk.corrn;
    int max_n = 0, smax_max = 0;
    for (TrkReq *r : grp) {
        if (r->n > max_n) max_n = r->n;
        const int sm = r->sdr->trk.corrp[r->sdr->trk.corrn - 1];
        if (sm > smax_max) smax_max = sm;
    }
    const int nseg = gc_trk_nseg(dtype, max_n);
    int rc = cmb_reserve(ctx, k, nseg, ntap);
    if (rc) return rc;
    const double tp0 = g_cprof.on ? CmbProf::now() : 0.0;
    const size_t plan_off = (sizeof(GcChan) * (size_t)k + 63) & ~(size_t)63;
    GcChan *hchan = reinterpret_cast<GcChan *>(q.hstage), *dchan = reinterpret_cast<GcChan *>(q.dstage);
    GcTrkPlan *hplan = reinterpret_cast<GcTrkPlan *>(q.hstage + plan_off), *dplan = reinterpret_cast<GcTrkPlan *>(q.dstage + plan_off);
    for (int i = 0; i < k; i++) {
        sdrch_t *sdr = grp[i]->sdr;
        const GcRing &ring = ctx->ring[sdr->ftype == FTYPE2 ? 1 : 0];
        CodeSlot *cs;
        rc = cmb_code(ctx, sdr, grp[i]->codesum, &cs);
        if (rc) return rc;
        GcChan &c = hchan[i];
        memset(&c, 0, sizeof(c));
        c.ring = ring.mem; c.ringlen = ring.ringlen; c.code = cs->dcode;
        c.dtype = dtype; c.clen = sdr->clen; c.nsamp = grp[i]->n; c.ntap = ntap;
        c.smax = sdr->trk.corrp[sdr->trk.corrn - 1];
        c.tapoff[0] = 0;
        for (int t = 0; t < sdr->trk.corrn; t++) { c.tapoff[1 + 2 * t] = -sdr->trk.corrp[t]; c.tapoff[2 + 2 * t] = sdr->trk.corrp[t]; }
        c.ti = sdr->ti;
        c.nedge = cs->nedge; c.pm1 = cs->pm1;
        GcTrkPlan &p = hplan[i];
        memset(&p, 0, sizeof(p));
        p.buffloc = grp[i]->buffloc; p.coff = sdr->trk.oldremcode; p.phi0 = sdr->trk.oldremcarr;
        p.carrfreq = sdr->trk.carrfreq; p.codefreq = sdr->trk.codefreq; p.n = grp[i]->n;
    }
    hipStream_t st = ctx->stream;
    const double tp1 = g_cprof.on ? CmbProf::now() : 0.0;
    GC_HIP(hipMemcpyAsync(q.dstage, q.hstage, plan_off + sizeof(GcTrkPlan) * k, hipMemcpyHostToDevice, st));
    rc = gc_launch_trk_expand(st, dchan, dplan, q.dunit, q.dsegs, nullptr, k, 1, q.drounds, nseg, max_n, q.hover_dev);
    if (rc) return rc;
    rc = gc_launch_trk_corr(st, dchan, q.dunit, q.dsegs, q.drounds, q.dpartial, k, 1, nseg, ntap, dtype, ntap, max_n, smax_max, nullptr);
    if (rc) return rc;
    // (the period's sums straight into pinned host memory: no copy back)
    double *cI = q.hres_dev, *cQ = q.hres_dev + (size_t)q.cap * ntap, *sI = q.dsum, *sQ = q.dsum + (size_t)q.cap * ntap;
    rc = gc_launch_trk_finish(st, q.dpartial, cI, cQ, sI, sQ, q.dfinish, k, 1, nseg, ntap);
    if (rc) return rc;
    const double tp2 = g_cprof.on ? CmbProf::now() : 0.0;
    GC_HIP(hipStreamSynchronize(st));
    const double tp3 = g_cprof.on ? CmbProf::now() : 0.0;
    if (*q.hover) {
        *q.hover = 0;
        return gc_fail(GNSSCORR_EINVAL, "sdrtracking: a period needs more NCO pieces than the tables hold");
    }
    for (int i = 0; i < k; i++) {
        memcpy(grp[i]->cI, q.hres + (size_t)i * ntap, sizeof(double) * ntap);
        memcpy(grp[i]->cQ, q.hres + (size_t)(q.cap + i) * ntap, sizeof(double) * ntap);
    }
    if (g_cprof.on) {
        const double tp4 = CmbProf::now();
        g_cprof.batches++; g_cprof.reqs += (unsigned long long)k;
        g_cprof.t[0] += tp1 - tp0; g_cprof.t[1] += tp2 - tp1; g_cprof.t[2] += tp3 - tp2; g_cprof.t[3] += tp4 - tp3;
    }
    return 0;
}

void cmb_run(std::vector<TrkReq *> &batch)
{
    const double tr0 = g_cprof.on ? CmbProf::now() : 0.0;
    struct Whole { double t0; ~Whole() { if (g_cprof.on) g_cprof.t[4] += CmbProf::now() - t0; } } whole{tr0};
    gnsscorr_ctx *ctx = gnsscorr_default_ctx();
    auto fail_all = [&](std::vector<TrkReq *> &v, const char *msg) {
        for (TrkReq *r : v) { r->rc = -1; snprintf(r->err, sizeof(r->err), "%s", msg); }
    };
    if (!ctx) { fail_all(batch, gnsscorr_last_error()); return; }
    std::lock_guard<std::mutex> lk(ctx->mtx);
    if (hipSetDevice(ctx->device) != hipSuccess) { fail_all(batch, "hipSetDevice"); return; }
    if (gc_ingest_fence(ctx)) { fail_all(batch, gnsscorr_last_error()); return; }      // behind the grabber's last block
    // groups of equal dtype and tap count (one [TRACK] section per receiver: normally a single group)
    std::vector<bool> taken(batch.size(), false);
    for (size_t i = 0; i < batch.size(); i++) {
        if (taken[i]) continue;
        std::vector<TrkReq *> grp;
        for (size_t j = i; j < batch.size(); j++) {
            if (taken[j]) continue;
            if (batch[j]->sdr->dtype == batch[i]->sdr->dtype && batch[j]->sdr->trk.corrn == batch[i]->sdr->trk.corrn) {
                const GcRing &ring = ctx->ring[batch[j]->sdr->ftype == FTYPE2 ? 1 : 0];
                taken[j] = true;
                if (!ring.mem || ring.dtype != batch[j]->sdr->dtype) {
                    batch[j]->rc = -1;
                    snprintf(batch[j]->err, sizeof(batch[j]->err), "IF ring %d is not mirrored on the GPU", batch[j]->sdr->ftype);
                    continue;
                }
                grp.push_back(batch[j]);
            }
        }
        if (grp.empty()) continue;
        if (cmb_run_group(ctx, grp)) fail_all(grp, gnsscorr_last_error());
    }
}

// enqueue, lead or wait
void cmb_submit(TrkReq *req)
{
    TrkCombiner &q = g_cmb;
    std::unique_lock<std::mutex> lk(q.qm);
    q.queue.push_back(req);
    while (!req->done) {
        if (!q.leader) {
            q.leader = true;
            std::vector<TrkReq *> batch;
            batch.swap(q.queue);
            lk.unlock();
            cmb_run(batch);
            lk.lock();
            for (TrkReq *r : batch) r->done = true;
            q.leader = false;
            q.cv.notify_all();
        } else {
            q.cv.wait(lk);
        }
    }
}

}  // namespace

extern "C" {

// ref src/sdrtrk.c:15-54.  Samples come from the HBM mirror of the ring (front end sdr->ftype) instead of
// rcvgetbuff(); concurrent callers are served by one launch chain (above).
uint64_t sdrtracking(sdrch_t *sdr, uint64_t buffloc, uint64_t cnt)
{
    uint64_t bufflocnow;
    sdr->flagtrk = OFF;

    mlock(hreadmtx);
    bufflocnow = sdrstat.fendbuffsize * sdrstat.buffcnt - sdr->nsamp;
    unmlock(hreadmtx);

    if (bufflocnow > buffloc) {
        sdr->currnsamp = (int)((sdr->clen - sdr->trk.remcode) / (sdr->trk.codefreq / sdr->f_sf));

        // the reference copies 1+2*corrn*sizeof(double) bytes here (:35-36)
        memcpy(sdr->trk.oldI, sdr->trk.II, 1 + 2 * sdr->trk.corrn * sizeof(double));
        memcpy(sdr->trk.oldQ, sdr->trk.QQ, 1 + 2 * sdr->trk.corrn * sizeof(double));
        sdr->trk.oldremcode = sdr->trk.remcode;
        sdr->trk.oldremcarr = sdr->trk.remcarr;

        TrkReq req;
        req.sdr = sdr; req.buffloc = buffloc; req.n = sdr->currnsamp; req.rc = 0; req.done = false; req.err[0] = 0;
        if (sdr->trk.corrn < 1 || 1 + 2 * sdr->trk.corrn > GNSSCORR_MAXTAPS || sdr->clen < 1 || sdr->clen > 1023 ||
            sdr->currnsamp < 1) {
            SDRPRINTF("error: sdrtracking: unsupported shape (corrn %d, code length %d, %d samples)\n", sdr->trk.corrn,
                      sdr->clen, sdr->currnsamp);
            return bufflocnow;
        }
        req.codesum = 0;                // (unsigned: the hash wraps)
        for (int i = 0; i < sdr->clen; i++) req.codesum = req.codesum * 31u + (unsigned long)(unsigned short)sdr->code[i];
        cmb_submit(&req);
        if (req.rc) {
            SDRPRINTF("error: sdrtracking: %s\n", req.err);
            return bufflocnow;
        }
        // correlator(..., sdr->trk.QQ, sdr->trk.II, ...): the swapped hand-over of :40-43
        const int ntap = 1 + 2 * sdr->trk.corrn;
        memcpy(sdr->trk.QQ, req.cI, sizeof(double) * ntap);
        memcpy(sdr->trk.II, req.cQ, sizeof(double) * ntap);
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
