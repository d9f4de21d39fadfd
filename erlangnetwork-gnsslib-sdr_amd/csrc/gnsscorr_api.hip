// gnsscorr_api.hip -- C-ABI layer of libgnsscorr.so: context, IF ring in HBM,
// channel tables, batched tracking entry points, per-kernel timing.
// (Acquisition entry points live in gnsscorr_acq.hip, the reference-named
// per-call symbols in gnsscorr_compat.hip.)
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gnsscorr_ctx.h"

static thread_local char g_err[512] = "";

int gc_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int gc_fail_hip(hipError_t e, const char *what, const char *file, int line)
{
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what,
             file, line);
    return GNSSCORR_EHIP;
}

extern "C" const char *gnsscorr_last_error(void) { return g_err; }

extern "C" int gnsscorr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int gnsscorr_create(gnsscorr_ctx **out, int device, void *stream)
{
    if (!out) return gc_fail(GNSSCORR_EINVAL, "gnsscorr_create: null out pointer");
    int ndev = 0;
    GC_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return gc_fail(GNSSCORR_EHIP, "gnsscorr_create: device %d not present (%d visible)", device, ndev);
    GC_HIP(hipSetDevice(device));
    gnsscorr_ctx *ctx = new gnsscorr_ctx();
    ctx->device = device;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return gc_fail_hip(e, "hipStreamCreate", __FILE__, __LINE__); }
        ctx->own_stream = true;
    }
    if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) ctx->stream2 = nullptr;
    if (!ctx->stream2 || hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess) ctx->stream3 = nullptr;
    if (ctx->stream2 && (hipStreamCreateWithFlags(&ctx->stream4, hipStreamNonBlocking) != hipSuccess ||
                         hipEventCreateWithFlags(&ctx->ev_spec, hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&ctx->ev_chain, hipEventDisableTiming) != hipSuccess)) {
        if (ctx->stream4) hipStreamDestroy(ctx->stream4);
        ctx->stream4 = nullptr;
    }
    for (int i = 0; i < 2 && ctx->stream2; i++) {
        hipEventCreateWithFlags(&ctx->ev_plan[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&ctx->ev_used[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&ctx->ev_corr[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&ctx->ev_fin[i], hipEventDisableTiming);
    }
    *out = ctx;
    return GNSSCORR_OK;
}

static void free_channels(gnsscorr_ctx *ctx)
{
    hipFree(ctx->dchan);  ctx->dchan = nullptr;
    hipFree(ctx->dcodes); ctx->dcodes = nullptr;
    hipFree(ctx->dfreqs); ctx->dfreqs = nullptr;
    for (int i = 0; i < 2; i++) { hipFree(ctx->dstate2[i]); ctx->dstate2[i] = nullptr; }
    hipFree(ctx->dloop); ctx->dloop = nullptr;
    hipFree(ctx->dloopdone); ctx->dloopdone = nullptr;
    hipFree(ctx->dlooplog); ctx->dlooplog = nullptr;
    hipFree(ctx->dstep_meta); ctx->dstep_meta = nullptr;
    hipFree(ctx->dstep_unit); ctx->dstep_unit = nullptr;
    hipFree(ctx->dstep_segs); ctx->dstep_segs = nullptr;
    hipFree(ctx->dstep_rounds); ctx->dstep_rounds = nullptr;
    hipFree(ctx->dstep_partial); ctx->dstep_partial = nullptr;
    ctx->step_nseg = 0;
    ctx->looplog_cap = 0;
    ctx->last_loop_nper = 0;
    ctx->state_cur = 0;
    ctx->ahead_valid = false;
    ctx->state_touched = true;
}

static void free_trk_buffers(gnsscorr_ctx *ctx)
{
    for (int i = 0; i < 2; i++) { hipFree(ctx->dplan2[i]); ctx->dplan2[i] = nullptr; }
    for (int i = 0; i < 2; i++) { hipFree(ctx->dspec2[i]); ctx->dspec2[i] = nullptr; }
    hipFree(ctx->detab); ctx->detab = nullptr;
    ctx->spec_ahead_valid = false;
    ctx->ahead_valid = false;
    hipFree(ctx->dcorrI); ctx->dcorrI = nullptr;
    hipFree(ctx->dcorrQ); ctx->dcorrQ = nullptr;
    hipFree(ctx->dsumI);  ctx->dsumI = nullptr;
    hipFree(ctx->dfinish); ctx->dfinish = nullptr;
    hipFree(ctx->dsumQ);  ctx->dsumQ = nullptr;
    for (int i = 0; i < 2; i++) { hipFree(ctx->dpartial2[i]); ctx->dpartial2[i] = nullptr; }
    ctx->fin_pending[0] = ctx->fin_pending[1] = false;
    for (int i = 0; i < 2; i++) {
        hipFree(ctx->dunit2[i]); ctx->dunit2[i] = nullptr;
        hipFree(ctx->drounds2[i]); ctx->drounds2[i] = nullptr;
        hipFree(ctx->dsegs2[i]); ctx->dsegs2[i] = nullptr;
        hipFree(ctx->dnsamp2[i]); ctx->dnsamp2[i] = nullptr;
    }
    hipFree(ctx->dnco_overflow); ctx->dnco_overflow = nullptr;
    hipFree(ctx->dring_viol); ctx->dring_viol = nullptr;
    ctx->plan_cap = 0;
}

extern "C" void gnsscorr_destroy(gnsscorr_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream_in) hipStreamSynchronize(ctx->stream_in);
    hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) hipStreamSynchronize(ctx->stream3);
    if (ctx->stream4) hipStreamSynchronize(ctx->stream4);
    gc_acq_free(ctx);
    free_trk_buffers(ctx);
    free_channels(ctx);
    for (auto &r : ctx->ring)
        if (r.owned && r.mem) hipFree(r.mem);
    for (auto &kv : ctx->timers)
        for (auto &p : kv.second.pending) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (int i = 0; i < 2; i++) {
        if (ctx->ev_plan[i]) hipEventDestroy(ctx->ev_plan[i]);
        if (ctx->ev_used[i]) hipEventDestroy(ctx->ev_used[i]);
        if (ctx->ev_corr[i]) hipEventDestroy(ctx->ev_corr[i]);
        if (ctx->ev_fin[i]) hipEventDestroy(ctx->ev_fin[i]);
    }
    if (ctx->stream_in) {
        hipStreamSynchronize(ctx->stream_in);
        for (int i = 0; i < 2; i++) { hipHostFree(ctx->pin[i]); hipFree(ctx->dstage[i]); if (ctx->ev_pin[i]) hipEventDestroy(ctx->ev_pin[i]); }
        if (ctx->ev_in) hipEventDestroy(ctx->ev_in);
        hipStreamDestroy(ctx->stream_in);
    }
    if (ctx->stream2) hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) hipStreamDestroy(ctx->stream3);
    if (ctx->stream4) { hipStreamSynchronize(ctx->stream4); hipStreamDestroy(ctx->stream4); }
    if (ctx->ev_spec) hipEventDestroy(ctx->ev_spec);
    if (ctx->ev_chain) hipEventDestroy(ctx->ev_chain);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" void *gnsscorr_stream(gnsscorr_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// Makes the main stream wait for the tracking outputs of the last batch (its finish runs on a stream
// of its own).
static int outputs_ready(gnsscorr_ctx *ctx)
{
    if (ctx->fin_pending[ctx->last_slot])
        GC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_fin[ctx->last_slot], 0));
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_sync(gnsscorr_ctx *ctx)
{
    if (!ctx) return gc_fail(GNSSCORR_EINVAL, "null context");
    GC_HIP(hipSetDevice(ctx->device));
    int rc = outputs_ready(ctx);
    if (rc) return rc;
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GNSSCORR_OK;
}

// ---------------------------------------------------------------------------
// IF ring
// ---------------------------------------------------------------------------
static GcRing *ring_of(gnsscorr_ctx *ctx, int ftype)
{
    if (!ctx || ftype < 1 || ftype > 2) return nullptr;
    return &ctx->ring[ftype - 1];
}

static void retarget_rings(gnsscorr_ctx *ctx);

extern "C" int gnsscorr_ring_create(gnsscorr_ctx *ctx, int ftype, int dtype, uint64_t ringlen,
                                    void *devmem)
{
    GcRing *r = ring_of(ctx, ftype);
    if (!r) return gc_fail(GNSSCORR_EINVAL, "ring_create: bad context or ftype %d", ftype);
    if (dtype != 1 && dtype != 2) return gc_fail(GNSSCORR_EINVAL, "ring_create: dtype %d not 1 or 2", dtype);
    if (ringlen == 0 || ((uint64_t)dtype * ringlen) % 16 != 0)
        return gc_fail(GNSSCORR_EINVAL, "ring_create: dtype*ringlen must be a positive multiple of 16");
    if (devmem && ((uintptr_t)devmem & 15))
        return gc_fail(GNSSCORR_EINVAL, "ring_create: device buffer must be 16-byte aligned");
    GC_HIP(hipSetDevice(ctx->device));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    if (r->owned && r->mem) hipFree(r->mem);
    r->mem = nullptr;
    r->owned = false;
    if (devmem) {
        r->mem = (int8_t *)devmem;
    } else {
        GC_HIP(hipMalloc((void **)&r->mem, (size_t)dtype * ringlen));
        GC_HIP(hipMemsetAsync(r->mem, 0, (size_t)dtype * ringlen, ctx->stream));
        r->owned = true;
    }
    r->dtype = dtype;
    r->ringlen = ringlen;
    r->wrpos = 0;
    retarget_rings(ctx);
    return GNSSCORR_OK;
}

int gc_ingest_fence(gnsscorr_ctx *ctx);

// ---- ingest ----------------------------------------------------------------------------------------------
#define GC_PIN_BYTES (8u << 20)         // per staging buffer

static int ingest_init(gnsscorr_ctx *ctx)
{
    if (ctx->stream_in) return GNSSCORR_OK;
    GC_HIP(hipStreamCreateWithFlags(&ctx->stream_in, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        GC_HIP(hipHostMalloc((void **)&ctx->pin[i], GC_PIN_BYTES));
        GC_HIP(hipMalloc((void **)&ctx->dstage[i], GC_PIN_BYTES));
        GC_HIP(hipEventCreateWithFlags(&ctx->ev_pin[i], hipEventDisableTiming));
    }
    GC_HIP(hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming));
    return GNSSCORR_OK;
}

// a free staging slot (waits for the transfer that last used it, never for the compute stream)
static int ingest_slot(gnsscorr_ctx *ctx, int *slot)
{
    const int s = ctx->pin_next;
    ctx->pin_next ^= 1;
    if (ctx->pin_busy[s]) GC_HIP(hipEventSynchronize(ctx->ev_pin[s]));
    ctx->pin_busy[s] = false;
    *slot = s;
    return GNSSCORR_OK;
}

// the compute stream reads the ring: order it behind the last transfer
int gc_ingest_fence(gnsscorr_ctx *ctx)
{
    if (ctx->in_pending) GC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_in, 0));
    return GNSSCORR_OK;
}

// ring bytes [pos, pos + bytes) <- device or pinned source, split at the end of the ring
static int ring_write(gnsscorr_ctx *ctx, GcRing *r, uint64_t bytepos, const void *src, uint64_t bytes, hipMemcpyKind kind)
{
    const uint64_t rb = (uint64_t)r->dtype * r->ringlen;
    const uint64_t pos = bytepos % rb;
    const uint64_t first = pos + bytes <= rb ? bytes : rb - pos;
    GC_HIP(hipMemcpyAsync(r->mem + pos, src, first, kind, ctx->stream_in));
    if (first < bytes) GC_HIP(hipMemcpyAsync(r->mem, (const int8_t *)src + first, bytes - first, kind, ctx->stream_in));
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_ring_push(gnsscorr_ctx *ctx, int ftype, const void *host, uint64_t nsamp)
{
    GcRing *r = ring_of(ctx, ftype);
    if (!r || !r->mem) return gc_fail(GNSSCORR_ESTATE, "ring_push: ring %d not created", ftype);
    if (nsamp > r->ringlen) return gc_fail(GNSSCORR_EINVAL, "ring_push: chunk larger than the ring");
    if (!host && nsamp) return gc_fail(GNSSCORR_EINVAL, "ring_push: null host buffer");
    GC_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(ctx->mtx);
    int rc = ingest_init(ctx);
    if (rc) return rc;
    const uint64_t d = (uint64_t)r->dtype;
    const int8_t *h = (const int8_t *)host;
    uint64_t done = 0, total = d * nsamp;
    while (done < total) {
        const uint64_t piece = total - done < GC_PIN_BYTES ? total - done : GC_PIN_BYTES;
        int s;
        rc = ingest_slot(ctx, &s);
        if (rc) return rc;
        memcpy(ctx->pin[s], h + done, piece);           // the caller's buffer is free again after this
        rc = ring_write(ctx, r, d * r->wrpos + done, ctx->pin[s], piece, hipMemcpyHostToDevice);
        if (rc) return rc;
        GC_HIP(hipEventRecord(ctx->ev_pin[s], ctx->stream_in));
        ctx->pin_busy[s] = true;
        done += piece;
    }
    GC_HIP(hipEventRecord(ctx->ev_in, ctx->stream_in));
    ctx->in_pending = true;
    r->wrpos += nsamp;
    return GNSSCORR_OK;
}

// ref src/rcv/stereo/stereo.c:160-205 (lut1 / lut2) and src/rcv/rtlsdr/rtlsdr.c:136-143
__global__ void unpack_stereo_kernel(const uint8_t *__restrict__ src, uint64_t n, int8_t *__restrict__ r1, uint64_t len1,
                                     uint64_t pos1, int8_t *__restrict__ r2, uint64_t len2, uint64_t pos2)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned b = src[i];
    // BASELUT1 = {-3,-1,+1,+3}; BASELUT2 = {+1,+3,+5,+7,-7,-5,-3,-1}
    if (r1) r1[(pos1 + i) % len1] = (int8_t)(2 * (int)((b >> 6) & 3) - 3);
    if (r2) {
        const int vi = (b >> 3) & 7, vq = b & 7;
        const uint64_t k = (pos2 + i) % len2;
        r2[2 * k] = (int8_t)(vi < 4 ? 2 * vi + 1 : 2 * vi - 15);
        r2[2 * k + 1] = (int8_t)(vq < 4 ? 2 * vq + 1 : 2 * vq - 15);
    }
}

__global__ void unpack_rtlsdr_kernel(const uint8_t *__restrict__ src, uint64_t nbytes, int8_t *__restrict__ ring,
                                     uint64_t ringbytes, uint64_t bytepos)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbytes) return;
    // (char)(buf[i] - 127.5): the double is truncated toward zero
    const int v = (int)src[i];
    ring[(bytepos + i) % ringbytes] = (int8_t)(v >= 128 ? v - 128 : v - 127);
}

extern "C" int gnsscorr_ring_push_packed(gnsscorr_ctx *ctx, int format, const void *host, uint64_t nsamp)
{
    if (!ctx || (!host && nsamp)) return gc_fail(GNSSCORR_EINVAL, "ring_push_packed: bad arguments");
    if (format != GNSSCORR_FMT_STEREO && format != GNSSCORR_FMT_RTLSDR)
        return gc_fail(GNSSCORR_EINVAL, "ring_push_packed: format %d", format);
    GcRing *r1 = &ctx->ring[0], *r2 = &ctx->ring[1];
    if (format == GNSSCORR_FMT_STEREO) {
        if (!r1->mem && !r2->mem) return gc_fail(GNSSCORR_ESTATE, "ring_push_packed: no ring created");
        if (r1->mem && r1->dtype != 1) return gc_fail(GNSSCORR_EINVAL, "ring_push_packed: Stereo front end 1 is real (ring 1 must be dtype 1)");
        if (r2->mem && r2->dtype != 2) return gc_fail(GNSSCORR_EINVAL, "ring_push_packed: Stereo front end 2 is IQ (ring 2 must be dtype 2)");
        if ((r1->mem && nsamp > r1->ringlen) || (r2->mem && nsamp > r2->ringlen))
            return gc_fail(GNSSCORR_EINVAL, "ring_push_packed: chunk larger than the ring");
    } else {
        if (!r1->mem || r1->dtype != 2) return gc_fail(GNSSCORR_ESTATE, "ring_push_packed: RTL-SDR feeds ring 1 as IQ (dtype 2)");
        if (nsamp > r1->ringlen) return gc_fail(GNSSCORR_EINVAL, "ring_push_packed: chunk larger than the ring");
    }
    GC_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(ctx->mtx);
    int rc = ingest_init(ctx);
    if (rc) return rc;
    const uint8_t *h = (const uint8_t *)host;
    const uint64_t total = format == GNSSCORR_FMT_STEREO ? nsamp : 2 * nsamp;       // packed bytes
    uint64_t done = 0;
    while (done < total) {
        const uint64_t piece = total - done < GC_PIN_BYTES ? total - done : GC_PIN_BYTES;
        int s;
        rc = ingest_slot(ctx, &s);
        if (rc) return rc;
        memcpy(ctx->pin[s], h + done, piece);
        GC_HIP(hipMemcpyAsync(ctx->dstage[s], ctx->pin[s], piece, hipMemcpyHostToDevice, ctx->stream_in));
        const unsigned blocks = (unsigned)((piece + 255) / 256);
        if (format == GNSSCORR_FMT_STEREO)
            hipLaunchKernelGGL(unpack_stereo_kernel, dim3(blocks), dim3(256), 0, ctx->stream_in, ctx->dstage[s], piece,
                               r1->mem, r1->mem ? r1->ringlen : 1, r1->wrpos + done, r2->mem, r2->mem ? r2->ringlen : 1,
                               r2->wrpos + done);
        else
            hipLaunchKernelGGL(unpack_rtlsdr_kernel, dim3(blocks), dim3(256), 0, ctx->stream_in, ctx->dstage[s], piece,
                               r1->mem, 2 * r1->ringlen, 2 * r1->wrpos + done);
        GC_HIP(hipGetLastError());
        GC_HIP(hipEventRecord(ctx->ev_pin[s], ctx->stream_in));
        ctx->pin_busy[s] = true;
        done += piece;
    }
    GC_HIP(hipEventRecord(ctx->ev_in, ctx->stream_in));
    ctx->in_pending = true;
    if (format == GNSSCORR_FMT_STEREO) {
        if (r1->mem) r1->wrpos += nsamp;
        if (r2->mem) r2->wrpos += nsamp;
    } else {
        r1->wrpos += nsamp;
    }
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_ring_read(gnsscorr_ctx *ctx, int ftype, uint64_t buffloc, int n, void *host)
{
    GcRing *r = ring_of(ctx, ftype);
    if (!r || !r->mem) return gc_fail(GNSSCORR_ESTATE, "ring_read: ring %d not created", ftype);
    if (n <= 0 || (uint64_t)n > r->ringlen || !host) return gc_fail(GNSSCORR_EINVAL, "ring_read: n %d", n);
    GC_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(ctx->mtx);
    if (ctx->in_pending) GC_HIP(hipEventSynchronize(ctx->ev_in));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    // ref src/sdrrcv.c:508-521
    const uint64_t d = (uint64_t)r->dtype, rb = d * r->ringlen, pos = (d * buffloc) % rb, nb = d * (uint64_t)n;
    const uint64_t first = pos + nb <= rb ? nb : rb - pos;
    GC_HIP(hipMemcpy(host, r->mem + pos, first, hipMemcpyDeviceToHost));
    if (first < nb) GC_HIP(hipMemcpy((int8_t *)host + first, r->mem, nb - first, hipMemcpyDeviceToHost));
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_ring_commit(gnsscorr_ctx *ctx, int ftype, uint64_t nsamp)
{
    GcRing *r = ring_of(ctx, ftype);
    if (!r || !r->mem) return gc_fail(GNSSCORR_ESTATE, "ring_commit: ring %d not created", ftype);
    std::lock_guard<std::mutex> lk(ctx->mtx);
    r->wrpos += nsamp;
    return GNSSCORR_OK;
}

extern "C" uint64_t gnsscorr_ring_wrpos(gnsscorr_ctx *ctx, int ftype)
{
    GcRing *r = ring_of(ctx, ftype);
    if (!r) return 0;
    std::lock_guard<std::mutex> lk(ctx->mtx);
    return r->wrpos;
}

extern "C" void *gnsscorr_ring_devptr(gnsscorr_ctx *ctx, int ftype)
{
    GcRing *r = ring_of(ctx, ftype);
    return r ? (void *)r->mem : nullptr;
}

// ---------------------------------------------------------------------------
// channels
// ---------------------------------------------------------------------------
static int upload_channels(gnsscorr_ctx *ctx)
{
    GC_HIP(hipMemcpyAsync(ctx->dchan, ctx->hchan.data(), sizeof(GcChan) * ctx->nch, hipMemcpyHostToDevice,
                          ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GNSSCORR_OK;
}

static void retarget_rings(gnsscorr_ctx *ctx)
{
    if (!ctx->nch || !ctx->dchan) return;
    for (int i = 0; i < ctx->nch; i++) {
        const GcRing &r = ctx->ring[ctx->hdesc[i].ftype - 1];
        ctx->hchan[i].ring = r.mem;
        ctx->hchan[i].ringlen = r.ringlen;
    }
    upload_channels(ctx);
}

extern "C" int gnsscorr_set_channels(gnsscorr_ctx *ctx, int nch, const gnsscorr_chan_t *ch)
{
    if (!ctx || nch <= 0 || !ch) return gc_fail(GNSSCORR_EINVAL, "set_channels: bad arguments");
    GC_HIP(hipSetDevice(ctx->device));
    // validate first: nothing is touched on failure
    for (int i = 0; i < nch; i++) {
        const gnsscorr_chan_t &c = ch[i];
        if (c.dtype != 1 && c.dtype != 2) return gc_fail(GNSSCORR_EINVAL, "channel %d: dtype %d", i, c.dtype);
        if (c.ftype != 1 && c.ftype != 2) return gc_fail(GNSSCORR_EINVAL, "channel %d: ftype %d", i, c.ftype);
        if (c.clen <= 0 || c.clen > 1023 || !c.code)
            return gc_fail(GNSSCORR_EINVAL, "channel %d: code length %d (1..1023 supported)", i, c.clen);
        if (c.corrn < 1 || 1 + 2 * c.corrn > GNSSCORR_MAXTAPS || !c.corrp)
            return gc_fail(GNSSCORR_EINVAL, "channel %d: corrn %d (1..16 supported)", i, c.corrn);
        if (c.corrn != ch[0].corrn)
            return gc_fail(GNSSCORR_EINVAL, "channel %d: corrn differs (one [TRACK] CORRN per receiver)", i);
        if (c.nfreq < 1 || c.nfreq > GNSSCORR_MAXFREQ || !c.freq)
            return gc_fail(GNSSCORR_EINVAL, "channel %d: nfreq %d", i, c.nfreq);
        // tracking takes any period length the int32 accumulators hold; acquisition (gnsscorr_acq_run)
        // additionally needs nsamp <= 16384 for its 32768-point transform and says so itself
        if (c.nsamp <= 0 || c.nsamp > 262144)
            return gc_fail(GNSSCORR_EINVAL, "channel %d: nsamp %d (1..262144 supported)", i, c.nsamp);
        for (int k = 0; k < c.corrn; k++)
            if (c.corrp[k] <= 0 || (k && c.corrp[k] <= c.corrp[k - 1]))
                return gc_fail(GNSSCORR_EINVAL, "channel %d: corrp must be positive and increasing", i);
        if (c.corrp[c.corrn - 1] > 64)
            return gc_fail(GNSSCORR_EINVAL, "channel %d: outermost tap at %d samples (<= 64 supported)", i, c.corrp[c.corrn - 1]);
        const GcRing &r = ctx->ring[c.ftype - 1];
        if (!r.mem) return gc_fail(GNSSCORR_ESTATE, "channel %d: ring %d not created", i, c.ftype);
        if (r.dtype != c.dtype)
            return gc_fail(GNSSCORR_EINVAL, "channel %d: dtype %d but ring %d holds dtype %d", i, c.dtype, c.ftype, r.dtype);
        // a code period (the reference's scratch is nsamp + 100 samples, ref src/sdrtrk.c:23) plus the 16-byte
        // groups around it must fit the ring once; acquisition looks (intg + 1) periods back
        if (r.ringlen < (uint64_t)c.nsamp + 100 + 32 / c.dtype)
            return gc_fail(GNSSCORR_EINVAL, "channel %d: ring %d (%llu samples) is shorter than a code period (%d + 100 + %d)",
                           i, c.ftype, (unsigned long long)r.ringlen, c.nsamp, 32 / c.dtype);
    }
    GC_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) GC_HIP(hipStreamSynchronize(ctx->stream2));
    if (ctx->stream3) GC_HIP(hipStreamSynchronize(ctx->stream3));
    if (ctx->stream4) GC_HIP(hipStreamSynchronize(ctx->stream4));
    gc_acq_free(ctx);
    free_trk_buffers(ctx);
    free_channels(ctx);

    ctx->nch = nch;
    ctx->hdesc.assign(ch, ch + nch);
    ctx->hcode.resize(nch);
    ctx->hfreq.resize(nch);
    ctx->hcorrp.resize(nch);
    ctx->hchan.assign(nch, GcChan());
    std::vector<int8_t> codes((size_t)nch * GC_CODEBLOCK, 0);
    std::vector<double> freqs;
    ctx->ntap = 1 + 2 * ch[0].corrn;
    ctx->smax_max = 0;
    ctx->max_n = 0;
    int ngrid = 0;
    for (int i = 0; i < nch; i++) {
        gnsscorr_chan_t &d = ctx->hdesc[i];
        ctx->hcode[i].assign(d.code, d.code + d.clen);
        ctx->hfreq[i].assign(d.freq, d.freq + d.nfreq);
        ctx->hcorrp[i].assign(d.corrp, d.corrp + d.corrn);
        d.code = ctx->hcode[i].data();
        d.freq = ctx->hfreq[i].data();
        d.corrp = ctx->hcorrp[i].data();
        GcChan &g = ctx->hchan[i];
        gc_build_codeblock(d.code, d.clen, codes.data() + (size_t)i * GC_CODEBLOCK, &g.nedge, &g.pm1);
        g.dtype = d.dtype; g.clen = d.clen; g.nsamp = d.nsamp; g.nsampchip = d.nsampchip;
        g.ntap = 1 + 2 * d.corrn;
        g.smax = d.corrp[d.corrn - 1];
        g.tapoff[0] = 0;
        for (int k = 0; k < d.corrn; k++) { g.tapoff[1 + 2 * k] = -d.corrp[k]; g.tapoff[2 + 2 * k] = d.corrp[k]; }
        g.ti = d.ti; g.f_sf = d.f_sf; g.crate = d.crate; g.ctime = d.ctime;
        g.nfreq = d.nfreq; g.intg = d.intg; g.nfft = d.nfft;
        g.freq_off = (int)freqs.size();
        freqs.insert(freqs.end(), d.freq, d.freq + d.nfreq);
        // acquisition grid = channels that share ring, sample grid and Doppler bins
        g.grid = -1;
        for (int j = 0; j < i && g.grid < 0; j++) {
            const gnsscorr_chan_t &o = ctx->hdesc[j];
            if (o.ftype == d.ftype && o.dtype == d.dtype && o.nsamp == d.nsamp && o.nfreq == d.nfreq &&
                o.intg == d.intg && o.ti == d.ti && o.nfft == d.nfft &&
                !memcmp(o.freq, d.freq, sizeof(double) * d.nfreq))
                g.grid = ctx->hchan[j].grid;
        }
        if (g.grid < 0) g.grid = ngrid++;
        if (g.smax > ctx->smax_max) ctx->smax_max = g.smax;
        if (d.nsamp + 100 > ctx->max_n) ctx->max_n = d.nsamp + 100;   // ref src/sdrtrk.c:23
    }
    GC_HIP(hipMalloc((void **)&ctx->dchan, sizeof(GcChan) * nch));
    GC_HIP(hipMalloc((void **)&ctx->dcodes, codes.size()));
    GC_HIP(hipMalloc((void **)&ctx->dfreqs, sizeof(double) * freqs.size()));
    for (int i = 0; i < 2; i++) {
        GC_HIP(hipMalloc((void **)&ctx->dstate2[i], sizeof(GcTrkState) * nch));
        GC_HIP(hipMemsetAsync(ctx->dstate2[i], 0, sizeof(GcTrkState) * nch, ctx->stream));
    }
    GC_HIP(hipMalloc((void **)&ctx->dloop, sizeof(gnsscorr_loop_t) * nch));
    GC_HIP(hipMemsetAsync(ctx->dloop, 0, sizeof(gnsscorr_loop_t) * nch, ctx->stream));
    GC_HIP(hipMalloc((void **)&ctx->dloopdone, sizeof(int) * nch + sizeof(uint64_t) * nch + 8));
    GC_HIP(hipMemsetAsync(ctx->dloopdone, 0, sizeof(int) * nch + sizeof(uint64_t) * nch + 8, ctx->stream));
    GC_HIP(hipMemcpyAsync(ctx->dcodes, codes.data(), codes.size(), hipMemcpyHostToDevice, ctx->stream));
    GC_HIP(hipMemcpyAsync(ctx->dfreqs, freqs.data(), sizeof(double) * freqs.size(), hipMemcpyHostToDevice,
                          ctx->stream));
    for (int i = 0; i < nch; i++) {
        const GcRing &r = ctx->ring[ctx->hdesc[i].ftype - 1];
        ctx->hchan[i].ring = r.mem;
        ctx->hchan[i].ringlen = r.ringlen;
        ctx->hchan[i].code = ctx->dcodes + (size_t)i * GC_CODEBLOCK;
    }
    return upload_channels(ctx);
}

extern "C" int gnsscorr_num_channels(gnsscorr_ctx *ctx) { return ctx ? ctx->nch : 0; }

static int nco_check(gnsscorr_ctx *ctx);

// ---------------------------------------------------------------------------
// tracking
// ---------------------------------------------------------------------------
extern "C" int gnsscorr_trk_set_state(gnsscorr_ctx *ctx, int ch0, int nch, const gnsscorr_trkstate_t *st)
{
    if (!ctx || !st || ch0 < 0 || nch <= 0 || ch0 + nch > ctx->nch)
        return gc_fail(GNSSCORR_EINVAL, "trk_set_state: channel range [%d,%d) of %d", ch0, ch0 + nch, ctx ? ctx->nch : 0);
    static_assert(sizeof(gnsscorr_trkstate_t) == sizeof(GcTrkState), "state layout");
    GC_HIP(hipSetDevice(ctx->device));
    if (ctx->stream2) GC_HIP(hipStreamSynchronize(ctx->stream2));     // a look-ahead plan may be in flight
    if (ctx->stream3) GC_HIP(hipStreamSynchronize(ctx->stream3));
    if (ctx->stream4) GC_HIP(hipStreamSynchronize(ctx->stream4));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->ahead_valid = false;                                         // ... and is dropped
    ctx->state_touched = true;
    GC_HIP(hipMemcpyAsync(ctx->dstate2[ctx->state_cur] + ch0, st, sizeof(GcTrkState) * nch, hipMemcpyHostToDevice,
                          ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_trk_get_state(gnsscorr_ctx *ctx, int ch0, int nch, gnsscorr_trkstate_t *st)
{
    if (!ctx || !st || ch0 < 0 || nch <= 0 || ch0 + nch > ctx->nch)
        return gc_fail(GNSSCORR_EINVAL, "trk_get_state: channel range [%d,%d) of %d", ch0, ch0 + nch, ctx ? ctx->nch : 0);
    GC_HIP(hipSetDevice(ctx->device));
    if (ctx->stream2) GC_HIP(hipStreamSynchronize(ctx->stream2));
    if (ctx->stream3) GC_HIP(hipStreamSynchronize(ctx->stream3));
    if (ctx->stream4) GC_HIP(hipStreamSynchronize(ctx->stream4));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    GC_HIP(hipMemcpyAsync(st, ctx->dstate2[ctx->state_cur] + ch0, sizeof(GcTrkState) * nch, hipMemcpyDeviceToHost,
                          ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GNSSCORR_OK;
}

static int ensure_trk_buffers(gnsscorr_ctx *ctx, int nepoch)
{
    const size_t units = (size_t)ctx->nch * nepoch;
    if (units <= ctx->plan_cap) return GNSSCORR_OK;
    if (ctx->stream2) GC_HIP(hipStreamSynchronize(ctx->stream2));
    if (ctx->stream3) GC_HIP(hipStreamSynchronize(ctx->stream3));
    if (ctx->stream4) GC_HIP(hipStreamSynchronize(ctx->stream4));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->ahead_valid) {            // a look-ahead plan advanced the state one batch: roll it back
        ctx->ahead_valid = false;
    }
    free_trk_buffers(ctx);
    for (int i = 0; i < 2; i++) GC_HIP(hipMalloc((void **)&ctx->dplan2[i], sizeof(GcTrkPlan) * units));
    for (int i = 0; i < 2; i++) GC_HIP(hipMalloc((void **)&ctx->dspec2[i], sizeof(int) * gc_trk_spec_ints(units)));
    GC_HIP(hipMalloc((void **)&ctx->detab, sizeof(unsigned short) * units * GC_EDGTAB));
    ctx->spec_ahead_valid = false;
    GC_HIP(hipMalloc((void **)&ctx->dcorrI, sizeof(double) * units * ctx->ntap));
    GC_HIP(hipMalloc((void **)&ctx->dcorrQ, sizeof(double) * units * ctx->ntap));
    GC_HIP(hipMalloc((void **)&ctx->dsumI, sizeof(double) * ctx->nch * ctx->ntap));
    GC_HIP(hipMalloc((void **)&ctx->dsumQ, sizeof(double) * ctx->nch * ctx->ntap));
    GC_HIP(hipMalloc((void **)&ctx->dfinish, sizeof(unsigned long long) * ctx->nch * GC_FINISH_SCRATCH));
    GC_HIP(hipMemsetAsync(ctx->dfinish, 0, sizeof(unsigned long long) * ctx->nch * GC_FINISH_SCRATCH, ctx->stream));
    ctx->nseg = 1;
    for (int i = 0; i < ctx->nch; i++) {
        const int s = gc_trk_nseg(ctx->hchan[i].dtype, ctx->max_n);
        if (s > ctx->nseg) ctx->nseg = s;
    }
    for (int i = 0; i < 2; i++)
        GC_HIP(hipMalloc((void **)&ctx->dpartial2[i], sizeof(int) * units * ctx->nseg * 2 * ctx->ntap));
    for (int i = 0; i < 2; i++) {
        GC_HIP(hipMalloc((void **)&ctx->dunit2[i], sizeof(GcTrkUnit) * units));
        GC_HIP(hipMalloc((void **)&ctx->dnsamp2[i], sizeof(int) * units));
        GC_HIP(hipMalloc((void **)&ctx->drounds2[i], sizeof(GcRound) * units * ctx->nseg * GC_MAXR));
        GC_HIP(hipMalloc((void **)&ctx->dsegs2[i], sizeof(GcUnitSegs) * units));
    }
    GC_HIP(hipMalloc((void **)&ctx->dnco_overflow, sizeof(int)));
    GC_HIP(hipMemsetAsync(ctx->dnco_overflow, 0, sizeof(int), ctx->stream));
    GC_HIP(hipMalloc((void **)&ctx->dring_viol, sizeof(int)));
    GC_HIP(hipMemsetAsync(ctx->dring_viol, 0, sizeof(int), ctx->stream));
    ctx->plan_cap = units;
    return GNSSCORR_OK;
}

// (tools/debug) the claims rows of the batch planned last: code rows, then carrier rows, GC_CLAIM_ROW ints each;
// returns the number of rows per NCO
extern "C" int gnsscorr_debug_spec_rows(gnsscorr_ctx *ctx, int *dst, int max_ints)
{
    if (!ctx || !ctx->dspec2[ctx->spec_last_buf] || !ctx->spec_last_units) return -1;
    GC_HIP(hipSetDevice(ctx->device));
    GC_HIP(hipDeviceSynchronize());
    const size_t ints = (size_t)2 * ctx->spec_last_units * GC_CLAIM_ROW;
    if ((size_t)max_ints < ints) return -2;
    GC_HIP(hipMemcpy(dst, ctx->dspec2[ctx->spec_last_buf], ints * sizeof(int), hipMemcpyDeviceToHost));
    return ctx->spec_last_units;
}

extern "C" int gnsscorr_trk_run(gnsscorr_ctx *ctx, int nepoch)
{
    if (!ctx || nepoch <= 0) return gc_fail(GNSSCORR_EINVAL, "trk_run: nepoch %d", nepoch);
    if (!ctx->nch) return gc_fail(GNSSCORR_ESTATE, "trk_run: no channels set");
    GC_HIP(hipSetDevice(ctx->device));
    int rc = ensure_trk_buffers(ctx, nepoch);
    if (rc) return rc;
    // the rings' write positions and the ingest fence together, under the lock (a grabber thread may be pushing): the
    // positions handed to the ring check cover only samples whose transfer the compute stream is ordered behind
    uint64_t wr0, wr1;
    {
        std::lock_guard<std::mutex> lk(ctx->mtx);
        wr0 = ctx->ring[0].wrpos;
        wr1 = ctx->ring[1].wrpos;
        rc = gc_ingest_fence(ctx);
        if (rc) return rc;
    }
    // ---- planner: use the look-ahead plan if it matches, else plan now ----
    // plan = the sequential NCO chain per channel (discovery pass + chain), on the planner stream into the
    // slot's plan buffer; ev_plan[slot] marks it ready
    hipStream_t ps = ctx->stream2 ? ctx->stream2 : ctx->stream;
    auto plan_into = [&](int s) -> int {
        // the slot's partial sums were last read by the finish of two batches ago: ordering the plan
        // behind it lets ev_plan[s] stand for "slot s is free and planned" on the main stream
        if (ctx->stream2 && ctx->fin_pending[s]) GC_HIP(hipStreamWaitEvent(ps, ctx->ev_fin[s], 0));
        // claims of this batch: discovered ahead (while the previous batch's chain ran, from that batch's
        // input state) if nothing has touched the state since, else discovered now from the state itself
        const GcTrkState *sin = ctx->dstate2[ctx->state_cur];
        int buf;
        if (ctx->stream4 && ctx->spec_pending) GC_HIP(hipStreamWaitEvent(ps, ctx->ev_spec, 0));
        if (ctx->spec_ahead_valid && !ctx->state_touched && ctx->spec_ahead_nepoch == nepoch && ctx->spec_ahead_state == (const void *)sin) {
            buf = ctx->spec_ahead_buf;
        } else {
            buf = 0;
            GcTimed t(ctx, "trk_spec", ps);
            int r2 = gc_launch_trk_spec(ps, ctx->dchan, sin, ctx->nch, nepoch, ctx->dspec2[buf], 0);
            if (r2) return r2;
        }
        ctx->spec_ahead_valid = false;
        ctx->spec_pending = false;
        ctx->spec_last_buf = buf;
        ctx->spec_last_units = ctx->nch * nepoch;
        if (ctx->stream4) {
            // the next batch's claims, from the same input state, beside this batch's chain (the other buffer
            // was last read by the chain in front of this one on the planner stream)
            GC_HIP(hipEventRecord(ctx->ev_chain, ps));
            GC_HIP(hipStreamWaitEvent(ctx->stream4, ctx->ev_chain, 0));
            GcTimed t(ctx, "trk_spec", ctx->stream4);
            int r2 = gc_launch_trk_spec(ctx->stream4, ctx->dchan, sin, ctx->nch, nepoch, ctx->dspec2[buf ^ 1], nepoch);
            if (r2) return r2;
            GC_HIP(hipEventRecord(ctx->ev_spec, ctx->stream4));
            ctx->spec_pending = true;
            ctx->spec_ahead_valid = true;
            ctx->spec_ahead_buf = buf ^ 1;
            ctx->spec_ahead_nepoch = nepoch;
            ctx->spec_ahead_state = (const void *)ctx->dstate2[ctx->state_cur ^ 1];
        }
        {
            GcTimed t(ctx, "trk_plan", ps);
            int r2 = gc_launch_trk_plan(ps, ctx->dchan, sin, ctx->dstate2[ctx->state_cur ^ 1],
                                        ctx->dplan2[s], ctx->nch, nepoch, ctx->dspec2[buf]);
            if (r2) return r2;
        }
        if (ctx->stream2) GC_HIP(hipEventRecord(ctx->ev_plan[s], ps));
        return 0;
    };
    const int slot = ctx->plan_slot;
    if (!(ctx->ahead_valid && ctx->ahead_nepoch == nepoch)) {
        if (ctx->ahead_valid) {        // planned for another batch length: the committed state is untouched
            GC_HIP(hipStreamSynchronize(ps));
            ctx->ahead_valid = false;
        }
        if (ctx->stream2) {            // order after whatever the main stream did to the state / buffers
            GC_HIP(hipEventRecord(ctx->ev_used[slot], ctx->stream));
            GC_HIP(hipStreamWaitEvent(ps, ctx->ev_used[slot], 0));
        }
        rc = plan_into(slot);
        if (rc) return rc;
    }
    if (ctx->stream2) GC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_plan[slot], 0));
    // the per-unit constants and NCO tables of the planned periods: on the main stream, in front of the
    // correlator that reads them (the planner stream carries nothing but the sequential chain)
    {
        const int s = slot;
        GcTimed t(ctx, "trk_expand");
        int r2 = gc_launch_trk_expand(ctx->stream, ctx->dchan, ctx->dplan2[s], ctx->dunit2[s], ctx->dsegs2[s], ctx->dnsamp2[s],
                                      ctx->nch, nepoch, ctx->drounds2[s], ctx->nseg, ctx->max_n, ctx->dnco_overflow);
        if (r2) return r2;
    }
    // the planned periods against what the rings hold now
    // (the write positions travel as kernel arguments: no copy, no host synchronisation per batch)
    rc = gc_launch_trk_ringcheck(ctx->stream, ctx->dchan, ctx->dplan2[slot], (const int8_t *)ctx->ring[0].mem, wr0, wr1, ctx->nch,
                                 nepoch, ctx->dring_viol);
    if (rc) return rc;
    bool have[3] = {false, false, false};
    for (int i = 0; i < ctx->nch; i++) have[ctx->hchan[i].dtype] = true;
    if (!ctx->stream2 && ctx->fin_pending[slot]) GC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_fin[slot], 0));
    {
        // start samples of the periods' chip edges, for the correlator's look-up phase
        GcTimed t(ctx, "trk_edges");
        rc = gc_launch_trk_edges(ctx->stream, ctx->dchan, ctx->dunit2[slot], ctx->dsegs2[slot], ctx->detab, ctx->nch, nepoch);
        if (rc) return rc;
    }
    for (int dtype = 1; dtype <= 2; dtype++) {
        if (!have[dtype]) continue;
        GcTimed t(ctx, "trk_corr");
        rc = gc_launch_trk_corr(ctx->stream, ctx->dchan, ctx->dunit2[slot], ctx->dsegs2[slot], ctx->drounds2[slot], ctx->dpartial2[slot],
                                ctx->nch, nepoch, ctx->nseg, ctx->ntap, dtype, ctx->ntap, ctx->max_n, ctx->smax_max, ctx->detab);
        if (rc) return rc;
    }
    if (ctx->stream2) GC_HIP(hipEventRecord(ctx->ev_corr[slot], ctx->stream));     // slot buffers consumed, partials ready
    // every launch of the batch was issued: only now the plan's output state becomes the committed one
    ctx->ahead_valid = false;
    ctx->state_cur ^= 1;
    ctx->plan_slot ^= 1;
    // ---- look ahead: plan the next batch of the same length while this one is correlated ----
    if (ctx->stream2 && !ctx->state_touched) {
        const int ns = ctx->plan_slot;
        GC_HIP(hipStreamWaitEvent(ps, ctx->ev_corr[ns], 0));        // its previous contents were consumed
        rc = plan_into(ns);
        if (rc) return rc;
        ctx->ahead_valid = true;
        ctx->ahead_nepoch = nepoch;
    }
    ctx->state_touched = false;
    // ---- finish on its own stream: the main stream goes straight from this batch's correlator to the
    // next one's, the outputs become valid at ev_fin[slot] ----
    {
        hipStream_t fs = ctx->stream3 ? ctx->stream3 : ctx->stream;
        if (ctx->stream3) GC_HIP(hipStreamWaitEvent(fs, ctx->ev_corr[slot], 0));
        GcTimed t(ctx, "trk_finish", fs);
        rc = gc_launch_trk_finish(fs, ctx->dpartial2[slot], ctx->dcorrI, ctx->dcorrQ, ctx->dsumI, ctx->dsumQ,
                                  ctx->dfinish, ctx->nch, nepoch, ctx->nseg, ctx->ntap);
        if (rc) return rc;
        if (ctx->stream3) {
            GC_HIP(hipEventRecord(ctx->ev_fin[slot], fs));
            ctx->fin_pending[slot] = true;
        }
    }
    ctx->last_slot = slot;
    ctx->last_nepoch = nepoch;
    ctx->last_loop_nper = 0;
    return GNSSCORR_OK;
}

// ---------------------------------------------------------------------------
// tracking, closed loop
// ---------------------------------------------------------------------------
static int loop_quiesce(gnsscorr_ctx *ctx)
{
    if (ctx->stream2) GC_HIP(hipStreamSynchronize(ctx->stream2));     // a look-ahead plan may be in flight
    if (ctx->stream3) GC_HIP(hipStreamSynchronize(ctx->stream3));
    if (ctx->stream4) GC_HIP(hipStreamSynchronize(ctx->stream4));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->ahead_valid = false;                                         // ... and is dropped
    ctx->state_touched = true;
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_loop_set(gnsscorr_ctx *ctx, int ch0, int nch, const gnsscorr_loop_t *lp)
{
    if (!ctx || !lp || ch0 < 0 || nch <= 0 || ch0 + nch > ctx->nch)
        return gc_fail(GNSSCORR_EINVAL, "loop_set: channel range [%d,%d) of %d", ch0, ch0 + nch, ctx ? ctx->nch : 0);
    for (int i = 0; i < nch; i++) {
        const gnsscorr_loop_t &l = lp[i];
        const int ntap = ctx->hchan[ch0 + i].ntap;
        if (l.ne < 0 || l.ne >= ntap || l.nl < 0 || l.nl >= ntap)
            return gc_fail(GNSSCORR_EINVAL, "loop_set: channel %d: early/late tap index %d/%d of %d taps", ch0 + i, l.ne, l.nl, ntap);
        if (l.loopms < 1 || l.rate < 1 || l.rate > 20)
            return gc_fail(GNSSCORR_EINVAL, "loop_set: channel %d: loopms %d, rate %d (rate 1..20)", ch0 + i, l.loopms, l.rate);
    }
    GC_HIP(hipSetDevice(ctx->device));
    { int rc = loop_quiesce(ctx); if (rc) return rc; }
    GC_HIP(hipMemcpyAsync(ctx->dloop + ch0, lp, sizeof(gnsscorr_loop_t) * nch, hipMemcpyHostToDevice, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nch; i++) {
        if (lp[i].loopms > ctx->loop_kmax) ctx->loop_kmax = lp[i].loopms < GC_STEP_KMAX ? lp[i].loopms : GC_STEP_KMAX;
        if (lp[i].flagsync) ctx->loop_sync_hint = true;
    }
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_loop_get(gnsscorr_ctx *ctx, int ch0, int nch, gnsscorr_loop_t *lp)
{
    if (!ctx || !lp || ch0 < 0 || nch <= 0 || ch0 + nch > ctx->nch)
        return gc_fail(GNSSCORR_EINVAL, "loop_get: channel range [%d,%d) of %d", ch0, ch0 + nch, ctx ? ctx->nch : 0);
    GC_HIP(hipSetDevice(ctx->device));
    { int rc = loop_quiesce(ctx); if (rc) return rc; }
    GC_HIP(hipMemcpyAsync(lp, ctx->dloop + ch0, sizeof(gnsscorr_loop_t) * nch, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GNSSCORR_OK;
}

// the step buffers: one filter interval (GC_STEP_KMAX periods at most) per channel
static int ensure_step_buffers(gnsscorr_ctx *ctx)
{
    if (ctx->dstep_meta) return GNSSCORR_OK;
    int nseg = 1;
    for (int i = 0; i < ctx->nch; i++) {
        const int s = gc_step_nseg(ctx->hchan[i].dtype, ctx->max_n);
        if (s > nseg) nseg = s;
    }
    if (nseg > 64) return gc_fail(GNSSCORR_EINVAL, "trk_run_loop: period of %d samples too long (%d rounds, 64 at most)", ctx->max_n, nseg);
    const size_t units = (size_t)ctx->nch * GC_STEP_KMAX;
    GC_HIP(hipMalloc((void **)&ctx->dstep_meta, sizeof(GcStepMeta) * ctx->nch));
    GC_HIP(hipMalloc((void **)&ctx->dstep_unit, sizeof(GcTrkUnit) * units));
    GC_HIP(hipMalloc((void **)&ctx->dstep_segs, sizeof(GcUnitSegs) * units));
    GC_HIP(hipMalloc((void **)&ctx->dstep_rounds, sizeof(GcRound) * units * nseg * 4));        // four rounds (one per wavefront) per workgroup
    GC_HIP(hipMalloc((void **)&ctx->dstep_partial, sizeof(int) * units * nseg * 2 * ctx->ntap));
    if (!ctx->hostflags) {
        GC_HIP(hipHostMalloc((void **)&ctx->hostflags, 64, hipHostMallocMapped));
        GC_HIP(hipHostGetDevicePointer((void **)&ctx->hostflags_dev, ctx->hostflags, 0));
    }
    ctx->step_nseg = nseg;
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_trk_run_loop(gnsscorr_ctx *ctx, int nperiod)
{
    if (!ctx || nperiod <= 0) return gc_fail(GNSSCORR_EINVAL, "trk_run_loop: nperiod %d", nperiod);
    if (!ctx->nch) return gc_fail(GNSSCORR_ESTATE, "trk_run_loop: no channels set");
    GC_HIP(hipSetDevice(ctx->device));
    int rc = ensure_trk_buffers(ctx, nperiod);
    if (rc) return rc;
    rc = ensure_step_buffers(ctx);
    if (rc) return rc;
    // the look-ahead planner of the batched interface works on the same state: stop it, drop its plan
    if (ctx->ahead_valid || ctx->fin_pending[0] || ctx->fin_pending[1]) { rc = loop_quiesce(ctx); if (rc) return rc; }
    ctx->ahead_valid = false;
    ctx->state_touched = true;
    const size_t units = (size_t)ctx->nch * nperiod;
    if (units > ctx->looplog_cap) {
        GC_HIP(hipStreamSynchronize(ctx->stream));
        hipFree(ctx->dlooplog); ctx->dlooplog = nullptr; ctx->looplog_cap = 0;
        GC_HIP(hipMalloc((void **)&ctx->dlooplog, sizeof(gnsscorr_trklog_t) * units));
        ctx->looplog_cap = units;
    }
    GC_HIP(hipMemsetAsync(ctx->dlooplog, 0, sizeof(gnsscorr_trklog_t) * units, ctx->stream));
    GC_HIP(hipMemsetAsync(ctx->dcorrI, 0, sizeof(double) * units * ctx->ntap, ctx->stream));
    GC_HIP(hipMemsetAsync(ctx->dcorrQ, 0, sizeof(double) * units * ctx->ntap, ctx->stream));
    GC_HIP(hipMemsetAsync(ctx->dnsamp2[0], 0, sizeof(int) * units, ctx->stream));
    GC_HIP(hipMemsetAsync(ctx->dstep_meta, 0, sizeof(GcStepMeta) * ctx->nch, ctx->stream));
    // write position of each channel's ring (ref src/sdrtrk.c:26-28: fendbuffsize*buffcnt), read together with the
    // ingest fence under the lock: the positions cover only samples whose transfer the compute stream is ordered behind
    std::vector<uint64_t> wp(ctx->nch);
    {
        std::lock_guard<std::mutex> lk(ctx->mtx);
        for (int i = 0; i < ctx->nch; i++) wp[i] = ctx->ring[ctx->hdesc[i].ftype - 1].wrpos;
        rc = gc_ingest_fence(ctx);
        if (rc) return rc;
    }
    uint64_t *dwp = reinterpret_cast<uint64_t *>(ctx->dloopdone + ctx->nch + (ctx->nch & 1));
    GC_HIP(hipMemcpyAsync(dwp, wp.data(), sizeof(uint64_t) * ctx->nch, hipMemcpyHostToDevice, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));          // (wp is a local; the flags below are host memory)
    ctx->hostflags[0] = 0;
    ctx->hostflags[1] = ctx->loop_sync_hint ? 1u : 0u;     // (some channel is known to be synchronised: steps of loopms periods from the start)
    bool have[3] = {false, false, false};
    for (int i = 0; i < ctx->nch; i++) have[ctx->hchan[i].dtype] = true;
    // One step = tail (close the previous interval, plan the next) + correlator.  Every step advances every channel
    // that still has work by at least one period, so nperiod steps always suffice; channels whose nav bit is
    // synchronised advance by up to loopms periods per step.  The host keeps a bounded number of steps ahead of the
    // device and stops as soon as the device says every channel is done (a pinned word the tail kernel updates).
    const int kmax = ctx->loop_kmax < 1 ? 1 : ctx->loop_kmax;
    const int BURST = 4, AHEAD = 3;
    hipEvent_t ev[AHEAD] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < AHEAD; i++) GC_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    auto cleanup = [&]() { for (int i = 0; i < AHEAD; i++) if (ev[i]) hipEventDestroy(ev[i]); };
    int steps = 0, burst = 0;
    bool done = false;
    while (!done && steps <= nperiod) {
        if (burst >= AHEAD) {                           // at most AHEAD bursts in flight
            hipError_t e = hipEventSynchronize(ev[burst % AHEAD]);
            if (e != hipSuccess) { cleanup(); return gc_fail_hip(e, "hipEventSynchronize", __FILE__, __LINE__); }
            if (ctx->hostflags[0] >= (unsigned)ctx->nch) break;
        }
        for (int b = 0; b < BURST && steps <= nperiod; b++, steps++) {
            // periods per step: 1 while no channel is synchronised (a performance hint only: the tail never plans more
            // than kcap periods, and any kcap >= 1 is correct)
            const int kcap = ctx->hostflags[1] ? kmax : 1;
            {
                GcTimed t(ctx, "trk_step_tail");
                rc = gc_launch_step_tail(ctx->stream, ctx->dchan, ctx->dstate2[ctx->state_cur], ctx->dloop, ctx->dstep_meta, dwp,
                                         ctx->dstep_partial, ctx->dstep_unit, ctx->dstep_segs, ctx->dstep_rounds, ctx->dcorrI,
                                         ctx->dcorrQ, ctx->dnsamp2[0], ctx->dlooplog, ctx->dloopdone, ctx->dnco_overflow,
                                         ctx->hostflags_dev, ctx->nch, nperiod, ctx->step_nseg, ctx->ntap, ctx->max_n, kcap, 1);
                if (rc) { cleanup(); return rc; }
            }
            for (int dtype = 1; dtype <= 2; dtype++) {
                if (!have[dtype]) continue;
                GcTimed t(ctx, "trk_step_corr");
                rc = gc_launch_step_corr(ctx->stream, ctx->dchan, ctx->dstep_meta, ctx->dstep_unit, ctx->dstep_segs, ctx->dstep_rounds,
                                         ctx->dstep_partial, ctx->nch, kcap, ctx->step_nseg, dtype, ctx->ntap, ctx->max_n,
                                         ctx->smax_max);
                if (rc) { cleanup(); return rc; }
            }
        }
        hipError_t e = hipEventRecord(ev[burst % AHEAD], ctx->stream);
        if (e != hipSuccess) { cleanup(); return gc_fail_hip(e, "hipEventRecord", __FILE__, __LINE__); }
        burst++;
    }
    // close whatever the last correlator launch produced
    rc = gc_launch_step_tail(ctx->stream, ctx->dchan, ctx->dstate2[ctx->state_cur], ctx->dloop, ctx->dstep_meta, dwp, ctx->dstep_partial,
                             ctx->dstep_unit, ctx->dstep_segs, ctx->dstep_rounds, ctx->dcorrI, ctx->dcorrQ, ctx->dnsamp2[0],
                             ctx->dlooplog, ctx->dloopdone, ctx->dnco_overflow, ctx->hostflags_dev, ctx->nch, nperiod, ctx->step_nseg,
                             ctx->ntap, ctx->max_n, 1, 0);
    cleanup();
    if (rc) return rc;
    if (ctx->hostflags[1]) ctx->loop_sync_hint = true;
    ctx->last_slot = 0;
    ctx->fin_pending[0] = ctx->fin_pending[1] = false;
    ctx->last_nepoch = nperiod;
    ctx->last_loop_nper = nperiod;
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_trk_fetch_log(gnsscorr_ctx *ctx, gnsscorr_trklog_t *log, int *ndone)
{
    if (!ctx || !ctx->last_loop_nper) return gc_fail(GNSSCORR_ESTATE, "trk_fetch_log: no completed trk_run_loop");
    GC_HIP(hipSetDevice(ctx->device));
    const size_t units = (size_t)ctx->nch * ctx->last_loop_nper;
    if (log) GC_HIP(hipMemcpyAsync(log, ctx->dlooplog, sizeof(gnsscorr_trklog_t) * units, hipMemcpyDeviceToHost, ctx->stream));
    if (ndone) GC_HIP(hipMemcpyAsync(ndone, ctx->dloopdone, sizeof(int) * ctx->nch, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return nco_check(ctx);
}

// Units whose NCO piece tables overflowed (a code step that wraps the code more than ~twice per call, a
// carrier that visits more than GC_NCAR binades) were not correlated: say so instead of handing out zeros.
static int nco_check(gnsscorr_ctx *ctx)
{
    int n = 0, v = 0;
    GC_HIP(hipMemcpy(&v, ctx->dring_viol, sizeof(int), hipMemcpyDeviceToHost));
    if (v) {
        GC_HIP(hipMemset(ctx->dring_viol, 0, sizeof(int)));
        return gc_fail(GNSSCORR_ESTATE, "tracking: %d (channel, period) units lay outside what the IF ring holds "
                       "(beyond the write position, or overwritten since): their sums are not the stream's", v);
    }
    GC_HIP(hipMemcpy(&n, ctx->dnco_overflow, sizeof(int), hipMemcpyDeviceToHost));
    if (!n) return GNSSCORR_OK;
    GC_HIP(hipMemset(ctx->dnco_overflow, 0, sizeof(int)));
    return gc_fail(GNSSCORR_EINVAL, "tracking: %d (channel, period) units need more NCO pieces than the tables hold "
                   "(more than two code periods per call, or a carrier crossing more than %d binades); their sums are zero",
                   n, GC_NCAR);
}

// trk.II <- correlator's QQ (sum dataQ*code), trk.QQ <- its II: ref src/sdrtrk.c:42
extern "C" int gnsscorr_trk_fetch(gnsscorr_ctx *ctx, double *trkII, double *trkQQ, int *nsamp_out)
{
    if (!ctx || !ctx->last_nepoch) return gc_fail(GNSSCORR_ESTATE, "trk_fetch: no completed trk_run");
    GC_HIP(hipSetDevice(ctx->device));
    { int rc = outputs_ready(ctx); if (rc) return rc; }
    const size_t units = (size_t)ctx->nch * ctx->last_nepoch;
    if (trkII)
        GC_HIP(hipMemcpyAsync(trkII, ctx->dcorrQ, sizeof(double) * units * ctx->ntap, hipMemcpyDeviceToHost, ctx->stream));
    if (trkQQ)
        GC_HIP(hipMemcpyAsync(trkQQ, ctx->dcorrI, sizeof(double) * units * ctx->ntap, hipMemcpyDeviceToHost, ctx->stream));
    if (nsamp_out)
        GC_HIP(hipMemcpyAsync(nsamp_out, ctx->dnsamp2[ctx->last_slot], sizeof(int) * units, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return nco_check(ctx);
}

extern "C" int gnsscorr_trk_fetch_sums(gnsscorr_ctx *ctx, double *sumI, double *sumQ)
{
    if (!ctx || !ctx->last_nepoch) return gc_fail(GNSSCORR_ESTATE, "trk_fetch_sums: no completed trk_run");
    if (ctx->last_loop_nper) return gc_fail(GNSSCORR_ESTATE, "trk_fetch_sums: the last run was closed loop (its sums are in gnsscorr_loop_get)");
    GC_HIP(hipSetDevice(ctx->device));
    { int rc = outputs_ready(ctx); if (rc) return rc; }
    const size_t n = (size_t)ctx->nch * ctx->ntap;
    if (sumI) GC_HIP(hipMemcpyAsync(sumI, ctx->dsumQ, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (sumQ) GC_HIP(hipMemcpyAsync(sumQ, ctx->dsumI, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_trk_devptrs(gnsscorr_ctx *ctx, void **trkII, void **trkQQ)
{
    if (!ctx || !ctx->dcorrI) return gc_fail(GNSSCORR_ESTATE, "trk_devptrs: no trk_run yet");
    { int rc = outputs_ready(ctx); if (rc) return rc; }     // work queued on the context stream after this sees them
    if (trkII) *trkII = ctx->dcorrQ;
    if (trkQQ) *trkQQ = ctx->dcorrI;
    return GNSSCORR_OK;
}

// ---------------------------------------------------------------------------
// timing
// ---------------------------------------------------------------------------
static void drain_timers(gnsscorr_ctx *ctx)
{
    for (auto &kv : ctx->timers) {
        for (auto &p : kv.second.pending) {
            float ms = 0.f;
            if (hipEventSynchronize(p.second) == hipSuccess &&
                hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                kv.second.total_ms += ms;
                kv.second.launches++;
            }
            hipEventDestroy(p.first);
            hipEventDestroy(p.second);
        }
        kv.second.pending.clear();
    }
}

extern "C" int gnsscorr_timing_enable(gnsscorr_ctx *ctx, int on)
{
    if (!ctx) return gc_fail(GNSSCORR_EINVAL, "null context");
    ctx->timing = on == 2 ? 2 : (on != 0);
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_timing_reset(gnsscorr_ctx *ctx)
{
    if (!ctx) return gc_fail(GNSSCORR_EINVAL, "null context");
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    drain_timers(ctx);
    ctx->timers.clear();
    return GNSSCORR_OK;
}

extern "C" int gnsscorr_timing_read(gnsscorr_ctx *ctx, const char *kernel, double *total_ms, int *launches)
{
    if (!ctx || !kernel) return gc_fail(GNSSCORR_EINVAL, "timing_read: bad arguments");
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    drain_timers(ctx);
    auto it = ctx->timers.find(kernel);
    if (total_ms) *total_ms = it == ctx->timers.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == ctx->timers.end() ? 0 : it->second.launches;
    return GNSSCORR_OK;
}

// ---------------------------------------------------------------------------
// process-wide context for the reference-named per-call symbols
// ---------------------------------------------------------------------------
extern "C" gnsscorr_ctx *gnsscorr_default_ctx(void)
{
    static std::mutex m;
    static gnsscorr_ctx *g = nullptr;
    std::lock_guard<std::mutex> lk(m);
    if (!g) {
        const char *e = getenv("GNSSCORR_DEVICE");
        int dev = e ? atoi(e) : 0;
        if (gnsscorr_create(&g, dev, nullptr) != GNSSCORR_OK) {
            fprintf(stderr, "error: gnsscorr: %s\n", gnsscorr_last_error());
            g = nullptr;
        }
    }
    return g;
}
