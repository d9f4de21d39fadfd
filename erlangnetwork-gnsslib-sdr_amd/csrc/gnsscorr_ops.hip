// gnsscorr_ops.hip -- the reference's op-level helpers (src/sdrcmn.c) and the
// acquisition entry points (src/sdracq.c) as per-call C symbols on the GPU.
//
// These keep the reference signatures: operands are host arrays, one call =
// one reference call.  They exist so that code written against src/sdr.h
// links unchanged; the batched interface in gnsscorr.h is the fast path.
//
// Transforms of arbitrary length m (the reference uses m = 2*nsamp = 32736 =
// 2^5*3*11*31 with FFTW) are evaluated here as a direct DFT with an fp64
// rotating twiddle: O(m^2) work, but exact to float rounding for every m and
// only a few hundred microseconds at m = 32736 on an MI355X.  The production
// acquisition path (gnsscorr_acq.hip) never uses it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unistd.h>
#include <vector>

#include "../../include/sdr_compat.h"
#include "gnsscorr_ctx.h"

#define SDRPRINTF printf

namespace {

__constant__ signed char oCos32[32] = {32, 31, 30, 27, 23, 18, 12, 6, 0, -6, -12, -18, -23, -27, -30, -31,
                                       -32, -31, -30, -27, -23, -18, -12, -6, 0, 6, 12, 18, 23, 27, 30, 31};
__constant__ signed char oSin32[32] = {0, 6, 12, 18, 23, 27, 30, 31, 32, 31, 30, 27, 23, 18, 12, 6,
                                       0, -6, -12, -18, -23, -27, -30, -31, -32, -31, -30, -27, -23, -18, -12, -6};

// out[f] = sum_j in[j] exp(sign*2*pi*i*f*j/m), one output per lane, inputs
// staged through LDS, twiddle advanced by rotation in fp64 and re-seeded from
// sincospi every tile so that the drift stays below float resolution.
#define DFT_TILE 1024
__global__ __launch_bounds__(256) void dft_direct_kernel(const float2 *__restrict__ in,
                                                         float2 *__restrict__ out, int m, int sign)
{
    __shared__ float2 tile[DFT_TILE];
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const double fm = (double)(f < m ? f : 0) / (double)m;
    double ws, wc;
    sincospi(2.0 * fm * (double)sign, &ws, &wc);        // w = exp(sign 2 pi i f/m)
    double ar = 0.0, ai = 0.0;
    for (int j0 = 0; j0 < m; j0 += DFT_TILE) {
        const int nt = min(DFT_TILE, m - j0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += blockDim.x) tile[i] = in[j0 + i];
        __syncthreads();
        // exact phase of the tile's first element: (f*j0 mod m)/m
        const long long r = ((long long)f * (long long)j0) % (long long)m;
        double ts, tc;
        sincospi(2.0 * (double)sign * (double)r / (double)m, &ts, &tc);
        for (int i = 0; i < nt; i++) {
            const double xr = (double)tile[i].x, xi = (double)tile[i].y;
            ar = fma(xr, tc, fma(-xi, ts, ar));
            ai = fma(xr, ts, fma(xi, tc, ai));
            const double nc = fma(tc, wc, -ts * ws), ns = fma(tc, ws, ts * wc);
            tc = nc; ts = ns;
        }
    }
    if (f < m) out[f] = make_float2((float)ar, (float)ai);
}

// ref src/sdrcmn.c:236-240 : a <- -a*conj(b)
__global__ void conjmul_kernel(float2 *__restrict__ a, const float2 *__restrict__ b, int m)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float2 p = a[i], q = b[i];
    float2 r;
    r.x = -p.x * q.x - p.y * q.y;
    r.y = p.x * q.y - p.y * q.x;
    a[i] = r;
}

// ref src/sdrcmn.c:244-251 / :268-275 : conv[i] (+)= (re^2+im^2)/scale2
__global__ void power_kernel(const float2 *__restrict__ a, double *__restrict__ conv, int n, float scale2,
                             int flagsum)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 p = a[i];
    const double v = (double)((p.x * p.x + p.y * p.y) / scale2);
    conv[i] = flagsum ? conv[i] + v : v;
}

// NCO piece tables of one mixcarr() / rescode() call (gnsscorr_nco.h), built by one device thread
#define GC_OPSEG 256
struct GcOpTables {
    int ncar, ncode, overflow, pad;
    int k0[GC_OPSEG];
    GcCarSeg car[GC_OPSEG];
    GcCodeSeg code[GC_OPSEG];
};

__global__ void op_tables_kernel(GcOpTables *__restrict__ t, int n, double phi0, double freq, double ti,
                                 int nt, double coff, int smax, double ci, int len)
{
    if (blockIdx.x || threadIdx.x) return;
    t->ncar = t->ncode = t->overflow = 0;
    if (n > 0) {
        GcCarTable ct{t->k0, t->car, GC_OPSEG, 0, 0};
        gc_carrier_walk(gc_carrier_phis(phi0), gc_carrier_ps(freq, ti), n, ct);
        t->ncar = ct.n;
        t->overflow |= ct.overflow;
    }
    if (nt > 0) {
        GcCodeTable dt{t->code, GC_OPSEG, 0, 0};
        gc_code_walk(gc_code_start(coff, smax, ci, len), ci, len, nt, dt);
        t->ncode = dt.n;
        t->overflow |= dt.overflow;
    }
}

// ref src/sdrcmn.c:633-669 (samples); optional float2 output scaled as cpxcpx() does (ref
// src/sdrcmn.c:185-195)
__global__ void mix_kernel(const int8_t *__restrict__ data, int dtype, int n, const GcOpTables *__restrict__ t,
                           short *__restrict__ I, short *__restrict__ Q,
                           float2 *__restrict__ cpx, float scale)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int idx = gc_carrier_idx_at(t->k0, t->car, t->ncar, k);
    const int c = oCos32[idx], s = oSin32[idx];
    int vi, vq;
    if (dtype == 2) {
        const int d0 = data[2 * k], d1 = data[2 * k + 1];
        vi = c * d0 - s * d1;
        vq = s * d0 + c * d1;
    } else {
        const int d0 = data[k];
        vi = c * d0;
        vq = s * d0;
    }
    if (I) I[k] = (short)vi;
    if (Q) Q[k] = (short)vq;
    if (cpx) cpx[k] = make_float2((float)vi * scale, (float)vq * scale);
}

// ref src/sdrcmn.c:608-621
__global__ void rescode_kernel(const short *__restrict__ code, const GcOpTables *__restrict__ t, int nt,
                               short *__restrict__ rcode)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nt) return;
    rcode[j] = code[gc_code_chip_at(t->code, t->ncode, j, nullptr)];
}

__global__ void cpxcpx_kernel(const short *__restrict__ I, const short *__restrict__ Q, float scale, int n,
                              float2 *__restrict__ cpx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cpx[i] = make_float2((float)I[i] * scale, Q ? (float)Q[i] * scale : 0.0f);
}

// maxvd()/meanvd() over one array (ref src/sdrcmn.c:461-497): one workgroup.
// out[0] = max (element 0 seeds it), out[1] = sum outside, out[2] = #inside, iout[0] = argmax
__global__ __launch_bounds__(1024) void vstat_kernel(const double *__restrict__ d, int n, int exs, int exe,
                                                     double *__restrict__ out, int *__restrict__ iout)
{
    __shared__ double smax[16], ssum[16];
    __shared__ int sidx[16], scnt[16];
    const int tid = threadIdx.x;
    double mx = -INFINITY, sum = 0.0;
    int mi = 0x7fffffff, cnt = 0;
    for (int i = tid; i < n; i += 1024) {
        const bool outside = (exs <= exe && (i < exs || i > exe)) || (exs > exe && (i < exs && i > exe));
        const double v = d[i];
        if (outside) sum += v; else cnt++;
        if ((outside || i == 0) && (v > mx || (v == mx && i < mi))) { mx = v; mi = i; }
    }
    // maxvd keeps data[0] unless a later candidate is strictly larger
    for (int o = 32; o > 0; o >>= 1) {
        const double om = __shfl_xor(mx, o, 64), os = __shfl_xor(sum, o, 64);
        const int oi = __shfl_xor(mi, o, 64), oc = __shfl_xor(cnt, o, 64);
        if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
        sum += os; cnt += oc;
    }
    if ((tid & 63) == 0) { smax[tid >> 6] = mx; sidx[tid >> 6] = mi; ssum[tid >> 6] = sum; scnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++) {
            if (smax[w] > mx || (smax[w] == mx && sidx[w] < mi)) { mx = smax[w]; mi = sidx[w]; }
            sum += ssum[w]; cnt += scnt[w];
        }
        out[0] = mx; out[1] = sum; out[2] = (double)cnt; iout[0] = mi;
    }
}

struct Scratch {           // per-call device scratch of the default context
    void *p[8] = {nullptr};
    size_t cap[8] = {0};
};
Scratch g_s;

void *need(int slot, size_t bytes)
{
    if (bytes <= g_s.cap[slot]) return g_s.p[slot];
    if (g_s.p[slot]) hipFree(g_s.p[slot]);
    g_s.p[slot] = nullptr; g_s.cap[slot] = 0;
    if (hipMalloc(&g_s.p[slot], bytes) != hipSuccess) return nullptr;
    g_s.cap[slot] = bytes;
    return g_s.p[slot];
}

int dft(gnsscorr_ctx *ctx, const float2 *in, float2 *out, int m, int sign)
{
    hipLaunchKernelGGL(dft_direct_kernel, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, in, out, m, sign);
    GC_HIP(hipGetLastError());
    return 0;
}

// cpxconv on device buffers: a (m, overwritten with the inverse transform), b (m), conv (n doubles)
int conv_dev(gnsscorr_ctx *ctx, float2 *a, float2 *tmp, const float2 *b, int m, int n, int flagsum,
             double *conv)
{
    int rc = dft(ctx, a, tmp, m, -1);
    if (rc) return rc;
    hipLaunchKernelGGL(conjmul_kernel, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, tmp, b, m);
    rc = dft(ctx, tmp, a, m, +1);
    if (rc) return rc;
    hipLaunchKernelGGL(power_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a, conv, n,
                       (float)m * (float)m, flagsum);
    GC_HIP(hipGetLastError());
    return 0;
}

struct Guard {             // default context + lock + device, or a printed error
    gnsscorr_ctx *ctx;
    std::unique_lock<std::mutex> lk;
    explicit Guard(const char *who) : ctx(gnsscorr_default_ctx())
    {
        if (!ctx) { SDRPRINTF("error: %s: no GPU context (%s)\n", who, gnsscorr_last_error()); return; }
        lk = std::unique_lock<std::mutex>(ctx->mtx);
        if (hipSetDevice(ctx->device) != hipSuccess) { SDRPRINTF("error: %s: hipSetDevice\n", who); ctx = nullptr; }
    }
    explicit operator bool() const { return ctx != nullptr; }
};

GcOpTables *g_optab = nullptr;

// builds the call's tables on the device; returns nonzero (and prints) when they do not fit
int op_tables(gnsscorr_ctx *ctx, const char *who, int n, double phi0, double freq, double ti, int nt, double coff,
              int smax, double ci, int len)
{
    if (!g_optab && hipMalloc((void **)&g_optab, sizeof(GcOpTables)) != hipSuccess) {
        SDRPRINTF("error: %s memory allocation\n", who);
        return -1;
    }
    hipLaunchKernelGGL(op_tables_kernel, dim3(1), dim3(1), 0, ctx->stream, g_optab, n, phi0, freq, ti, nt, coff, smax, ci, len);
    int over = 0;
    if (hipMemcpyAsync(&over, &g_optab->overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        SDRPRINTF("error: %s: HIP failure\n", who);
        return -1;
    }
    if (over) SDRPRINTF("error: %s: the call needs more than %d NCO pieces (code periods / carrier binades per call)\n", who, GC_OPSEG);
    return over;
}

}  // namespace

extern "C" {

// ref src/sdrcmn.c:185-195
void cpxcpx(const short *II, const short *QQ, double scale, int n, cpx_t *cpx)
{
    Guard g("cpxcpx");
    if (!g || n <= 0) return;
    short *dI = (short *)need(0, sizeof(short) * n), *dQ = QQ ? (short *)need(1, sizeof(short) * n) : nullptr;
    float2 *dc = (float2 *)need(2, sizeof(float2) * n);
    if (!dI || (QQ && !dQ) || !dc) { SDRPRINTF("error: cpxcpx memory allocation\n"); return; }
    hipMemcpyAsync(dI, II, sizeof(short) * n, hipMemcpyHostToDevice, g.ctx->stream);
    if (QQ) hipMemcpyAsync(dQ, QQ, sizeof(short) * n, hipMemcpyHostToDevice, g.ctx->stream);
    hipLaunchKernelGGL(cpxcpx_kernel, dim3((n + 255) / 256), dim3(256), 0, g.ctx->stream, dI, dQ, (float)scale, n, dc);
    hipMemcpyAsync(cpx, dc, sizeof(float2) * n, hipMemcpyDeviceToHost, g.ctx->stream);
    if (hipStreamSynchronize(g.ctx->stream) != hipSuccess) SDRPRINTF("error: cpxcpx: HIP failure\n");
}

// ref src/sdrcmn.c:134-175 (plan argument ignored: there is nothing to plan)
void cpxfft(void *plan, cpx_t *cpx, int n)
{
    (void)plan;
    Guard g("cpxfft");
    if (!g || n <= 0) return;
    float2 *a = (float2 *)need(2, sizeof(float2) * n), *b = (float2 *)need(3, sizeof(float2) * n);
    if (!a || !b) { SDRPRINTF("error: cpxfft memory allocation\n"); return; }
    hipMemcpyAsync(a, cpx, sizeof(float2) * n, hipMemcpyHostToDevice, g.ctx->stream);
    if (dft(g.ctx, a, b, n, -1)) { SDRPRINTF("error: cpxfft: %s\n", gnsscorr_last_error()); return; }
    hipMemcpyAsync(cpx, b, sizeof(float2) * n, hipMemcpyDeviceToHost, g.ctx->stream);
    if (hipStreamSynchronize(g.ctx->stream) != hipSuccess) SDRPRINTF("error: cpxfft: HIP failure\n");
}

void cpxifft(void *plan, cpx_t *cpx, int n)
{
    (void)plan;
    Guard g("cpxifft");
    if (!g || n <= 0) return;
    float2 *a = (float2 *)need(2, sizeof(float2) * n), *b = (float2 *)need(3, sizeof(float2) * n);
    if (!a || !b) { SDRPRINTF("error: cpxifft memory allocation\n"); return; }
    hipMemcpyAsync(a, cpx, sizeof(float2) * n, hipMemcpyHostToDevice, g.ctx->stream);
    if (dft(g.ctx, a, b, n, +1)) { SDRPRINTF("error: cpxifft: %s\n", gnsscorr_last_error()); return; }
    hipMemcpyAsync(cpx, b, sizeof(float2) * n, hipMemcpyDeviceToHost, g.ctx->stream);
    if (hipStreamSynchronize(g.ctx->stream) != hipSuccess) SDRPRINTF("error: cpxifft: HIP failure\n");
}

// ref src/sdrcmn.c:228-251 (cpxa is left holding the inverse transform, as in the reference)
void cpxconv(void *plan, void *iplan, cpx_t *cpxa, cpx_t *cpxb, int m, int n, int flagsum, double *conv)
{
    (void)plan; (void)iplan;
    Guard g("cpxconv");
    if (!g || m <= 0 || n <= 0 || n > m) return;
    float2 *a = (float2 *)need(2, sizeof(float2) * m), *t = (float2 *)need(3, sizeof(float2) * m);
    float2 *b = (float2 *)need(4, sizeof(float2) * m);
    double *c = (double *)need(5, sizeof(double) * n);
    if (!a || !t || !b || !c) { SDRPRINTF("error: cpxconv memory allocation\n"); return; }
    hipStream_t st = g.ctx->stream;
    hipMemcpyAsync(a, cpxa, sizeof(float2) * m, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(b, cpxb, sizeof(float2) * m, hipMemcpyHostToDevice, st);
    if (flagsum) hipMemcpyAsync(c, conv, sizeof(double) * n, hipMemcpyHostToDevice, st);
    if (conv_dev(g.ctx, a, t, b, m, n, flagsum, c)) { SDRPRINTF("error: cpxconv: %s\n", gnsscorr_last_error()); return; }
    hipMemcpyAsync(cpxa, a, sizeof(float2) * m, hipMemcpyDeviceToHost, st);
    hipMemcpyAsync(conv, c, sizeof(double) * n, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) SDRPRINTF("error: cpxconv: HIP failure\n");
}

// ref src/sdrcmn.c:261-276; 16384/32768 points run on the LDS-resident FFT
void cpxpspec(void *plan, cpx_t *cpx, int n, int flagsum, double *pspec)
{
    (void)plan;
    if (n == 16384 || n == 32768) {
        gnsscorr_ctx *ctx = gnsscorr_default_ctx();
        if (!ctx) { SDRPRINTF("error: cpxpspec: no GPU context (%s)\n", gnsscorr_last_error()); return; }
        std::lock_guard<std::mutex> lk(ctx->mtx);
        if (gnsscorr_pspec(ctx, (const float *)cpx, n, flagsum, pspec))
            SDRPRINTF("error: cpxpspec: %s\n", gnsscorr_last_error());
        return;
    }
    Guard g("cpxpspec");
    if (!g || n <= 0) return;
    float2 *a = (float2 *)need(2, sizeof(float2) * n), *b = (float2 *)need(3, sizeof(float2) * n);
    double *c = (double *)need(5, sizeof(double) * n);
    if (!a || !b || !c) { SDRPRINTF("error: cpxpspec memory allocation\n"); return; }
    hipStream_t st = g.ctx->stream;
    hipMemcpyAsync(a, cpx, sizeof(float2) * n, hipMemcpyHostToDevice, st);
    if (flagsum) hipMemcpyAsync(c, pspec, sizeof(double) * n, hipMemcpyHostToDevice, st);
    if (dft(g.ctx, a, b, n, -1)) { SDRPRINTF("error: cpxpspec: %s\n", gnsscorr_last_error()); return; }
    hipLaunchKernelGGL(power_kernel, dim3((n + 255) / 256), dim3(256), 0, st, b, c, n, 1.0f, flagsum);
    hipMemcpyAsync(cpx, b, sizeof(float2) * n, hipMemcpyDeviceToHost, st);     // in-place FFT side effect
    hipMemcpyAsync(pspec, c, sizeof(double) * n, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) SDRPRINTF("error: cpxpspec: HIP failure\n");
}

// ref src/sdrcmn.c:633-669
double mixcarr(const char *data, int dtype, double ti, int n, double freq, double phi0, short *II, short *QQ)
{
    Guard g("mixcarr");
    if (!g || n <= 0 || (dtype != 1 && dtype != 2)) return 0.0;
    int8_t *d = (int8_t *)need(0, (size_t)n * dtype);
    short *dI = (short *)need(1, sizeof(short) * n), *dQ = (short *)need(6, sizeof(short) * n);
    if (!d || !dI || !dQ) { SDRPRINTF("error: mixcarr memory allocation\n"); return 0.0; }
    hipStream_t st = g.ctx->stream;
    hipMemcpyAsync(d, data, (size_t)n * dtype, hipMemcpyHostToDevice, st);
    if (op_tables(g.ctx, "mixcarr", n, phi0, freq, ti, 0, 0.0, 0, 0.0, 1)) return 0.0;
    hipLaunchKernelGGL(mix_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d, dtype, n, g_optab, dI, dQ,
                       (float2 *)nullptr, 0.0f);
    hipMemcpyAsync(II, dI, sizeof(short) * n, hipMemcpyDeviceToHost, st);
    hipMemcpyAsync(QQ, dQ, sizeof(short) * n, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) SDRPRINTF("error: mixcarr: HIP failure\n");
    GcNoEmit ne;
    return gc_carrier_prem(gc_carrier_walk(gc_carrier_phis(phi0), gc_carrier_ps(freq, ti), n, ne));
}

// ref src/sdrcmn.c:608-621
double rescode(const short *code, int len, double coff, int smax, double ci, int n, short *rcode)
{
    Guard g("rescode");
    const int nt = n + 2 * smax;
    if (!g || nt <= 0 || len <= 0) return 0.0;
    short *dc = (short *)need(1, sizeof(short) * len), *dr = (short *)need(6, sizeof(short) * nt);
    if (!dc || !dr) { SDRPRINTF("error: rescode memory allocation\n"); return 0.0; }
    hipStream_t st = g.ctx->stream;
    hipMemcpyAsync(dc, code, sizeof(short) * len, hipMemcpyHostToDevice, st);
    if (!(ci > 0.0 && ci < (double)len)) { SDRPRINTF("error: rescode: chip step %g outside (0, %d)\n", ci, len); return 0.0; }
    if (op_tables(g.ctx, "rescode", 0, 0.0, 0.0, 0.0, nt, coff, smax, ci, len)) return 0.0;
    hipLaunchKernelGGL(rescode_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, dc, g_optab, nt, dr);
    hipMemcpyAsync(rcode, dr, sizeof(short) * nt, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) SDRPRINTF("error: rescode: HIP failure\n");
    GcNoEmit ne;
    return gc_code_rem(gc_code_walk(gc_code_start(coff, smax, ci, len), ci, len, nt, ne), smax, ci);
}

// ref src/sdrcmn.c:738-773 with the reference's own m-point transforms
void pcorrelator(const char *data, int dtype, double ti, int n, double *freq, int nfreq, double crate, int m,
                 cpx_t *codex, double *P)
{
    (void)crate;
    Guard g("pcorrelator");
    if (!g) return;
    if (n <= 0 || nfreq <= 0 || m < 2 * n || (dtype != 1 && dtype != 2)) {
        SDRPRINTF("error: pcorrelator: unsupported shape n=%d m=%d dtype=%d\n", n, m, dtype);
        return;
    }
    int8_t *d = (int8_t *)need(0, (size_t)m * dtype);
    float2 *x = (float2 *)need(2, sizeof(float2) * m), *t = (float2 *)need(3, sizeof(float2) * m);
    float2 *cx = (float2 *)need(4, sizeof(float2) * m);
    double *dP = (double *)need(5, sizeof(double) * (size_t)n * nfreq);
    if (!d || !x || !t || !cx || !dP) { SDRPRINTF("error: pcorrelator memory allocation\n"); return; }
    hipStream_t st = g.ctx->stream;
    hipMemsetAsync(d, 0, (size_t)m * dtype, st);                       // zero padding (:756)
    hipMemcpyAsync(d, data, (size_t)2 * n * dtype, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(cx, codex, sizeof(float2) * m, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(dP, P, sizeof(double) * (size_t)n * nfreq, hipMemcpyHostToDevice, st);
    const float sc = (float)(CSCALE / m);
    for (int i = 0; i < nfreq; i++) {
        if (op_tables(g.ctx, "pcorrelator", m, 0.0, freq[i], ti, 0, 0.0, 0, 0.0, 1)) return;
        hipLaunchKernelGGL(mix_kernel, dim3((m + 255) / 256), dim3(256), 0, st, d, dtype, m, g_optab,
                           (short *)nullptr, (short *)nullptr, x, sc);
        if (conv_dev(g.ctx, x, t, cx, m, n, 1, dP + (size_t)i * n)) {
            SDRPRINTF("error: pcorrelator: %s\n", gnsscorr_last_error());
            return;
        }
    }
    hipMemcpyAsync(P, dP, sizeof(double) * (size_t)n * nfreq, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) SDRPRINTF("error: pcorrelator: HIP failure\n");
}

static int vstat(gnsscorr_ctx *ctx, const double *ddev, int n, int exs, int exe, double *o3, int *oi)
{
    double *out = (double *)need(7, 64);
    if (!out) return gc_fail(GNSSCORR_EHIP, "vstat: hipMalloc");
    hipLaunchKernelGGL(vstat_kernel, dim3(1), dim3(1024), 0, ctx->stream, ddev, n, exs, exe, out, (int *)(out + 4));
    GC_HIP(hipGetLastError());
    double h[5];
    GC_HIP(hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    o3[0] = h[0]; o3[1] = h[1]; o3[2] = h[2];
    memcpy(oi, &h[4], sizeof(int));
    return 0;
}

// ref src/sdrcmn.c:461-476
double maxvd(const double *data, int n, int exinds, int exinde, int *ind)
{
    Guard g("maxvd");
    if (!g || n <= 0) return 0.0;
    double *d = (double *)need(5, sizeof(double) * n);
    if (!d) { SDRPRINTF("error: maxvd memory allocation\n"); return 0.0; }
    hipMemcpyAsync(d, data, sizeof(double) * n, hipMemcpyHostToDevice, g.ctx->stream);
    double o[3]; int oi = 0;
    if (vstat(g.ctx, d, n, exinds, exinde, o, &oi)) { SDRPRINTF("error: maxvd: %s\n", gnsscorr_last_error()); return 0.0; }
    *ind = oi;
    return o[0];
}

// ref src/sdrcmn.c:487-497
double meanvd(const double *data, int n, int exinds, int exinde)
{
    Guard g("meanvd");
    if (!g || n <= 0) return 0.0;
    double *d = (double *)need(5, sizeof(double) * n);
    if (!d) { SDRPRINTF("error: meanvd memory allocation\n"); return 0.0; }
    hipMemcpyAsync(d, data, sizeof(double) * n, hipMemcpyHostToDevice, g.ctx->stream);
    double o[3]; int oi = 0;
    if (vstat(g.ctx, d, n, exinds, exinde, o, &oi)) { SDRPRINTF("error: meanvd: %s\n", gnsscorr_last_error()); return 0.0; }
    return o[1] / ((double)n - o[2]);
}

// ref src/sdracq.c:71-95
int checkacquisition(double *P, sdrch_t *sdr)
{
    Guard g("checkacquisition");
    const int n = sdr->nsamp, nf = sdr->acq.nfreq;
    if (!g || n <= 0 || nf <= 0) return 0;
    double *d = (double *)need(5, sizeof(double) * (size_t)n * nf);
    if (!d) { SDRPRINTF("error: checkacquisition memory allocation\n"); return 0; }
    hipMemcpyAsync(d, P, sizeof(double) * (size_t)n * nf, hipMemcpyHostToDevice, g.ctx->stream);
    double o[3]; int maxi = 0, dummy = 0;
    if (vstat(g.ctx, d, n * nf, -1, -1, o, &maxi)) { SDRPRINTF("error: checkacquisition: %s\n", gnsscorr_last_error()); return 0; }
    const double maxP = o[0];
    int codei, freqi;
    ind2sub(maxi, n, nf, &codei, &freqi);
    int exinds = codei - 2 * sdr->nsampchip; if (exinds < 0) exinds += n;
    int exinde = codei + 2 * sdr->nsampchip; if (exinde >= n) exinde -= n;
    if (vstat(g.ctx, d + (size_t)freqi * n, n, exinds, exinde, o, &dummy)) {
        SDRPRINTF("error: checkacquisition: %s\n", gnsscorr_last_error());
        return 0;
    }
    const double meanP = o[1] / ((double)n - o[2]), maxP2 = o[0];
    sdr->acq.cn0 = 10 * log10(maxP / meanP / sdr->ctime);
    sdr->acq.peakr = maxP / maxP2;
    sdr->acq.acqcodei = codei;
    sdr->acq.freqi = freqi;
    sdr->acq.acqfreq = sdr->acq.freq[freqi];
    return sdr->acq.peakr > ACQTH;
}

// ---- sdracquisition ---------------------------------------------------------
// One private engine per channel struct: it shares the HBM ring of the default
// context and keeps the channel's replica spectrum and work buffers resident
// between attempts (a failed acquisition is retried every ACQSLEEP ms forever,
// ref src/sdracq.c:57-59).
static std::map<sdrch_t *, gnsscorr_ctx *> g_acqctx;
static std::mutex g_acqctx_mtx;

static gnsscorr_ctx *acq_engine(sdrch_t *sdr, gnsscorr_ctx *def)
{
    std::lock_guard<std::mutex> lk(g_acqctx_mtx);
    auto it = g_acqctx.find(sdr);
    if (it != g_acqctx.end()) return it->second;
    const GcRing &r = def->ring[sdr->ftype == FTYPE2 ? 1 : 0];
    if (!r.mem || r.dtype != sdr->dtype) {
        gc_fail(GNSSCORR_ESTATE, "IF ring %d is not mirrored on the GPU", sdr->ftype);
        return nullptr;
    }
    gnsscorr_ctx *c = nullptr;
    if (gnsscorr_create(&c, def->device, nullptr)) return nullptr;
    gnsscorr_chan_t d;
    memset(&d, 0, sizeof(d));
    d.prn = sdr->prn; d.ctype = sdr->ctype; d.dtype = sdr->dtype; d.ftype = sdr->ftype;
    d.clen = sdr->clen; d.nsamp = sdr->nsamp; d.nsampchip = sdr->nsampchip;
    d.f_sf = sdr->f_sf; d.f_if = sdr->f_if; d.foffset = sdr->foffset;
    d.crate = sdr->crate; d.ctime = sdr->ctime; d.ti = sdr->ti;
    d.code = sdr->code; d.intg = sdr->acq.intg; d.nfreq = sdr->acq.nfreq; d.freq = sdr->acq.freq;
    d.nfft = sdr->acq.nfft; d.corrn = sdr->trk.corrn; d.corrp = sdr->trk.corrp;
    if (gnsscorr_ring_create(c, sdr->ftype, sdr->dtype, r.ringlen, r.mem) || gnsscorr_set_channels(c, 1, &d)) {
        gnsscorr_destroy(c);
        return nullptr;
    }
    g_acqctx[sdr] = c;
    return c;
}

// ref src/sdracq.c:14-62
uint64_t sdracquisition(sdrch_t *sdr, double *power)
{
    uint64_t wrpos;
    mlock(hreadmtx);
    wrpos = (uint64_t)sdrstat.fendbuffsize * sdrstat.buffcnt;
    unmlock(hreadmtx);
    uint64_t buffloc = wrpos - (uint64_t)(sdr->acq.intg + 1) * sdr->nsamp;

    gnsscorr_ctx *def = gnsscorr_default_ctx();
    if (def) {      // the private engine shares the default context's ring: its last block must have landed
        std::lock_guard<std::mutex> lk(def->mtx);
        if (def->in_pending) hipEventSynchronize(def->ev_in);
    }
    gnsscorr_ctx *eng = def ? acq_engine(sdr, def) : nullptr;
    gnsscorr_acqres_t r;
    if (!eng || gnsscorr_acq_run(eng, wrpos) || gnsscorr_acq_fetch(eng, &r)) {
        SDRPRINTF("error: sdracquisition: %s\n", gnsscorr_last_error());
        return buffloc;
    }
    sdr->acq.cn0 = r.cn0; sdr->acq.peakr = r.peakr; sdr->acq.acqcodei = r.acqcodei;
    sdr->acq.freqi = r.freqi; sdr->acq.acqfreq = r.acqfreq;
    if (r.flagacq) sdr->flagacq = ON;
    if (power) {            // the reference accumulates into the caller's zeroed array
        const size_t ne = (size_t)sdr->nsamp * sdr->acq.nfreq;
        std::vector<double> tmp(ne);
        if (gnsscorr_acq_power(eng, 0, tmp.data())) SDRPRINTF("error: sdracquisition: %s\n", gnsscorr_last_error());
        else for (size_t i = 0; i < ne; i++) power[i] += tmp[i];
    }
    SDRPRINTF("%s, C/N0=%4.1f, peak=%3.1f, codei=%5d, freq=%8.1f\n", sdr->satstr, sdr->acq.cn0, sdr->acq.peakr,
              sdr->acq.acqcodei, sdr->acq.acqfreq - sdr->f_if - sdr->foffset);
    if (sdr->flagacq) {
        sdr->trk.carrfreq = sdr->acq.acqfreq;
        sdr->trk.codefreq = sdr->crate;
    } else {
        const char *e = getenv("GNSSCORR_ACQSLEEP_MS");     // default ACQSLEEP = 2000 ms
        usleep(1000u * (unsigned)(e ? atoi(e) : ACQSLEEP));
    }
    return r.buffloc;
}

}  // extern "C"
