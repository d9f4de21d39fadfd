// Micro-benchmark: cost of the planner's exact NCO chain per code period on one wavefront.
// hipcc -O3 --offload-arch=gfx950 -I../../erlangnetwork-gnsslib-sdr_amd/csrc nco_chain.hip -o nco_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "gnsscorr_nco.h"
#include "code_period_prof.h"

struct Cnt { int n = 0; __host__ __device__ void operator()(int, double, double, int) { n++; } __host__ __device__ void operator()(int, double, double, int, int) { n++; } };

__global__ __launch_bounds__(64) void chain_kernel(int mode, int nper, double carrfreq, double codefreq, double *out, long long *clk)
{
    __shared__ int Ks[GC_NB + 2];
    const int lane = threadIdx.x;
    if (threadIdx.x && mode < 16) return;
    const double ti = 1 / 16.368e6, f_sf = 16.368e6;
    const int len = 1023, smax = 6;
    const double ci = __dmul_rn(ti, codefreq), spc = __ddiv_rn(codefreq, f_sf), ps = gc_carrier_ps(carrfreq, ti);
    GcNcoFast fcar, fcode, fprem;
    gc_fast_init(fcar, ps);
    gc_fast_init(fcode, ci);
    gc_fast_init(fprem, -GC_NCO_DPI);
    const double yspc = 1.0 / spc, ydpi = 1.0 / GC_NCO_DPI, smaxci = smax * ci;
    double remcode = 0, remcarr = 0;
    Cnt c1, c2;
    long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    GcCodePlan PC;
    GcCarPlan PK;
    gc_code_plan_init(PC, ci, len, smax);
    gc_car_plan_init(PK, ps);
    GcFillLanes fill{lane};
    const long long t0 = wall_clock64();
    for (int p = 0; p < nper; p++) {
        const int n = (int)gc_div_y(len - remcode, spc, yspc);
        if (mode & 1) remcarr = gc_fast_prem(fprem, gc_fast_carrier_walk(fcar, gc_div_y(remcarr * 32.0, GC_NCO_DPI, ydpi), n, c1));
        if (mode & 2) remcode = gc_fast_code_walk(fcode, gc_code_start_fast(remcode, smaxci, len), len, n + 2 * smax, c2) - smaxci;
        if (mode & 4) remcarr = gc_carrier_prem(gc_carrier_walk(gc_carrier_phis(remcarr), ps, n, c1));
        if (mode & 16) {
            const double phis = gc_div_y(remcarr * 32.0, GC_NCO_DPI, ydpi);
            double xn;
            GcNoEmit ne;
            if (!plan_carrier_dev(fcar, phis, n, Ks, lane, &xn)) { xn = gc_fast_carrier_walk(fcar, phis, n, ne); c1.n++; }
            remcarr = gc_fast_prem(fprem, xn);
        }
        if (mode & 32) {
            const double c0 = gc_code_start_fast(remcode, smaxci, len);
            double cend;
            GcNoEmit ne;
            if (!plan_code_dev(fcode, c0, len, n + 2 * smax, Ks, lane, &cend)) { cend = gc_fast_code_walk(fcode, c0, len, n + 2 * smax, ne); c2.n++; }
            remcode = cend - smaxci;
        }
        if (mode & 64) { double rp; if (gc_carrier_period(PK, remcarr, n, fill, &rp)) remcarr = rp; else { GcNoEmit ne; c1.n++; remcarr = gc_fast_prem(fprem, gc_fast_carrier_walk(fcar, gc_div_y(remcarr * 32.0, GC_NCO_DPI, ydpi), n, ne)); } }
        if (mode & 128) { double rc; if (gc_code_period(PC, remcode, n + 2 * smax, fill, &rc)) remcode = rc; else { GcNoEmit ne; c2.n++; remcode = gc_fast_code_walk(fcode, gc_code_start_fast(remcode, smaxci, len), len, n + 2 * smax, ne) - smaxci; } }
        if (mode & 256) { double rc; GcNoEmit ne; if (gc_code_period_prof<11>(T, PC, remcode, n + 2 * smax, fill, &rc, ne)) remcode = rc; else c2.n++; }
        if (mode & 8) remcode = gc_code_rem(gc_code_walk(gc_code_start(remcode, smax, ci, len), ci, len, n + 2 * smax, c2), smax, ci);
    }
    const long long t1 = wall_clock64();
    if (lane == 0 && (mode & 256)) for (int i = 0; i < 5; i++) out[4 + i] = (double)T[i] / nper;
    if (lane == 0) { out[0] = remcode; out[1] = remcarr; out[2] = c1.n; out[3] = c2.n; clk[0] = t1 - t0; }
}

int main()
{
    double *out; long long *clk;
    hipMalloc(&out, 128); hipMalloc(&clk, 8);
    const int nper = 2000;
    for (int mode : {128, 256}) for (double cf : {2345.6, -2345.6, 4.0932e6}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, mode, 10, cf, 1.023e6 + 1.3, out, clk);
        hipEventRecord(a);
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, mode, nper, cf, 1.023e6 + 1.3, out, clk);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        double h[10]; long long c; hipMemcpy(h, out, 80, hipMemcpyDeviceToHost); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        if (mode & 256) printf("  shader clocks/period: head %.0f literal %.0f fill %.0f chain %.0f tail %.0f\n", h[4], h[5], h[6], h[7], h[8]);
        printf("mode %d carr %.1f: %.3f us/period (%.0f wall-clock ticks/period), pieces/period car %.1f code %.1f, rem %.6g %.6g\n", mode, cf,
               ms * 1e3 / nper, (double)c / nper, h[2] / nper, h[3] / nper, h[0], h[1]);
    }
    return 0;
}
