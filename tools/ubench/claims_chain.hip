// What one period of the planner's evaluating chain costs a lone wavefront: the code / carrier step on claims with
// its checks (as round 2's chain ran it), and the same step with only the VALUE kept (the checks' result unused: the
// compiler drops them) -- the split behind "one wavefront chains the values, others check them".
// hipcc -O3 --offload-arch=gfx950 -I../../erlangnetwork-gnsslib-sdr_amd/csrc claims_chain.hip -o claims_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "gnsscorr_nco.h"

#define NPER 2000
// mode 0: discover (exact chain start) and store the claims; 1: evaluate with checks; 2: values only;
// which: 1 code, 2 carrier
template <int ITOP, int mode>
__device__ void run(int which, const GcCodePlan &PC, const GcCarPlan &PK, double spc, int len, int smax, int nsamp,
                    GcCodeClaims *ccl, GcCarClaims *kcl, int *ns, double *out, long long *clk)
{
    GcCodeStepC<ITOP> SC;
    gc_code_stepc_init(SC, PC);
    GcCarStepC CK;
    gc_car_stepc_init(CK, PK, nsamp + 16);
    const double yspc = 1.0 / spc;
    double remcode = 0.3, remcarr = 0.2;
    int fails = 0;
    const long long t0 = wall_clock64();
    for (int p = 0; p < NPER; p++) {
        if (mode == 0) {
            const int n = (int)gc_div_y(len - remcode, spc, yspc);
            ns[p] = n;
            GcCodeClaims cl;
            GcCarClaims kl;
            double r, k;
            if (!gc_code_claims_step<ITOP, 8, true>(PC, SC, remcode, n + 2 * smax, cl, &r)) fails++;
            if (!gc_carrier_claims_step<true>(PK, CK, remcarr, n, kl, &k)) fails += 1000;
            ccl[p] = cl; kcl[p] = kl;
            remcode = r; remcarr = k;
        } else {
            const int n = ns[p];
            if (which & 1) {
                GcCodeClaims cl = ccl[p];
                double r;
                const bool ok = gc_code_claims_step<ITOP, 8, false>(PC, SC, remcode, n + 2 * smax, cl, &r);
                if (mode == 1 && !ok) fails++;
                remcode = r;
            }
            if (which & 2) {
                GcCarClaims kl = kcl[p];
                double k;
                const bool ok = gc_carrier_claims_step<false>(PK, CK, remcarr, n, kl, &k);
                if (mode == 1 && !ok) fails += 1000;
                remcarr = k;
            }
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = remcode; out[1] = remcarr; out[2] = fails; clk[0] = t1 - t0; }
}

template <int mode>
__global__ __launch_bounds__(64) void k(int which, double carrfreq, double codefreq, GcCodeClaims *ccl, GcCarClaims *kcl, int *ns,
                                        double *out, long long *clk)
{
    const double ti = 1 / 16.368e6, f_sf = 16.368e6;
    const int len = 1023, smax = 6;
    const double ci = __dmul_rn(ti, codefreq), spc = __ddiv_rn(codefreq, f_sf), ps = gc_carrier_ps(carrfreq, ti);
    GcCodePlan PC;
    GcCarPlan PK;
    gc_code_plan_init(PC, ci, len, smax);
    gc_car_plan_init(PK, ps);
    if (PC.itop == 11) run<11, mode>(which, PC, PK, spc, len, smax, 16368, ccl, kcl, ns, out, clk);
    else if (PC.itop == 12) run<12, mode>(which, PC, PK, spc, len, smax, 16368, ccl, kcl, ns, out, clk);
    else if (threadIdx.x == 0) out[2] = -1;
}

int main()
{
    double *out; long long *clk; GcCodeClaims *ccl; GcCarClaims *kcl; int *ns;
    hipMalloc(&out, 128); hipMalloc(&clk, 64); hipMalloc(&ccl, sizeof(GcCodeClaims) * NPER); hipMalloc(&kcl, sizeof(GcCarClaims) * NPER); hipMalloc(&ns, 4 * NPER);
    for (double cf : {2345.6, -2345.6}) for (double df : {1.3, -1.3}) {
        printf("carr %.1f code %+.1f:\n", cf, df);
        const char *names[] = {"discover (both, exact start)", "code, with checks", "code, value only", "carrier, with checks", "carrier, value only"};
        const int modes[] = {0, 1, 2, 1, 2}, whichs[] = {3, 1, 1, 2, 2};
        for (int i = 0; i < 5; i++) {
            for (int rep = 0; rep < 2; rep++) {
                if (modes[i] == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, whichs[i], cf, 1.023e6 + df, ccl, kcl, ns, out, clk);
                if (modes[i] == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, whichs[i], cf, 1.023e6 + df, ccl, kcl, ns, out, clk);
                if (modes[i] == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, whichs[i], cf, 1.023e6 + df, ccl, kcl, ns, out, clk);
            }
            hipDeviceSynchronize();
            double h[4]; long long c[2];
            hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            printf("   %-32s %7.1f ns/period (100 MHz clock: %.0f ticks)  failed checks %g  end %.9g %.9g\n", names[i], (double)c[0] * 10.0 / NPER, (double)c[0] / NPER, h[2], h[0], h[1]);
        }
    }
    return 0;
}
