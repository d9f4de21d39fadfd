// dependent-instruction latency of a lone wavefront (shader clocks per operation)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ void k(double *out, long long *clk, double a, double b)
{
    double x = a + threadIdx.x * 1e-9;
    float xf = (float)x;
    int xi = (int)threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 256; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (OP == 0) x = __fma_rn(x, b, a);
            if (OP == 1) x = __dadd_rn(x, b);
            if (OP == 2) xf = __fmaf_rn(xf, (float)b, (float)a);
            if (OP == 3) xi = xi * 3 + 1;
            if (OP == 4) { x = __fma_rn(x, b, a); x = __dadd_rn(x, a); }
            if (OP == 5) x = floor(x * b) + a;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + xf + xi;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int OP> void run(const char *name, double *out, long long *clk)
{
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, clk, 1.0000001, 0.9999999);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, clk, 1.0000001, 0.9999999);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%-28s %.1f clocks per iteration\n", name, (double)c / 4096.0);
}
int main()
{
    double *out; long long *clk;
    hipMalloc(&out, 64 * 8); hipMalloc(&clk, 8);
    run<0>("v_fma_f64 dependent", out, clk);
    run<1>("v_add_f64 dependent", out, clk);
    run<2>("v_fma_f32 dependent", out, clk);
    run<3>("v_mad_u32 dependent", out, clk);
    run<4>("fma_f64 + add_f64 pair", out, clk);
    run<5>("mul, floor, add f64", out, clk);
    return 0;
}
