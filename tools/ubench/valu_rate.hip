// Micro-benchmark: issue rate of the VALU instructions the tracking correlator is made of.
// Prints wave-instructions per cycle per CU (4 SIMDs) at full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s2 __attribute__((ext_vector_type(2)));
#define ITER 4096
#define UNROLL 16
template <int OP>
__global__ __launch_bounds__(256) void k(int *out, int seed)
{
    int a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * (i + 1);
    int b = seed * 3 + 1;
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = (float)a[i];
    unsigned long long u[4];
    for (int i = 0; i < 4; i++) u[i] = ((unsigned long long)a[i] << 32) | a[i + 4];
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int r = j & 7;
            if (OP == 0) a[r] = a[r] + b;
            if (OP == 1) a[r] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a[(r + 1) & 7]), __builtin_bit_cast(s2, b), a[r], false);
            if (OP == 2) a[r] = __builtin_amdgcn_sdot4(a[(r + 1) & 7], b, a[r], false);
            if (OP == 3) a[r] = __builtin_amdgcn_perm(a[r], b, 0x05040100u + it);
            if (OP == 4) u[r & 3] += ((unsigned long long)b << 20) + 12345;
            if (OP == 5) f[r] = fmaf(f[r], 1.0001f, 0.5f);
            if (OP == 6) a[r] = (a[r] >> 27) ^ b;
            if (OP == 7) a[r] = __mul24(a[r], b) + it;
        }
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (int)f[i];
    for (int i = 0; i < 4; i++) s += (int)u[i] + (int)(u[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;            // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITER * UNROLL;
    printf("%-28s %8.3f ms  %7.2f wave-instr/us/CU  (=%5.2f per cycle per CU at 2.4 GHz)\n", name, ms,
           winstr / (ms * 1e3) / 256, winstr / (ms * 1e-3) / 256 / 2.4e9);
}
int main()
{
    int *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_dot2c_i32_i16", d); run<2>("v_dot4c_i32_i8", d); run<3>("v_perm_b32", d);
    run<4>("64-bit add (2 instr)", d); run<5>("v_fma_f32", d); run<6>("shift+xor (2 instr)", d); run<7>("mul24+add (2)", d);
    return 0;
}
