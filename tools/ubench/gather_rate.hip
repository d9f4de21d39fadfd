// What a divergent gather from an L2-resident window costs on gfx950: one wavefront per "unit" reads K entries
// per lane from a window of a prefix table (the access shape of the shared-prefix correlator's look-ups).
//   pattern 0: lane = edge, taps in turn: 64 lanes of one instruction in 64 different 128-byte lines
//   pattern 1: 8 lanes per edge (taps side by side): one instruction touches ~8-16 lines
//   pattern 2: coalesced (reference)
// ES = entry size in bytes (8: int2 per sample, 4: packed int16 pair per sample)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int ES, int PAT>
__global__ __launch_bounds__(64) void k(const char *__restrict__ tab, unsigned tabentries, int window, int iters, int *out)
{
    const int lane = threadIdx.x;
    // every 32 consecutive blocks ("channels") walk the same window, like the channels of one code period
    const unsigned base = (unsigned)((blockIdx.x / 32) * (unsigned)window) % (tabentries - 2 * window);
    int acc = 0;
    unsigned pos = 7 + (PAT == 1 ? (lane >> 3) * 32 : lane * 32);
    const int toff = PAT == 1 ? ((lane & 7) < 5 ? ((lane & 7) * 3 - 6) : 0) : 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 5; t++) {
            unsigned e;
            if (PAT == 0) e = pos + t * 3;                   // taps of one edge in consecutive instructions
            else if (PAT == 1) e = pos + toff + 6 + t * 256; // five edges per lane group and iteration
            else e = (it * 5 + t) * 64 + lane;
            e = base + (e % (unsigned)window);
            if (ES == 8) {
                const int2 v = *(const int2 *)(tab + (size_t)e * 8);
                acc += v.x ^ v.y;
            } else {
                acc += *(const int *)(tab + (size_t)e * 4);
            }
        }
        pos += PAT == 1 ? 8 * 32 * 5 : 64 * 32;
    }
    if (acc == 0x12345678) out[0] = acc;
}

template <int ES, int PAT> void run(const char *name, const char *tab, unsigned n, int *out)
{
    const int units = 32000, iters = PAT == 1 ? 13 : 8, window = 16384;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<ES, PAT>), dim3(units), dim3(64), 0, 0, tab, n, window, iters, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)units * iters * 5;
    printf("%-40s %.3f ms  %.1f ns per wave-instruction per CU (256 CUs)  lookups/unit %d\n", name, ms,
           ms * 1e6 / (instr / 256.0), iters * 5 * 64);
}

int main()
{
    const unsigned n = 1u << 25;   // entries
    char *tab; int *out;
    hipMalloc(&tab, (size_t)n * 8); hipMalloc(&out, 64);
    hipMemset(tab, 1, (size_t)n * 8);
    run<8, 0>("8B lane=edge (64 lines/instr)", tab, n, out);
    run<8, 1>("8B 8 lanes/edge", tab, n, out);
    run<8, 2>("8B coalesced", tab, n, out);
    run<4, 0>("4B lane=edge", tab, n, out);
    run<4, 1>("4B 8 lanes/edge", tab, n, out);
    run<4, 2>("4B coalesced", tab, n, out);
    return 0;
}
