// Does a VALU instruction cost less when only part of the wavefront is live?  8 waves per SIMD, independent chains of
// v_perm_b32 / v_add_u32 under an EXEC mask of `live` lanes.  Build: hipcc --offload-arch=gfx950 -O2 -o exec_mask exec_mask.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP>
__global__ __launch_bounds__(256) void masked(uint32_t* out, int iters, uint32_t seed, int live)
{
    uint32_t a[8];
    for (int c = 0; c < 8; ++c) a[c] = threadIdx.x * 2654435761u + c * 40503u + seed;
    uint32_t s = seed | 1u;
    if ((int)(threadIdx.x & 63) < live) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(s));
                    if (OP == 1) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(s));
                }
            }
        }
    }
    uint32_t x = 0;
    for (int c = 0; c < 8; ++c) x ^= a[c];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int OP>
static void run(const char* name, uint32_t* out, int live)
{
    const int iters = 2000, blocks = 256 * 8;
    masked<OP><<<blocks, 256>>>(out, 10, 1, live);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        masked<OP><<<blocks, 256>>>(out, iters, 1, live);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: 8 waves x iters x 64 instructions
    double ns = best * 1e6 / (8.0 * iters * 64);
    printf("%-12s live lanes %2d: %8.1f us  %.3f ns per wave-instr per SIMD = %.2f cycles @2.4GHz\n", name, live, best * 1e3, ns, ns * 2.4);
}

int main()
{
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int live : {64, 48, 32, 16, 8, 1}) { run<0>("v_add_u32", out, live); run<1>("v_perm_b32", out, live); }
    return 0;
}
