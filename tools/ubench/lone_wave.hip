// Dependent-issue latency of a lone wavefront on gfx950: CHAINS independent chains of one VALU op, interleaved,
// one wave per SIMD (or fewer).  cycles per instruction = s_memtime ticks (100 MHz) scaled by the shader clock.
// Build: hipcc --offload-arch=gfx950 -O2 -o lone_wave lone_wave.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>

template <int CHAINS, int OP>
__global__ __launch_bounds__(64) void chain_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[8];
  for (int c = 0; c < 8; ++c) a[c] = threadIdx.x * 2654435761u + c * 40503u + seed;
  uint32_t s = seed | 1u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 64 / CHAINS; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(s));
        if (OP == 1) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(s));
        if (OP == 2) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(s));
      }
    }
  }
  uint32_t x = 0;
  for (int c = 0; c < 8; ++c) x ^= a[c];
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <int CHAINS, int OP>
static void run(const char* name, uint32_t* out, int blocks) {
  const int iters = 4000;
  chain_kernel<CHAINS, OP><<<blocks, 64>>>(out, 10, 1);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    chain_kernel<CHAINS, OP><<<blocks, 64>>>(out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double ns_per = best * 1e6 / (double(iters) * 64);
  printf("%-14s chains %d  blocks %5d: %8.1f us  %.3f ns per instr = %.2f cycles @2.4GHz\n", name, CHAINS, blocks, best * 1e3, ns_per, ns_per * 2.4);
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 8192 * 64 * 4);
  for (int blocks : {256, 1024, 4096}) {
    run<1, 0>("v_add_u32", out, blocks);
    run<2, 0>("v_add_u32", out, blocks);
    run<4, 0>("v_add_u32", out, blocks);
    run<8, 0>("v_add_u32", out, blocks);
    run<1, 1>("v_perm_b32", out, blocks);
    run<2, 1>("v_perm_b32", out, blocks);
    run<4, 1>("v_perm_b32", out, blocks);
    run<1, 2>("v_and_or_b32", out, blocks);
    run<2, 2>("v_and_or_b32", out, blocks);
  }
  return 0;
}
