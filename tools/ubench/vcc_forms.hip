// Cost of reading a lane mask in VOP2 (implicit VCC) and VOP3 (explicit SGPR pair) form, depending on who wrote the mask
// last (VALU compare or SALU).  8 waves per SIMD, 8 independent chains.  hipcc --offload-arch=gfx950 -O2 -o vcc_forms vcc_forms.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int CASE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a[8];
    for (int c = 0; c < 8; ++c) a[c] = threadIdx.x * 2654435761u + c * 40503u + seed;
    uint32_t s = seed | 1u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (CASE == 0)   // VALU compare -> vcc, VOP2 cndmask reads vcc
                    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(s) : "vcc", "scc");
                if (CASE == 1)   // SALU writes vcc, VOP2 cndmask reads vcc
                    asm volatile("s_not_b64 vcc, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(s) : "vcc", "scc");
                if (CASE == 2)   // SALU writes an SGPR pair, VOP3 cndmask reads it
                    asm volatile("s_not_b64 s[20:21], s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[c]) : "v"(s) : "s20", "s21", "scc");
                if (CASE == 3)   // SALU writes vcc, VOP3-encoded cndmask reads vcc explicitly
                    asm volatile("s_not_b64 vcc, vcc\n\tv_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(s) : "vcc", "scc");
                if (CASE == 4)   // SALU writes vcc, VOP2 addc reads and writes vcc
                    asm volatile("s_not_b64 vcc, vcc\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[c]) : "v"(s) : "vcc", "scc");
                if (CASE == 5)   // VALU compare -> vcc, VOP2 addc
                    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[c]) : "v"(s) : "vcc", "scc");
                if (CASE == 6)   // SALU writes an SGPR pair, VOP3 addc
                    asm volatile("s_not_b64 s[20:21], s[20:21]\n\tv_addc_co_u32_e64 %0, s[22:23], %0, %1, s[20:21]" : "+v"(a[c]) : "v"(s) : "s20", "s21", "s22", "s23", "scc");
                if (CASE == 7)   // nobody writes vcc, VOP2 cndmask (the r01 22-cycle case)
                    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(s));
                if (CASE == 9)   // one VALU compare -> vcc, then FOUR VOP2 cndmask reading it (counted as one pair)
                    asm volatile("v_cmp_lt_u32 vcc, %0, %4\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\t"
                                 "v_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc"
                                 : "+v"(a[c]), "+v"(a[(c + 1) & 7]), "+v"(a[(c + 2) & 7]), "+v"(a[(c + 3) & 7]) : "v"(s) : "vcc");
                if (CASE == 10)  // the same with VOP3-encoded cndmask reading vcc
                    asm volatile("v_cmp_lt_u32 vcc, %0, %4\n\tv_cndmask_b32_e64 %0, %0, %4, vcc\n\tv_cndmask_b32_e64 %1, %1, %4, vcc\n\t"
                                 "v_cndmask_b32_e64 %2, %2, %4, vcc\n\tv_cndmask_b32_e64 %3, %3, %4, vcc"
                                 : "+v"(a[c]), "+v"(a[(c + 1) & 7]), "+v"(a[(c + 2) & 7]), "+v"(a[(c + 3) & 7]) : "v"(s) : "vcc");
                if (CASE == 11)  // VALU compare -> vcc, four unrelated VALU instructions, then one VOP2 cndmask
                    asm volatile("v_cmp_lt_u32 vcc, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\t"
                                 "v_add_u32 %3, %3, %4\n\tv_add_u32 %1, %1, %4\n\tv_cndmask_b32 %0, %0, %4, vcc"
                                 : "+v"(a[c]), "+v"(a[(c + 1) & 7]), "+v"(a[(c + 2) & 7]), "+v"(a[(c + 3) & 7]) : "v"(s) : "vcc");
                if (CASE == 8)   // VALU compare into an SGPR pair, VOP3 cndmask
                    asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[c]) : "v"(s) : "s20", "s21", "scc");
            }
        }
    }
    uint32_t x = 0;
    for (int c = 0; c < 8; ++c) x ^= a[c];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int CASE>
static void run(const char* name, uint32_t* out)
{
    const int iters = 1000, blocks = 256 * 8;
    k<CASE><<<blocks, 256>>>(out, 10, 1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k<CASE><<<blocks, 256>>>(out, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double ns = best * 1e6 / (8.0 * iters * 32);       // per (mask write + mask read) pair per wave per SIMD
    printf("%-58s %8.1f us  %.2f cycles per pair @2.4GHz\n", name, best * 1e3, ns * 2.4);
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
#define run if (only >= 0 && only != __COUNTER__ - base) {} else run
    const int base = __COUNTER__ + 1;
    run<0>("v_cmp -> vcc ; v_cndmask (VOP2, vcc)", out);
    run<1>("s_not vcc    ; v_cndmask (VOP2, vcc)", out);
    run<2>("s_not s[20:21]; v_cndmask_e64 s[20:21]", out);
    run<3>("s_not vcc    ; v_cndmask_e64 vcc", out);
    run<4>("s_not vcc    ; v_addc_co (VOP2, vcc)", out);
    run<5>("v_cmp -> vcc ; v_addc_co (VOP2, vcc)", out);
    run<6>("s_not s[20:21]; v_addc_co_e64 s[20:21]", out);
    run<7>("(no mask write); v_cndmask (VOP2, vcc)  [single instr]", out);
    run<8>("v_cmp_e64 -> s[20:21]; v_cndmask_e64 s[20:21]", out);
    run<9>("v_cmp -> vcc ; 4 x v_cndmask (VOP2, vcc)   [5 instr]", out);
    run<10>("v_cmp -> vcc ; 4 x v_cndmask_e64 vcc       [5 instr]", out);
    run<11>("v_cmp -> vcc ; 4 x v_add ; v_cndmask (VOP2) [6 instr]", out);
#undef run
    return 0;
}
