// step_floor.hip -- floors for a 1,048,576-board g2048_step launch on this GPU: an empty kernel with the same grid,
// and a pure copy kernel that moves exactly the step kernel's seven streams (46 B per board) with trivial ALU work.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k_empty(const uint4*, const uint8_t*, uint4*, uint32_t*, float*, uint8_t*, size_t) {}
__global__ __launch_bounds__(256) void k_copy(const uint4* bi, const uint8_t* a, uint4* bo, uint32_t* sc, float* rw, uint8_t* fl, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    uint4 b = bi[i]; uint32_t act = a[i]; uint32_t s = sc[i];
    b.x ^= act; s += b.y & 3u;
    bo[i] = b; sc[i] = s; rw[i] = (float)(b.z & 0xff); fl[i] = (uint8_t)(b.w ^ act);
}
template <int ALU> __global__ __launch_bounds__(256) void k_alu(const uint4* bi, const uint8_t* a, uint4* bo, uint32_t* sc, float* rw, uint8_t* fl, size_t n)
{   // the copy plus ALU dependent fast integer ops x 4 registers (ALU*8 VALU instructions per board)
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    uint4 b = bi[i]; uint32_t act = a[i]; uint32_t s = sc[i];
#pragma unroll
    for (int k = 0; k < ALU; ++k) { b.x = (b.x ^ act) + 0x9E3779B9u; b.y = (b.y + b.x) ^ 0x85EBCA6Bu; b.z = (b.z ^ b.y) + 0xC2B2AE35u; b.w = (b.w + b.z) ^ 0x27D4EB2Fu; }
    bo[i] = b; sc[i] = s + (b.y & 3u); rw[i] = (float)(b.z & 0xff); fl[i] = (uint8_t)(b.w ^ act);
}
int main()
{
    const size_t n = 1 << 20; const int K = 200;
    uint4 *bi, *bo; uint8_t *a, *fl; uint32_t* sc; float* rw;
    hipMalloc(&bi, n * 16); hipMalloc(&bo, n * 16); hipMalloc(&a, n); hipMalloc(&fl, n); hipMalloc(&sc, n * 4); hipMalloc(&rw, n * 4);
    hipMemset(bi, 1, n * 16); hipMemset(a, 2, n); hipMemset(sc, 0, n * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    typedef void (*kern_t)(const uint4*, const uint8_t*, uint4*, uint32_t*, float*, uint8_t*, size_t);
    struct { const char* name; kern_t fn; } ks[] = {{"empty", k_empty}, {"copy (46 B/board)", k_copy}, {"copy + 200 fast VALU", k_alu<25>}, {"copy + 400 fast VALU", k_alu<50>}, {"copy + 800 fast VALU", k_alu<100>}, {"copy + 1200 fast VALU", k_alu<150>}};
    for (auto& k : ks) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int t = 0; t < K; ++t) hipLaunchKernelGGL(k.fn, dim3(n / 256), dim3(256), 0, st, bi, a, bo, sc, rw, fl, n);
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        std::vector<float> ts;
        for (int r = 0; r < 7; ++r) { hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f / K); }
        std::sort(ts.begin(), ts.end());
        printf("%-24s %7.2f us per launch (median of 7 graph replays of %d launches)  -> %.2f TB/s at 46 B/board\n", k.name, ts[3], K, n * 46.0 / ts[3] / 1e6);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
