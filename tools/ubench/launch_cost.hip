// Host cost of one kernel launch on this box, by launch API, for a kernel with step_kernel's argument list (15 arguments, 88 bytes):
//   hipLaunchKernelGGL (what the library uses), hipModuleLaunchKernel on the hipFunction_t of the same kernel (hipGetFuncBySymbol),
//   the same with the arguments passed as ONE packed buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER), and a replay of a linear hipGraph.
// K launches alternating between two streams, the GPU work negligible (one block): what is timed is the host's enqueue loop.
// hipcc --offload-arch=gfx950 -O2 -o launch_cost launch_cost.hip && ./launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k15(size_t n, const uint32_t *kb, const uint4 *in, const uint8_t *act, uint32_t *score, uint64_t id_base, uint32_t k0, uint32_t k1,
                    uint4 *out, void *reward, uint8_t *flags, uint32_t e0, uint32_t e1, uint32_t a0, uint32_t a1)
{
    if (threadIdx.x == 0 && n == 12345) score[0] = k0 + k1 + e0 + e1 + a0 + a1 + (uint32_t)id_base;
    if (a1 > 6u) {                               // busy variant: ~a1 x 64 ns of work per launch, so that the queues are never empty
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)a1 * 6400ull / 1000ull) { }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int K = 2000;
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    uint32_t *buf;
    CK(hipMalloc(&buf, 4096));
    size_t n = 1; const uint32_t *kb = nullptr; const uint4 *in = (const uint4 *)buf; const uint8_t *act = (const uint8_t *)buf; uint32_t *score = buf;
    uint64_t idb = 0; uint32_t k0 = 1, k1 = 2, e0 = 3, e1 = 4, a0 = 5, a1 = 6; uint4 *out = (uint4 *)buf; void *rw = buf; uint8_t *fl = (uint8_t *)buf;
    for (int rep = 0; rep < 3; ++rep) {
        for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(k15, dim3(1), dim3(64), 0, s[w & 1], n, kb, in, act, score, idb, k0, k1, out, rw, fl, e0, e1, a0, a1);
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int w = 0; w < K; ++w) hipLaunchKernelGGL(k15, dim3(1), dim3(64), 0, s[w & 1], n, kb, in, act, score, idb, k0, k1, out, rw, fl, e0, e1, a0, a1);
        double t1 = now();
        CK(hipDeviceSynchronize());
        printf("hipLaunchKernelGGL, alternating streams:        %.2f us per launch\n", (t1 - t0) * 1e6 / K);
        t0 = now();
        for (int w = 0; w < K; ++w) hipLaunchKernelGGL(k15, dim3(1), dim3(64), 0, s[0], n, kb, in, act, score, idb, k0, k1, out, rw, fl, e0, e1, a0, a1);
        t1 = now();
        CK(hipDeviceSynchronize());
        printf("hipLaunchKernelGGL, one stream:                 %.2f us per launch\n", (t1 - t0) * 1e6 / K);

        // the same with stream 0 of the pair being the NULL stream (what torch.cuda.current_stream() is outside a stream context)
        hipStream_t sn[2] = {nullptr, s[1]};
        t0 = now();
        for (int w = 0; w < K; ++w) hipLaunchKernelGGL(k15, dim3(1), dim3(64), 0, sn[w & 1], n, kb, in, act, score, idb, k0, k1, out, rw, fl, e0, e1, a0, a1);
        t1 = now();
        CK(hipDeviceSynchronize());
        printf("hipLaunchKernelGGL, NULL stream / non-blocking stream alternating: %.2f us per launch\n", (t1 - t0) * 1e6 / K);
        // busy kernels (~5 us each): 40 launches from idle, as the bench's timed region
        for (int busy = 0; busy < 2; ++busy) {
            hipStream_t *ss = busy ? sn : s;
            double acc = 0;
            for (int r = 0; r < 50; ++r) {
                CK(hipDeviceSynchronize());
                t0 = now();
                for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(k15, dim3(1), dim3(64), 0, ss[w & 1], n, kb, in, act, score, idb, k0, k1, out, rw, fl, e0, e1, a0, 800u);
                t1 = now();
                acc += t1 - t0;
            }
            CK(hipDeviceSynchronize());
            printf("  40 busy (5 us) launches from idle, %s: %.2f us of host time per launch\n", busy ? "NULL / non-blocking" : "two non-blocking streams", acc * 1e6 / 50 / 40);
        }

        hipFunction_t f;
        CK(hipGetFuncBySymbol(&f, (const void *)k15));
        void *args[] = {&n, &kb, &in, &act, &score, &idb, &k0, &k1, &out, &rw, &fl, &e0, &e1, &a0, &a1};
        t0 = now();
        for (int w = 0; w < K; ++w) CK(hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, s[w & 1], args, nullptr));
        t1 = now();
        CK(hipDeviceSynchronize());
        printf("hipModuleLaunchKernel (args array), alternating: %.2f us per launch\n", (t1 - t0) * 1e6 / K);

        struct __attribute__((packed, aligned(8))) Packed { size_t n; const void *kb, *in, *act, *score; uint64_t idb; uint32_t k0, k1; void *out, *rw, *fl; uint32_t e0, e1, a0, a1; } p =
            {n, kb, in, act, score, idb, k0, k1, out, rw, fl, e0, e1, a0, a1};
        size_t psz = sizeof p;
        void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psz, HIP_LAUNCH_PARAM_END};
        t0 = now();
        for (int w = 0; w < K; ++w) CK(hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, s[w & 1], nullptr, cfg));
        t1 = now();
        CK(hipDeviceSynchronize());
        printf("hipModuleLaunchKernel (packed buffer), alternating: %.2f us per launch\n", (t1 - t0) * 1e6 / K);

        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k15, dim3(1), dim3(64), 0, s[0], n, kb, in, act, score, idb, k0, k1, out, rw, fl, e0, e1, a0, a1);
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s[0])); CK(hipDeviceSynchronize());
        t0 = now();
        for (int w = 0; w < 100; ++w) CK(hipGraphLaunch(ge, s[0]));
        t1 = now();
        CK(hipDeviceSynchronize());
        printf("hipGraphLaunch of a 20-kernel linear graph:      %.2f us per graph = %.2f us per kernel\n", (t1 - t0) * 1e6 / 100, (t1 - t0) * 1e6 / 2000);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
