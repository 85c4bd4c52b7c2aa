// Where do the blocks of a 4096 x 64-thread launch land?  Records (XCC, SE, CU, SIMD, wave slot) per block for a kernel
// with the beam kernel's footprint (one wavefront per block, ~4.4 KB LDS), blocks kept alive ~20 us so that all are
// resident together.  Build: hipcc --offload-arch=gfx950 -O2 -o placement placement.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(64) void probe(uint32_t* hw, uint32_t* xcc, unsigned long long* t_start, int spin)
{
    __shared__ uint32_t lds[1100];
    const uint32_t h = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
    const uint32_t x = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x] = h;
    uint32_t acc = 0;
    while (wall_clock64() - t0 < (unsigned long long)spin) acc += lds[(threadIdx.x + acc) & 63];
    if (threadIdx.x == 0) { hw[blockIdx.x] = h + (acc & 0u); xcc[blockIdx.x] = x; t_start[blockIdx.x] = t0; }
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    uint32_t *hw, *xcc; unsigned long long* ts;
    hipMalloc(&hw, n * 4); hipMalloc(&xcc, n * 4); hipMalloc(&ts, n * 8);
    std::vector<uint32_t> h(n), x(n); std::vector<unsigned long long> t(n);
    for (int rep = 0; rep < 3; ++rep) {
        probe<<<n, 64>>>(hw, xcc, ts, 2000);   // 20 us
        hipDeviceSynchronize();
        hipMemcpy(h.data(), hw, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(x.data(), xcc, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(t.data(), ts, n * 8, hipMemcpyDeviceToHost);
        std::map<uint32_t, std::vector<int>> per_simd;
        for (int b = 0; b < n; ++b) {
            const uint32_t simd = (h[b] >> 4) & 3, cu = (h[b] >> 8) & 15, sh = (h[b] >> 12) & 1, se = (h[b] >> 13) & 7, xc = x[b] & 15;
            per_simd[(xc << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd].push_back(b);
        }
        std::map<size_t, int> hist;
        for (auto& kv : per_simd) hist[kv.second.size()]++;
        printf("rep %d: %zu distinct SIMDs;", rep, per_simd.size());
        for (auto& kv : hist) printf("  %d SIMDs with %zu blocks", kv.second, kv.first);
        printf("\n");
        if (rep == 2) {
            printf("first 48 blocks: block -> xcc se sh cu simd wave\n");
            for (int b = 0; b < 48; ++b)
                printf("  %4d -> %u %u %u %2u %u %u\n", b, x[b] & 15, (h[b] >> 13) & 7, (h[b] >> 12) & 1, (h[b] >> 8) & 15, (h[b] >> 4) & 3, h[b] & 15);
            int shown = 0;
            for (auto& kv : per_simd) {
                if (shown++ >= 12) break;
                printf("  simd %05x:", kv.first);
                for (int b : kv.second) printf(" %d", b);
                printf("\n");
            }
            // does block b share its SIMD with b + 1024k?
            int same = 0, tot = 0;
            for (auto& kv : per_simd) for (size_t i = 1; i < kv.second.size(); ++i) { tot++; same += ((kv.second[i] - kv.second[0]) % 1024 == 0); }
            printf("blocks sharing a SIMD whose index differs by a multiple of 1024: %d of %d\n", same, tot);
            std::map<int,int> d;
            for (auto& kv : per_simd) for (size_t i = 1; i < kv.second.size(); ++i) d[kv.second[i] - kv.second[i-1]]++;
            printf("index gaps between consecutive blocks of one SIMD:");
            int c = 0; for (auto& kv : d) { if (c++ < 16) printf(" %d:%d", kv.first, kv.second); }
            printf("\n");
        }
    }
    return 0;
}
