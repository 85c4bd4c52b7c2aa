/* abi_smoke.c -- drives include/g2048.h from plain C with nothing but the HIP runtime (no Python, no torch):
 * synthesises N boards and actions on the device, steps them T times in place, runs one beam decision for the
 * first G boards, and prints FNV-1a checksums of every output array. tests/test_gpu_abi_c.py compares the
 * checksums with the oracle's for the same seeds. Build: gcc -std=c11 (tests/abi_c/Makefile). */
#include <stdint.h>
#include <stddef.h>

/* the handful of HIP runtime entry points this client needs, declared by hand so that the file stays plain C with no
 * HIP headers (libamdhip64's C API) */
typedef int hipError_t;
typedef void *hipStream_t;
#define hipSuccess 0
#define hipMemcpyDeviceToHost 2
hipError_t hipMalloc(void **ptr, size_t size);
hipError_t hipMemset(void *dst, int value, size_t size);
hipError_t hipMemcpy(void *dst, const void *src, size_t size, int kind);
hipError_t hipStreamCreate(hipStream_t *stream);
hipError_t hipStreamSynchronize(hipStream_t stream);
const char *hipGetErrorString(hipError_t e);

#include <stdio.h>
#include <stdlib.h>

#include "g2048.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_G(x) do { int r_ = (x); if (r_ != G2048_OK) { fprintf(stderr, "g2048 error %d: %s\n", r_, g2048_last_error()); return 3; } } while (0)

static uint64_t fnv1a(const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    size_t n = argc > 1 ? (size_t)atol(argv[1]) : 100000;
    int steps = argc > 2 ? atoi(argv[2]) : 5;
    size_t games = argc > 3 ? (size_t)atol(argv[3]) : 256;
    uint64_t seed = 0x2048;
    if (g2048_abi_version() != G2048_ABI_VERSION || g2048_device_count() < 1) { fprintf(stderr, "no device / ABI mismatch\n"); return 1; }
    void *boards; uint8_t *actions, *flags, *beam_a; uint32_t *score, *expd; double *reward; float *beam_p;
    CHECK_HIP(hipMalloc(&boards, n * 16)); CHECK_HIP(hipMalloc((void **)&actions, n)); CHECK_HIP(hipMalloc((void **)&flags, n));
    CHECK_HIP(hipMalloc((void **)&score, n * 4)); CHECK_HIP(hipMalloc((void **)&reward, n * 8));
    CHECK_HIP(hipMalloc((void **)&beam_a, games)); CHECK_HIP(hipMalloc((void **)&beam_p, games * 4)); CHECK_HIP(hipMalloc((void **)&expd, games * 4));
    CHECK_HIP(hipMemset(score, 0, n * 4));
    hipStream_t st; CHECK_HIP(hipStreamCreate(&st));
    CHECK_G(g2048_synth_boards(boards, seed, 0, n, 19661 /* 0.30 */, 11, st));
    for (int t = 0; t < steps; ++t) {
        CHECK_G(g2048_synth_actions(actions, seed, (uint64_t)t, 0, n, st));
        CHECK_G(g2048_step(boards, actions, boards, score, reward, flags, seed, (uint64_t)t, 0, n, G2048_STEP_REWARD_F64, st));
    }
    CHECK_G(g2048_beam_get_action(boards, NULL, beam_a, beam_p, expd, 20, 30, 512, 1024, seed, 0, 0, games, 0, st));
    CHECK_HIP(hipStreamSynchronize(st));
    uint8_t *hb = (uint8_t *)malloc(n * 16), *hf = (uint8_t *)malloc(n), *ha = (uint8_t *)malloc(games);
    uint32_t *hs = (uint32_t *)malloc(n * 4), *he = (uint32_t *)malloc(games * 4); double *hr = (double *)malloc(n * 8);
    CHECK_HIP(hipMemcpy(hb, boards, n * 16, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(hf, flags, n, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hs, score, n * 4, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(hr, reward, n * 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(ha, beam_a, games, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(he, expd, games * 4, hipMemcpyDeviceToHost));
    printf("boards %016llx score %016llx reward %016llx flags %016llx beam_action %016llx beam_expanded %016llx\n",
           (unsigned long long)fnv1a(hb, n * 16), (unsigned long long)fnv1a(hs, n * 4), (unsigned long long)fnv1a(hr, n * 8),
           (unsigned long long)fnv1a(hf, n), (unsigned long long)fnv1a(ha, games), (unsigned long long)fnv1a(he, games * 4));
    return 0;
}
