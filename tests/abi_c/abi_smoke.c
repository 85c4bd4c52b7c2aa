/* abi_smoke.c -- drives include/g2048.h from plain C with nothing but the HIP runtime (no Python, no torch):
 * synthesises N boards and actions on the device, steps them T times in place, runs one beam decision for the
 * first G boards, and prints FNV-1a checksums of every output array. tests/test_gpu_abi_c.py compares the
 * checksums with the oracle's for the same seeds. Round 4 (ABI 3): P complete games with their move-sets recorded
 * (g2048_play_games, actions_out), replayed into histories (g2048_replay_games) and checked here against the kernel's own
 * final boards / scores, and one env driven through g2048_env_step (reset, a move, an out-of-range action). Build: gcc -std=c11 (tests/abi_c/Makefile). */
#include <stdint.h>
#include <stddef.h>

/* the handful of HIP runtime entry points this client needs, declared by hand so that the file stays plain C with no
 * HIP headers (libamdhip64's C API) */
typedef int hipError_t;
typedef void *hipStream_t;
#define hipSuccess 0
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
hipError_t hipMalloc(void **ptr, size_t size);
hipError_t hipMemset(void *dst, int value, size_t size);
hipError_t hipMemcpy(void *dst, const void *src, size_t size, int kind);
hipError_t hipStreamCreate(hipStream_t *stream);
hipError_t hipStreamSynchronize(hipStream_t stream);
const char *hipGetErrorString(hipError_t e);

#include <stdio.h>
#include <stdlib.h>

#include "g2048.h"
#include "g2048_testing.h"   /* synthetic benchmark inputs (g2048_synth_*) */

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

    /* ---- round 4: move-sets, replayed histories, the one-launch env step ---- */
    size_t pg = argc > 4 ? (size_t)atol(argv[4]) : 32;
    const int cap = 300, hist = cap + 1;
    void *pb, *pb0, *bh; uint32_t *ps, *sh; int32_t *mv, *va, *iv, *ms; uint8_t *alive, *acts, *fh;
    CHECK_HIP(hipMalloc(&pb, pg * 16)); CHECK_HIP(hipMalloc(&pb0, pg * 16)); CHECK_HIP(hipMalloc((void **)&ps, pg * 4));
    CHECK_HIP(hipMalloc((void **)&mv, pg * 4)); CHECK_HIP(hipMalloc((void **)&va, pg * 4)); CHECK_HIP(hipMalloc((void **)&iv, pg * 4));
    CHECK_HIP(hipMalloc((void **)&ms, pg * 32)); CHECK_HIP(hipMalloc((void **)&alive, pg)); CHECK_HIP(hipMalloc((void **)&acts, pg * cap));
    CHECK_HIP(hipMalloc(&bh, pg * hist * 16)); CHECK_HIP(hipMalloc((void **)&sh, pg * hist * 4)); CHECK_HIP(hipMalloc((void **)&fh, pg * hist));
    CHECK_G(g2048_reset(pb, ps, seed, 0, 1000, pg, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(pb0, pb, pg * 16, hipMemcpyDeviceToDevice));
    CHECK_G(g2048_play_games(pb, ps, mv, va, iv, ms, NULL, alive, acts, 8, 6, 512, 1024, cap, seed, 1000, pg, 0, st));
    CHECK_G(g2048_replay_games(pb0, NULL, NULL, 1000, acts, (size_t)cap, mv, bh, sh, fh, (size_t)hist, seed, pg, st));
    CHECK_HIP(hipStreamSynchronize(st));
    uint8_t *hpb = (uint8_t *)malloc(pg * 16), *hacts = (uint8_t *)malloc(pg * cap), *hbh = (uint8_t *)malloc(pg * hist * 16);
    uint32_t *hps = (uint32_t *)malloc(pg * 4), *hsh = (uint32_t *)malloc(pg * hist * 4); int32_t *hmv = (int32_t *)malloc(pg * 4);
    CHECK_HIP(hipMemcpy(hpb, pb, pg * 16, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(hacts, acts, pg * cap, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hbh, bh, pg * hist * 16, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(hps, ps, pg * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hsh, sh, pg * hist * 4, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(hmv, mv, pg * 4, hipMemcpyDeviceToHost));
    int replay_ok = 1;
    for (size_t g = 0; g < pg; ++g) {       /* entry moves[g] of a replayed history = what the fused kernel ended with */
        const size_t at = g * hist + (size_t)hmv[g];
        for (int k = 0; k < 16; ++k) if (hbh[at * 16 + k] != hpb[g * 16 + k]) replay_ok = 0;
        if (hsh[at] != hps[g]) replay_ok = 0;
        for (int t = 0; t < cap; ++t) if ((t < hmv[g]) != (hacts[g * cap + t] <= 3)) replay_ok = 0;      /* 0..3 up to the end, 0xFF after it */
    }
    void *eb, *rec; uint32_t *es;
    CHECK_HIP(hipMalloc(&eb, 16)); CHECK_HIP(hipMalloc((void **)&es, 4)); CHECK_HIP(hipMalloc(&rec, 3 * G2048_ENV_RECORD_BYTES));
    CHECK_HIP(hipMemset(eb, 0, 16)); CHECK_HIP(hipMemset(es, 0, 4));
    CHECK_G(g2048_env_step(eb, es, 0, G2048_ENV_OP_RESET, rec, seed, 0, 7, st));
    CHECK_G(g2048_env_step(eb, es, 1, G2048_ENV_OP_STEP, (char *)rec + G2048_ENV_RECORD_BYTES, seed, 0, 7, st));
    CHECK_G(g2048_env_step(eb, es, 7, G2048_ENV_OP_STEP, (char *)rec + 2 * G2048_ENV_RECORD_BYTES, seed, 1, 7, st));
    CHECK_HIP(hipStreamSynchronize(st));
    uint8_t hrec[3 * G2048_ENV_RECORD_BYTES];
    CHECK_HIP(hipMemcpy(hrec, rec, sizeof hrec, hipMemcpyDeviceToHost));
    printf("replay_ok %d play_actions %016llx play_moves %016llx play_scores %016llx env_records %016llx\n", replay_ok,
           (unsigned long long)fnv1a(hacts, pg * cap), (unsigned long long)fnv1a(hmv, pg * 4), (unsigned long long)fnv1a(hps, pg * 4),
           (unsigned long long)fnv1a(hrec, sizeof hrec));
    return 0;
}
