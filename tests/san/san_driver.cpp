// san_driver.cpp -- TEST ONLY. Built with -fsanitize=address,undefined together with the oracle (C) and the host
// build of csrc/g2048_board.h; runs every routine over a spread of random and degenerate boards so that
// out-of-bounds accesses, invalid shifts, signed overflow or clz(0) in either implementation abort the run.
// (GPU sanitizers are not available on this pool; this covers the same arithmetic on the CPU.)
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "../hostsim/hostsim_intrinsics.h"
#include "g2048_board.h"
#include "g2048_rng.h"
extern "C" {
#include "g2048_oracle.h"
}

using namespace g2048;

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }

int main()
{
    uint32_t s = 12345u;
    size_t mism = 0, n = 0;
    std::vector<uint8_t> boards;
    for (int rep = 0; rep < 60000; ++rep) {
        uint8_t c[16];
        const uint32_t style = lcg(s) >> 29;
        for (int i = 0; i < 16; ++i) {
            const uint32_t r = lcg(s) >> 8;
            c[i] = style == 0 ? 0 : style == 1 ? (uint8_t)(r % 18) : style == 2 ? 17 : (r % 3 == 0 ? 0 : (uint8_t)(1 + r % (style * 3)));
        }
        boards.insert(boards.end(), c, c + 16);
    }
    n = boards.size() / 16;
    for (size_t i = 0; i < n; ++i) {
        Board b; memcpy(b.w, &boards[16 * i], 16);
        int32_t t[16]; g2048o_unpack(&boards[16 * i], t, 1);
        const uint32_t h = lcg(s), a = lcg(s) >> 30;
        // step
        const StepOut o = step_board(b, a, h);
        int32_t tb[16]; memcpy(tb, t, sizeof tb);
        int32_t sc = 0; double r; int done;
        const int valid = g2048o_env_step(tb, &sc, (int)a, h, &r, &done, NULL);
        uint8_t pk[16]; g2048o_pack(tb, pk, 1);
        mism += memcmp(pk, o.board.w, 16) != 0;
        mism += (uint32_t)sc != o.gain || (o.flags & 1u) != (uint32_t)done || ((o.flags >> 1) & 1u) != (uint32_t)valid;
        mism += !(r == o.reward || (r != r && o.reward != o.reward));
        // masks, agent move, evals, simulate, fill
        mism += valid_mask_env(b) != (uint32_t)g2048o_env_valid_mask(t);
        mism += valid_mask_agent(b, false) != (uint32_t)g2048o_agent_valid_mask(t);
        uint32_t g;
        const Board am = move_agent(b, a, g, false);
        int32_t ao[16], asc; int av;
        g2048o_agent_move(t, (int)a, ao, &asc, &av);
        g2048o_pack(ao, pk, 1);
        mism += memcmp(pk, am.w, 16) != 0 || (uint32_t)asc != g;
        mism += eval_fast(b) != g2048o_fast_eval(t);
        mism += eval_full(b, i % 3) != g2048o_full_eval(t, (int)(i % 3));
        mism += eval_ppo_heuristic(b) != g2048o_ppo_heuristic(t);
        mism += eval_ppo_shaping(b, 0.0) != g2048o_ppo_shaping(t, 0.0);
        Board moved; uint32_t gain;
        const uint32_t ns = simulate_count(b, a, moved, gain);
        int32_t succ[32 * 16]; double rw[32]; uint8_t dn[32];
        const int ons = g2048o_simulate_move(t, (int)a, 1 << (int)(1 + i % 17), succ, rw, dn);
        mism += (int)ns != ons;
        for (uint32_t k = 0; k < ns && (int)k < ons; ++k) {
            const SimOut so = simulate_successor(b, moved, gain, k, 1 + (uint32_t)(i % 17));
            g2048o_pack(succ + 16 * k, pk, 1);
            mism += memcmp(pk, so.board.w, 16) != 0 || so.done != (dn[k] != 0);
            mism += !(so.reward == rw[k] || (so.reward != so.reward && rw[k] != rw[k]));
        }
        float pa, opa; const float pr[4] = {0.1f * (float)(i % 7), 0.2f, 0.0f, 0.3f};
        mism += sample_action(pr[0], pr[1], pr[2], pr[3], (uint32_t)(i % 16), h, pa) != (uint32_t)g2048o_sample_action(pr, (int)(i % 16), h, &opa);
        mism += pa != opa;
        const Keys k = rng_keys(i * 7919ull, (uint32_t)(i % 7), i);
        uint32_t k0, k1; g2048o_rng_keys(i * 7919ull, (uint32_t)(i % 7), i, &k0, &k1);
        mism += k.k0 != k0 || k.k1 != k1 || rng_draw(k0, k1, (uint64_t)i << 29, (uint32_t)i) != g2048o_rng_draw(k0, k1, (uint64_t)i << 29, (uint32_t)i);
    }
    // beam search through the oracle on a few hundred roots (bounds of its internal arrays, width up to the maximum)
    for (size_t i = 0; i < 300; ++i) {
        int32_t t[16]; g2048o_unpack(&boards[16 * (i * 37 % n)], t, 1);
        int a; float p; uint32_t nc, ne;
        g2048o_beam_get_action(t, (i & 1) ? (int)(i % 16) : -1, 1 + (int)(i % 32), 4 + (int)(i % 27), 512, 1024, NULL, 0, 1, i, i, &a, &p, &nc, &ne, NULL, NULL, 0);
        mism += a < 0 || a > 3;
    }
    printf("boards %zu mismatches %zu\n", n, mism);
    return mism ? 1 : 0;
}
