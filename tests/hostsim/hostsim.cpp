// hostsim.cpp -- TEST HARNESS ONLY (lives under tests/, never shipped, never loaded by the product).
//
// Compiles csrc/g2048_board.h -- the exact per-board arithmetic the HIP kernels run, with the two
// gfx950 builtins (v_perm_b32, v_dot4_u32_u8) replaced by portable stand-ins (hostsim_intrinsics.h) -- for the host CPU,
// so the `-m "not gpu"` suite can check every SWAR routine against the oracle without a GPU. It is not
// a CPU back-end: include/g2048.h has no entry point that reaches this code.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "hostsim_intrinsics.h"      // portable stand-ins for v_perm_b32 / v_dot4_u32_u8, then the product header
#include "g2048_board.h"
#include "g2048_rng.h"

using namespace g2048;

static const uint32_t kDirTable[G2048_DIR_TABLE_WORDS] = G2048_DIR_TABLE_INIT;
static DirSel dir_sel(uint32_t action)
{
    const uint32_t *t = kDirTable + 8 * (action & 3u);
    return DirSel{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]};
}

static Board ld(const uint8_t *p) { Board b; memcpy(b.w, p, 16); return b; }
static void st(uint8_t *p, const Board &b) { memcpy(p, b.w, 16); }

extern "C" {

void hs_rng_keys(uint64_t seed, uint32_t domain, uint64_t index, uint32_t *k0, uint32_t *k1)
{
    const Keys k = rng_keys(seed, domain, index);
    *k0 = k.k0; *k1 = k.k1;
}

uint32_t hs_rng_draw(uint32_t k0, uint32_t k1, uint64_t id, uint32_t ctr) { return rng_draw(k0, k1, id, ctr); }
// the form the step kernel uses: the id's high word folded into the key once per launch, the lane hashes the low word
uint32_t hs_rng_draw_split(uint32_t k0, uint32_t k1, uint64_t id, uint32_t ctr) { return rng_draw_lo(k0, k1 + rng_hi_term(id), (uint32_t)id, ctr); }

void hs_move(const uint8_t *in, const uint8_t *actions, int agent, uint8_t *out, uint32_t *gain, uint8_t *valid, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const Board b = ld(in + 16 * i);
        uint32_t g;
        Board o = agent ? move_agent(b, actions[i] & 3u, g, false) : move_env(b, actions[i] & 3u, g);
        if (!agent) {       // the table-driven direction network must agree with the select-based move on everything
            uint32_t g2, m1, m2;
            const Board o1 = move_env(b, actions[i] & 3u, g, m1), o2 = move_env_sel(b, dir_sel(actions[i]), g2, m2);
            if (!same(o1, o2) || g2 != g || m1 != m2) o.w[0] = 0xffffffffu;
        }
        if (agent) {        // the beam kernel's axis-pair move through loop-invariant selectors must give the same boards,
            Board f, r;     // with the DOWN quirk (move_agent) and without it (the env's move)
            move_axis_sel(b, axis_sel((actions[i] & 1u) != 0u, false), f, r);
            if (!same((actions[i] & 2u) ? r : f, o)) o.w[0] = 0xffffffffu;
            uint32_t g3;
            const Board e = move_env(b, actions[i] & 3u, g3);
            move_axis_sel(b, axis_sel((actions[i] & 1u) != 0u, true), f, r);
            if (!same((actions[i] & 2u) ? r : f, e)) o.w[0] = 0xffffffffu;
        }
        st(out + 16 * i, o); gain[i] = g; valid[i] = !same(o, b);
    }
}

void hs_valid(const uint8_t *in, int agent, uint8_t *mask, size_t n)
{
    for (size_t i = 0; i < n; ++i) mask[i] = (uint8_t)(agent ? valid_mask_agent(ld(in + 16 * i), false) : valid_mask_env(ld(in + 16 * i)));
}

void hs_spawn(const uint8_t *in, const uint32_t *h, uint8_t *out, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        Board b = ld(in + 16 * i), c = b;
        uint32_t za[4], zb[4];
        const uint32_t na = spawn_rowprefix(b, h[i], true, za), nb = spawn_prefix(c, h[i], true, zb);
        // the two formulations (beam kernel / step kernel) must agree on everything; a disagreement poisons the output so the test fails
        if (!same(b, c) || na != nb || memcmp(za, zb, sizeof za) != 0) b.w[0] = 0xffffffffu;
        st(out + 16 * i, b);
    }
}

void hs_step(const uint8_t *in, const uint8_t *actions, const uint32_t *h, uint8_t *out, uint32_t *score,
             double *reward, uint8_t *flags, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const StepOut o = step_board_sel(ld(in + 16 * i), dir_sel(actions[i]), h[i]);      // what step_kernel runs
        const StepOut o2 = step_board(ld(in + 16 * i), actions[i] & 3u, h[i]);             // what play_kernel runs
        if (!same(o.board, o2.board) || o.gain != o2.gain || o.flags != o2.flags || memcmp(&o.reward, &o2.reward, 8) != 0) {
            st(out + 16 * i, Board{{0xffffffffu, 0, 0, 0}});
            continue;
        }
        st(out + 16 * i, o.board); score[i] += o.gain; reward[i] = o.reward; flags[i] = (uint8_t)o.flags;
    }
}

void hs_reset(const uint32_t *h0, const uint32_t *h1, uint8_t *out, size_t n)
{
    for (size_t i = 0; i < n; ++i) st(out + 16 * i, fresh_board(h0[i], h1[i]));
}

void hs_eval(const uint8_t *in, int kind, const uint8_t *phase, double *out, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const Board b = ld(in + 16 * i);
        out[i] = kind == 0 ? eval_fast(b)
               : kind == 1 ? eval_full(b, phase ? phase[i] : phase_of(max_code(b), 512u, 1024u))
               : kind == 2 ? eval_ppo_heuristic(b) : kind == 7 ? eval_ppo_shaping(b, 0.0) : kind == 8 ? eval_pattern(b)
               : kind == 9 ? (double)max_corner_code(b) * 2.0 : kind == 10 ? (double)merge_potential(b)
               : eval_monotonicity(b, kind - 3);
    }
}

void hs_simulate(const uint8_t *in, const uint8_t *actions, const uint8_t *hc, uint8_t *succ, double *reward, uint8_t *done,
                 uint8_t *count, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const Board state = ld(in + 16 * i);
        Board moved; uint32_t gain;
        const uint32_t ns = simulate_count(state, actions[i] & 3u, moved, gain);
        count[i] = (uint8_t)ns;
        for (uint32_t k = 0; k < ns; ++k) {
            const SimOut o = simulate_successor(state, moved, gain, k, hc[i]);
            st(succ + (i * 32 + k) * 16, o.board); reward[i * 32 + k] = o.reward; done[i * 32 + k] = o.done;
        }
    }
}

// the hybrid agent's sampled simulate_move: 8 slots per state, as the kernel lays them out
void hs_simulate_sampled(const uint8_t *in, const uint8_t *actions, const uint32_t *h3, uint8_t *succ, double *reward, uint8_t *done,
                         uint8_t *count, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const Board state = ld(in + 16 * i);
        uint32_t gain;
        const Board moved = move_env(state, actions[i] & 3u, gain);
        const uint32_t ne = count_empty(moved);
        const bool valid = !same(moved, state);
        const uint32_t picks = ne < 3u ? ne : 3u;
        const uint32_t ns = (!valid || ne == 0u) ? 1u : 2u * picks;
        count[i] = (uint8_t)ns;
        for (uint32_t k = 0; k < 8; ++k) { st(succ + (i * 8 + k) * 16, Board{{0, 0, 0, 0}}); reward[i * 8 + k] = 0.0; done[i * 8 + k] = 0; }
        if (!valid) { st(succ + i * 8 * 16, moved); reward[i * 8] = -1.0; continue; }
        if (ne == 0u) { st(succ + i * 8 * 16, moved); done[i * 8] = 1; continue; }
        for (uint32_t k = 0; k < ns; ++k) {
            const SampledOut o = simulate_sampled_successor(state, moved, k, ne, h3[3 * i], h3[3 * i + 1], h3[3 * i + 2]);
            st(succ + (i * 8 + k) * 16, o.board); reward[i * 8 + k] = o.reward;
        }
    }
}

void hs_sample(const float *probs, const uint8_t *mask, const uint32_t *h, uint8_t *actions, float *prob, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        actions[i] = (uint8_t)sample_action(probs[4 * i], probs[4 * i + 1], probs[4 * i + 2], probs[4 * i + 3], mask[i], h[i], prob[i]);
}

void hs_transpose(const uint8_t *in, uint8_t *out, uint8_t *rot, size_t n)
{
    for (size_t i = 0; i < n; ++i) { st(out + 16 * i, transpose(ld(in + 16 * i))); st(rot + 16 * i, rot180(ld(in + 16 * i))); }
}

// the keyed bijection behind g2048_minibatch_gather (g2048_rng.h): indices of samples 0 .. batch-1 out of n
void hs_minibatch_indices(uint64_t n, uint64_t batch, uint32_t k0, uint32_t k1, uint64_t *out)
{
    const uint32_t hb = minibatch_half_bits(n);
    for (uint64_t j = 0; j < batch; ++j) out[j] = minibatch_index(j, n, hb, k0, k1);
}

}  // extern "C"
