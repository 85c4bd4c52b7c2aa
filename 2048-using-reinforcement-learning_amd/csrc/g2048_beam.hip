// g2048_beam.hip -- BeamSearchAgent.get_action (reference agents/beam_search_agent.py:71-181) on gfx950.
//
// One wavefront owns one game. The beam (<= 32 boards, 16 B each) lives in LDS. Each level has two stages:
//   A. lane 2p + axis makes BOTH moves of one axis of parent p (g2048_board.h move_axis: one transpose in, two slides,
//      two transposes out; the same SWAR slide the env kernel uses, plus the reference's rot180-DOWN quirk), so
//      2 * beam <= 64 lanes cover the four moves of every parent in one round; the children that changed the board are
//      packed into LDS at their ballot-prefix index, i.e. in the reference's generation order (parent rank, action);
//   B. one lane per valid child: the spawn -- child j of the decision takes draw j, exactly as the Python loop
//      consumes its RNG -- and the heuristic score, fed the empty count and max code the kernel already knows;
//      then the top-k: each candidate counts the candidates that sort before it (score descending, generation order
//      ascending = Python's stable sorted(reverse=True)) with broadcast LDS reads, and the first `width` write
//      themselves back to the beam at their rank. On _fast_evaluate levels the score is a small exact integer, so
//      (score, order) is one unique u32 key; levels 1..3 (_evaluate_state) rank f64 scores.
// Scores are computed in the reference's operation order (bit-exact with the oracle); no MFMA, no global memory
// traffic inside the search (root in, action out). beam_decide() is the search as a device function; beam_kernel
// runs it once per game (g2048_beam_get_action), play_kernel loops it with the env step (g2048_play_games).
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/g2048.h"
#include "g2048_board.h"
#include "g2048_rng.h"

using namespace g2048;

namespace {

constexpr int kMaxWidth = G2048_BEAM_MAX_WIDTH;

__device__ __forceinline__ uint32_t prefix_count(unsigned long long ballot)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

// LDS of one search (one wavefront = one game)
template <int PASSES>
struct BeamShared {
    static constexpr int kWidth = 16 * PASSES;      // widest beam this instance can hold (4 * width <= 64 * PASSES children)
    uint4 board[kWidth];                            // the beam, rank order
    uint32_t root[kWidth];                          // root action | (max code << 8) of each beam entry
    uint4 cboard[64 * PASSES];                      // moved (pre-spawn) boards of the VALID children,
    uint32_t croot[64 * PASSES];                    //   compacted in generation order, + root action | parent max << 8
    alignas(16) double score[64 * PASSES + 2];      // f64 scores (levels 1..3) or, reinterpreted, u32 keys
};

struct Decision { uint32_t action; float prob; uint32_t expanded; };

// BeamSearchAgent.get_action for the game this wavefront owns. mask_in < 0: no caller mask. Every lane returns the
// same Decision. Must be called by all 64 lanes (it contains workgroup barriers).
template <int PASSES>
__device__ __forceinline__ Decision beam_decide(BeamShared<PASSES> &sh, const Board &root, int mask_in, int width, int depth,
                                                uint32_t early_thr, uint32_t mid_thr, uint32_t k0, uint32_t k1, uint64_t gid,
                                                bool fixed_down)
{
    uint4 *const s_board = sh.board;
    uint32_t *const s_root = sh.root;
    uint4 *const s_cboard = sh.cboard;
    uint32_t *const s_croot = sh.croot;
    double *const s_score = sh.score;
    const uint32_t lane = threadIdx.x;

    // :82-93 -- caller mask or the agent's own validity; 0 or 1 valid move short-circuit
    const uint32_t mask = mask_in >= 0 ? (uint32_t)(mask_in & 15) : valid_mask_agent(root, fixed_down);
    const uint32_t nvalid = popc(mask);
    if (nvalid <= 1u) return Decision{nvalid ? (uint32_t)__builtin_ctz(mask) : 0u, nvalid ? 1.0f : 0.5f, 0u};
    // :96-106 -- phase and depth are fixed from the ROOT board
    const uint32_t root_max = max_code(root);
    const uint32_t phase = phase_of(root_max, early_thr, mid_thr);
    const uint32_t root_empty = count_empty(root);
    int actual_depth;
    if (root_empty <= 4u) actual_depth = min(depth + 5, 25);
    else if (root_empty >= 10u) actual_depth = min(depth - 5, 10);
    else actual_depth = depth;

    int nb = 0;                    // current beam size
    uint32_t draws = 0, expanded = 0;

    for (int level = 0; level == 0 || level < actual_depth; ++level) {
        const bool fast = level == 0 || level > 3;             // :122, :139
        // ---- stage A: lane 2p + axis makes BOTH moves of one axis of parent p (axis 0: LEFT, RIGHT; axis 1: UP,
        // DOWN), so 2 * beam <= 64 lanes cover all four moves of every parent in one round. Valid children are
        // compacted into LDS in generation order (parent rank, then action 0..3) = the order the reference draws in.
        uint32_t total_valid = 0;
        const int n_parents = level == 0 ? 1 : nb;
        for (int round = 0; round * 32 < n_parents; ++round) {               // one round unless the beam is wider than 32
            const uint32_t par = (uint32_t)round * 32u + (lane >> 1);
            const bool vertical = (lane & 1u) != 0u;
            const bool on = (int)par < n_parents;
            Board P = root;
            uint32_t ra_f = (vertical ? 1u : 0u) | (root_max << 8), ra_r = (vertical ? 3u : 2u) | (root_max << 8);
            bool en_f = on, en_r = on;
            if (level == 0) {
                en_f = on && ((mask >> (vertical ? 1 : 0)) & 1u);
                en_r = on && ((mask >> (vertical ? 3 : 2)) & 1u);
            } else if (on) {
                const uint4 pv = s_board[par];
                P = Board{{pv.x, pv.y, pv.z, pv.w}};
                ra_f = ra_r = s_root[par];
            }
            Board cf, cr;
            move_axis(P, vertical, cf, cr);                                      // :115 / :152
            if (!fixed_down) {                                                   // the agent's DOWN = rot180(true DOWN)
                const Board q = rot180(cr);
                cr.w[0] = vertical ? q.w[0] : cr.w[0]; cr.w[1] = vertical ? q.w[1] : cr.w[1];
                cr.w[2] = vertical ? q.w[2] : cr.w[2]; cr.w[3] = vertical ? q.w[3] : cr.w[3];
            }
            const bool vf = en_f && !same(cf, P), vr = en_r && !same(cr, P);
            const unsigned long long bf = __ballot(vf), br = __ballot(vr);
            const uint32_t before = total_valid + prefix_count(bf) + prefix_count(br);     // valid children generated earlier
            // partner lane (same parent, other axis) through a DPP quad swap
            const uint32_t mine = (vf ? 1u : 0u) | (vr ? 2u : 0u);
            const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
            // order inside a parent: LEFT(h,f) UP(v,f) RIGHT(h,r) DOWN(v,r)
            const uint32_t idx_f = vertical ? before - ((other >> 1) & 1u) : before;
            const uint32_t idx_r = vertical ? before + (mine & 1u) : before + (mine & 1u) + (other & 1u);
            total_valid += (uint32_t)__popcll(bf) + (uint32_t)__popcll(br);
            if (vf) { s_cboard[idx_f] = make_uint4(cf.w[0], cf.w[1], cf.w[2], cf.w[3]); s_croot[idx_f] = ra_f; }
            if (vr) { s_cboard[idx_r] = make_uint4(cr.w[0], cr.w[1], cr.w[2], cr.w[3]); s_croot[idx_r] = ra_r; }
        }
        expanded += total_valid;
        if (total_valid == 0u) {
            if (level == 0) {                                                   // :126-128 random valid action, prob 0.5
                uint32_t idx = ((rng_draw(k0, k1, gid, draws) >> 16) * nvalid) >> 16;
                uint32_t m = mask;
                while (idx--) m &= m - 1u;
                return Decision{(uint32_t)__builtin_ctz(m), 0.5f, 0u};
            }
            break;                                                              // :170-171 keep the previous beam
        }
        __syncthreads();
        // ---- stage B: spawn + score of the compacted children (one per lane; a second round only when
        // more than 64 children are valid)
        Board child[PASSES];
        double score[PASSES];
        uint32_t ikey[PASSES], cmaxv[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            score[p] = 0.0; ikey[p] = 0u; cmaxv[p] = 0u; child[p] = root;
            if ((uint32_t)(p * 64) < total_valid) {                              // wave-uniform
                const uint32_t ci = (uint32_t)(p * 64) + lane;
                const bool live = ci < total_valid;
                const uint4 cv = s_cboard[live ? ci : 0u];
                Board c = {{cv.x, cv.y, cv.z, cv.w}};
                const uint32_t pmax = s_croot[live ? ci : 0u] >> 8;
                const uint32_t n_moved = count_empty(c);
                const bool consume = live && n_moved != 0u;                      // :262-263
                const unsigned long long bc = __ballot(consume);
                const uint32_t j = draws + prefix_count(bc);
                draws += (uint32_t)__popcll(bc);
                {
                    Board s = c;
                    spawn(s, rng_draw(k0, k1, gid, j));                          // :118 / :155
                    c.w[0] = consume ? s.w[0] : c.w[0]; c.w[1] = consume ? s.w[1] : c.w[1];
                    c.w[2] = consume ? s.w[2] : c.w[2]; c.w[3] = consume ? s.w[3] : c.w[3];
                }
                // :122 / :158-161. _fast_evaluate is integer-valued (< 2^19), so on those levels the sort key
                // (score desc, generation order asc) is the single unique integer score * 512 + (511 - index);
                // the f64 path is only needed for _evaluate_state (levels 1..3).
                // what the evaluators need is already known: the empty count (one fewer after a spawn) and the
                // max code -- a move raises the parent's max by at most one, exactly when some cell now holds
                // parent max + 1 (two max tiles merged, or a 2/4 spawned onto a board whose max was lower)
                const uint32_t n_child = n_moved - (consume ? 1u : 0u);
                const uint32_t cmax = pmax + (has_code(c, pmax + 1u) ? 1u : 0u);
                cmaxv[p] = cmax;
                if (fast) {
                    ikey[p] = live ? (eval_fast_u32_known(c, n_child, cmax) << 9) + (511u - ci) : 0u;
                    reinterpret_cast<uint32_t *>(s_score)[ci] = ikey[p];         // ci < 64 * PASSES always
                } else {
                    score[p] = eval_full_known(c, phase, n_child, cmax);
                    s_score[ci] = live ? score[p] : -INFINITY;
                }
                child[p] = c;
            }
        }
        __syncthreads();
        // ---- stable descending rank (:131, :174) among the valid children
        uint32_t rank[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) rank[p] = 0u;
        if (fast) {
            const uint4 *keys = reinterpret_cast<const uint4 *>(s_score);
            for (uint32_t j = 0; j < total_valid; j += 4) {
                const uint4 kj = keys[j >> 2];
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    rank[p] += (kj.x > ikey[p] ? 1u : 0u) + (kj.y > ikey[p] ? 1u : 0u) +
                               (kj.z > ikey[p] ? 1u : 0u) + (kj.w > ikey[p] ? 1u : 0u);
                }
            }
        } else {
            for (uint32_t j = 0; j < total_valid; j += 2) {
                const double2 sj = *reinterpret_cast<const double2 *>(&s_score[j]);
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const uint32_t ci = (uint32_t)(p * 64) + lane;
                    rank[p] += (sj.x > score[p] || (sj.x == score[p] && j < ci)) ? 1u : 0u;
                    rank[p] += (sj.y > score[p] || (sj.y == score[p] && j + 1 < ci)) ? 1u : 0u;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const uint32_t ci = (uint32_t)(p * 64) + lane;
            if (ci < total_valid && rank[p] < (uint32_t)width) {                  // :132 / :175
                s_board[rank[p]] = make_uint4(child[p].w[0], child[p].w[1], child[p].w[2], child[p].w[3]);
                s_root[rank[p]] = (s_croot[ci] & 0xffu) | (cmaxv[p] << 8);
            }
        }
        nb = (int)min(total_valid, (uint32_t)width);
        __syncthreads();
    }
    const Decision d = {s_root[0] & 0xffu, 1.0f, expanded};                     // :178-181
    __syncthreads();                                                            // s_root[0] read before any reuse of the LDS
    return d;
}

template <int PASSES>
__global__ __launch_bounds__(64) void beam_kernel(const uint4 *__restrict__ roots, const uint8_t *__restrict__ mask_in,
                                                 uint8_t *__restrict__ action_out, float *__restrict__ prob_out,
                                                 uint32_t *__restrict__ expanded_out, int width, int depth,
                                                 uint32_t early_thr, uint32_t mid_thr, uint32_t k0, uint32_t k1,
                                                 uint64_t id_base, bool fixed_down, const uint32_t *__restrict__ keyblock)
{
    if (keyblock) { k0 = keyblock[4]; k1 = keyblock[5]; }            // KB_BEAM of the device key block
    __shared__ BeamShared<PASSES> sh;
    const size_t g = blockIdx.x;
    const uint4 rv = roots[g];
    const Board root = {{rv.x, rv.y, rv.z, rv.w}};
    const Decision d = beam_decide<PASSES>(sh, root, mask_in ? (int)(mask_in[g] & 15u) : -1, width, depth, early_thr, mid_thr,
                                           k0, k1, id_base + g, fixed_down);
    if (threadIdx.x == 0) {
        action_out[g] = (uint8_t)d.action;
        prob_out[g] = d.prob;
        if (expanded_out) expanded_out[g] = d.expanded;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Widths 17..20 (the reference's evaluation setting is 20): four games per 256-thread block, one wavefront each.
// A level can have up to 4 * width = 80 valid children, 16 more than a wavefront has lanes. Instead of every game paying
// a second 64-lane pass for its <= 16 leftover children, ONE wavefront of the block (the duty rotates with the level)
// spawns and scores the leftovers of all four games at once, 16 lanes per game, and hands boards and scores back through
// LDS. The ranking no longer compares every child with every other child: on the _fast_evaluate levels (unique u32
// keys) the top `width` are found by a radix select over the key bits with wave ballots (two VALU instructions per
// bit), and only those <= 20 winners are ranked against each other. Levels 1..3 (_evaluate_state, f64) keep the
// all-pairs count. Results are identical to beam_decide's (same generation order, same draws, same total order).
constexpr int kSharedGames = 4;
constexpr int kSharedMaxWidth = 20;
constexpr int kSharedChildren = 4 * kSharedMaxWidth;      // 80
constexpr int kSharedLeft = kSharedChildren - 64;         // 16 leftover children per game, at most

struct SharedGame {
    uint4 board[kSharedMaxWidth];                 // the beam, rank order
    uint32_t root[kSharedMaxWidth];               // root action | (max code << 8)
    uint4 cboard[kSharedChildren];                // valid children in generation order: moved boards; the leftovers
    uint32_t croot[kSharedChildren];              //   (index >= 64) are replaced by their spawned boards by the duty wave
    alignas(16) double score[kSharedChildren + 2];    // f64 scores of levels 1..3
    alignas(16) uint32_t lkey[kSharedLeft];       // u32 keys of the leftovers (fast levels)
    uint32_t lmax[kSharedLeft];                   // max code of the spawned leftovers
    alignas(16) uint32_t skey[kSharedMaxWidth + 4];   // keys of the selected children, for the final ranking
    uint32_t total_valid, draws, lconsumed, phase, levels;      // draws: index of the first leftover's draw
    uint32_t gid_lo, gid_hi;
};

__device__ __forceinline__ uint32_t popc64(unsigned long long m) { return (uint32_t)__popcll(m); }

// spawn + score of one valid child (moved board c, parent's max code pmax, draw j): what stage B does per lane
struct Scored { Board board; uint32_t key; double score; uint32_t cmax; };

__device__ __forceinline__ Scored spawn_and_score(Board c, uint32_t n_moved, uint32_t pmax, uint32_t draw, uint32_t ci, bool fast,
                                                   uint32_t phase)
{
    Scored o;
    spawn(c, draw);                                     // :118 / :155 -- a no-op on a full board (:262-263)
    const uint32_t n_child = n_moved - (n_moved ? 1u : 0u);
    const uint32_t cmax = pmax + (has_code(c, pmax + 1u) ? 1u : 0u);
    o.board = c; o.cmax = cmax; o.key = 0u; o.score = 0.0;
    if (fast) o.key = (eval_fast_u32_known(c, n_child, cmax) << 8) + (255u - ci);
    else o.score = eval_full_known(c, phase, n_child, cmax);
    return o;
}

__global__ __launch_bounds__(64 * kSharedGames) void beam_shared_kernel(
    const uint4 *__restrict__ roots, const uint8_t *__restrict__ mask_in, uint8_t *__restrict__ action_out,
    float *__restrict__ prob_out, uint32_t *__restrict__ expanded_out, int width, int depth, uint32_t early_thr,
    uint32_t mid_thr, uint32_t k0, uint32_t k1, uint64_t id_base, size_t n_games, bool fixed_down,
    const uint32_t *__restrict__ keyblock)
{
    if (keyblock) { k0 = keyblock[4]; k1 = keyblock[5]; }
    __shared__ SharedGame sh[kSharedGames];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    SharedGame &G = sh[wv];
    const size_t g = (size_t)blockIdx.x * kSharedGames + wv;
    const bool have = g < n_games;
    const uint64_t gid = id_base + g;
    Board root = {{1u, 0u, 0u, 0u}};
    if (have) { const uint4 rv = roots[g]; root = Board{{rv.x, rv.y, rv.z, rv.w}}; }
    const int mask_arg = (have && mask_in) ? (int)(mask_in[g] & 15u) : -1;
    const uint32_t mask = mask_arg >= 0 ? (uint32_t)mask_arg : valid_mask_agent(root, fixed_down);     // :82-84
    const uint32_t nvalid = popc(mask);
    const uint32_t root_max = max_code(root);
    const uint32_t phase = phase_of(root_max, early_thr, mid_thr);                                      // :96-97
    const uint32_t root_empty = count_empty(root);
    int actual_depth;                                                                                    // :100-106
    if (root_empty <= 4u) actual_depth = min(depth + 5, 25);
    else if (root_empty >= 10u) actual_depth = min(depth - 5, 10);
    else actual_depth = depth;
    bool active = have && nvalid > 1u;                              // :86-93: 0 or 1 valid move needs no search
    uint32_t result_action = nvalid ? (uint32_t)__builtin_ctz(mask) : 0u;
    float result_prob = nvalid == 1u ? 1.0f : 0.5f;
    if (lane == 0) {
        G.levels = active ? (uint32_t)max(actual_depth, 1) : 0u;
        G.phase = phase; G.gid_lo = (uint32_t)gid; G.gid_hi = (uint32_t)(gid >> 32);
        G.total_valid = 0u; G.draws = 0u;
    }
    __syncthreads();
    const uint32_t block_levels = max(max(sh[0].levels, sh[1].levels), max(sh[2].levels, sh[3].levels));
    const int my_levels = active ? max(actual_depth, 1) : 0;

    int nb = 0;
    uint32_t draws = 0, expanded = 0, hi = root_max;
    for (uint32_t level = 0; level < block_levels; ++level) {
        const bool fast = level == 0 || level > 3;             // :122, :139 (block-uniform)
        const bool run = active && (int)level < my_levels;     // wave-uniform
        uint32_t total_valid = 0;
        // ---- stage A (as beam_decide): both moves of one axis per lane, valid children compacted in generation order
        if (run) {
            const uint32_t par = lane >> 1;
            const bool vertical = (lane & 1u) != 0u;
            const bool on = (int)par < (level == 0 ? 1 : nb);
            Board P = root;
            uint32_t ra_f = (vertical ? 1u : 0u) | (root_max << 8), ra_r = (vertical ? 3u : 2u) | (root_max << 8);
            bool en_f = on, en_r = on;
            if (level == 0) {
                en_f = on && ((mask >> (vertical ? 1 : 0)) & 1u);
                en_r = on && ((mask >> (vertical ? 3 : 2)) & 1u);
            } else if (on) {
                const uint4 pv = G.board[par];
                P = Board{{pv.x, pv.y, pv.z, pv.w}};
                ra_f = ra_r = G.root[par];
            }
            Board cf, cr;
            move_axis(P, vertical, cf, cr);
            if (!fixed_down) {
                const Board q = rot180(cr);
                cr.w[0] = vertical ? q.w[0] : cr.w[0]; cr.w[1] = vertical ? q.w[1] : cr.w[1];
                cr.w[2] = vertical ? q.w[2] : cr.w[2]; cr.w[3] = vertical ? q.w[3] : cr.w[3];
            }
            const bool vf = en_f && !same(cf, P), vr = en_r && !same(cr, P);
            const unsigned long long bf = __ballot(vf), br = __ballot(vr);
            const uint32_t before = prefix_count(bf) + prefix_count(br);
            const uint32_t mine = (vf ? 1u : 0u) | (vr ? 2u : 0u);
            const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);
            const uint32_t idx_f = vertical ? before - ((other >> 1) & 1u) : before;
            const uint32_t idx_r = vertical ? before + (mine & 1u) : before + (mine & 1u) + (other & 1u);
            total_valid = popc64(bf) + popc64(br);
            if (vf) { G.cboard[idx_f] = make_uint4(cf.w[0], cf.w[1], cf.w[2], cf.w[3]); G.croot[idx_f] = ra_f; }
            if (vr) { G.cboard[idx_r] = make_uint4(cr.w[0], cr.w[1], cr.w[2], cr.w[3]); G.croot[idx_r] = ra_r; }
            expanded += total_valid;
            if (total_valid == 0u) {
                if (level == 0) {                                                   // :126-128 random valid action, prob 0.5
                    uint32_t idx = ((rng_draw(k0, k1, gid, draws) >> 16) * nvalid) >> 16;
                    uint32_t m = mask;
                    while (idx--) m &= m - 1u;
                    result_action = (uint32_t)__builtin_ctz(m); result_prob = 0.5f;
                    nb = -1;                                                        // marks "answer already final"
                }
                active = false;                                                     // :170-171 keep the previous beam
            }
        }
        const bool work = run && total_valid != 0u;
        // ---- stage B, children 0..63: one per lane. A child that has an empty cell consumes the next draw, in generation
        // order (:262-269); only the reference's rot180-DOWN quirk can produce a "changed" board without one
        Scored c1;
        c1.key = 0u; c1.score = -INFINITY; c1.cmax = 0u; c1.board = root;
        const bool live1 = work && lane < total_valid;
        uint32_t consumed1 = 0u;
        if (work) {                                                                   // own LDS writes are visible in order
            const uint4 cv = G.cboard[live1 ? lane : 0u];
            const uint32_t pmax = G.croot[live1 ? lane : 0u] >> 8;
            const Board c = {{cv.x, cv.y, cv.z, cv.w}};
            const uint32_t n_moved = count_empty(c);
            const unsigned long long bc = __ballot(live1 && n_moved != 0u);
            consumed1 = popc64(bc);
            c1 = spawn_and_score(c, n_moved, pmax, rng_draw(k0, k1, gid, draws + prefix_count(bc)), lane, fast, phase);
            if (!live1) { c1.key = 0u; c1.score = -INFINITY; }
            if (!fast) G.score[lane] = c1.score;
        }
        if (lane == 0) { G.total_valid = work ? total_valid : 0u; G.draws = draws + consumed1; }
        __syncthreads();
        // ---- leftovers (children 64..79) of all four games by the duty wave, 16 lanes per game
        if (wv == (level & 3u)) {
            const uint32_t q = lane >> 4, k = lane & 15u, ci = 64u + k;
            SharedGame &Q = sh[q];
            const bool livel = ci < Q.total_valid;
            const uint4 cv = Q.cboard[livel ? ci : 0u];
            const uint32_t pmax = Q.croot[livel ? ci : 0u] >> 8;
            const Board c = {{cv.x, cv.y, cv.z, cv.w}};
            const uint32_t n_moved = count_empty(c);
            const unsigned long long bc = __ballot(livel && n_moved != 0u);
            const uint32_t grp = (uint32_t)(bc >> (16u * q)) & 0xffffu;
            const uint64_t qgid = (uint64_t)Q.gid_lo | ((uint64_t)Q.gid_hi << 32);
            if (__ballot(livel)) {                                                    // wave-uniform: somebody has leftovers
                const Scored o = spawn_and_score(c, n_moved, pmax, rng_draw(k0, k1, qgid, Q.draws + popc(grp & ((1u << k) - 1u))), ci,
                                                 fast, Q.phase);
                if (livel) {
                    Q.cboard[ci] = make_uint4(o.board.w[0], o.board.w[1], o.board.w[2], o.board.w[3]);
                    Q.lmax[k] = o.cmax;
                }
                Q.lkey[k] = livel ? o.key : 0u;
                if (!fast) Q.score[ci] = livel ? o.score : -INFINITY;
            } else {
                Q.lkey[k] = 0u;
                if (!fast) Q.score[ci] = -INFINITY;
            }
            if (k == 0u) Q.lconsumed = popc(grp);
        }
        __syncthreads();
        // ---- stable descending top-`width` (:131-132, :174-175)
        if (work) {
            draws += consumed1 + G.lconsumed;
            const bool live2 = lane < (uint32_t)kSharedLeft && 64u + lane < total_valid;
            const uint32_t key2 = lane < (uint32_t)kSharedLeft ? G.lkey[lane] : 0u;
            uint32_t rank1 = 0xffffu, rank2 = 0xffffu;
            if (fast) {
                // radix select of the k largest keys over the two slots, from the highest bit any key can have set: keys
                // are unique, key = score * 256 + (255 - index) with score < 2 * 2^cc + 512 <= 2^(max(cc, 7) + 2), cc = the
                // largest corner code <= hi = the largest code on any board of this level (tracked: a move raises a
                // board's maximum by at most one)
                const uint32_t lmax2 = live2 ? G.lmax[lane] : 0u;
                if (__ballot(c1.cmax > hi || lmax2 > hi)) hi += 1u;               // wave-uniform
                const int top = (int)min(30u, max(hi, 7u) + 9u);
                unsigned long long A1 = __ballot(live1), A2 = __ballot(live2), S1 = 0ull, S2 = 0ull;
                uint32_t k = min((uint32_t)width, total_valid);
                if (total_valid <= (uint32_t)width) { S1 = A1; S2 = A2; k = 0u; }
                uint32_t s1 = c1.key << (31 - top), s2 = key2 << (31 - top);      // bit `top` -> bit 31
                for (int bit = top; bit >= 0 && k != 0u; --bit) {
                    const unsigned long long M1 = __ballot((int32_t)s1 < 0) & A1, M2 = __ballot((int32_t)s2 < 0) & A2;
                    s1 += s1; s2 += s2;
                    const uint32_t c = popc64(M1) + popc64(M2);
                    if (c >= k) { A1 = M1; A2 = M2; }
                    else { S1 |= M1; S2 |= M2; k -= c; A1 &= ~M1; A2 &= ~M2; }
                    if (popc64(A1) + popc64(A2) == k) { S1 |= A1; S2 |= A2; k = 0u; }
                }
                // rank the <= width selected keys against each other
                const bool sel1 = (S1 >> lane) & 1ull, sel2 = (S2 >> lane) & 1ull;
                const uint32_t p1 = prefix_count(S1), p2 = popc64(S1) + prefix_count(S2);
                if (lane < (uint32_t)(kSharedMaxWidth + 4)) G.skey[lane] = 0u;
                if (sel1) G.skey[p1] = c1.key;
                if (sel2) G.skey[p2] = key2;
                uint32_t r1 = 0u, r2 = 0u;
                const uint4 *sk = reinterpret_cast<const uint4 *>(G.skey);
#pragma unroll
                for (int j = 0; j < kSharedMaxWidth / 4; ++j) {
                    const uint4 kj = sk[j];
                    r1 += (kj.x > c1.key ? 1u : 0u) + (kj.y > c1.key ? 1u : 0u) + (kj.z > c1.key ? 1u : 0u) + (kj.w > c1.key ? 1u : 0u);
                    r2 += (kj.x > key2 ? 1u : 0u) + (kj.y > key2 ? 1u : 0u) + (kj.z > key2 ? 1u : 0u) + (kj.w > key2 ? 1u : 0u);
                }
                if (sel1) rank1 = r1;
                if (sel2) rank2 = r2;
            } else {
                const uint32_t lmax2 = live2 ? G.lmax[lane] : 0u;
                if (__ballot(c1.cmax > hi || lmax2 > hi)) hi += 1u;
                const double sc2 = live2 ? G.score[64u + lane] : -INFINITY;
                uint32_t r1 = 0u, r2 = 0u;
                const uint32_t ci2 = 64u + lane;
                for (uint32_t j = 0; j < total_valid; j += 2) {
                    const double2 sj = *reinterpret_cast<const double2 *>(&G.score[j]);
                    r1 += (sj.x > c1.score || (sj.x == c1.score && j < lane)) ? 1u : 0u;
                    r1 += (sj.y > c1.score || (sj.y == c1.score && j + 1 < lane)) ? 1u : 0u;
                    r2 += (sj.x > sc2 || (sj.x == sc2 && j < ci2)) ? 1u : 0u;
                    r2 += (sj.y > sc2 || (sj.y == sc2 && j + 1 < ci2)) ? 1u : 0u;
                }
                if (live1) rank1 = r1;
                if (live2) rank2 = r2;
            }
            if (rank1 < (uint32_t)width) {
                G.board[rank1] = make_uint4(c1.board.w[0], c1.board.w[1], c1.board.w[2], c1.board.w[3]);
                G.root[rank1] = (G.croot[lane] & 0xffu) | (c1.cmax << 8);
            }
            if (rank2 < (uint32_t)width) {                                       // a leftover made the beam: LDS -> LDS
                G.board[rank2] = G.cboard[64u + lane];
                G.root[rank2] = (G.croot[64u + lane] & 0xffu) | (G.lmax[lane] << 8);
            }
            nb = (int)min(total_valid, (uint32_t)width);
        }
        // the next level's stage A reads only this wave's own beam (ordered in its LDS queue); the barriers above
        // separate the duty wave's accesses from the owners'
    }
    if (have && lane == 0) {
        uint32_t a = result_action; float p = result_prob;
        if (nvalid > 1u && nb >= 0) { a = G.root[0] & 0xffu; p = 1.0f; }            // :178-181
        action_out[g] = (uint8_t)a;
        prob_out[g] = p;
        if (expanded_out) expanded_out[g] = (nvalid > 1u && nb >= 0) ? expanded : 0u;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's evaluation loop (run_evaluation.py:48-69, evaluate_beam_search.py:16-98) fused per game: the
// wavefront that owns a game alternates get_action (above) and Game2048Env.step until the game is over or the move cap
// is reached, with the per-move bookkeeping (milestones, valid / invalid counters) in registers. No launch, no host,
// no other game is involved between two moves; draws are the ones the step-by-step driver uses -- move t of game g
// takes (seed, BEAM, t, g, j) for the search and (seed, STEP, t, g) for the spawn -- so the results are identical.
template <int PASSES>
__global__ __launch_bounds__(64) void play_kernel(uint4 *__restrict__ boards, uint32_t *__restrict__ score,
                                                 int32_t *__restrict__ moves_out, int32_t *__restrict__ valid_out,
                                                 int32_t *__restrict__ invalid_out, int4 *__restrict__ milestone_out,
                                                 unsigned long long *__restrict__ expanded_out, uint8_t *__restrict__ alive_out,
                                                 int width, int depth, uint32_t early_thr, uint32_t mid_thr, int max_moves,
                                                 uint64_t seed, uint64_t id_base, bool fixed_down)
{
    __shared__ BeamShared<PASSES> sh;
    const size_t g = blockIdx.x;
    const uint64_t gid = id_base + g;
    const uint4 rv = boards[g];
    Board b = {{rv.x, rv.y, rv.z, rv.w}};
    uint32_t sc = score[g];
    int32_t ms[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int32_t nvalid = 0, ninvalid = 0, t = 0;
    unsigned long long expanded = 0ull;
    bool alive = true;
    for (; t < max_moves && alive; ++t) {
        const Keys kb = rng_keys(seed, DOM_BEAM, (uint64_t)t), ks = rng_keys(seed, DOM_STEP, (uint64_t)t);
        const Decision d = beam_decide<PASSES>(sh, b, -1, width, depth, early_thr, mid_thr, kb.k0, kb.k1, gid, fixed_down);
        const StepOut o = step_board(b, d.action, rng_draw(ks.k0, ks.k1, gid, 0u));
        b = o.board;
        sc += o.gain;
        expanded += d.expanded;
        const int32_t maxcode = (int32_t)(o.flags >> G2048_FLAG_MAXCODE_SHIFT);
#pragma unroll
        for (int k = 0; k < 8; ++k) if (ms[k] < 0 && maxcode >= 6 + k) ms[k] = t;          // evaluate_beam_search.py:60-64
        if (o.flags & G2048_FLAG_VALID) ++nvalid; else ++ninvalid;
        alive = !(o.flags & G2048_FLAG_DONE);
    }
    if (threadIdx.x == 0) {
        boards[g] = make_uint4(b.w[0], b.w[1], b.w[2], b.w[3]);
        score[g] = sc;
        moves_out[g] = t; valid_out[g] = nvalid; invalid_out[g] = ninvalid;
        milestone_out[2 * g] = make_int4(ms[0], ms[1], ms[2], ms[3]);
        milestone_out[2 * g + 1] = make_int4(ms[4], ms[5], ms[6], ms[7]);
        if (expanded_out) expanded_out[g] = expanded;
        alive_out[g] = alive ? 1 : 0;
    }
}

}  // namespace

extern "C" {

// defined in g2048_kernels.hip; the beam entry point reports through the same thread-local string
const char *g2048_last_error(void);
void g2048_set_last_error_(const char *msg);

static int beam_impl(const void *root_boards, const uint8_t *valid_mask_or_null, uint8_t *action_out,
                     float *prob_out, uint32_t *expanded_out_or_null, int width, int depth,
                     int early_threshold, int mid_threshold, uint64_t seed, uint64_t step_index,
                     uint64_t game_id_base, size_t n_games, uint32_t opts, void *stream, const uint32_t *keyblock)
{
    if (n_games == 0) return G2048_OK;
    if (!root_boards || !action_out || !prob_out) { g2048_set_last_error_("g2048_beam_get_action: null pointer"); return G2048_ERR_ARG; }
    if (reinterpret_cast<uintptr_t>(root_boards) & 15u) { g2048_set_last_error_("g2048_beam_get_action: root array must be 16-byte aligned"); return G2048_ERR_ARG; }
    if (width < 1 || width > kMaxWidth) { g2048_set_last_error_("g2048_beam_get_action: width must be in 1..128"); return G2048_ERR_ARG; }
    if (opts & ~(G2048_BEAM_FIXED_DOWN | G2048_BEAM_ONE_WAVE_PER_GAME)) { g2048_set_last_error_("g2048_beam_get_action: unknown opts"); return G2048_ERR_ARG; }
    if (n_games > 0x7fffffffu) { g2048_set_last_error_("g2048_beam_get_action: too many games for one launch"); return G2048_ERR_ARG; }
    if (early_threshold < 0 || mid_threshold < 0) { g2048_set_last_error_("g2048_beam_get_action: negative threshold"); return G2048_ERR_ARG; }
    const Keys k = rng_keys(seed, DOM_BEAM, step_index);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)n_games), block(64);
    const uint4 *roots = static_cast<const uint4 *>(root_boards);
    const bool fd = (opts & G2048_BEAM_FIXED_DOWN) != 0;
    if (width > 16 && width <= kSharedMaxWidth && !(opts & G2048_BEAM_ONE_WAVE_PER_GAME))
        hipLaunchKernelGGL(beam_shared_kernel, dim3((unsigned)((n_games + kSharedGames - 1) / kSharedGames)), dim3(64 * kSharedGames), 0, s,
                           roots, valid_mask_or_null, action_out, prob_out, expanded_out_or_null, width, depth,
                           (uint32_t)early_threshold, (uint32_t)mid_threshold, k.k0, k.k1, game_id_base, n_games, fd, keyblock);
    else {
#define G2048_LAUNCH_BEAM(P) hipLaunchKernelGGL(beam_kernel<P>, grid, block, 0, s, roots, valid_mask_or_null, action_out, prob_out, \
                           expanded_out_or_null, width, depth, (uint32_t)early_threshold, (uint32_t)mid_threshold, \
                           k.k0, k.k1, game_id_base, fd, keyblock)
        if (width <= 16) G2048_LAUNCH_BEAM(1);
        else if (width <= 32) G2048_LAUNCH_BEAM(2);
        else if (width <= 64) G2048_LAUNCH_BEAM(4);
        else G2048_LAUNCH_BEAM(8);
#undef G2048_LAUNCH_BEAM
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g2048_set_last_error_(hipGetErrorString(e)); return G2048_ERR_HIP; }
    return G2048_OK;
}

int g2048_play_games(void *boards_inout, uint32_t *score_inout, int32_t *moves_out, int32_t *valid_out, int32_t *invalid_out,
                     int32_t *milestone_move_out, unsigned long long *expanded_sum_out_or_null, uint8_t *alive_out,
                     int width, int depth, int early_threshold, int mid_threshold, int max_moves, uint64_t seed,
                     uint64_t game_id_base, size_t n_games, uint32_t opts, void *stream)
{
    if (n_games == 0) return G2048_OK;
    if (!boards_inout || !score_inout || !moves_out || !valid_out || !invalid_out || !milestone_move_out || !alive_out) {
        g2048_set_last_error_("g2048_play_games: null pointer"); return G2048_ERR_ARG;
    }
    if ((reinterpret_cast<uintptr_t>(boards_inout) & 15u) || (reinterpret_cast<uintptr_t>(milestone_move_out) & 15u)) {
        g2048_set_last_error_("g2048_play_games: board / milestone arrays must be 16-byte aligned"); return G2048_ERR_ARG;
    }
    if (width < 1 || width > kMaxWidth || max_moves < 0 || early_threshold < 0 || mid_threshold < 0 || n_games > 0x7fffffffu ||
        (opts & ~G2048_BEAM_FIXED_DOWN)) {
        g2048_set_last_error_("g2048_play_games: bad width / max_moves / thresholds / opts / n_games"); return G2048_ERR_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)n_games), block(64);
    const bool fd = (opts & G2048_BEAM_FIXED_DOWN) != 0;
#define G2048_LAUNCH_PLAY(P) hipLaunchKernelGGL(play_kernel<P>, grid, block, 0, s, static_cast<uint4 *>(boards_inout), score_inout, moves_out, \
                           valid_out, invalid_out, reinterpret_cast<int4 *>(milestone_move_out), expanded_sum_out_or_null, alive_out, \
                           width, depth, (uint32_t)early_threshold, (uint32_t)mid_threshold, max_moves, seed, game_id_base, fd)
    if (width <= 16) G2048_LAUNCH_PLAY(1);
    else if (width <= 32) G2048_LAUNCH_PLAY(2);
    else if (width <= 64) G2048_LAUNCH_PLAY(4);
    else G2048_LAUNCH_PLAY(8);
#undef G2048_LAUNCH_PLAY
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g2048_set_last_error_(hipGetErrorString(e)); return G2048_ERR_HIP; }
    return G2048_OK;
}

int g2048_beam_get_action(const void *root_boards, const uint8_t *valid_mask_or_null, uint8_t *action_out,
                          float *prob_out, uint32_t *expanded_out_or_null, int width, int depth,
                          int early_threshold, int mid_threshold, uint64_t seed, uint64_t step_index,
                          uint64_t game_id_base, size_t n_games, uint32_t opts, void *stream)
{
    return beam_impl(root_boards, valid_mask_or_null, action_out, prob_out, expanded_out_or_null, width, depth, early_threshold,
                     mid_threshold, seed, step_index, game_id_base, n_games, opts, stream, nullptr);
}

int g2048_beam_get_action_dyn(const void *root_boards, const uint8_t *valid_mask_or_null, uint8_t *action_out,
                              float *prob_out, uint32_t *expanded_out_or_null, int width, int depth,
                              int early_threshold, int mid_threshold, const uint32_t *keyblock,
                              uint64_t game_id_base, size_t n_games, uint32_t opts, void *stream)
{
    if (!keyblock) { g2048_set_last_error_("g2048_beam_get_action_dyn: null key block"); return G2048_ERR_ARG; }
    return beam_impl(root_boards, valid_mask_or_null, action_out, prob_out, expanded_out_or_null, width, depth, early_threshold,
                     mid_threshold, 0, 0, game_id_base, n_games, opts, stream, keyblock);
}

}  // extern "C"
