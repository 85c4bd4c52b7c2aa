/*
 * g2048_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See g2048_oracle.h for the rules. Parity status: PINNED by tests/golden/.
 *
 * Style: deliberately literal. Boards are int32 real tile values in a 4x4
 * array, view transforms are spelled out as the reference spells them
 * (transpose / fliplr), rows are compacted and merged with the same scan the
 * reference uses. Nothing here is shared with the HIP kernels, which use a
 * different (packed, SWAR) formulation -- that independence is the point.
 */
#include "g2048_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ RNG -- */
/* DESIGN.md "RNG": keys = splitmix64 chain over (seed, domain, index);      */
/* draw = two 32-bit xorshift-multiply finalizers over (id, ctr).            */
static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void g2048o_rng_keys(uint64_t seed, uint32_t domain, uint64_t index, uint32_t *k0, uint32_t *k1)
{
    uint64_t a = splitmix64(seed ^ ((uint64_t)domain * 0xD1B54A32D192ED03ull));
    uint64_t b = splitmix64(a ^ splitmix64(index + 0x2048204820482048ull));
    *k0 = (uint32_t)b;
    *k1 = (uint32_t)(b >> 32);
}

uint32_t g2048o_rng_draw(uint32_t k0, uint32_t k1, uint64_t id, uint32_t ctr)
{
    uint32_t h = (uint32_t)id ^ k0;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += k1 + (uint32_t)(id >> 32) * 0x9E3779B1u + ctr * 0x85EBCA77u;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}

uint32_t g2048o_draw_index(uint32_t h, uint32_t n) { return ((h >> 16) * n) >> 16; }
int      g2048o_draw_is4(uint32_t h) { return (h & 0xFFFFu) >= 58982u; } /* u < 0.9 -> tile 2 */

/* ------------------------------------------------------ pack / unpack ---- */
void g2048o_pack(const int32_t *tiles, uint8_t *codes, size_t n)
{
    for (size_t i = 0; i < n * 16; ++i) {
        int32_t v = tiles[i];
        uint8_t c = 0;
        while (v > 1) { v >>= 1; ++c; }
        codes[i] = c;
    }
}

void g2048o_unpack(const uint8_t *codes, int32_t *tiles, size_t n)
{
    for (size_t i = 0; i < n * 16; ++i)
        tiles[i] = codes[i] ? (int32_t)1 << codes[i] : 0;
}

/* ------------------------------------------------------ view helpers ----- */
static void transpose4(int32_t g[4][4])
{
    for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j) { int32_t t = g[i][j]; g[i][j] = g[j][i]; g[j][i] = t; }
}

static void fliplr4(int32_t g[4][4])
{
    for (int i = 0; i < 4; ++i) {
        int32_t t = g[i][0]; g[i][0] = g[i][3]; g[i][3] = t;
        t = g[i][1]; g[i][1] = g[i][2]; g[i][2] = t;
    }
}

/* environment/game_2048.py:116-168 (_move_left) and, identically,
 * agents/beam_search_agent.py:213-242: per row drop zeros, scan left to right,
 * equal neighbours merge once into 2v (score += 2v), pad with zeros.        */
static void move_left_rows(int32_t g[4][4], int32_t *gain)
{
    for (int i = 0; i < 4; ++i) {
        int32_t row[4]; int len = 0;
        for (int j = 0; j < 4; ++j) if (g[i][j] != 0) row[len++] = g[i][j];
        if (len == 0) continue;
        int32_t new_row[4] = {0, 0, 0, 0}; int m = 0; int skip_next = 0;
        for (int j = 0; j < len; ++j) {
            if (skip_next) { skip_next = 0; continue; }
            if (j + 1 < len && row[j] == row[j + 1]) {
                int32_t merged = row[j] * 2;
                new_row[m++] = merged;
                *gain += merged;
                skip_next = 1;
            } else {
                new_row[m++] = row[j];
            }
        }
        for (int j = 0; j < 4; ++j) g[i][j] = new_row[j];
    }
}

/* ------------------------------------------------------ environment ------ */
/* environment/game_2048.py:97-114 (_execute_move): UP = T,left,T; RIGHT =
 * fliplr,left,fliplr; DOWN = T,fliplr,left,fliplr,T.                        */
void g2048o_env_move(int32_t b[16], int action, int32_t *score_gain)
{
    int32_t (*g)[4] = (int32_t (*)[4])b;
    int32_t gain = 0;
    if (action == 0) {
        move_left_rows(g, &gain);
    } else if (action == 1) {
        transpose4(g); move_left_rows(g, &gain); transpose4(g);
    } else if (action == 2) {
        fliplr4(g); move_left_rows(g, &gain); fliplr4(g);
    } else if (action == 3) {
        transpose4(g); fliplr4(g); move_left_rows(g, &gain); fliplr4(g); transpose4(g);
    }
    if (score_gain) *score_gain = gain;
}

/* environment/game_2048.py:69-95 (get_valid_moves): try each move, valid iff
 * the board changed. Bit a of the result = action a.                        */
int g2048o_env_valid_mask(const int32_t b[16])
{
    int mask = 0;
    for (int a = 0; a < 4; ++a) {
        int32_t t[16]; memcpy(t, b, sizeof t);
        g2048o_env_move(t, a, NULL);
        if (memcmp(t, b, sizeof t) != 0) mask |= 1 << a;
    }
    return mask;
}

/* environment/game_2048.py:59-67 (add_new_tile) and
 * agents/beam_search_agent.py:260-269 (_add_random_tile): empties enumerated
 * row-major, pick the idx-th, place 2 (p=0.9) or 4. Returns 1 iff a draw was
 * consumed (there was an empty cell).                                       */
int g2048o_spawn(int32_t b[16], uint32_t h)
{
    int pos[16]; int n = 0;
    for (int i = 0; i < 16; ++i) if (b[i] == 0) pos[n++] = i;
    if (n == 0) return 0;
    uint32_t idx = g2048o_draw_index(h, (uint32_t)n);
    b[pos[idx]] = g2048o_draw_is4(h) ? 4 : 2;
    return 1;
}

static int32_t max_tile16(const int32_t b[16])
{
    int32_t m = b[0];
    for (int i = 1; i < 16; ++i) if (b[i] > m) m = b[i];
    return m;
}

static int count_zero16(const int32_t b[16])
{
    int n = 0;
    for (int i = 0; i < 16; ++i) n += (b[i] == 0);
    return n;
}

/* environment/game_2048.py:212-277 (_calculate_reward), same f64 operation
 * order. highest_tile is the env attribute as it stands when the reward is
 * computed, i.e. BEFORE step() updates it (:195 vs :200-203), which is why the
 * milestone branch (:229-241) never fires from step(). It is kept
 * here because it is what the reference executes.                           */
static double env_reward_full(const int32_t prev[16], const int32_t cur[16],
                              int32_t score_diff, int valid, int32_t highest_tile)
{
    double reward = (double)score_diff / 4.0;
    if (highest_tile > max_tile16(prev)) {
        reward += 2.0 * log2((double)highest_tile);
        if (highest_tile >= 256) reward += 50;
        if (highest_tile >= 512) reward += 100;
        if (highest_tile >= 1024) reward += 200;
        if (highest_tile >= 2048) reward += 500;
    }
    if (!valid) reward -= 2.0;
    int empty_before = count_zero16(prev);
    int empty_after = count_zero16(cur);
    reward += (double)(empty_after - empty_before) * 0.5;

    const int32_t (*g)[4] = (const int32_t (*)[4])cur;
    int64_t edge_sum = 0, total = 0;
    for (int j = 0; j < 4; ++j) edge_sum += g[0][j];
    for (int j = 0; j < 4; ++j) edge_sum += g[3][j];
    for (int i = 0; i < 4; ++i) edge_sum += g[i][0];
    for (int i = 0; i < 4; ++i) edge_sum += g[i][3];
    for (int i = 0; i < 16; ++i) total += cur[i];
    reward += ((double)edge_sum / (double)total) * 1.0;   /* 0/0 -> NaN like numpy */

    if (empty_after <= 2) reward -= 2.0;

    for (int i = 0; i < 4; ++i) {
        int row_ordered = 0, col_ordered = 0;
        for (int j = 1; j < 4; ++j) {
            if (g[i][j] > 0 && g[i][j - 1] > 0) row_ordered += (g[i][j] >= g[i][j - 1]);
            if (g[j][i] > 0 && g[j - 1][i] > 0) col_ordered += (g[j][i] >= g[j - 1][i]);
        }
        reward += (double)(row_ordered + col_ordered) * 0.1;
    }
    return reward;
}

double g2048o_env_reward(const int32_t prev[16], const int32_t cur[16], int32_t score_diff, int valid)
{
    return env_reward_full(prev, cur, score_diff, valid, max_tile16(prev));
}

/* environment/game_2048.py:170-210 (step): move -> valid -> spawn iff valid ->
 * reward (post-spawn board, pre-update highest_tile) -> game over (post-spawn)
 * -> highest_tile update.                                                   */
int g2048o_env_step(int32_t b[16], int32_t *score, int action, uint32_t h,
                    double *reward, int *done, int32_t *highest_tile)
{
    int32_t prev[16]; memcpy(prev, b, sizeof prev);
    int32_t prev_score = *score;
    int32_t gain = 0;
    g2048o_env_move(b, action, &gain);
    *score += gain;
    int valid = memcmp(prev, b, sizeof prev) != 0;
    if (valid) g2048o_spawn(b, h);
    int32_t hi = highest_tile ? *highest_tile : max_tile16(prev);
    *reward = env_reward_full(prev, b, *score - prev_score, valid, hi);
    *done = (g2048o_env_valid_mask(b) == 0);                /* :279-288 */
    int32_t cur_hi = max_tile16(b);
    if (highest_tile && cur_hi > *highest_tile) *highest_tile = cur_hi;
    return valid;
}

/* environment/game_2048.py:29-48 (reset): zero board, two spawns.           */
void g2048o_env_reset(int32_t b[16], uint32_t h0, uint32_t h1)
{
    memset(b, 0, 16 * sizeof(int32_t));
    g2048o_spawn(b, h0);
    g2048o_spawn(b, h1);
}

/* environment/game_2048.py:341-387 (simulate_move), restated with its actual behaviour:
 *   - the move is executed on a copy of `state`; nothing is produced when the board does not change;
 *   - `empty` is taken once from the moved board M (row-major); for every empty cell, tiles 2 then 4:
 *       new_state = self.board.copy()   -- self.board is the PREVIOUS iteration's new_state (:371 after :378), so
 *                                          tiles accumulate: earlier cells end up holding 4;
 *       reward    = _calculate_reward(True, state, original_score) evaluated while self.board is still the
 *                   previous iteration's board (M for the first) and self.highest_tile is the env's own attribute,
 *                   so the milestone branch (:229-241) can fire here;
 *       done      = is_game_over() on new_state.
 * Returns the number of successors (0 or 2 * #empty(M)).                                                   */
int g2048o_simulate_move(const int32_t state[16], int action, int32_t highest_tile,
                         int32_t *succ /* [32][16] */, double *reward /* [32] */, uint8_t *done /* [32] */)
{
    int32_t board[16]; memcpy(board, state, sizeof board);
    int32_t gain = 0;
    g2048o_env_move(board, action, &gain);
    if (memcmp(board, state, sizeof board) == 0) return 0;
    int pos[16]; int ne = 0;
    for (int i = 0; i < 16; ++i) if (board[i] == 0) pos[ne++] = i;
    int k = 0;
    for (int i = 0; i < ne; ++i) {
        for (int t = 0; t < 2; ++t) {
            int32_t ns[16]; memcpy(ns, board, sizeof ns);
            ns[pos[i]] = t ? 4 : 2;
            reward[k] = env_reward_full(state, board, gain, 1, highest_tile);
            memcpy(board, ns, sizeof ns);
            done[k] = (uint8_t)(g2048o_env_valid_mask(board) == 0);
            memcpy(succ + 16 * k, ns, sizeof ns);
            ++k;
        }
    }
    return k;
}

/* ------------------------------------------------------ beam agent ------- */
/* agents/beam_search_agent.py:194-258 (_make_move). Pre-transform for DOWN is
 * fliplr(board.T) (:210); post-transform is board.T then fliplr (:252-253),
 * which is NOT the inverse -- the returned DOWN board is rot180 of the true
 * result. Reproduced on purpose.                                */
/* fixed_down != 0 is NOT the reference: it replaces the quirky post-transform of DOWN by the true inverse (the
 * product's G2048_BEAM_FIXED_DOWN option). Everything pinned by the goldens runs with fixed_down == 0.        */
static void agent_move_opt(const int32_t in[16], int action, int32_t out[16], int32_t *score, int *valid, int fixed_down)
{
    if (fixed_down && action == 3) {
        int32_t w[16]; memcpy(w, in, sizeof w);
        int32_t gain = 0;
        g2048o_env_move(w, 3, &gain);
        memcpy(out, w, sizeof w);
        if (score) *score = gain;
        if (valid) *valid = memcmp(in, w, sizeof w) != 0;
        return;
    }
    g2048o_agent_move(in, action, out, score, valid);
}

static int agent_valid_mask_opt(const int32_t b[16], int fixed_down)
{
    int mask = 0;
    for (int a = 0; a < 4; ++a) {
        int32_t t[16]; int v;
        agent_move_opt(b, a, t, NULL, &v, fixed_down);
        if (v) mask |= 1 << a;
    }
    return mask;
}

void g2048o_agent_move(const int32_t in[16], int action, int32_t out[16], int32_t *score, int *valid)
{
    int32_t w[16]; memcpy(w, in, sizeof w);
    int32_t (*g)[4] = (int32_t (*)[4])w;
    int32_t gain = 0;
    if (action == 1) transpose4(g);
    else if (action == 2) fliplr4(g);
    else if (action == 3) { transpose4(g); fliplr4(g); }      /* fliplr(board.T) */
    move_left_rows(g, &gain);
    if (action == 1) transpose4(g);
    else if (action == 2) fliplr4(g);
    else if (action == 3) { transpose4(g); fliplr4(g); }      /* board.T ; fliplr */
    memcpy(out, w, sizeof w);
    if (score) *score = gain;
    if (valid) *valid = memcmp(in, w, sizeof w) != 0;
}

/* agents/beam_search_agent.py:183-192 (_check_valid_moves).                 */
int g2048o_agent_valid_mask(const int32_t b[16])
{
    int mask = 0;
    for (int a = 0; a < 4; ++a) {
        int32_t t[16]; int v;
        g2048o_agent_move(b, a, t, NULL, &v);
        if (v) mask |= 1 << a;
    }
    return mask;
}

/* agents/beam_search_agent.py:271-278: 0 early, 1 mid, 2 late.              */
int g2048o_phase(int32_t max_tile, int32_t early_thr, int32_t mid_thr)
{
    if (max_tile < early_thr) return 0;
    if (max_tile < mid_thr) return 1;
    return 2;
}

/* agents/beam_search_agent.py:280-314 (_fast_evaluate).                     */
double g2048o_fast_eval(const int32_t b[16])
{
    const int32_t (*g)[4] = (const int32_t (*)[4])b;
    double empty_score = (double)count_zero16(b) * 10.0;
    int32_t mx = max_tile16(b);
    double max_score = mx > 0 ? log2((double)mx) * 2.0 : 0.0;
    static const int cr[4] = {0, 0, 3, 3}, cc[4] = {0, 3, 0, 3};
    int64_t corner_score = 0;
    for (int k = 0; k < 4; ++k) {
        int64_t s = (int64_t)g[cr[k]][cc[k]] * 2;
        if (s > 0 && s > corner_score) corner_score = s;
    }
    int merge_score = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j)
            if (g[i][j] == g[i][j + 1] && g[i][j] > 0) merge_score += 1;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            if (g[i][j] == g[i + 1][j] && g[i][j] > 0) merge_score += 1;
    return ((empty_score + max_score) + (double)corner_score) + (double)(merge_score * 2);
}

/* agents/beam_search_agent.py:375-385 (_calculate_corner_bonus): log2 of the largest corner times 2.0, 0 if all four are empty */
double g2048o_corner_bonus(const int32_t b[16])
{
    const int32_t (*g)[4] = (const int32_t (*)[4])b;
    int32_t max_corner = g[0][0];
    if (g[0][3] > max_corner) max_corner = g[0][3];
    if (g[3][0] > max_corner) max_corner = g[3][0];
    if (g[3][3] > max_corner) max_corner = g[3][3];
    return max_corner <= 0 ? 0.0 : log2((double)max_corner) * 2.0;
}

/* agents/beam_search_agent.py:387-403 (_calculate_merge_potential): sum of log2 over equal non-zero neighbours, rows then columns */
double g2048o_merge_potential(const int32_t b[16])
{
    const int32_t (*g)[4] = (const int32_t (*)[4])b;
    double mp = 0.0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j)
            if (g[i][j] > 0 && g[i][j] == g[i][j + 1]) mp += log2((double)g[i][j]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            if (g[i][j] > 0 && g[i][j] == g[i + 1][j]) mp += log2((double)g[i][j]);
    return mp;
}

/* agents/beam_search_agent.py:316-403 (_evaluate_state, _calculate_corner_bonus,
 * _calculate_merge_potential), f64, left-to-right as written.               */
double g2048o_full_eval(const int32_t b[16], int phase)
{
    static const double W[3][4] = { {15.0, 1.0, 2.0, 2.0}, {10.0, 1.5, 2.5, 1.5}, {8.0, 2.0, 3.0, 1.0} };
    static const int snake0[4][4] = { {15, 14, 13, 12}, {8, 9, 10, 11}, {7, 6, 5, 4}, {0, 1, 2, 3} }; /* :37-42 */
    const int32_t (*g)[4] = (const int32_t (*)[4])b;
    const double *w = W[phase];
    int empty_count = count_zero16(b);
    double empty_score = (double)empty_count * w[0];
    if (empty_count <= 2) empty_score -= 10.0;
    int32_t mx = max_tile16(b);
    double max_score = mx > 0 ? log2((double)mx) * w[1] : 0.0;
    if (mx >= 512) max_score *= 1.2;
    if (mx >= 1024) max_score *= 1.5;
    if (mx >= 2048) max_score *= 2.0;
    double corner_bonus = g2048o_corner_bonus(b) * w[2];             /* :358-359 */
    double merge_potential = g2048o_merge_potential(b) * w[3];       /* :362-363 */
    double snake = 0.0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (g[i][j] > 0) snake += log2((double)g[i][j]) * (double)snake0[i][j];
    snake /= 100.0;
    return (((empty_score + max_score) + corner_bonus) + merge_potential) + snake;
}

typedef struct { int32_t board[16]; int root_action; double score; } cand_t;

typedef struct {
    const uint32_t *explicit_draws; size_t n_explicit;
    uint32_t k0, k1; uint64_t id; uint32_t pos;
} draw_src_t;

static uint32_t next_draw(draw_src_t *s)
{
    uint32_t h;
    if (s->explicit_draws) h = s->pos < s->n_explicit ? s->explicit_draws[s->pos] : 0u;
    else h = g2048o_rng_draw(s->k0, s->k1, s->id, s->pos);
    s->pos++;
    return h;
}

/* Python's sorted(key=score, reverse=True) is stable: equal scores keep their
 * generation order (:131, :174). Insertion sort with strict '>' reproduces it. */
static void stable_sort_desc(cand_t *c, int n)
{
    for (int i = 1; i < n; ++i) {
        cand_t t = c[i]; int j = i - 1;
        while (j >= 0 && c[j].score < t.score) { c[j + 1] = c[j]; --j; }
        c[j + 1] = t;
    }
}

#define G2048O_MAX_WIDTH 256

/* agents/beam_search_agent.py:71-181 (get_action).                          */
static int beam_impl(const int32_t root[16], int valid_mask4, int width, int depth, int32_t early_thr, int32_t mid_thr,
                     const uint32_t *draws, size_t n_draws, uint64_t seed, uint64_t step_index, uint64_t game_id,
                     int *action_out, float *prob_out, uint32_t *n_consumed, uint32_t *n_expanded,
                     double *trace_scores, int32_t *trace_counts, int trace_levels, int fixed_down);

int g2048o_beam_get_action(const int32_t root[16], int valid_mask4,
                           int width, int depth, int32_t early_thr, int32_t mid_thr,
                           const uint32_t *draws, size_t n_draws,
                           uint64_t seed, uint64_t step_index, uint64_t game_id,
                           int *action_out, float *prob_out, uint32_t *n_consumed,
                           uint32_t *n_expanded,
                           double *trace_scores, int32_t *trace_counts, int trace_levels)
{
    return beam_impl(root, valid_mask4, width, depth, early_thr, mid_thr, draws, n_draws, seed, step_index, game_id,
                     action_out, prob_out, n_consumed, n_expanded, trace_scores, trace_counts, trace_levels, 0);
}

static int beam_impl(const int32_t root[16], int valid_mask4, int width, int depth, int32_t early_thr, int32_t mid_thr,
                     const uint32_t *draws, size_t n_draws, uint64_t seed, uint64_t step_index, uint64_t game_id,
                     int *action_out, float *prob_out, uint32_t *n_consumed, uint32_t *n_expanded,
                     double *trace_scores, int32_t *trace_counts, int trace_levels, int fixed_down)
{
    if (width < 1 || width > G2048O_MAX_WIDTH) return -1;
    draw_src_t src; memset(&src, 0, sizeof src);
    src.explicit_draws = draws; src.n_explicit = n_draws; src.id = game_id;
    if (!draws) g2048o_rng_keys(seed, G2048O_DOM_BEAM, step_index, &src.k0, &src.k1);
    uint32_t expanded = 0;
    if (trace_counts) for (int l = 0; l < trace_levels; ++l) trace_counts[l] = 0;

    int mask = valid_mask4 < 0 ? agent_valid_mask_opt(root, fixed_down) : (valid_mask4 & 15);   /* :82-84 */
    int nvalid = (mask & 1) + ((mask >> 1) & 1) + ((mask >> 2) & 1) + ((mask >> 3) & 1);
    if (n_consumed) *n_consumed = 0;
    if (n_expanded) *n_expanded = 0;
    if (nvalid == 0) { *action_out = 0; *prob_out = 0.5f; return 0; }                    /* :86-88 */
    if (nvalid == 1) {                                                                   /* :91-93 */
        for (int a = 0; a < 4; ++a) if (mask & (1 << a)) { *action_out = a; break; }
        *prob_out = 1.0f; return 0;
    }
    int phase = g2048o_phase(max_tile16(root), early_thr, mid_thr);                      /* :96-97 */
    int empty_count = count_zero16(root);                                                /* :100-106 */
    int actual_depth;
    if (empty_count <= 4) actual_depth = depth + 5 < 25 ? depth + 5 : 25;
    else if (empty_count >= 10) actual_depth = depth - 5 < 10 ? depth - 5 : 10;
    else actual_depth = depth;

    static _Thread_local cand_t beam[G2048O_MAX_WIDTH], next[4 * G2048O_MAX_WIDTH];
    int nb = 0;
    for (int a = 0; a < 4; ++a) {                                                        /* :112-123 */
        if (!(mask & (1 << a))) continue;
        int32_t nbrd[16]; int v;
        agent_move_opt(root, a, nbrd, NULL, &v, fixed_down);
        if (!v) continue;
        if (count_zero16(nbrd) > 0) g2048o_spawn(nbrd, next_draw(&src));
        ++expanded;
        memcpy(next[nb].board, nbrd, sizeof nbrd);
        next[nb].root_action = a;
        next[nb].score = g2048o_fast_eval(nbrd);
        ++nb;
    }
    if (nb == 0) {                                                                       /* :126-128 */
        int va[4]; int nv = 0;
        for (int a = 0; a < 4; ++a) if (mask & (1 << a)) va[nv++] = a;
        *action_out = va[g2048o_draw_index(next_draw(&src), (uint32_t)nv)];
        *prob_out = 0.5f;
        if (n_consumed) *n_consumed = src.pos;
        return 0;
    }
    stable_sort_desc(next, nb);                                                          /* :131-132 */
    if (nb > width) nb = width;
    memcpy(beam, next, (size_t)nb * sizeof(cand_t));
    if (trace_counts && trace_levels > 0) {
        trace_counts[0] = nb;
        for (int i = 0; i < nb; ++i) trace_scores[i] = beam[i].score;
    }

    for (int d = 1; d < actual_depth; ++d) {                                             /* :135-175 */
        int use_fast_eval = d > 3;
        int nn = 0;
        for (int c = 0; c < nb; ++c) {
            int cmask = agent_valid_mask_opt(beam[c].board, fixed_down);
            for (int a = 0; a < 4; ++a) {
                if (!(cmask & (1 << a))) continue;
                int32_t nbrd[16]; int v;
                agent_move_opt(beam[c].board, a, nbrd, NULL, &v, fixed_down);
                if (!v) continue;
                if (count_zero16(nbrd) > 0) g2048o_spawn(nbrd, next_draw(&src));
                ++expanded;
                memcpy(next[nn].board, nbrd, sizeof nbrd);
                next[nn].root_action = beam[c].root_action;
                next[nn].score = use_fast_eval ? g2048o_fast_eval(nbrd) : g2048o_full_eval(nbrd, phase);
                ++nn;
            }
        }
        if (nn == 0) break;
        stable_sort_desc(next, nn);
        nb = nn > width ? width : nn;
        memcpy(beam, next, (size_t)nb * sizeof(cand_t));
        if (trace_counts && d < trace_levels) {
            trace_counts[d] = nb;
            for (int i = 0; i < nb; ++i) trace_scores[(size_t)d * width + i] = beam[i].score;
        }
    }
    *action_out = beam[0].root_action;                                                   /* :178-181 */
    *prob_out = 1.0f;
    if (n_consumed) *n_consumed = src.pos;
    if (n_expanded) *n_expanded = expanded;
    return 0;
}

/* ------------------------------------------------------ PPO-side --------- */
/* agents/ppo_agent.py:184-195: float32 array, log2 of tiles, / 15.0 in f32. */
void g2048o_normalize_state(const int32_t b[16], float out[16])
{
    float mx = 0.0f;
    for (int i = 0; i < 16; ++i) {
        out[i] = b[i] > 0 ? (float)log2((double)b[i]) : 0.0f;
        if (out[i] > mx) mx = out[i];
    }
    if (mx > 0.0f) for (int i = 0; i < 16; ++i) out[i] = out[i] / 15.0f;
}

/* agents/ppo_agent.py:300-333.                                              */
double g2048o_monotonicity(const int32_t b[16], int row_dir, int col_dir)
{
    const int32_t (*g)[4] = (const int32_t (*)[4])b;
    int score = 0;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 3; ++c)
            if (g[r][c] > 0 && g[r][c + 1] > 0)
                score += row_dir > 0 ? (g[r][c] <= g[r][c + 1]) : (g[r][c] >= g[r][c + 1]);
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 3; ++r)
            if (g[r][c] > 0 && g[r + 1][c] > 0)
                score += col_dir > 0 ? (g[r][c] <= g[r + 1][c]) : (g[r][c] >= g[r + 1][c]);
    return (double)score / 24.0;
}

/* agents/ppo_agent.py:271-298.                                              */
double g2048o_ppo_heuristic(const int32_t b[16])
{
    const int32_t (*g)[4] = (const int32_t (*)[4])b;
    double score = 0.0;
    double m = g2048o_monotonicity(b, 1, 1), t;
    t = g2048o_monotonicity(b, 1, -1);  if (t > m) m = t;
    t = g2048o_monotonicity(b, -1, 1);  if (t > m) m = t;
    t = g2048o_monotonicity(b, -1, -1); if (t > m) m = t;
    score += 2.0 * m;
    int32_t max_corner = g[0][0];
    if (g[0][3] > max_corner) max_corner = g[0][3];
    if (g[3][0] > max_corner) max_corner = g[3][0];
    if (g[3][3] > max_corner) max_corner = g[3][3];
    if (max_corner == max_tile16(b)) score += 1.0;
    int high = 0;
    for (int i = 0; i < 16; ++i) high += (b[i] >= 8);
    if (high > 0) score += -0.1 * (double)high;
    return score;
}

/* agents/ppo_agent.py:253-266 (remember): the two PURE per-transition shaping terms, in the order the
 * reference adds them to the reward (the stateful terms in between -- highest_tile_seen :241-246,
 * regression :249-251 (unreachable), novelty :259-262 -- contribute 0 here):
 *   reward += 0.1 * sum(log2(t) for t in sorted(next_state)[-4:] if t > 0)      :254-256
 *   reward += 0.3 * evaluate_heuristic(next_state)                               :265-266          */
/* ---- the hybrid agent's simulate_move (agents/hybrid.py:578-692; patched onto its own copy of the env, :694-697) ---- */
static void hybrid_move_left(int32_t g[4][4])                                     /* _simulate_move_left, :631-669 */
{
    for (int i = 0; i < 4; ++i) {
        int32_t nz[4]; int n = 0;
        for (int j = 0; j < 4; ++j) if (g[i][j] != 0) nz[n++] = g[i][j];
        if (n == 0) continue;
        int32_t out[4] = {0, 0, 0, 0}; int m = 0, skip = 0;
        for (int j = 0; j < n; ++j) {
            if (skip) { skip = 0; continue; }
            if (j + 1 < n && nz[j] == nz[j + 1]) { out[m++] = nz[j] * 2; skip = 1; }
            else out[m++] = nz[j];
        }
        for (int j = 0; j < 4; ++j) g[i][j] = out[j];
    }
}

static void rot90_ccw(int32_t g[4][4], int times)                                  /* np.rot90(m, k) */
{
    for (int t = 0; t < times; ++t) {
        int32_t r[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r[i][j] = g[j][3 - i];
        memcpy(g, r, sizeof r);
    }
}

static double hybrid_reward(const int32_t nb[16], const int32_t ob[16])            /* _calculate_simulation_reward, :671-692 */
{
    int64_t old_sum = 0, new_sum = 0; int32_t old_max = ob[0], new_max = nb[0]; int zeros = 0;
    for (int i = 0; i < 16; ++i) {
        old_sum += ob[i]; new_sum += nb[i];
        if (ob[i] > old_max) old_max = ob[i];
        if (nb[i] > new_max) new_max = nb[i];
        zeros += nb[i] == 0;
    }
    const int64_t merge_reward = new_sum - old_sum;
    const int64_t max_tile_bonus = new_max > old_max ? new_max : 0;
    const double empty_bonus = (double)zeros * 0.1;
    return (double)(merge_reward + max_tile_bonus) + empty_bonus;
}

/* picks[j]: index into the row-major list of empty cells of the moved board (what random.sample returns, in its order).
 * Returns the number of (successor, reward, done) triples; succ must hold 6 boards. */
int g2048o_hybrid_simulate_move(const int32_t board[16], int action, const int *picks, int32_t *succ, double *reward, uint8_t *done)
{
    int32_t g[4][4];
    memcpy(g, board, sizeof g);
    if (action == 0) hybrid_move_left(g);                                         /* :588-601 */
    else if (action == 1) { rot90_ccw(g, 1); hybrid_move_left(g); rot90_ccw(g, 3); }
    else if (action == 2) { for (int i = 0; i < 4; ++i) { int32_t t = g[i][0]; g[i][0] = g[i][3]; g[i][3] = t; t = g[i][1]; g[i][1] = g[i][2]; g[i][2] = t; }
                            hybrid_move_left(g);
                            for (int i = 0; i < 4; ++i) { int32_t t = g[i][0]; g[i][0] = g[i][3]; g[i][3] = t; t = g[i][1]; g[i][1] = g[i][2]; g[i][2] = t; } }
    else if (action == 3) { rot90_ccw(g, 3); hybrid_move_left(g); rot90_ccw(g, 1); }
    const int32_t *m = &g[0][0];
    if (memcmp(m, board, 16 * sizeof(int32_t)) == 0) {                              /* :604-608 */
        memcpy(succ, m, 16 * sizeof(int32_t)); reward[0] = -1.0; done[0] = 0;
        return 1;
    }
    int empty[16], n = 0;
    for (int i = 0; i < 16; ++i) if (m[i] == 0) empty[n++] = i;                     /* :611 row-major */
    if (n == 0) { memcpy(succ, m, 16 * sizeof(int32_t)); reward[0] = 0.0; done[0] = 1; return 1; }     /* :612-614 */
    const int sample_size = n < 3 ? n : 3;                                          /* :620 */
    int k = 0;
    for (int j = 0; j < sample_size; ++j) {                                         /* :623-633 */
        const int pos = empty[picks[j]];
        for (int four = 0; four < 2; ++four) {
            int32_t *nb = succ + 16 * k;
            memcpy(nb, m, 16 * sizeof(int32_t));
            nb[pos] = four ? 4 : 2;
            reward[k] = hybrid_reward(nb, board) * (four ? 0.1 : 0.9);
            done[k] = 0;
            ++k;
        }
    }
    return k;
}

/* the product's draw -> picks mapping (sampling without replacement, see include/g2048.h): pick j = the idx(h_j, n - j)-th
 * empty cell among those not picked before; returned as indices into the ORIGINAL empty list */
void g2048o_sample_picks(const uint32_t h[3], int n_empty, int picks[3])
{
    int taken[3]; int nt = 0;
    const int k = n_empty < 3 ? n_empty : 3;
    for (int j = 0; j < k; ++j) {
        int r = (int)g2048o_draw_index(h[j], (uint32_t)(n_empty - j));
        /* r-th among the untaken: walk the original order */
        int idx = -1;
        for (int c = 0; c < n_empty; ++c) {
            int used = 0;
            for (int t = 0; t < nt; ++t) used |= taken[t] == c;
            if (used) continue;
            if (r-- == 0) { idx = c; break; }
        }
        picks[j] = idx; taken[nt++] = idx;
    }
}

void g2048o_hybrid_simulate_batch(const uint8_t *boards, const uint8_t *actions, uint8_t *succ, double *reward, uint8_t *done,
                                  uint8_t *count, uint64_t seed, uint64_t step_index, uint64_t id_base, size_t n)
{
    uint32_t k0, k1;
    g2048o_rng_keys(seed, 8u /* SIMULATE */, step_index, &k0, &k1);
    for (size_t i = 0; i < n; ++i) {
        int32_t b[16], m[16], sc;
        g2048o_unpack(boards + 16 * i, b, 1);
        memcpy(m, b, sizeof m);
        g2048o_env_move(m, actions[i] & 3, &sc);
        int ne = 0;
        for (int c = 0; c < 16; ++c) ne += m[c] == 0;
        uint32_t h[3];
        for (uint32_t j = 0; j < 3; ++j) h[j] = g2048o_rng_draw(k0, k1, id_base + i, j);
        int picks[3] = {0, 0, 0};
        if (ne > 0) g2048o_sample_picks(h, ne, picks);
        int32_t s6[6 * 16]; double r6[6]; uint8_t d6[6];
        const int k = g2048o_hybrid_simulate_move(b, actions[i] & 3, picks, s6, r6, d6);
        memset(succ + i * 8 * 16, 0, 8 * 16); 
        for (int q = 0; q < 8; ++q) { reward[i * 8 + q] = 0.0; done[i * 8 + q] = 0; }
        for (int q = 0; q < k; ++q) {
            g2048o_pack(s6 + 16 * q, succ + (i * 8 + q) * 16, 1);
            reward[i * 8 + q] = r6[q]; done[i * 8 + q] = d6[q];
        }
        count[i] = (uint8_t)k;
    }
}

double g2048o_ppo_shaping(const int32_t b[16], double reward_in)
{
    int32_t s[16]; memcpy(s, b, sizeof s);
    for (int i = 1; i < 16; ++i) {                      /* np.sort ascending */
        int32_t t = s[i]; int j = i - 1;
        while (j >= 0 && s[j] > t) { s[j + 1] = s[j]; --j; }
        s[j + 1] = t;
    }
    double sum = 0.0;
    for (int i = 12; i < 16; ++i) if (s[i] > 0) sum += log2((double)s[i]);
    double reward = reward_in;
    reward += 0.1 * sum;
    reward += 0.3 * g2048o_ppo_heuristic(b);
    return reward;
}

/* PPOAgent.remember (agents/ppo_agent.py:234-269), the whole method with its agent-level state: highest_tile_seen (:171,
 * starts at tile 2) and seen_states (:175; here a set of the boards themselves -- the reference stores Python's hash() of
 * the board bytes, see DESIGN.md for that one deviation). Sequential by construction: call i sees the state calls 0..i-1
 * left behind. */
struct g2048o_remember_state {
    int32_t highest_tile_seen;
    size_t cap, count;              /* open addressing over whole boards; cap is a power of two */
    int32_t *keys;                  /* cap x 16 */
    uint8_t *used;
};

g2048o_remember_state *g2048o_remember_new(void)
{
    g2048o_remember_state *st = (g2048o_remember_state *)calloc(1, sizeof *st);
    st->highest_tile_seen = 2;
    st->cap = 1024;
    st->keys = (int32_t *)calloc(st->cap * 16, sizeof(int32_t));
    st->used = (uint8_t *)calloc(st->cap, 1);
    return st;
}

void g2048o_remember_free(g2048o_remember_state *st)
{
    if (!st) return;
    free(st->keys); free(st->used); free(st);
}

int32_t g2048o_remember_highest(const g2048o_remember_state *st) { return st->highest_tile_seen; }
size_t g2048o_remember_seen(const g2048o_remember_state *st) { return st->count; }

static size_t remember_hash(const int32_t b[16])
{
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 16; ++i) { h ^= (uint64_t)(uint32_t)b[i]; h *= 1099511628211ull; }
    return (size_t)(h ^ (h >> 29));
}

/* returns 1 if the board was not in the set (and adds it) */
static int remember_add(g2048o_remember_state *st, const int32_t b[16])
{
    if (2 * (st->count + 1) > st->cap) {
        const size_t ncap = st->cap * 2;
        int32_t *nk = (int32_t *)calloc(ncap * 16, sizeof(int32_t));
        uint8_t *nu = (uint8_t *)calloc(ncap, 1);
        for (size_t i = 0; i < st->cap; ++i) {
            if (!st->used[i]) continue;
            size_t p = remember_hash(st->keys + 16 * i) & (ncap - 1);
            while (nu[p]) p = (p + 1) & (ncap - 1);
            memcpy(nk + 16 * p, st->keys + 16 * i, 16 * sizeof(int32_t)); nu[p] = 1;
        }
        free(st->keys); free(st->used);
        st->keys = nk; st->used = nu; st->cap = ncap;
    }
    size_t p = remember_hash(b) & (st->cap - 1);
    while (st->used[p]) {
        if (memcmp(st->keys + 16 * p, b, 16 * sizeof(int32_t)) == 0) return 0;
        p = (p + 1) & (st->cap - 1);
    }
    memcpy(st->keys + 16 * p, b, 16 * sizeof(int32_t)); st->used[p] = 1; st->count++;
    return 1;
}

static int32_t board_max(const int32_t b[16])
{
    int32_t m = b[0];
    for (int i = 1; i < 16; ++i) if (b[i] > m) m = b[i];
    return m;
}

double g2048o_remember(g2048o_remember_state *st, const int32_t state[16], const int32_t next_state[16], double reward,
                       int *novel_out)
{
    const int32_t current_max_tile = board_max(state), next_max_tile = board_max(next_state);       /* :237-238 */
    if (next_max_tile > st->highest_tile_seen) {                                                    /* :241-246 */
        const double tile_bonus = 5.0 * (log2((double)next_max_tile) - log2((double)st->highest_tile_seen));
        st->highest_tile_seen = next_max_tile;
        reward += tile_bonus;
    }
    if (next_max_tile < current_max_tile) {                                                         /* :249-251 */
        const double regression_penalty = -2.0 * (log2((double)current_max_tile) - log2((double)next_max_tile));
        reward += regression_penalty;
    }
    {                                                                                               /* :254-256 */
        int32_t s[16]; memcpy(s, next_state, sizeof s);
        for (int i = 1; i < 16; ++i) {
            int32_t t = s[i]; int j = i - 1;
            while (j >= 0 && s[j] > t) { s[j + 1] = s[j]; --j; }
            s[j + 1] = t;
        }
        double sum = 0.0;
        for (int i = 12; i < 16; ++i) if (s[i] > 0) sum += log2((double)s[i]);
        reward += 0.1 * sum;
    }
    const int novel = remember_add(st, next_state);                                                 /* :259-262 */
    if (novel) reward += 0.2;
    if (novel_out) *novel_out = novel;
    reward += 0.3 * g2048o_ppo_heuristic(next_state);                                               /* :265-266 */
    return reward;
}

/* n transitions in order over the packed layout */
void g2048o_remember_batch(g2048o_remember_state *st, const uint8_t *state_codes, const uint8_t *next_codes,
                           const double *reward_in, double *reward_out, uint8_t *novel_out, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        int32_t a[16], b[16];
        g2048o_unpack(state_codes + 16 * i, a, 1);
        g2048o_unpack(next_codes + 16 * i, b, 1);
        int nv;
        reward_out[i] = g2048o_remember(st, a, b, reward_in[i], &nv);
        if (novel_out) novel_out[i] = (uint8_t)nv;
    }
}

/* Masked sampling of PPOAgent.get_action (agents/ppo_agent.py:211-221): weights p_a + 1e-10 on the valid actions
 * (what Categorical(logits = log(p + 1e-10) + mask) samples from); the build's sampler is an f32 inverse CDF over
 * one 32-bit draw (DESIGN.md "RNG"), restated here operation by operation.                                   */
int g2048o_sample_action(const float p[4], int mask4, uint32_t h, float *prob)
{
    int m = (mask4 & 15) ? (mask4 & 15) : 15;
    float w[4], c[4];
    for (int a = 0; a < 4; ++a) w[a] = ((m >> a) & 1) ? p[a] + 1e-10f : 0.0f;
    c[0] = w[0]; c[1] = c[0] + w[1]; c[2] = c[1] + w[2]; c[3] = c[2] + w[3];
    float u = (float)(h >> 8) * 5.9604644775390625e-08f;          /* 2^-24 */
    float t = u * c[3];
    int a = 3;
    if (t < c[2]) a = 2;
    if (t < c[1]) a = 1;
    if (t < c[0]) a = 0;
    if (!((m >> a) & 1)) { a = 3; while (!((m >> a) & 1)) --a; }
    *prob = w[a] / c[3];
    return a;
}

void g2048o_sample_batch(const float *probs, const uint8_t *mask_or_null, uint8_t *actions, float *prob,
                         uint64_t seed, uint64_t step_index, uint64_t id_base, size_t n)
{
    uint32_t k0, k1;
    g2048o_rng_keys(seed, 7 /* POLICY */, step_index, &k0, &k1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
        actions[i] = (uint8_t)g2048o_sample_action(probs + 4 * i, mask_or_null ? mask_or_null[i] : 15,
                                                   g2048o_rng_draw(k0, k1, id_base + i, 0), prob + i);
}

/* ------------------------------------------------------ batched forms ---- */
void g2048o_synth_boards(uint8_t *codes, uint64_t seed, uint64_t id_base, size_t n,
                         uint32_t p_empty_u16, uint32_t max_code)
{
    uint32_t k0, k1;
    g2048o_rng_keys(seed, G2048O_DOM_SYNTH_BOARD, 0, &k0, &k1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint8_t *c = codes + i * 16; int any = 0;
        for (uint32_t cell = 0; cell < 16; ++cell) {
            uint32_t h = g2048o_rng_draw(k0, k1, id_base + i, cell);
            if ((h >> 16) < p_empty_u16) c[cell] = 0;
            else { c[cell] = (uint8_t)(1 + (((h & 0xFFFFu) * max_code) >> 16)); any = 1; }
        }
        if (!any) c[0] = 1;
    }
}

void g2048o_synth_actions(uint8_t *actions, uint64_t seed, uint64_t step_index,
                          uint64_t id_base, size_t n)
{
    uint32_t k0, k1;
    g2048o_rng_keys(seed, G2048O_DOM_SYNTH_ACTION, step_index, &k0, &k1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
        actions[i] = (uint8_t)(g2048o_rng_draw(k0, k1, id_base + i, 0) >> 30);
}

static uint8_t max_code16(const uint8_t *c)
{
    uint8_t m = 0;
    for (int i = 0; i < 16; ++i) if (c[i] > m) m = c[i];
    return m;
}

void g2048o_step_batch(const uint8_t *boards_in, const uint8_t *actions, uint8_t *boards_out,
                       uint32_t *score_inout, double *reward_out, uint8_t *flags_out,
                       uint64_t seed, uint64_t step_index, uint64_t id_base, size_t n,
                       uint32_t opts)
{
    uint32_t k0, k1, e0, e1;
    g2048o_rng_keys(seed, G2048O_DOM_STEP, step_index, &k0, &k1);
    g2048o_rng_keys(seed, G2048O_DOM_EPISODE, step_index, &e0, &e1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        int32_t b[16]; double r; int done;
        g2048o_unpack(boards_in + i * 16, b, 1);
        int32_t score = (int32_t)score_inout[i];
        uint32_t h = g2048o_rng_draw(k0, k1, id_base + i, 0);
        int valid = g2048o_env_step(b, &score, (opts & 2u) ? (int)actions[i] : (actions[i] & 3), h, &r, &done, NULL);
        uint8_t out[16];
        g2048o_pack(b, out, 1);
        uint8_t flags = (uint8_t)((done ? 1 : 0) | (valid ? 2 : 0) | (max_code16(out) << 3));
        if ((opts & 1u) && done) {
            g2048o_env_reset(b, g2048o_rng_draw(e0, e1, id_base + i, 0),
                                g2048o_rng_draw(e0, e1, id_base + i, 1));
            g2048o_pack(b, out, 1);
            score = 0;
        }
        memcpy(boards_out + i * 16, out, 16);
        score_inout[i] = (uint32_t)score;
        reward_out[i] = r;
        flags_out[i] = flags;
    }
}

void g2048o_reset_batch(uint8_t *boards_out, uint32_t *score_out, uint64_t seed,
                        uint64_t epoch, uint64_t id_base, size_t n)
{
    uint32_t k0, k1;
    g2048o_rng_keys(seed, G2048O_DOM_RESET, epoch, &k0, &k1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        int32_t b[16];
        g2048o_env_reset(b, g2048o_rng_draw(k0, k1, id_base + i, 0),
                            g2048o_rng_draw(k0, k1, id_base + i, 1));
        g2048o_pack(b, boards_out + i * 16, 1);
        if (score_out) score_out[i] = 0;
    }
}

void g2048o_valid_moves_batch(const uint8_t *boards, uint8_t *mask4, size_t n, int agent_semantics)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        int32_t b[16];
        g2048o_unpack(boards + i * 16, b, 1);
        mask4[i] = (uint8_t)(agent_semantics ? g2048o_agent_valid_mask(b) : g2048o_env_valid_mask(b));
    }
}

/* kind: 0 fast, 1 full (phase[i] in 0..2), 2 ppo heuristic, 3..6 monotonicity
 * (+,+) (+,-) (-,+) (-,-), 7 pure PPO shaping terms (reward_in = 0), 8 pattern, 9 corner bonus, 10 merge potential. */
/* environment/game_2048.py:313-339 (_evaluate_pattern; unused by the reference itself): np.sum(board * snake) / 100.0 and
 * np.sum(board * corner) / 100.0 -- the int products summed in int64, the float products (weights down to 0.25: exact binary
 * fractions, so numpy's pairwise order and this sequential one give the same sum) in float64 -- then max(). */
double g2048o_pattern(const int32_t b[16])
{
    static const int32_t snake[16] = {16, 15, 14, 13, 9, 10, 11, 12, 8, 7, 6, 5, 1, 2, 3, 4};                    /* :319-324 */
    static const double corner[16] = {16, 8, 4, 2, 8, 4, 2, 1, 4, 2, 1, 0.5, 2, 1, 0.5, 0.25};                   /* :327-332 */
    int64_t s = 0;
    double c = 0.0;
    for (int i = 0; i < 16; ++i) { s += (int64_t)b[i] * snake[i]; c += (double)b[i] * corner[i]; }
    const double snake_score = (double)s / 100.0, corner_score = c / 100.0;                                      /* :335-336 */
    return snake_score > corner_score ? snake_score : corner_score;                                              /* :339 */
}

void g2048o_eval_batch(const uint8_t *boards, int kind, const uint8_t *phase, double *out, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        int32_t b[16];
        g2048o_unpack(boards + i * 16, b, 1);
        double v;
        switch (kind) {
        case 0: v = g2048o_fast_eval(b); break;
        case 1: v = g2048o_full_eval(b, phase ? phase[i] : 0); break;
        case 2: v = g2048o_ppo_heuristic(b); break;
        case 3: v = g2048o_monotonicity(b, 1, 1); break;
        case 4: v = g2048o_monotonicity(b, 1, -1); break;
        case 5: v = g2048o_monotonicity(b, -1, 1); break;
        case 6: v = g2048o_monotonicity(b, -1, -1); break;
        case 8: v = g2048o_pattern(b); break;
        case 9: v = g2048o_corner_bonus(b); break;
        case 10: v = g2048o_merge_potential(b); break;
        default: v = g2048o_ppo_shaping(b, 0.0); break;
        }
        out[i] = v;
    }
}

void g2048o_obs_batch(const uint8_t *boards, float *obs, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        int32_t b[16];
        g2048o_unpack(boards + i * 16, b, 1);
        g2048o_normalize_state(b, obs + i * 16);
    }
}

void g2048o_beam_batch_opt(const uint8_t *roots, const uint8_t *mask_or_null, uint8_t *action_out,
                           float *prob_out, uint32_t *expanded_out, int width, int depth,
                           int32_t early_thr, int32_t mid_thr,
                           uint64_t seed, uint64_t step_index, uint64_t game_id_base, size_t n, int fixed_down)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (size_t i = 0; i < n; ++i) {
        int32_t b[16]; int a = 0; float p = 0.0f; uint32_t ne = 0;
        g2048o_unpack(roots + i * 16, b, 1);
        beam_impl(b, mask_or_null ? (int)(mask_or_null[i] & 15) : -1, width, depth, early_thr, mid_thr, NULL, 0, seed,
                  step_index, game_id_base + i, &a, &p, NULL, &ne, NULL, NULL, 0, fixed_down);
        action_out[i] = (uint8_t)a;
        prob_out[i] = p;
        if (expanded_out) expanded_out[i] = ne;
    }
}

void g2048o_beam_batch(const uint8_t *roots, const uint8_t *mask_or_null, uint8_t *action_out,
                       float *prob_out, uint32_t *expanded_out, int width, int depth,
                       int32_t early_thr, int32_t mid_thr,
                       uint64_t seed, uint64_t step_index, uint64_t game_id_base, size_t n)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (size_t i = 0; i < n; ++i) {
        int32_t b[16]; int a = 0; float p = 0.0f; uint32_t ne = 0;
        g2048o_unpack(roots + i * 16, b, 1);
        g2048o_beam_get_action(b, mask_or_null ? (int)(mask_or_null[i] & 15) : -1, width, depth,
                               early_thr, mid_thr, NULL, 0, seed, step_index, game_id_base + i,
                               &a, &p, NULL, &ne, NULL, NULL, 0);
        action_out[i] = (uint8_t)a;
        prob_out[i] = p;
        if (expanded_out) expanded_out[i] = ne;
    }
}

void g2048o_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int g2048o_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
