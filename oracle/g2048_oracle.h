/*
 * g2048_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's 2048 hot path, written from scratch
 * from a reading of the reference's Python:
 *     environment/game_2048.py      (Game2048Env)
 *     agents/beam_search_agent.py   (BeamSearchAgent)
 *     agents/ppo_agent.py           (normalize_state / evaluate_heuristic / monotonicity)
 * Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library. The product (lib g2048 HIP) never links or calls it.
 *
 * Parity status: PINNED. tests/golden/ holds vectors captured by running the
 * reference itself in the build container (tests/golden/gen_golden.py);
 * tests/test_oracle_golden.py checks every function below against them.
 *
 * Boards here are int32[16] of REAL tile values, row-major (exactly the
 * reference's np.int32[4,4]); the *_batch functions take the product's packed
 * layout (16 x uint8 log2 codes per board) and convert at the edge.
 */
#ifndef G2048_ORACLE_H
#define G2048_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- counter RNG (spec in DESIGN.md "RNG"; restated, not shared, with csrc/) ---- */
enum { G2048O_DOM_STEP = 1, G2048O_DOM_RESET = 2, G2048O_DOM_BEAM = 3,
       G2048O_DOM_SYNTH_BOARD = 4, G2048O_DOM_SYNTH_ACTION = 5, G2048O_DOM_EPISODE = 6,
       G2048O_DOM_POLICY = 7, G2048O_DOM_SIMULATE = 8 };
void     g2048o_rng_keys(uint64_t seed, uint32_t domain, uint64_t index, uint32_t *k0, uint32_t *k1);
uint32_t g2048o_rng_draw(uint32_t k0, uint32_t k1, uint64_t id, uint32_t ctr);
/* spawn decision from one 32-bit draw: idx in [0,n), is4 */
uint32_t g2048o_draw_index(uint32_t h, uint32_t n);
int      g2048o_draw_is4(uint32_t h);

/* ---- packed <-> tiles ---- */
void g2048o_pack(const int32_t *tiles, uint8_t *codes, size_t n);     /* n boards */
void g2048o_unpack(const uint8_t *codes, int32_t *tiles, size_t n);

/* ---- environment (environment/game_2048.py) ---- */
void   g2048o_env_move(int32_t b[16], int action, int32_t *score_gain);          /* :97-168 */
int    g2048o_env_valid_mask(const int32_t b[16]);                               /* :69-95  */
int    g2048o_spawn(int32_t b[16], uint32_t h);                                  /* :59-67 / agent :260-269 */
double g2048o_env_reward(const int32_t prev[16], const int32_t cur[16],
                         int32_t score_diff, int valid);                          /* :212-277 */
/* one full step with an explicit 32-bit draw h (:170-210). Returns valid. */
int    g2048o_env_step(int32_t b[16], int32_t *score, int action, uint32_t h,
                       double *reward, int *done, int32_t *highest_tile);
void   g2048o_env_reset(int32_t b[16], uint32_t h0, uint32_t h1);               /* :29-48 */
int    g2048o_simulate_move(const int32_t state[16], int action, int32_t highest_tile,
                            int32_t *succ, double *reward, uint8_t *done);          /* :341-387 */

/* the hybrid agent's sampled simulate_move (agents/hybrid.py:578-692); picks = what random.sample returned, as indices into
 * the row-major empty-cell list of the moved board; succ holds up to 6 boards. Returns the number of triples. */
int    g2048o_hybrid_simulate_move(const int32_t board[16], int action, const int *picks, int32_t *succ, double *reward,
                                   uint8_t *done);
void   g2048o_sample_picks(const uint32_t h[3], int n_empty, int picks[3]);          /* the product's draws -> picks */
void   g2048o_hybrid_simulate_batch(const uint8_t *boards, const uint8_t *actions, uint8_t *succ, double *reward, uint8_t *done,
                                    uint8_t *count, uint64_t seed, uint64_t step_index, uint64_t id_base, size_t n);

/* ---- beam agent (agents/beam_search_agent.py) ---- */
void   g2048o_agent_move(const int32_t in[16], int action, int32_t out[16],
                         int32_t *score, int *valid);                             /* :194-258 (Q1 kept) */
int    g2048o_agent_valid_mask(const int32_t b[16]);                             /* :183-192 */
int    g2048o_phase(int32_t max_tile, int32_t early_thr, int32_t mid_thr);       /* :271-278 */
double g2048o_fast_eval(const int32_t b[16]);                                    /* :280-314 */
double g2048o_full_eval(const int32_t b[16], int phase);                         /* :316-373 */
double g2048o_pattern(const int32_t b[16]);                                      /* environment/game_2048.py:313-339 */
/* get_action (:71-181). draws: explicit stream (may be NULL -> hashed from
 * (seed, step_index, game_id, counter)). valid_mask4 < 0 means "None".
 * trace_scores[(depth)*width] / trace_counts[depth] optional (NULL ok).     */
int    g2048o_beam_get_action(const int32_t root[16], int valid_mask4,
                              int width, int depth, int32_t early_thr, int32_t mid_thr,
                              const uint32_t *draws, size_t n_draws,
                              uint64_t seed, uint64_t step_index, uint64_t game_id,
                              int *action_out, float *prob_out, uint32_t *n_consumed,
                              uint32_t *n_expanded,
                              double *trace_scores, int32_t *trace_counts, int trace_levels);

/* ---- PPO-side per-board functions (agents/ppo_agent.py) ---- */
void   g2048o_normalize_state(const int32_t b[16], float out[16]);               /* :184-195 */
double g2048o_monotonicity(const int32_t b[16], int row_dir, int col_dir);       /* :300-333 */
double g2048o_ppo_heuristic(const int32_t b[16]);                                /* :271-298 */
double g2048o_ppo_shaping(const int32_t b[16], double reward_in);                /* :253-266, pure terms */

/* PPOAgent.remember (:234-269) complete, with the agent's running state (highest_tile_seen :171, seen_states :175);
 * returns the reward remember() stores; sequential -- call order is the semantics */
typedef struct g2048o_remember_state g2048o_remember_state;
g2048o_remember_state *g2048o_remember_new(void);
void    g2048o_remember_free(g2048o_remember_state *st);
int32_t g2048o_remember_highest(const g2048o_remember_state *st);
size_t  g2048o_remember_seen(const g2048o_remember_state *st);
double  g2048o_remember(g2048o_remember_state *st, const int32_t state[16], const int32_t next_state[16], double reward,
                        int *novel_out);
void    g2048o_remember_batch(g2048o_remember_state *st, const uint8_t *state_codes, const uint8_t *next_codes,
                              const double *reward_in, double *reward_out, uint8_t *novel_out, size_t n);

/* ---- batched forms over the packed layout (full-size checks, cpu_baseline) ---- */
void g2048o_synth_boards(uint8_t *codes, uint64_t seed, uint64_t id_base, size_t n,
                         uint32_t p_empty_u16, uint32_t max_code);
void g2048o_synth_actions(uint8_t *actions, uint64_t seed, uint64_t step_index,
                          uint64_t id_base, size_t n);
/* flags: bit0 done, bit1 valid, bits 3..7 = max log2 code after the step.
 * opts bit0: auto-reset finished boards (new episode drawn in DOM_EPISODE); bit1: action bytes above 3 move
 * nothing, as Game2048Env._execute_move (:97-114) treats them (default: the low two bits are the action).      */
void g2048o_step_batch(const uint8_t *boards_in, const uint8_t *actions, uint8_t *boards_out,
                       uint32_t *score_inout, double *reward_out, uint8_t *flags_out,
                       uint64_t seed, uint64_t step_index, uint64_t id_base, size_t n,
                       uint32_t opts);
void g2048o_reset_batch(uint8_t *boards_out, uint32_t *score_out, uint64_t seed,
                        uint64_t epoch, uint64_t id_base, size_t n);
void g2048o_valid_moves_batch(const uint8_t *boards, uint8_t *mask4, size_t n, int agent_semantics);
void g2048o_eval_batch(const uint8_t *boards, int kind, const uint8_t *phase, double *out, size_t n);
void g2048o_obs_batch(const uint8_t *boards, float *obs, size_t n);
void g2048o_beam_batch(const uint8_t *roots, const uint8_t *mask_or_null, uint8_t *action_out,
                       float *prob_out, uint32_t *expanded_out, int width, int depth,
                       int32_t early_thr, int32_t mid_thr,
                       uint64_t seed, uint64_t step_index, uint64_t game_id_base, size_t n);
/* fixed_down != 0: the product's non-parity option (true DOWN instead of the reference's rot180 quirk) */
void g2048o_beam_batch_opt(const uint8_t *roots, const uint8_t *mask_or_null, uint8_t *action_out,
                           float *prob_out, uint32_t *expanded_out, int width, int depth,
                           int32_t early_thr, int32_t mid_thr,
                           uint64_t seed, uint64_t step_index, uint64_t game_id_base, size_t n, int fixed_down);
int  g2048o_sample_action(const float p[4], int mask4, uint32_t h, float *prob);     /* ppo_agent.py:211-221 */
void g2048o_sample_batch(const float *probs, const uint8_t *mask_or_null, uint8_t *actions, float *prob,
                         uint64_t seed, uint64_t step_index, uint64_t id_base, size_t n);
int  g2048o_num_threads(void);
void g2048o_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
