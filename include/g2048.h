/*
 * g2048.h -- C-ABI of the MI355X (gfx950) batched 2048 rollout / beam-search engine.
 *
 * The reference (vivek-tiwari-vt/2048-Using-Reinforcement-Learning) has NO FFI,
 * plugin or operator interface: its boundary is two Python classes. This header
 * is therefore the boundary the build defines for that hot path; each entry
 * point names the reference method it replaces (file:line, relative to the
 * reference root). INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (e.g. a
 *     torch.Tensor's data_ptr()); scratch is caller-provided too (the *_ws entry
 *     points with their *_workspace queries). ONE entry point allocates:
 *     g2048_play_games, the convenience form of g2048_play_games_ws, takes its
 *     helper workspace from the stream-ordered pool (hipMallocAsync / hipFreeAsync
 *     on `stream`). Nothing ever synchronises: work is enqueued on `stream` (a
 *     hipStream_t passed as void*, NULL = the default stream) and is stream-ordered;
 *   - a board is 16 bytes: 16 x uint8 log2 codes, row-major (0 = empty,
 *     1 = tile 2, ... 17 = tile 131072). Board arrays must be 16-byte aligned;
 *   - return value: 0 = G2048_OK, negative = error (g2048_last_error() gives
 *     the text for the calling thread). Nothing throws, nothing prints;
 *   - randomness is a counter RNG keyed by (seed, domain, step/epoch index,
 *     GLOBAL board id, counter): results do not depend on launch geometry or
 *     on how boards are sharded over GPUs (DESIGN.md "RNG");
 *   - thread-safe: no global mutable state apart from the thread-local error
 *     string; no environment variable is read (round 2's G2048_PLAY_TUNE hook is
 *     gone: g2048_play_games_tuned takes the same numbers as an argument);
 *   - launch geometry that depends on the chip's size (how many helper wavefronts a
 *     launch may carry, how a beam batch is dealt to the SIMDs) is derived per call
 *     from the compute-unit count of the current device (g2048_launch_plan shows
 *     the arithmetic).
 * There is no CPU implementation behind this ABI: without a HIP device every
 * compute entry point fails with G2048_ERR_HIP.
 */
#ifndef G2048_H
#define G2048_H
#include <stddef.h>
#include <stdint.h>
/* Every entry point below is exported (the library is built with -fvisibility=hidden: nothing else is). */
#ifndef G2048_API
#define G2048_API __attribute__((visibility("default")))
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define G2048_ABI_VERSION 5        /* 5: round 5, second half (ops MOVE / SPAWN / MOVE_AGENT of g2048_env_step; g2048_eval kinds CORNER_BONUS and
                                      MERGE_POTENTIAL; no entry point added or removed)
                                      4: round 5 (export table = this header + g2048_testing.h exactly: test / measurement hooks moved there,
                                      internal symbols hidden; g2048_replay_games also clamps a game's length to actions_stride)
                                      3: round 4 (actions_out of the g2048_play_games family, g2048_replay_games, g2048_env_step,
                                      g2048_minibatch_gather, g2048_build_flags; the A/B variants of g2048_step / g2048_sort_selftest gone)
                                      2: round 3 (g2048_step_many, g2048_play_games_tuned, g2048_launch_plan, g2048_device_plan,
                                      g2048_beam_get_action_hist; no env hook) */

enum {
    G2048_OK = 0,
    G2048_ERR_ARG = -1,        /* bad argument (null pointer, misaligned board array, bad enum) */
    G2048_ERR_HIP = -2         /* HIP runtime error (no device, launch failure) */
};

/* flags_out byte of g2048_step */
#define G2048_FLAG_DONE        0x01u   /* game over after this step (game_2048.py:198, :279-288) */
#define G2048_FLAG_VALID       0x02u   /* the move changed the board (game_2048.py:188)          */
#define G2048_FLAG_MAXCODE_SHIFT 3     /* bits 3..7: max log2 code after the step -> info["highest_tile"] */

/* opts of g2048_step */
#define G2048_STEP_REWARD_F64  0x01u   /* reward_out is double[n] (parity mode); default float[n] = (float)f64 reward */
#define G2048_STEP_AUTO_RESET  0x02u   /* finished boards are replaced by a fresh episode (score 0); flags keep DONE */
#define G2048_STEP_RANDOM_ACTIONS 0x04u /* random playouts: `actions` is ignored (may be NULL); board i moves in direction
                                          draw(seed, SYNTH_ACTION, step_index, id) >> 30, exactly what g2048_synth_actions
                                          writes for the same (seed, step_index, id) */

#define G2048_STEP_NOOP_ACTIONS 0x08u  /* an action byte above 3 moves nothing -- an invalid move -- exactly as the reference's
                                          _execute_move (game_2048.py:97-114) treats values other than 0..3; without this
                                          flag only the low two bits of the byte are looked at */

/* tuning only (results identical): bits 8..9 pick the boards-per-lane variant: 0 = library default (one board per lane, two
 * from 4 Mi boards per launch on), 1 = one, 2 = two */
#define G2048_STEP_TUNE_SHIFT  8

/* opts of g2048_valid_moves */
#define G2048_VALID_ENV        0x00u   /* Game2048Env.get_valid_moves semantics (game_2048.py:69-95) */
#define G2048_VALID_AGENT      0x01u   /* BeamSearchAgent._check_valid_moves semantics, incl. its DOWN quirk
                                          (beam_search_agent.py:183-192, :209-210 vs :251-253) */

/* kind of g2048_eval */
enum {
    G2048_EVAL_FAST = 0,       /* BeamSearchAgent._fast_evaluate   beam_search_agent.py:280-314 */
    G2048_EVAL_FULL = 1,       /* BeamSearchAgent._evaluate_state  beam_search_agent.py:316-403 (phase per board) */
    G2048_EVAL_PPO_HEURISTIC = 2, /* PPOAgent.evaluate_heuristic   ppo_agent.py:271-298 */
    G2048_EVAL_MONO_PP = 3,    /* PPOAgent.monotonicity(board, +1, +1)  ppo_agent.py:300-333 */
    G2048_EVAL_MONO_PM = 4,    /*                        (+1, -1) */
    G2048_EVAL_MONO_MP = 5,    /*                        (-1, +1) */
    G2048_EVAL_MONO_MM = 6,    /*                        (-1, -1) */
    G2048_EVAL_PPO_SHAPING = 7,/* the pure per-transition terms of PPOAgent.remember, ppo_agent.py:253-266:
                                  0.1 * sum(log2(top-4 tiles)) + 0.3 * evaluate_heuristic (stateful terms excluded) */
    G2048_EVAL_PATTERN = 8,    /* Game2048Env._evaluate_pattern  game_2048.py:313-339 (snake / corner weights on tile values) */
    G2048_EVAL_CORNER_BONUS = 9,     /* BeamSearchAgent._calculate_corner_bonus     beam_search_agent.py:375-385 (unweighted) */
    G2048_EVAL_MERGE_POTENTIAL = 10  /* BeamSearchAgent._calculate_merge_potential  beam_search_agent.py:387-403 (unweighted) */
};

/* opts of g2048_beam_get_action */
#define G2048_BEAM_FIXED_DOWN  0x01u   /* use the true DOWN move instead of the reference's rot180 quirk (not parity) */

#define G2048_BEAM_RANK_BY_COUNTING 0x04u /* rank every level by the counting loop instead of the sorting network (same
                                              decisions; an A/B switch for tests and measurements; also g2048_play_games) */
/* further opts of g2048_play_games */
#define G2048_PLAY_ONE_PHASE   0x02u   /* every game on its one wavefront only, no speculative helper wavefronts; the games
                                          are the same either way -- an A/B switch for tests and measurements */

#define G2048_BEAM_MAX_WIDTH   128

G2048_API const char *g2048_last_error(void);
G2048_API int g2048_abi_version(void);
/* 0 for the product library. Non-zero: a measurement build (csrc/g2048_instrument.h: bit 0 beam timeline, bit 1 evaluation
 * timeline, bit 2 step timeline) that overwrites real outputs with clock ticks -- never use its results. */
G2048_API unsigned g2048_build_flags(void);
/* number of visible HIP devices (0 on a CPU-only host); never fails */
G2048_API int g2048_device_count(void);

/* Game2048Env.step for n boards (environment/game_2048.py:170-210): move, validity, spawn iff valid,
 * shaped reward (:212-277), done (:279-288). boards_out may alias boards_in. Draw for board i:
 * (seed, STEP, step_index, board_id_base + i). */
G2048_API int g2048_step(const void *boards_in, const uint8_t *actions, void *boards_out,
               uint32_t *score_inout, void *reward_out, uint8_t *flags_out,
               uint64_t seed, uint64_t step_index, uint64_t board_id_base, size_t n,
               uint32_t opts, void *stream);

/* `steps` consecutive Game2048Env.step calls (environment/game_2048.py:170-210) of every board in ONE launch: the board and
 * its score stay in registers between the steps instead of making a 46-byte round trip through memory per step. Step t
 * (0 <= t < steps) is bit-for-bit g2048_step(step_index = step_index0 + t) with the same opts. The actions are either the
 * in-kernel uniform draws (G2048_STEP_RANDOM_ACTIONS: random playouts, SURVEY 8d C2 "rollout" variant; actions_stream_or_null
 * is ignored and may be NULL) or explicit, step-major: actions_stream_or_null[t * n + i] (low two bits, as in g2048_step) --
 * a recorded move sequence such as the reference's checkpoints/ move-set files (BeamSearchAgent_best_moveset_tile_N.txt), or an open-loop plan; a policy that needs the
 * state of step t to choose action t uses g2048_step / g2048_rollout_step. G2048_STEP_AUTO_RESET and G2048_STEP_REWARD_F64
 * are optional. boards_out may alias boards_in.
 * Outputs: the boards and scores after the last step, flags_last_out[i] = the flags byte of the last step; optional per-step
 * streams, step-major: reward_stream_out_or_null[t * n + i] (float, or double with G2048_STEP_REWARD_F64) and
 * flags_stream_out_or_null[t * n + i]; episodes_out_or_null[i] = episodes board i finished (auto-resets taken). */
G2048_API int g2048_step_many(const void *boards_in, const uint8_t *actions_stream_or_null, void *boards_out, uint32_t *score_inout,
                    void *reward_stream_out_or_null, uint8_t *flags_stream_out_or_null, uint8_t *flags_last_out,
                    uint32_t *episodes_out_or_null, uint64_t seed, uint64_t step_index0, uint32_t steps,
                    uint64_t board_id_base, size_t n, uint32_t opts, void *stream);

/* Game2048Env.reset for n boards (environment/game_2048.py:29-48). score_out may be NULL. */
G2048_API int g2048_reset(void *boards_out, uint32_t *score_out, uint64_t seed, uint64_t epoch,
                uint64_t board_id_base, size_t n, void *stream);

/* get_valid_moves / _check_valid_moves: mask4_out[i] bit a = action a valid (0 LEFT 1 UP 2 RIGHT 3 DOWN). */
G2048_API int g2048_valid_moves(const void *boards, uint8_t *mask4_out, size_t n, uint32_t opts, void *stream);

/* board heuristics, f64 out. phase_or_null: per-board 0 early / 1 mid / 2 late for G2048_EVAL_FULL
 * (NULL = derive from the board's own max tile with thresholds 512 / 1024, beam_search_agent.py:271-278). */
G2048_API int g2048_eval(const void *boards, int kind, const uint8_t *phase_or_null, double *out, size_t n, void *stream);

/* PPOAgent.normalize_state (agents/ppo_agent.py:184-195): obs_out[i*16+j] = float32(code)/float32(15). */
G2048_API int g2048_obs_f32(const void *boards, float *obs_out, size_t n, void *stream);

/* the same observation in 16 bits: obs_out is n*16 IEEE half (bf16 = 0) or bfloat16 (bf16 = 1) values, each the f32
 * quotient above rounded to nearest even -- for policies that run in reduced precision (32 B per board instead of 64) */
G2048_API int g2048_obs_16(const void *boards, void *obs_out, int bf16, size_t n, void *stream);

/* BeamSearchAgent.get_action for n_games roots (agents/beam_search_agent.py:71-181).
 * valid_mask_or_null: caller-supplied masks (the `valid_moves` argument), NULL = None.
 * expanded_out_or_null: children generated per game (calls of _add_random_tile).
 * Draw j of game g: (seed, BEAM, step_index, game_id_base + g, j) in the reference's generation order. */
G2048_API int g2048_beam_get_action(const void *root_boards, const uint8_t *valid_mask_or_null,
                          uint8_t *action_out, float *prob_out, uint32_t *expanded_out_or_null,
                          int width, int depth, int early_threshold, int mid_threshold,
                          uint64_t seed, uint64_t step_index, uint64_t game_id_base, size_t n_games,
                          uint32_t opts, void *stream);
/* The same with g2048_beam_workspace_bytes(n_games) bytes of caller-provided device scratch: from 4096 games per
 * call on, the blocks then take the games in a depth-balanced order (deep and shallow searches mixed on every SIMD) -- same
 * results, a shorter launch. workspace NULL, or a batch for which the query returns 0: exactly g2048_beam_get_action. */
G2048_API size_t g2048_beam_workspace_bytes(size_t n_games);
G2048_API int g2048_beam_get_action_ws(const void *root_boards, const uint8_t *valid_mask_or_null, uint8_t *action_out,
                             float *prob_out, uint32_t *expanded_out_or_null, int width, int depth,
                             int early_threshold, int mid_threshold, uint64_t seed, uint64_t step_index,
                             uint64_t game_id_base, size_t n_games, uint32_t opts, void *workspace,
                             size_t workspace_bytes, void *stream);
/* The same order without its own launch, for callers that search batch after batch (an evaluation loop): every block of a
 * call files its game into per-class lists in `history`, and the NEXT call deals the games from them -- the balanced order of
 * the previous batch's roots (any order gives the same results; this one is balanced as far as a game keeps its depth class
 * from one call to the next). history: g2048_beam_history_bytes(n_games) bytes of device memory, ZERO-FILLED before its first
 * use and again whenever a call is not the direct successor (call_index + 1, same n_games, same buffer, stream-ordered after
 * it) of the last call that used it; call_index counts 1, 2, 3, ... . A call that finds no usable history takes the games in
 * caller order. history NULL, or a batch for which the query returns 0: exactly g2048_beam_get_action. One buffer serves one
 * stream of calls; it must not be shared by calls that may run concurrently. */
G2048_API size_t g2048_beam_history_bytes(size_t n_games);
G2048_API int g2048_beam_get_action_hist(const void *root_boards, const uint8_t *valid_mask_or_null, uint8_t *action_out,
                               float *prob_out, uint32_t *expanded_out_or_null, int width, int depth,
                               int early_threshold, int mid_threshold, uint64_t seed, uint64_t step_index,
                               uint64_t game_id_base, size_t n_games, uint32_t opts, void *history, size_t history_bytes,
                               uint32_t call_index, void *stream);

/* Per-move bookkeeping of the reference's evaluation loops (evaluate_beam_search.py:42-64, run_evaluation.py:56-69)
 * for n games after a g2048_step: for games still alive, milestone_move_inout[i][k] (k = 0..7 for tiles 64..8192,
 * -1 = not reached yet) records move_index the first time the max tile reaches it, the valid / invalid / total move
 * counters advance, expanded_or_null[i] is added to expanded_sum, and the game leaves `alive` when its DONE flag is
 * set. Games not alive are untouched. */
G2048_API int g2048_track_episodes(const uint8_t *flags, const uint32_t *expanded_or_null, uint8_t *alive_inout, int32_t *moves_inout,
                         int32_t *valid_inout, int32_t *invalid_inout, int32_t *milestone_move_inout,
                         unsigned long long *expanded_sum_inout_or_null, int32_t move_index, size_t n, void *stream);

/* The masked sampling of PPOAgent.get_action (agents/ppo_agent.py:211-221) for n envs: probs is float32 [n][4]
 * (the actor's softmax output), mask4 as g2048_valid_moves writes it (NULL = all valid). The action is drawn from
 * weights p_a + 1e-10 over the valid actions (what Categorical(logits = log(p + 1e-10) + mask) samples) by inverse
 * CDF with draw (seed, POLICY, step_index, env_id_base + i); prob_out[i] = its probability (take the log for
 * the reference's `action_prob`). */
G2048_API int g2048_sample_actions(const float *probs, const uint8_t *mask4_or_null, uint8_t *actions_out, float *prob_out,
                         uint64_t seed, uint64_t step_index, uint64_t env_id_base, size_t n, void *stream);

/* Game2048Env.simulate_move (environment/game_2048.py:341-387) for n (state, action) pairs: every successor the
 * reference lists -- 2 per empty cell of the moved board, at most 30 -- in its order and with its behaviour (each
 * successor is built on top of the previous one; its reward is computed on the previous successor's board and
 * includes the milestone bonus against highest_code, the env's highest_tile attribute as a log2 code; NULL = the
 * state's own max, as inside an episode). Outputs are 32 slots per state: succ_boards_out n*32 boards,
 * reward_out n*32 f64, done_out n*32 bytes; count_out[i] = number of valid slots (0 when the move is invalid). */
G2048_API int g2048_simulate_move(const void *boards, const uint8_t *actions, const uint8_t *highest_code_or_null,
                        void *succ_boards_out, double *reward_out, uint8_t *done_out, uint8_t *count_out,
                        size_t n, void *stream);

/* The hybrid agent's simulate_move (agents/hybrid.py:578-629, the function it monkey-patches onto its own copy of the
 * env, :694-697) for n (state, action) pairs: the move, then up to three distinct empty cells of the moved board
 * (random.sample), each as a 2-successor and a 4-successor whose reward (_calculate_simulation_reward, :671-692) is
 * weighted by 0.9 / 0.1. Outputs are 8 slots per state: succ_boards_out n*8 boards, reward_out n*8 f64, done_out n*8
 * bytes; count_out[i] = valid slots: 1 for a move that changes nothing (the board itself, reward -1.0), else 2 * min(3,
 * empty cells), successor 2j / 2j+1 = pick j with a 2 / a 4. Pick j of state i is the idx(h_j, n_empty - j)-th empty
 * cell (row-major) not picked before, h_j = draw (seed, SIMULATE, step_index, state_id_base + i, j). */
G2048_API int g2048_simulate_move_sampled(const void *boards, const uint8_t *actions, void *succ_boards_out, double *reward_out,
                                uint8_t *done_out, uint8_t *count_out, uint64_t seed, uint64_t step_index,
                                uint64_t state_id_base, size_t n, void *stream);

/* The reference's evaluation loop (run_evaluation.py:48-69, evaluate_beam_search.py:16-98) fused per game: every game
 * (one wavefront) alternates BeamSearchAgent.get_action (no caller mask) and Game2048Env.step from boards_inout /
 * score_inout until it is over or max_moves is reached, entirely on the device. Move t of game g uses the draws of
 * g2048_beam_get_action(step_index = t, game id g) and g2048_step(step_index = t, board id g), so the outcome equals
 * the step-by-step loop. Unless opts has G2048_PLAY_ONE_PHASE (or n_games > 65,536) the launch also carries helper
 * wavefronts that search the roots a game's next moves can start from ahead of time (same decisions, less latency for the
 * last games). They need g2048_play_games_workspace(n_games) bytes of device scratch: g2048_play_games_ws takes it from
 * the caller (64-byte aligned; NULL = play without helpers), g2048_play_games from hipMallocAsync on `stream`.
 * Outputs per game: final board / score (in place), moves played, valid / invalid move counts,
 * milestone_move_out[g][0..8) = move at which tiles 64..8192 first appeared (-1 = never), total children expanded
 * (optional), alive_out[g] = 1 if the game hit max_moves without finishing.
 * actions_out_or_null (ABI 3): the move-set of every game -- what train.py:51,67 collects in `moveset` and :140-142 writes to
 * *_best_moveset_tile_N.txt --, n_games rows of max_moves bytes: actions_out[g * max_moves + t] = the action of move t (0..3),
 * 0xFF from the game's end on (the library fills the array with 0xFF before the launch). One byte per move is also all that is
 * needed to rebuild the per-move histories of evaluate_beam_search.py:44-50, :72-75 afterwards: g2048_replay_games. The bytes do
 * not depend on helper wavefronts, tuning or the ranking switch. */
G2048_API int g2048_play_games(void *boards_inout, uint32_t *score_inout, int32_t *moves_out, int32_t *valid_out,
                     int32_t *invalid_out, int32_t *milestone_move_out, unsigned long long *expanded_sum_out_or_null,
                     uint8_t *alive_out, uint8_t *actions_out_or_null, int width, int depth, int early_threshold,
                     int mid_threshold, int max_moves, uint64_t seed, uint64_t game_id_base, size_t n_games, uint32_t opts,
                     void *stream);
G2048_API size_t g2048_play_games_workspace(size_t n_games);
G2048_API int g2048_play_games_ws(void *boards_inout, uint32_t *score_inout, int32_t *moves_out, int32_t *valid_out,
                        int32_t *invalid_out, int32_t *milestone_move_out, unsigned long long *expanded_sum_out_or_null,
                        uint8_t *alive_out, uint8_t *actions_out_or_null, int width, int depth, int early_threshold,
                        int mid_threshold, int max_moves, uint64_t seed, uint64_t game_id_base, size_t n_games, uint32_t opts,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Recorded games replayed into the per-move histories of the reference's run_game (evaluate_beam_search.py:44-50, :72-75,
 * :88-97: board_history, max_tiles_history, scores_history; :185-196 game_N_data.json): game k starts from boards0[k] with score
 * score0_or_null[k] (NULL: 0) and plays actions[k * actions_stride + t], t = 0 .. n_moves[k] - 1, with the draw of
 * g2048_step(step_index = t, board id = game_ids_or_null[k], or game_id_base + k) -- exactly the moves g2048_play_games made
 * when it wrote those bytes; an action byte above 3 (0xFF = no move) ends that game's replay early. Outputs, hist_stride
 * entries per game (hist_stride > the longest game): boards_hist_out[k * hist_stride + t] = the board BEFORE move t (t = 0 the
 * start, t = n_moves[k] the final board), score_hist_out likewise, flags_hist_out[k * hist_stride + t] = the flags byte of move
 * t (DONE / VALID / max code after the move: bits 3..7 give max_tiles_history). Entries past a game's end are left untouched.
 * A game's length is clamped to min(n_moves[k], hist_stride - 1, actions_stride): the kernel never reads past a game's row of
 * action bytes nor writes past its row of the history; actions_stride = 0 or hist_stride = 0 is refused. */
G2048_API int g2048_replay_games(const void *boards0, const uint32_t *score0_or_null, const uint64_t *game_ids_or_null, uint64_t game_id_base,
                       const uint8_t *actions, size_t actions_stride, const int32_t *n_moves, void *boards_hist_out,
                       uint32_t *score_hist_out_or_null, uint8_t *flags_hist_out_or_null, size_t hist_stride, uint64_t seed,
                       size_t n, void *stream);

/* ONE env driven from a host loop -- the drop-in Game2048Env of train.py:55-75 -- in one launch per iteration: op STEP =
 * Game2048Env.step(action) (environment/game_2048.py:170-210; an action outside 0..3 moves nothing, :97-114; draw (seed, STEP,
 * index, board_id)), op RESET = Game2048Env.reset() (:29-48; draws (seed, RESET, index, board_id)), op PEEK = nothing moves.
 * board_inout (16 bytes) / score_inout are updated in place, and record_out -- G2048_ENV_RECORD_BYTES bytes of device memory or
 * of pinned, device-visible host memory, 16-byte aligned -- receives everything the host mirrors of the env need, so that an
 * iteration costs one launch and one copy: [0,64) the state as int32 tile values (get_state, :50-57), [64,68) int32 score,
 * [68] the flags byte (DONE / VALID / max code), [69] the valid-move mask of the NEW state (get_valid_moves, :69-95, for the
 * next iteration), [70,72) a 16-bit token, [72,80) the f64 reward (:212-277; 0.0 for every op but STEP). The token is bits
 * 8..23 of `op` (G2048_ENV_TOKEN_SHIFT; 0 if the caller passes a bare op) and is written LAST, behind a system-scope fence: a host
 * that passes a fresh non-zero token and polls [70,72) of a pinned record has the complete record when it reads the token back
 * -- no stream synchronisation in the loop (the drop-in Game2048Env does this: ~2x the iterations per second).
 * The pieces of a step as ops of their own (the reference's methods of the same names, for callers that drive them directly):
 * op MOVE = Game2048Env._execute_move(action) (:97-114; action 0 is _move_left, :116-168): the slide / merge alone -- score +=
 * merged tiles, no spawn; flags: VALID = the board changed, DONE = is_game_over() of the result. op MOVE_AGENT = the same with
 * BeamSearchAgent._make_move's semantics (agents/beam_search_agent.py:194-258: DOWN returns rot180 of the true result); with
 * *score_inout = 0 before the call the record's score is its merge_score. op SPAWN = Game2048Env.add_new_tile() (:59-67) /
 * BeamSearchAgent._add_random_tile (:260-269): one 2 / 4 on an empty cell by the draw (seed, STEP, index, board_id, counter 1)
 * -- counter 0 is the step's own spawn --, nothing on a full board; flags: VALID = a tile was placed. */
#define G2048_ENV_RECORD_BYTES 80
#define G2048_ENV_OP_STEP  0u
#define G2048_ENV_OP_RESET 1u
#define G2048_ENV_OP_PEEK  2u
#define G2048_ENV_OP_MOVE  3u
#define G2048_ENV_OP_SPAWN 4u
#define G2048_ENV_OP_MOVE_AGENT 5u
#define G2048_ENV_TOKEN_SHIFT 8        /* op | (token << 8), token 0 .. 65535 */
G2048_API int g2048_env_step(void *board_inout, uint32_t *score_inout, uint32_t action, uint32_t op, void *record_out, uint64_t seed,
                   uint64_t index, uint64_t board_id, void *stream);

/* reference state layout (np.int32[16] real tile values, game_2048.py:36,57) <-> packed codes */
G2048_API int g2048_pack_i32(const int32_t *tiles, void *boards_out, size_t n, void *stream);
G2048_API int g2048_unpack_i32(const void *boards, int32_t *tiles_out, size_t n, void *stream);

/* per-shard metrics for the multi-GPU reduction: out[0] = n, out[1] = sum(score), out[2] = #done,
 * out[3] = sum(expanded or 0), out[4..22) = histogram of max code 0..17. out must hold 24 uint64,
 * and is accumulated into (zero it first). flags/expanded may be NULL. */
G2048_API int g2048_metrics(const void *boards, const uint32_t *score, const uint8_t *flags_or_null,
                  const uint32_t *expanded_or_null, unsigned long long *out24, size_t n, void *stream);

/* ---- graph-replayable loops -------------------------------------------------------------------------------
 * The entry points above take the step / decision index as a host scalar, so a captured hipGraph bakes it in. For
 * loops that replay ONE captured move many times (evaluation driver, rollouts), the per-move keys can live in a
 * device key block instead: g2048_keys_advance (one thread) writes the keys of every per-move RNG domain for index
 * *counter into keyblock_out[G2048_KEYBLOCK_WORDS] and increments *counter; the *_dyn variants read their keys (and
 * g2048_track_episodes_dyn its move index) from that block. Results are identical to the scalar forms called with
 * step_index = the counter's value. */
#define G2048_KEYBLOCK_WORDS 16
G2048_API int g2048_keys_advance(uint32_t *keyblock_out, unsigned long long *counter_inout, uint64_t seed, void *stream);
G2048_API int g2048_step_dyn(const void *boards_in, const uint8_t *actions, void *boards_out, uint32_t *score_inout,
                   void *reward_out, uint8_t *flags_out, const uint32_t *keyblock, uint64_t board_id_base, size_t n,
                   uint32_t opts, void *stream);
G2048_API int g2048_beam_get_action_dyn(const void *root_boards, const uint8_t *valid_mask_or_null, uint8_t *action_out,
                              float *prob_out, uint32_t *expanded_out_or_null, int width, int depth,
                              int early_threshold, int mid_threshold, const uint32_t *keyblock,
                              uint64_t game_id_base, size_t n_games, uint32_t opts, void *stream);
G2048_API int g2048_sample_actions_dyn(const float *probs, const uint8_t *mask4_or_null, uint8_t *actions_out, float *prob_out,
                             const uint32_t *keyblock, uint64_t env_id_base, size_t n, void *stream);
G2048_API int g2048_track_episodes_dyn(const uint8_t *flags, const uint32_t *expanded_or_null, uint8_t *alive_inout,
                             int32_t *moves_inout, int32_t *valid_inout, int32_t *invalid_inout,
                             int32_t *milestone_move_inout, unsigned long long *expanded_sum_inout_or_null,
                             const uint32_t *keyblock, size_t n, void *stream);

/* ---- PPO rollout step ---------------------------------------------------------------------------------------
 * One step of a PPO rollout for n envs in ONE launch -- what PPOAgent.get_action's sampling (agents/ppo_agent.py:211-221),
 * Game2048Env.step (environment/game_2048.py:170-210) and the preparation of the next policy call do between two forward
 * passes of the policy: the action is drawn from `probs` (float32 [n][4]) under the valid-move mask exactly as
 * g2048_sample_actions does (draw (seed, POLICY, index, env id)), the env is stepped exactly as g2048_step does (draw
 * (seed, STEP, index, env id); auto-reset in the EPISODE domain), and from the board still in registers the kernel
 * writes the NEXT observation (PPOAgent.normalize_state, :184-195; dtype by the OBS bits of opts) and the NEXT env
 * valid-move mask (game_2048.py:69-95). mask4_in_or_null = NULL: the mask of the current board is computed in place.
 * index = step_index + (step_counter_or_null ? *step_counter_or_null : 0): with a device counter the launch carries no
 * host-side step number, so a hipGraph of a whole T-step rollout (step_index = 0..T-1 baked in, counter += T per replay)
 * can be replayed. Optional outputs for the reward shaping below: next_boards_out (the next state BEFORE any auto-reset)
 * and state_maxcode_out (max log2 code of the state the action was taken in). */
#define G2048_ROLLOUT_OBS_SHIFT 4       /* opts bits 4..5: dtype of obs_next_out */
#define G2048_OBS_F32  0u
#define G2048_OBS_F16  1u
#define G2048_OBS_BF16 2u
G2048_API int g2048_rollout_step(const void *boards_in, const float *probs, const uint8_t *mask4_in_or_null, void *boards_out,
                       uint32_t *score_inout, uint8_t *actions_out, float *prob_out, void *reward_out, uint8_t *flags_out,
                       void *obs_next_out_or_null, uint8_t *mask4_next_out_or_null, void *next_boards_out_or_null,
                       uint8_t *state_maxcode_out_or_null, uint64_t seed, uint64_t step_index,
                       const unsigned long long *step_counter_or_null, uint64_t env_id_base, size_t n, uint32_t opts,
                       void *stream);

/* ---- PPOMemory.sample (agents/ppo_agent.py:21-50) + the head of PPOAgent.update (:342-354) on a device-resident buffer -----
 * `batch` DISTINCT transitions out of n_transitions (np.random.choice(len, batch, replace=False)): sample j is transition P(j),
 * P a bijection of 0 .. n_transitions-1 keyed by (seed, MINIBATCH, sample_index) -- a Feistel network walked until it lands in
 * range --, drawn and gathered by ONE launch with no host round trip. Inputs are the flattened trajectory arrays of a rollout:
 * obs ([n][16] float32 / float16 / bfloat16 per obs_kind = G2048_OBS_*, the normalized states the policy saw), actions (uint8),
 * log_probs (float32), rewards (float32, or float64 when rewards_f64 -- e.g. the shaped reward remember() stores), next_boards
 * (the next states BEFORE any auto-reset, as g2048_rollout_step writes them) and flags. Outputs, what update() turns the sample
 * into: states_out float32 [batch][16], actions_out int64, old_log_probs_out float32, rewards_out float32, next_states_out
 * float32 [batch][16] = normalize_state(next state) (:184-195), dones_out float32 (1.0 = done), and optionally the indices. */
G2048_API int g2048_minibatch_gather(const void *obs, uint32_t obs_kind, const uint8_t *actions, const float *log_probs, const void *rewards,
                           uint32_t rewards_f64, const void *next_boards, const uint8_t *flags, size_t n_transitions, size_t batch,
                           uint64_t seed, uint64_t sample_index, float *states_out, int64_t *actions_out, float *old_log_probs_out,
                           float *rewards_out, float *next_states_out, float *dones_out, int64_t *indices_out_or_null, void *stream);

/* ---- PPOAgent.remember reward shaping (agents/ppo_agent.py:234-269) for an ORDERED batch of n transitions ------------
 * The reference calls remember() once per transition, and two of its terms carry state from call to call: the
 * "new highest tile" bonus (:241-246, self.highest_tile_seen) and the novelty bonus (:259-262, self.seen_states).
 * Batched with exactly the sequential semantics for the order i = 0..n-1 (a rollout buffer [T][N] flattened: step t,
 * then env id), continued across calls through highest_code_inout / the table / index_base:
 *   g2048_shaping_scan    prev_highest_out[i] = max(*highest_code_inout, max log2 code after transitions 0..i-1) -- the
 *                         value highest_tile_seen has when the reference reaches transition i -- and then
 *                         *highest_code_inout = the maximum over everything (device uint32; a fresh agent starts at 1
 *                         = tile 2, ppo_agent.py:171). flags = the flags byte g2048_step / g2048_rollout_step wrote
 *                         (bits 3..7 = max code of the next state). workspace: g2048_shaping_scan_workspace(n) bytes.
 *   g2048_seen_insert     every next state is looked up / inserted in an open-addressing hash set keyed by the whole
 *                         16-byte board (table: 2^capacity_log2 slots of G2048_SEEN_SLOT_BYTES bytes, zero-initialised by
 *                         the caller; keep it at most half full); each key keeps the smallest transition index
 *                         index_base + i that presented it. slot_out[i] = the key's slot. *count_inout grows by the
 *                         number of new keys; *overflow_flag becomes non-zero if the table filled up.
 *   g2048_shaping_apply   shaped_out[i] = the reward remember() stores: env_reward[i] (+ 5.0 * (log2 next_max - log2
 *                         highest so far) if a new highest tile) (+ -2.0 * (log2 current_max - log2 next_max) if the
 *                         max tile fell) + 0.1 * sum(log2 of the 4 largest tiles) (+ 0.2 if transition i is the FIRST
 *                         in order to present its next state) + 0.3 * evaluate_heuristic(next_state), added in that
 *                         order in f64. Call after g2048_seen_insert of the same batch (stream order is enough).
 *   g2048_seen_rehash     re-inserts every key of a table into a larger zeroed one (first indices kept).
 * The only deviation from the reference: its set holds Python hash() values of the board bytes, so two different boards
 * whose 64-bit hashes collide count as one there; here keys are compared in full. */
#define G2048_SEEN_SLOT_BYTES 32
G2048_API size_t g2048_shaping_scan_workspace(size_t n);
G2048_API int g2048_shaping_scan(const uint8_t *flags, uint8_t *prev_highest_out, uint32_t *highest_code_inout, void *workspace,
                       size_t n, void *stream);
G2048_API int g2048_seen_insert(const void *next_boards, uint64_t index_base, void *table, uint32_t capacity_log2,
                      unsigned long long *count_inout, uint32_t *overflow_flag, uint32_t *slot_out, size_t n, void *stream);
G2048_API int g2048_seen_rehash(const void *old_table, uint32_t old_capacity_log2, void *new_table, uint32_t new_capacity_log2,
                      uint32_t *overflow_flag, void *stream);
G2048_API int g2048_shaping_apply(const void *next_boards, const uint8_t *state_maxcode, const uint8_t *flags, const double *env_reward,
                        const uint8_t *prev_highest, const void *table, const uint32_t *slots, uint64_t index_base,
                        double *shaped_out, uint8_t *novel_out_or_null, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif
