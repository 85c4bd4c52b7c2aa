/*
 * g2048_testing.h -- test, measurement and benchmark-input hooks of libg2048_hip.so.
 *
 * Not part of the drop-in boundary (include/g2048.h): nothing here replaces a reference method. These entry points exist so
 * that tests can reach pieces of the kernels on their own (the ranking network, the instruction-level self-test), so that
 * measurements can override the evaluation's helper-wavefront parameters, and so that the benchmark's synthetic inputs
 *  are generated on the device by the same counter RNG the CPU oracle uses. Same conventions as g2048.h.
 */
#ifndef G2048_TESTING_H
#define G2048_TESTING_H
#include "g2048.h"
#ifdef __cplusplus
extern "C" {
#endif

/* g2048_play_games_ws with the helper-wavefront parameters given explicitly -- a measurement / test interface (the games are
 * the same for every setting; only the time changes). tuning4 = { helper wavefronts (clamped to 8 per game and to the
 * device's cap, g2048_launch_plan), games left at which every remaining game registers for helpers (>= n_games: at once),
 * "stuck" threshold = invalid minus valid moves at which a game registers early (clamped to 1 .. 2^20), microseconds an
 * owner polls for a posted result (clamped to 1000) }. The defaults g2048_play_games uses: { min(8 n, max(n / 2, 1024), cap),
 * max(n / 8, 256), 16, 150 }. */
G2048_API int g2048_play_games_tuned(void *boards_inout, uint32_t *score_inout, int32_t *moves_out, int32_t *valid_out,
                           int32_t *invalid_out, int32_t *milestone_move_out, unsigned long long *expanded_sum_out_or_null,
                           uint8_t *alive_out, uint8_t *actions_out_or_null, int width, int depth, int early_threshold,
                           int mid_threshold, int max_moves, uint64_t seed, uint64_t game_id_base, size_t n_games, uint32_t opts,
                           void *workspace, size_t workspace_bytes, const uint32_t *tuning4, void *stream);

/* The launch arithmetic the library derives from the device's size, as a pure host function (no launch, no allocation):
 * out4 = { SIMD row length beam batches are dealt in (4 per compute unit), smallest batch that gets the depth-balanced
 * order (four searches per SIMD), most helper wavefronts a g2048_play_games launch may carry (a quarter of the wavefronts
 * the device holds at once: compute_units x resident_blocks_per_cu / 4), default helper wavefronts for n_games }.
 * compute_units = 0: the current device's count; resident_blocks_per_cu = 0: 32 (the hardware cap for 64-thread blocks). */
G2048_API int g2048_launch_plan(int compute_units, int resident_blocks_per_cu, size_t n_games, uint32_t *out4);

/* What the library asks the CURRENT device before a beam / evaluation launch of this width and size (host only):
 * out6 = { compute units, blocks of the evaluation kernel one compute unit holds (occupancy query), helper-wavefront cap,
 * default helper wavefronts for n_games, blocks of the beam kernel the device holds at once, 1 if a g2048_beam_get_action
 * launch of n_games runs with issue priority by remaining levels (every block resident at once), else 0 }. */
G2048_API int g2048_device_plan(int width, size_t n_games, uint32_t *out6);

/* synthetic inputs of the benchmark configs, generated on the device:
 * each cell empty with probability p_empty_u16/65536 else code uniform in 1..max_code; an all-empty
 * draw gets code 1 at cell 0. actions: uniform 0..3. */
G2048_API int g2048_synth_boards(void *boards_out, uint64_t seed, uint64_t board_id_base, size_t n,
                       uint32_t p_empty_u16, uint32_t max_code, void *stream);
G2048_API int g2048_synth_actions(uint8_t *actions_out, uint64_t seed, uint64_t step_index,
                        uint64_t board_id_base, size_t n, void *stream);

/* device self-test of the instruction-level assumptions the kernels rely on (v_perm_b32 byte order,
 * udot4, f64 contraction off). Writes 0 to *result_out (device uint32) when all hold. */
G2048_API int g2048_selftest(uint32_t *result_out, void *stream);
/* the beam kernel's ranking network on its own (tests): every 64 keys of keys_inout become the 64 largest, descending, of
 * those 64 and -- if extra_or_null is given -- 16 more per block (0 = no key; all other keys distinct and > 0).
 * key_bits = 32: uint32 keys; 64: uint64 keys stored as (low word, high word), the network of the f64-score levels. */
G2048_API int g2048_sort_selftest(uint32_t *keys_inout, const uint32_t *extra_or_null, size_t n_waves, int key_bits, void *stream);

#ifdef __cplusplus
}
#endif
#endif
