// g2048_instrument.h -- the library's one compile-time switch: measurement builds (tools/*_timeline.py).
//
//   -DG2048_INSTRUMENT=1   beam timeline: g2048_beam_get_action writes, per search, its start tick into prob_out and
//                          duration | SIMD << 18 into expanded_out (tools/beam_timeline.py)
//   -DG2048_INSTRUMENT=2   evaluation timeline: g2048_play_games writes start / end tick, registration move and tick,
//                          owner searches, helper results taken / late into the milestone record (tools/play_timeline.py)
//   -DG2048_INSTRUMENT=4   step timeline: lanes 0..2 of every wavefront of g2048_step write start tick, end tick and
//                          SIMD over the f32 reward (tools/step_timeline.py)
// Such a build OVERWRITES real outputs. It reports itself through g2048_build_flags() (0 for the product build), and the
// Python loader refuses it unless the caller opts in (g2048/_lib.py). Everything else that used to be an A/B switch is gone
// from the source: the measured losers are recorded under profiles/ and live in the git history.
#pragma once
#ifndef G2048_INSTRUMENT
#define G2048_INSTRUMENT 0
#endif
namespace g2048 {
constexpr unsigned kInstrument = G2048_INSTRUMENT;
constexpr bool kBeamTiming = (kInstrument & 1u) != 0u, kPlayTiming = (kInstrument & 2u) != 0u, kStepTiming = (kInstrument & 4u) != 0u;
// 13-bit SIMD number of the executing wavefront from HW_ID (simd [5:4], cu [11:8], sh [12], se [15:13]) and XCC_ID [3:0]
__device__ __forceinline__ unsigned simd_id()
{
    const unsigned h = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), x = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    return ((h >> 4) & 3u) | (((h >> 8) & 15u) << 2) | (((h >> 12) & 1u) << 6) | (((h >> 13) & 7u) << 7) | ((x & 15u) << 10);
}
}  // namespace g2048
