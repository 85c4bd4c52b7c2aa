// g2048_kernels.hip -- batched env / heuristic kernels for gfx950 and their C-ABI (include/g2048.h).
//
// Mapping: one lane owns one board for the compute kernels (16-byte vector load/store per lane,
// 1 KiB per wave-instruction, fully coalesced); the pure format kernels (obs / pack / unpack) give
// one board ROW to a lane so that their wide side is a 16-byte access per lane as well.
// No LDS, no MFMA: the work is integer SWAR on four VGPRs per board (g2048_board.h) plus ~20 f64
// operations for the shaped reward. Compile with -ffp-contract=off (reward / eval operation order).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>

#include "../../include/g2048.h"
#include "../../include/g2048_testing.h"
#include "g2048_board.h"
#include "g2048_instrument.h"
#include "g2048_rng.h"

using namespace g2048;

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(G2048_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return G2048_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

// device key block (uint32[G2048_KEYBLOCK_WORDS]) read by the *_dyn entry points
enum { KB_STEP = 0, KB_EPISODE = 2, KB_BEAM = 4, KB_POLICY = 6, KB_INDEX = 8 };

constexpr int kBlock = 256;
constexpr int kStepBoardsPerLane = 1;     // default of g2048_step: measured fastest (profiles/r01_step_tune.txt)
inline unsigned blocks_for(size_t n, int per_block = kBlock) { return (unsigned)((n + per_block - 1) / per_block); }

// selector words of the direction network (g2048_board.h, "direction by table"); every wave copies them into the
// block's LDS table itself -- all waves write the same 32 words, and a wave's own write precedes its reads in its
// LDS queue, so no block barrier is needed -- and a lane then fetches the eight words of its action with two
// ds_read_b128.
__device__ const uint32_t kDirTable[G2048_DIR_TABLE_WORDS] = G2048_DIR_TABLE_INIT;

#include "g2048_step_table.h"

__device__ __forceinline__ void dir_table_to_lds(uint4 *s_dir)
{
    const uint32_t l = threadIdx.x & (G2048_DIR_TABLE_WORDS - 1u);          // both halves of the wave write the same 32 words:
    reinterpret_cast<uint32_t *>(s_dir)[l] = kDirTable[l];                    // no exec masking, no branch
}

__device__ __forceinline__ DirSel dir_sel(const uint4 *s_dir, uint32_t action)
{
    const uint4 i = s_dir[2u * action], o = s_dir[2u * action + 1u];
    return DirSel{i.x, i.y, i.z, i.w, o.x, o.y, o.z, o.w};
}

__device__ __forceinline__ Board load_board(const uint4 *p, size_t i)
{
    const uint4 v = p[i];
    return Board{{v.x, v.y, v.z, v.w}};
}

__device__ __forceinline__ void store_board(uint4 *p, size_t i, const Board &b)
{
    p[i] = make_uint4(b.w[0], b.w[1], b.w[2], b.w[3]);
}

// ------------------------------------------------------------------ step ------
// Game2048Env.step (environment/game_2048.py:170-210), one board per lane per pass, B passes per lane.
// All B loads of a lane are issued before the first board is computed: with ~400 VALU instructions per
// board the kernel sits between the HBM and the VALU roofline, and a wave that only ever has one board
// in flight serialises load latency -> compute -> store. A block owns 256*B consecutive boards; pass k
// of a wave touches 64 consecutive boards (1 KiB per wave-instruction).
// NOOP_ACTIONS: action bytes above 3 move nothing, as in the reference (drop-in class); otherwise the low two bits count
template <bool REWARD_F64, bool AUTO_RESET, int B, int BLOCK, bool RANDOM_ACTIONS = false, bool NOOP_ACTIONS = false>
__global__ __launch_bounds__(BLOCK) void step_kernel(size_t n, const uint32_t *__restrict__ keyblock,
                                                     const uint4 *boards_in,       // may alias boards_out
                                                     const uint8_t *__restrict__ actions,
                                                     uint32_t *__restrict__ score,
                                                     uint32_t id_lo_base, uint32_t id_hi_term, uint32_t k0, uint32_t k1,
                                                     // (the first 14 dwords -- everything the head of a wavefront needs before its loads
                                                     // can go out -- arrive preloaded in SGPRs: -amdgpu-kernarg-preload-count, g2048/_build.py)
                                                     uint4 *boards_out,
                                                     void *__restrict__ reward_out,
                                                     uint8_t *__restrict__ flags_out,
                                                     uint32_t e0, uint32_t e1,
                                                     uint32_t a0 = 0, uint32_t a1 = 0)
{
    // Streaming launches (two boards per lane, from 4 Mi boards on): a new wavefront issues its loads ahead of the arithmetic of
    // the older ones on its SIMD (the arbiter would serve those first) -- 152.8 -> 147.5 us per 16 Mi boards. Nothing to gain at
    // 1 Mi boards, where five phase-priority schemes stayed within +-1 % (profiles/r03_beam_priority.txt, section 10).
    [[maybe_unused]] const unsigned long long tm0 = kStepTiming ? wall_clock64() : 0ull;       // (measurement builds only, tools/step_timeline.py)
    if (B > 1) __builtin_amdgcn_s_setprio(3);
    if (keyblock) { k0 = keyblock[KB_STEP]; k1 = keyblock[KB_STEP + 1]; e0 = keyblock[KB_EPISODE]; e1 = keyblock[KB_EPISODE + 1]; }
    // the ids of one launch share their high word (step_impl splits a launch that would cross a multiple of 2^32), so its term of
    // the draw is folded into the second key word here, on the scalar unit, and a lane hashes its id's low word only (rng_draw_lo)
    k1 += id_hi_term; e1 += id_hi_term; a1 += id_hi_term;
    __shared__ uint4 s_dir[kStepTableWords / 4];
    const uint2 dir_word = step_table_word();                     // (stored to LDS below, once the lane's own loads are on their way)
    const TenthFromLds tenth{reinterpret_cast<const StepTable *>(s_dir)};
    // per-block scalar bases + a 32-bit lane offset: the 7 streams are addressed as SGPR base + VGPR offset
    const size_t block0 = (size_t)blockIdx.x * (BLOCK * B);
    const uint4 *bin = boards_in + block0;
    uint4 *bout = boards_out + block0;
    const uint8_t *act = actions + block0;
    uint32_t *scp = score + block0;
    uint8_t *flp = flags_out + block0;
    float *rw32 = static_cast<float *>(reward_out) + block0;
    double *rw64 = static_cast<double *>(reward_out) + block0;
    // boards of this block that exist (32-bit scalar arithmetic: a 64-bit unsigned compare would run on the VALU)
    const uint64_t rem = (uint64_t)n - block0;
    uint32_t rem_hi = (uint32_t)(rem >> 32);
    const uint32_t rem_lo = (uint32_t)rem;
    asm volatile("" : "+s"(rem_hi));             // (kept apart on purpose: recombined, the test is a 64-bit compare, which only the VALU has)
    const uint32_t lim = rem_hi != 0u ? (uint32_t)(BLOCK * B) : min(rem_lo, (uint32_t)(BLOCK * B));
    const bool full = lim == (uint32_t)(BLOCK * B);               // wave-uniform: every lane of the block is in range
    // Every lane loads -- the lanes past the end of a ragged last block a clamped (valid) index whose data they drop -- so that
    // the loads need no exec region and go out BEFORE the direction table's LDS write waits for its word: one memory round trip
    // at the head of a wavefront, not two.
    Board prev[B];
    uint32_t action[B], sc[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
        const uint32_t j = min(threadIdx.x + (uint32_t)k * BLOCK, lim - 1u);
        prev[k] = load_board(bin, j);
        if (!RANDOM_ACTIONS) action[k] = act[j];
        sc[k] = scp[j];
    }
    step_table_store(s_dir, dir_word);
    if (B > 1) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int k = 0; k < B; ++k) {
        // (B == 1: the clamped index serves the stores too -- the same offsets as the loads' -- since the lanes it differs for leave
        // before them)
        const uint32_t j = B == 1 ? min(threadIdx.x, lim - 1u) : threadIdx.x + (uint32_t)k * BLOCK;
        if (B > 1 && !(full || j < lim)) break;
        const uint32_t id = id_lo_base + (uint32_t)block0 + j;           // low word of the board id (the launch never wraps it)
        if (RANDOM_ACTIONS) action[k] = rng_draw_lo(a0, a1, id, 0u) >> 30;    // what g2048_synth_actions would write
        const StepOut o = NOOP_ACTIONS ? step_board_sel_noop(prev[k], dir_sel(s_dir, action[k] & 3u), action[k] > 3u, rng_draw_lo(k0, k1, id, 0u), tenth)
                                       : step_board_sel(prev[k], dir_sel(s_dir, action[k] & 3u), rng_draw_lo(k0, k1, id, 0u), tenth);
        Board cur = o.board;
        uint32_t s = sc[k] + o.gain;
        if (AUTO_RESET) {
            if (o.flags & G2048_FLAG_DONE) {
                cur = fresh_board(rng_draw_lo(e0, e1, id, 0u), rng_draw_lo(e0, e1, id, 1u));
                s = 0u;
            }
        }
        // B == 1: the lanes past the end of a ragged last block ran the arithmetic on their clamped loads (no exec region around
        // the body: the compiler would sink the loads into it, behind the table's write); they store nothing
        if (B == 1 && !(full || threadIdx.x < lim)) return;
        store_board(bout, j, cur);
        scp[j] = s;
        if (REWARD_F64) rw64[j] = o.reward;
        else rw32[j] = (float)o.reward;
        flp[j] = (uint8_t)o.flags;        // bit0 DONE, bit1 VALID, bits 3..7 max code (include/g2048.h)
        if constexpr (kStepTiming && !REWARD_F64) {       // lanes 0..2 of every wavefront: start tick, end tick, SIMD over the reward
            const uint32_t l = threadIdx.x & 63u;
            if (k == 0 && l < 3u) reinterpret_cast<uint32_t *>(rw32)[j] = l == 0u ? (uint32_t)tm0 : l == 1u ? (uint32_t)wall_clock64() : simd_id();
        }
    }
}

// ------------------------------------------------------------- step, T times --
// T consecutive env steps of every board in ONE launch (random playouts: SURVEY 8d C2 "rollout" variant). Consecutive
// g2048_step launches re-read and re-write 46 B per board and pay the launch's ramp-up every step; here the board, its score
// and the direction table stay in registers / LDS for all T steps, the per-step keys of the three RNG domains are derived
// on the scalar unit from (seed, domain, step index) -- the same keys the host derives for g2048_step -- and only what
// the caller asks for per step (reward / flags streams) is written. Step t is bit-for-bit g2048_step(step_index0 + t,
// G2048_STEP_RANDOM_ACTIONS [| G2048_STEP_AUTO_RESET]) -- or, with RANDOM false, g2048_step with the explicit action
// actions[t * n + i] (a recorded move sequence, an open-loop plan): the next step's action byte is requested before this
// step is computed, so the load never sits on the dependent chain.
template <bool REWARD_F64, bool AUTO_RESET, int BLOCK, bool RANDOM = true>
__global__ __launch_bounds__(BLOCK) void step_many_kernel(const uint4 *boards_in, const uint8_t *__restrict__ actions,
                                                          uint4 *boards_out, uint32_t *__restrict__ score,
                                                          void *__restrict__ reward_stream, uint8_t *__restrict__ flags_stream,
                                                          uint8_t *__restrict__ flags_last, uint32_t *__restrict__ episodes_out,
                                                          uint64_t seed, uint64_t step_index0, uint32_t steps, uint64_t id_base,
                                                          size_t n)
{
    // (the reward's constants stay arithmetic here: with them read from LDS the compiler no longer sinks the reward into the
    // `if (reward_stream)` branch, and a launch without a reward stream pays for it -- 7.9 -> 8.9 us per step, measured)
    __shared__ uint4 s_dir[G2048_DIR_TABLE_WORDS / 4];
    dir_table_to_lds(s_dir);
    // per-block scalar bases + a 32-bit lane offset, as in step_kernel; the streams advance by n per step on the scalar unit
    const size_t block0 = (size_t)blockIdx.x * BLOCK;
    if (block0 + threadIdx.x >= n) return;
    const uint32_t j = threadIdx.x;
    const uint64_t id = id_base + block0 + j;
    Board cur = load_board(boards_in + block0, j);
    uint32_t sc = score[block0 + j], flags = 0u, episodes = 0u;
    float *rw32 = static_cast<float *>(reward_stream) + block0;
    double *rw64 = static_cast<double *>(reward_stream) + block0;
    uint8_t *fl = flags_stream + block0;
    const uint8_t *act = actions + block0;
    uint32_t next_action = RANDOM ? 0u : (uint32_t)act[j];
    for (uint32_t t = 0; t < steps; ++t) {
        const uint64_t index = step_index0 + t;                          // uniform: scalar unit
        const Keys ks = rng_keys(seed, DOM_STEP, index);
        uint32_t action;
        if (RANDOM) {
            const Keys ka = rng_keys(seed, DOM_SYNTH_ACTION, index);
            action = rng_draw(ka.k0, ka.k1, id, 0u) >> 30;                  // what g2048_synth_actions would write
        } else {
            action = next_action & 3u;                                   // (the low two bits count, as in g2048_step)
            act += n;
            if (t + 1u < steps) next_action = (uint32_t)act[j];
        }
        const StepOut o = step_board_sel(cur, dir_sel(s_dir, action), rng_draw(ks.k0, ks.k1, id, 0u));
        cur = o.board;
        sc += o.gain;
        flags = o.flags;
        if (AUTO_RESET) {
            if (o.flags & G2048_FLAG_DONE) {
                const Keys ke = rng_keys(seed, DOM_EPISODE, index);
                cur = fresh_board(rng_draw(ke.k0, ke.k1, id, 0u), rng_draw(ke.k0, ke.k1, id, 1u));
                sc = 0u;
                ++episodes;
            }
        }
        if (reward_stream) {
            if (REWARD_F64) rw64[j] = o.reward;
            else rw32[j] = (float)o.reward;
            rw64 += n; rw32 += n;
        }
        if (flags_stream) { fl[j] = (uint8_t)o.flags; fl += n; }
    }
    store_board(boards_out + block0, j, cur);
    score[block0 + j] = sc;
    flags_last[block0 + j] = (uint8_t)flags;
    if (episodes_out) episodes_out[block0 + j] = episodes;
}

// ---------------------------------------------------------- one env, one record --
// The drop-in Game2048Env (environment/game_2048.py:29-48 reset, :170-210 step, :50-57 get_state, :69-95 get_valid_moves) is
// ONE board driven from a Python loop (train.py:55-75 calls get_valid_moves and step every iteration): what costs there is
// launches and host round trips, not arithmetic. One launch does the whole iteration's device work: the action arrives by value
// (no fill launch), the board and score are updated in place, and ONE 80-byte record receives everything the host mirrors need:
// the state in the reference's layout (int32 tile values), the score, the flags, the NEXT state's valid-move mask (so that
// get_valid_moves() is a cache read) and the f64 reward. One wavefront, lane 0 stores.
struct EnvRecord { int32_t tiles[16]; int32_t score; uint8_t flags, valid_next; uint16_t token; double reward; };
static_assert(sizeof(EnvRecord) == G2048_ENV_RECORD_BYTES && offsetof(EnvRecord, reward) == 72, "record layout is part of the ABI");

__global__ __launch_bounds__(64) void env_step_kernel(uint4 *board_inout, uint32_t *score_inout, uint32_t action, uint32_t op,
                                                     EnvRecord *record, uint32_t k0, uint32_t k1, uint64_t id, uint32_t token)
{
    const uint4 bv = board_inout[0];
    Board cur = {{bv.x, bv.y, bv.z, bv.w}};
    uint32_t sc = score_inout[0], flags;
    double reward = 0.0;
    if (op == G2048_ENV_OP_RESET) {                               // :29-48 (keys of the RESET domain, index = epoch)
        cur = fresh_board(rng_draw(k0, k1, id, 0u), rng_draw(k0, k1, id, 1u));
        sc = 0u;
        flags = max_code(cur) << G2048_FLAG_MAXCODE_SHIFT;
    } else if (op == G2048_ENV_OP_PEEK) {                         // the record of the board as it is (after `board` was assigned)
        flags = (max_code(cur) << G2048_FLAG_MAXCODE_SHIFT) | (game_over(cur) ? G2048_FLAG_DONE : 0u);
    } else if (op == G2048_ENV_OP_MOVE || op == G2048_ENV_OP_MOVE_AGENT) {
        // _execute_move (:97-114) / BeamSearchAgent._make_move (agents/beam_search_agent.py:194-258): the move alone
        uint32_t gain = 0u;
        Board moved = cur;
        if (action <= 3u) moved = op == G2048_ENV_OP_MOVE ? move_env(cur, action, gain) : move_agent(cur, action, gain, false);
        const bool changed = !same(moved, cur);
        cur = moved; sc += gain;
        flags = (max_code(cur) << G2048_FLAG_MAXCODE_SHIFT) | (changed ? G2048_FLAG_VALID : 0u) | (game_over(cur) ? G2048_FLAG_DONE : 0u);
    } else if (op == G2048_ENV_OP_SPAWN) {                        // add_new_tile (:59-67) / _add_random_tile (agent :260-269)
        const uint32_t n_empty = spawn_prefix(cur, rng_draw(k0, k1, id, 1u));
        flags = (max_code(cur) << G2048_FLAG_MAXCODE_SHIFT) | (n_empty ? G2048_FLAG_VALID : 0u) | (game_over(cur) ? G2048_FLAG_DONE : 0u);
    } else {                                                      // :170-210; an action outside 0..3 moves nothing (:97-114)
        const uint32_t *d = kDirTable + 8u * (action & 3u);
        const StepOut o = step_board_sel_noop(cur, DirSel{d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]}, action > 3u,
                                              rng_draw(k0, k1, id, 0u));
        cur = o.board; sc += o.gain; flags = o.flags; reward = o.reward;
    }
    const uint32_t valid_next = valid_mask_env(cur);
    if (threadIdx.x != 0) return;
    board_inout[0] = make_uint4(cur.w[0], cur.w[1], cur.w[2], cur.w[3]);
    score_inout[0] = sc;
    int4 *tiles = reinterpret_cast<int4 *>(record->tiles);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t x = cur.w[r];
        auto tile = [](uint32_t c) -> int { return c ? (int)(1u << c) : 0; };
        tiles[r] = make_int4(tile(x & 0xffu), tile((x >> 8) & 0xffu), tile((x >> 16) & 0xffu), tile(x >> 24));
    }
    record->score = (int32_t)sc;
    record->flags = (uint8_t)flags; record->valid_next = (uint8_t)valid_next;
    record->reward = reward;
    // the caller's token LAST, behind a system-scope fence: a host polling it on a pinned record has the whole record when it
    // sees the token, without a stream synchronisation (whose wake-up costs more than this launch)
    __threadfence_system();
    __hip_atomic_store(&record->token, (uint16_t)token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------ recorded games --
// The per-move histories of the reference's run_game (evaluate_beam_search.py:44-50, :72-75, :88-97: board_history,
// max_tiles_history, scores_history) rebuilt from a game's action bytes: the env's draws are counter-based -- move t of game g
// spawns with (seed, STEP, t, g) whoever plays it --, so one byte per move is all g2048_play_games has to keep, and replaying
// them reproduces every intermediate board. One lane per game; entry t of a game's history is the state BEFORE move t (entry 0:
// the start, entry n_moves: the final state), which is the reference's list (its entry t + 1 = the state after move t).
__global__ __launch_bounds__(kBlock) void replay_kernel(const uint4 *__restrict__ boards0, const uint32_t *__restrict__ score0,
                                                       const unsigned long long *__restrict__ game_ids, uint64_t id_base,
                                                       const uint8_t *__restrict__ actions, size_t actions_stride,
                                                       const int32_t *__restrict__ n_moves, uint4 *__restrict__ boards_hist,
                                                       uint32_t *__restrict__ score_hist, uint8_t *__restrict__ flags_hist,
                                                       size_t hist_stride, uint64_t seed, size_t n)
{
    __shared__ uint4 s_dir[G2048_DIR_TABLE_WORDS / 4];
    dir_table_to_lds(s_dir);
    const size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool in_range = k < n;
    const uint64_t id = in_range ? (game_ids ? (uint64_t)game_ids[k] : id_base + k) : 0ull;
    Board cur = in_range ? load_board(boards0, k) : Board{{0u, 0u, 0u, 0u}};
    uint32_t sc = (in_range && score0) ? score0[k] : 0u;
    // (never beyond the history's room nor beyond the game's row of action bytes, whatever the caller's counts say)
    const uint32_t len = in_range ? (uint32_t)min(min((size_t)max(n_moves[k], 0), hist_stride - 1u), actions_stride) : 0u;
    const uint8_t *act = actions + k * actions_stride;
    uint4 *bh = boards_hist + k * hist_stride;
    uint32_t *sh = score_hist ? score_hist + k * hist_stride : nullptr;
    uint8_t *fh = flags_hist ? flags_hist + k * hist_stride : nullptr;
    uint32_t next_action = len ? (uint32_t)act[0] : 0xffu;
    uint32_t t = 0;
    for (;; ++t) {                                               // t is wave-uniform: the keys of move t come from the scalar unit
        const bool live = t < len && next_action <= 3u;          // (a byte above 3 -- 0xFF = no move -- ends this game's replay)
        if (__ballot(live) == 0ull) break;
        if (!live) continue;
        const Keys ks = rng_keys(seed, DOM_STEP, (uint64_t)t);
        const uint32_t action = next_action;
        if (t + 1u < len) next_action = (uint32_t)act[t + 1u];   // requested before this move is computed
        bh[t] = make_uint4(cur.w[0], cur.w[1], cur.w[2], cur.w[3]);
        if (sh) sh[t] = sc;
        const StepOut o = step_board_sel(cur, dir_sel(s_dir, action), rng_draw(ks.k0, ks.k1, id, 0u));
        cur = o.board;
        sc += o.gain;
        if (fh) fh[t] = (uint8_t)o.flags;
        if (t + 1u == len || next_action > 3u) {                 // the state after the last move replayed
            bh[t + 1u] = make_uint4(cur.w[0], cur.w[1], cur.w[2], cur.w[3]);
            if (sh) sh[t + 1u] = sc;
        }
    }
    if (in_range && (len == 0u || (uint32_t)act[0] > 3u)) {      // nothing to replay: the history is the start state
        bh[0] = make_uint4(cur.w[0], cur.w[1], cur.w[2], cur.w[3]);
        if (sh) sh[0] = sc;
    }
}

// ------------------------------------------------------------------ reset -----
__global__ __launch_bounds__(kBlock) void reset_kernel(uint4 *__restrict__ boards_out, uint32_t *__restrict__ score_out,
                                                      uint32_t k0, uint32_t k1, uint64_t id_base, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    store_board(boards_out, i, fresh_board(rng_draw(k0, k1, id_base + i, 0u), rng_draw(k0, k1, id_base + i, 1u)));
    if (score_out) score_out[i] = 0u;
}

// ------------------------------------------------------------ valid moves -----
template <bool AGENT>
__global__ __launch_bounds__(kBlock) void valid_kernel(const uint4 *__restrict__ boards, uint8_t *__restrict__ mask_out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Board b = load_board(boards, i);
    mask_out[i] = (uint8_t)(AGENT ? valid_mask_agent(b, false) : valid_mask_env(b));
}

// ------------------------------------------------------- episode tracking -----
// The per-move bookkeeping of the reference's evaluation loop (evaluate_beam_search.py:42-64, run_evaluation.py
// :56-69) for every game of a batch, after a step: milestone tiles 64..8192 record the move index at which they
// were first reached, valid / invalid move counters, move counter, and the game leaves the `alive` set when done.
__global__ __launch_bounds__(kBlock) void track_kernel(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ expanded,
                                                      uint8_t *__restrict__ alive, int32_t *__restrict__ moves,
                                                      int32_t *__restrict__ valid_cnt, int32_t *__restrict__ invalid_cnt,
                                                      int4 *__restrict__ milestone, unsigned long long *__restrict__ expanded_sum,
                                                      int32_t move_index, size_t n, const uint32_t *__restrict__ keyblock)
{
    if (keyblock) move_index = (int32_t)keyblock[KB_INDEX];
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (!alive[i]) return;
    const uint32_t f = flags[i];
    const int32_t maxcode = (int32_t)(f >> G2048_FLAG_MAXCODE_SHIFT);
    int4 lo = milestone[2 * i], hi = milestone[2 * i + 1];               // codes 6..9 and 10..13
    if (lo.x < 0 && maxcode >= 6) lo.x = move_index;
    if (lo.y < 0 && maxcode >= 7) lo.y = move_index;
    if (lo.z < 0 && maxcode >= 8) lo.z = move_index;
    if (lo.w < 0 && maxcode >= 9) lo.w = move_index;
    if (hi.x < 0 && maxcode >= 10) hi.x = move_index;
    if (hi.y < 0 && maxcode >= 11) hi.y = move_index;
    if (hi.z < 0 && maxcode >= 12) hi.z = move_index;
    if (hi.w < 0 && maxcode >= 13) hi.w = move_index;
    milestone[2 * i] = lo; milestone[2 * i + 1] = hi;
    if (f & G2048_FLAG_VALID) valid_cnt[i] += 1; else invalid_cnt[i] += 1;
    moves[i] += 1;
    if (expanded && expanded_sum) expanded_sum[i] += expanded[i];
    if (f & G2048_FLAG_DONE) alive[i] = 0;
}

// ----------------------------------------------------------------- policy -----
__global__ __launch_bounds__(kBlock) void sample_kernel(const float4 *__restrict__ probs, const uint8_t *__restrict__ mask,
                                                       uint8_t *__restrict__ actions, float *__restrict__ prob_out,
                                                       uint32_t k0, uint32_t k1, uint64_t id_base, size_t n,
                                                       const uint32_t *__restrict__ keyblock)
{
    if (keyblock) { k0 = keyblock[KB_POLICY]; k1 = keyblock[KB_POLICY + 1]; }
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = probs[i];
    float pa;
    const uint32_t a = sample_action(p.x, p.y, p.z, p.w, mask ? (uint32_t)mask[i] : 15u, rng_draw(k0, k1, id_base + i, 0u), pa);
    actions[i] = (uint8_t)a;
    prob_out[i] = pa;
}

// ---------------------------------------------------------- simulate_move -----
// 32 lanes per board (at most 15 empty cells x 2 tiles = 30 successors), lane k builds successor k.
__global__ __launch_bounds__(kBlock) void simulate_kernel(const uint4 *__restrict__ boards, const uint8_t *__restrict__ actions,
                                                         const uint8_t *__restrict__ highest_code, uint4 *__restrict__ succ,
                                                         double *__restrict__ reward, uint8_t *__restrict__ done,
                                                         uint8_t *__restrict__ count, size_t n)
{
    const size_t gidx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t b = gidx >> 5;
    const uint32_t k = (uint32_t)gidx & 31u;
    if (b >= n) return;
    const Board state = load_board(boards, b);
    Board moved;
    uint32_t gain;
    const uint32_t nsucc = simulate_count(state, actions[b] & 3u, moved, gain);
    const uint32_t hc = highest_code ? (uint32_t)highest_code[b] : max_code(state);
    SimOut o;
    o.board = Board{{0u, 0u, 0u, 0u}}; o.reward = 0.0; o.done = false;
    if (k < nsucc) o = simulate_successor(state, moved, gain, k, hc);
    store_board(succ, gidx, o.board);
    reward[gidx] = o.reward;
    done[gidx] = o.done ? 1 : 0;
    if (k == 0) count[b] = (uint8_t)nsucc;
}

// the hybrid agent's sampled variant (agents/hybrid.py:578-629): 8 lanes per state, lane k < count builds successor k
__global__ __launch_bounds__(kBlock) void simulate_sampled_kernel(const uint4 *__restrict__ boards, const uint8_t *__restrict__ actions,
                                                                 uint4 *__restrict__ succ, double *__restrict__ reward,
                                                                 uint8_t *__restrict__ done, uint8_t *__restrict__ count,
                                                                 uint32_t k0, uint32_t k1, uint64_t id_base, size_t n)
{
    const size_t gidx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t b = gidx >> 3;
    const uint32_t k = (uint32_t)gidx & 7u;
    if (b >= n) return;
    const Board state = load_board(boards, b);
    uint32_t gain;
    const Board moved = move_env(state, actions[b] & 3u, gain);
    const uint32_t n_empty = count_empty(moved);
    const bool valid = !same(moved, state);
    const uint32_t picks = n_empty < 3u ? n_empty : 3u;
    const uint32_t nsucc = (!valid || n_empty == 0u) ? 1u : 2u * picks;
    Board ob = {{0u, 0u, 0u, 0u}};
    double orw = 0.0;
    uint32_t od = 0u;
    if (!valid) { if (k == 0u) { ob = moved; orw = -1.0; } }                       // :601-603
    else if (n_empty == 0u) { if (k == 0u) { ob = moved; od = 1u; } }            // :606-609 (a changed board always has an empty cell)
    else if (k < nsucc) {
        const uint64_t id = id_base + b;
        const SampledOut o = simulate_sampled_successor(state, moved, k, n_empty, rng_draw(k0, k1, id, 0u), rng_draw(k0, k1, id, 1u),
                                                        rng_draw(k0, k1, id, 2u));
        ob = o.board; orw = o.reward;
    }
    store_board(succ, gidx, ob);
    reward[gidx] = orw;
    done[gidx] = (uint8_t)od;
    if (k == 0u) count[b] = (uint8_t)nsucc;
}

// ------------------------------------------------------------------- eval -----
template <int KIND>
__global__ __launch_bounds__(kBlock) void eval_kernel(const uint4 *__restrict__ boards, const uint8_t *__restrict__ phase,
                                                     double *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Board b = load_board(boards, i);
    double v;
    if (KIND == G2048_EVAL_FAST) v = eval_fast(b);
    else if (KIND == G2048_EVAL_FULL) v = eval_full(b, phase ? (uint32_t)phase[i] : phase_of(max_code(b), 512u, 1024u));
    else if (KIND == G2048_EVAL_PPO_HEURISTIC) v = eval_ppo_heuristic(b);
    else if (KIND == G2048_EVAL_PPO_SHAPING) v = eval_ppo_shaping(b, 0.0);
    else if (KIND == G2048_EVAL_PATTERN) v = eval_pattern(b);
    else if (KIND == G2048_EVAL_CORNER_BONUS) v = (double)max_corner_code(b) * 2.0;          // log2(max corner) * 2.0, 0 if none
    else if (KIND == G2048_EVAL_MERGE_POTENTIAL) v = (double)merge_potential(b);            // a sum of small integers: exact in any order
    else v = eval_monotonicity(b, KIND - G2048_EVAL_MONO_PP);
    out[i] = v;
}

// -------------------------------------------------- format kernels (row/lane) -
// PPOAgent.normalize_state (agents/ppo_agent.py:184-195): float32(code) / float32(15)
__global__ __launch_bounds__(kBlock) void obs_kernel(const uint32_t *__restrict__ rows, float4 *__restrict__ obs, size_t n_rows)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t x = rows[i];
    obs[i] = make_float4((float)(x & 0xffu) / 15.0f, (float)((x >> 8) & 0xffu) / 15.0f,
                         (float)((x >> 16) & 0xffu) / 15.0f, (float)(x >> 24) / 15.0f);
}

// the same observation rounded once more, to f16 or bf16 (round to nearest even of the f32 quotient): 8 bytes per row
template <bool BF16>
__global__ __launch_bounds__(kBlock) void obs16_kernel(const uint32_t *__restrict__ rows, uint2 *__restrict__ obs, size_t n_rows)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t x = rows[i];
    uint32_t h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float v = (float)((x >> (8 * k)) & 0xffu) / 15.0f;
        if (BF16) {
            const uint32_t u = __float_as_uint(v);
            h[k] = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;          // v is a finite non-negative number: RNE by integer add
        } else {
            h[k] = (uint32_t)__half_as_ushort(__float2half_rn(v));
        }
    }
    obs[i] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
}

__global__ __launch_bounds__(kBlock) void pack_kernel(const int4 *__restrict__ tiles, uint32_t *__restrict__ rows, size_t n_rows)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_rows) return;
    const int4 t = tiles[i];
    auto code = [](int v) -> uint32_t { return v > 0 ? 31u - (uint32_t)__builtin_clz((uint32_t)v) : 0u; };
    rows[i] = code(t.x) | (code(t.y) << 8) | (code(t.z) << 16) | (code(t.w) << 24);
}

__global__ __launch_bounds__(kBlock) void unpack_kernel(const uint32_t *__restrict__ rows, int4 *__restrict__ tiles, size_t n_rows)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t x = rows[i];
    auto tile = [](uint32_t c) -> int { return c ? (int)(1u << c) : 0; };
    tiles[i] = make_int4(tile(x & 0xffu), tile((x >> 8) & 0xffu), tile((x >> 16) & 0xffu), tile(x >> 24));
}

// ------------------------------------------------------------- synthetic ------
__global__ __launch_bounds__(kBlock) void synth_boards_kernel(uint4 *__restrict__ boards, uint32_t k0, uint32_t k1,
                                                             uint64_t id_base, size_t n, uint32_t p_empty, uint32_t max_code_)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Board b = {{0u, 0u, 0u, 0u}};
#pragma unroll
    for (uint32_t cell = 0; cell < 16; ++cell) {
        const uint32_t h = rng_draw(k0, k1, id_base + i, cell);
        const uint32_t c = (h >> 16) < p_empty ? 0u : 1u + (((h & 0xffffu) * max_code_) >> 16);
        b.w[cell >> 2] |= c << (8 * (cell & 3));
    }
    if ((b.w[0] | b.w[1] | b.w[2] | b.w[3]) == 0u) b.w[0] = 1u;
    store_board(boards, i, b);
}

__global__ __launch_bounds__(kBlock) void synth_actions_kernel(uint8_t *__restrict__ actions, uint32_t k0, uint32_t k1,
                                                              uint64_t id_base, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    actions[i] = (uint8_t)(rng_draw(k0, k1, id_base + i, 0u) >> 30);
}

// --------------------------------------------------------------- metrics ------
__global__ __launch_bounds__(kBlock) void metrics_kernel(const uint4 *__restrict__ boards, const uint32_t *__restrict__ score,
                                                        const uint8_t *__restrict__ flags, const uint32_t *__restrict__ expanded,
                                                        unsigned long long *__restrict__ out, size_t n)
{
    __shared__ unsigned long long acc[24];
    if (threadIdx.x < 24) acc[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long cnt = 0, ssum = 0, dsum = 0, esum = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const Board b = load_board(boards, i);
        cnt += 1ull;
        ssum += score ? score[i] : 0u;
        dsum += flags ? (flags[i] & G2048_FLAG_DONE) : 0u;
        esum += expanded ? expanded[i] : 0u;
        atomicAdd(&acc[4 + max_code(b)], 1ull);
    }
    // wave reduction of the four scalars, then one LDS atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off); ssum += __shfl_down(ssum, off);
        dsum += __shfl_down(dsum, off); esum += __shfl_down(esum, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], cnt); atomicAdd(&acc[1], ssum); atomicAdd(&acc[2], dsum); atomicAdd(&acc[3], esum);
    }
    __syncthreads();
    if (threadIdx.x < 22 && acc[threadIdx.x]) atomicAdd(&out[threadIdx.x], acc[threadIdx.x]);
}

// ------------------------------------------------------------ key block -------
// One thread: keys of every per-move domain at index *counter, then *counter += 1. Put in front of the *_dyn
// kernels of one move, it makes that move's kernel sequence replayable from a captured hipGraph.
__global__ void keys_advance_kernel(uint32_t *keyblock, unsigned long long *counter, uint64_t seed)
{
    const uint64_t idx = *counter;
    const Keys s = rng_keys(seed, DOM_STEP, idx), e = rng_keys(seed, DOM_EPISODE, idx);
    const Keys b = rng_keys(seed, DOM_BEAM, idx), p = rng_keys(seed, DOM_POLICY, idx);
    keyblock[KB_STEP] = s.k0; keyblock[KB_STEP + 1] = s.k1; keyblock[KB_EPISODE] = e.k0; keyblock[KB_EPISODE + 1] = e.k1;
    keyblock[KB_BEAM] = b.k0; keyblock[KB_BEAM + 1] = b.k1; keyblock[KB_POLICY] = p.k0; keyblock[KB_POLICY + 1] = p.k1;
    keyblock[KB_INDEX] = (uint32_t)idx; keyblock[KB_INDEX + 1] = (uint32_t)(idx >> 32);
    *counter = idx + 1ull;
}

// -------------------------------------------------------------- self-test -----
__global__ void selftest_kernel(uint32_t *result, uint32_t a, uint32_t b, double x, double y, double z)
{
    uint32_t bad = 0;
    // v_perm_b32: selector 0..3 -> bytes of the SECOND operand, 4..7 -> bytes of the FIRST, 0x0c -> 0
    if (perm(a, b, 0x07040300u) != (((a >> 24) << 24) | ((a & 0xffu) << 16) | ((b >> 24) << 8) | (b & 0xffu))) bad |= 1u;
    if (perm(a, b, 0x0c0c0c05u) != ((a >> 8) & 0xffu)) bad |= 2u;
    if (dot4(a, b, 7u) != 7u + (a & 0xffu) * (b & 0xffu) + ((a >> 8) & 0xffu) * ((b >> 8) & 0xffu) +
                              ((a >> 16) & 0xffu) * ((b >> 16) & 0xffu) + (a >> 24) * (b >> 24)) bad |= 4u;
    // contraction must be off: x*y is rounded before the add
    if (x * y + z != 0.0) bad |= 8u;
    // transpose / rot180 of a counting board
    const Board c = {{0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu}};
    const Board t = transpose(c);
    if (t.w[0] != 0x0c080400u || t.w[1] != 0x0d090501u || t.w[2] != 0x0e0a0602u || t.w[3] != 0x0f0b0703u) bad |= 16u;
    const Board r = rot180(c);
    if (r.w[0] != 0x0c0d0e0fu || r.w[3] != 0x00010203u) bad |= 32u;
    *result = bad;
}

}  // namespace

// ==================================================================== C-ABI ====
extern "C" {

const char *g2048_last_error(void) { return g_err; }
// internal: lets the other translation units of this library report through the same string
void g2048_set_last_error_(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
int g2048_abi_version(void) { return G2048_ABI_VERSION; }
unsigned g2048_build_flags(void) { return kInstrument; }

int g2048_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

static int step_impl(const void *boards_in, const uint8_t *actions, void *boards_out, uint32_t *score_inout,
                     void *reward_out, uint8_t *flags_out, uint64_t seed, uint64_t step_index,
                     uint64_t board_id_base, size_t n, uint32_t opts, void *stream, const uint32_t *keyblock)
{
    if (n == 0) return G2048_OK;
    const bool random_actions = (opts & G2048_STEP_RANDOM_ACTIONS) != 0u;
    if (!boards_in || (!actions && !random_actions) || !boards_out || !score_inout || !reward_out || !flags_out)
        return fail(G2048_ERR_ARG, "g2048_step: null pointer");
    {   // The kernel hashes the low word of a board id and takes the high word's term as a launch constant: a range of ids that
        // crosses a multiple of 2^32 becomes two launches, cut there (every per-board array is offset by the same number of boards).
        const uint64_t to_wrap = 0x100000000ull - (board_id_base & 0xffffffffull);
        if ((uint64_t)n > to_wrap) {
            const size_t n1 = (size_t)to_wrap, rb = (opts & G2048_STEP_REWARD_F64) ? 8u : 4u;
            const int rc = step_impl(boards_in, actions, boards_out, score_inout, reward_out, flags_out, seed, step_index, board_id_base, n1,
                                     opts, stream, keyblock);
            if (rc != G2048_OK) return rc;
            return step_impl(static_cast<const char *>(boards_in) + 16u * n1, actions ? actions + n1 : nullptr,
                             static_cast<char *>(boards_out) + 16u * n1, score_inout + n1, static_cast<char *>(reward_out) + rb * n1,
                             flags_out + n1, seed, step_index, board_id_base + n1, n - n1, opts, stream, keyblock);
        }
    }
    if (random_actions && keyblock) return fail(G2048_ERR_ARG, "g2048_step_dyn: RANDOM_ACTIONS needs the scalar form");
    if (!aligned16(boards_in) || !aligned16(boards_out)) return fail(G2048_ERR_ARG, "g2048_step: board arrays must be 16-byte aligned");
    if (!aligned4(score_inout) || !aligned4(reward_out) || ((opts & G2048_STEP_REWARD_F64) && (reinterpret_cast<uintptr_t>(reward_out) & 7u)))
        return fail(G2048_ERR_ARG, "g2048_step: score/reward arrays misaligned");
    if (opts & ~(G2048_STEP_REWARD_F64 | G2048_STEP_AUTO_RESET | G2048_STEP_RANDOM_ACTIONS | G2048_STEP_NOOP_ACTIONS | (3u << G2048_STEP_TUNE_SHIFT))) return fail(G2048_ERR_ARG, "g2048_step: unknown opts 0x%x", opts);
    const Keys k = rng_keys(seed, DOM_STEP, step_index), e = rng_keys(seed, DOM_EPISODE, step_index);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint4 *in = static_cast<const uint4 *>(boards_in);
    uint4 *out = static_cast<uint4 *>(boards_out);
    const bool f64 = opts & G2048_STEP_REWARD_F64, ar = opts & G2048_STEP_AUTO_RESET;
    const unsigned tune = (opts >> G2048_STEP_TUNE_SHIFT) & 3u;            // 0 = default
    if (tune == 3u) return fail(G2048_ERR_ARG, "g2048_step: tune 3 (four boards per lane) was removed: measured slower at every size");
    // default: one board per lane -- measured fastest up to a few Mi boards per launch, where the launch is short and
    // wave-level parallelism hides the load latency; from 4 Mi boards on (beyond the Infinity Cache) two boards per lane,
    // both loads in flight before the first is computed, stream 4-5 % faster (profiles/r02_step_tune.txt)
    const int per_lane = tune == 1 ? 1 : tune == 2 ? 2 : (n >= ((size_t)1 << 22) ? 2 : kStepBoardsPerLane);
#define G2048_LAUNCH_STEP(F, A, BB) \
    hipLaunchKernelGGL((step_kernel<F, A, BB, kBlock>), dim3(blocks_for(n, kBlock * BB)), dim3(kBlock), 0, s, n, keyblock, in, actions, \
                       score_inout, (uint32_t)board_id_base, rng_hi_term(board_id_base), k.k0, k.k1, out, reward_out, flags_out, e.k0, e.k1)
#define G2048_LAUNCH_STEP_B(F, A) \
    do { if (per_lane == 1) G2048_LAUNCH_STEP(F, A, 1); else G2048_LAUNCH_STEP(F, A, 2); } while (0)
    if (opts & G2048_STEP_NOOP_ACTIONS) {            // reference semantics for action values outside 0..3 (drop-in class)
        if (random_actions) return fail(G2048_ERR_ARG, "g2048_step: NOOP_ACTIONS needs explicit actions");
#define G2048_LAUNCH_NOOP(F, A) hipLaunchKernelGGL((step_kernel<F, A, 1, kBlock, false, true>), dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, s, \
                           n, keyblock, in, actions, score_inout, (uint32_t)board_id_base, rng_hi_term(board_id_base), k.k0, k.k1, out, reward_out, flags_out, e.k0, e.k1)
        if (f64 && ar) G2048_LAUNCH_NOOP(true, true); else if (f64) G2048_LAUNCH_NOOP(true, false);
        else if (ar) G2048_LAUNCH_NOOP(false, true); else G2048_LAUNCH_NOOP(false, false);
#undef G2048_LAUNCH_NOOP
        return check_launch("g2048_step");
    }
    if (random_actions) {           // uniform actions drawn in the kernel: (seed, SYNTH_ACTION, step_index, board id) >> 30
        const Keys ak = rng_keys(seed, DOM_SYNTH_ACTION, step_index);
#define G2048_LAUNCH_RANDOM(F, A) \
        hipLaunchKernelGGL((step_kernel<F, A, 1, kBlock, true>), dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, s, n, keyblock, in, actions, \
                           score_inout, (uint32_t)board_id_base, rng_hi_term(board_id_base), k.k0, k.k1, out, reward_out, flags_out, e.k0, e.k1, ak.k0, ak.k1)
        if (f64 && ar) G2048_LAUNCH_RANDOM(true, true);
        else if (f64) G2048_LAUNCH_RANDOM(true, false);
        else if (ar) G2048_LAUNCH_RANDOM(false, true);
        else G2048_LAUNCH_RANDOM(false, false);
#undef G2048_LAUNCH_RANDOM
        return check_launch("g2048_step");
    }
    if (f64 && ar) G2048_LAUNCH_STEP_B(true, true);
    else if (f64) G2048_LAUNCH_STEP_B(true, false);
    else if (ar) G2048_LAUNCH_STEP_B(false, true);
    else G2048_LAUNCH_STEP_B(false, false);
#undef G2048_LAUNCH_STEP_B
#undef G2048_LAUNCH_STEP
    return check_launch("g2048_step");
}

int g2048_step(const void *boards_in, const uint8_t *actions, void *boards_out, uint32_t *score_inout,
               void *reward_out, uint8_t *flags_out, uint64_t seed, uint64_t step_index,
               uint64_t board_id_base, size_t n, uint32_t opts, void *stream)
{
    return step_impl(boards_in, actions, boards_out, score_inout, reward_out, flags_out, seed, step_index, board_id_base, n,
                     opts, stream, nullptr);
}

int g2048_step_dyn(const void *boards_in, const uint8_t *actions, void *boards_out, uint32_t *score_inout,
                   void *reward_out, uint8_t *flags_out, const uint32_t *keyblock, uint64_t board_id_base, size_t n,
                   uint32_t opts, void *stream)
{
    if (!keyblock) return fail(G2048_ERR_ARG, "g2048_step_dyn: null key block");
    return step_impl(boards_in, actions, boards_out, score_inout, reward_out, flags_out, 0, 0, board_id_base, n, opts, stream,
                     keyblock);
}

int g2048_step_many(const void *boards_in, const uint8_t *actions_stream_or_null, void *boards_out, uint32_t *score_inout,
                    void *reward_stream_out_or_null, uint8_t *flags_stream_out_or_null, uint8_t *flags_last_out,
                    uint32_t *episodes_out_or_null, uint64_t seed, uint64_t step_index0, uint32_t steps,
                    uint64_t board_id_base, size_t n, uint32_t opts, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards_in || !boards_out || !score_inout || !flags_last_out) return fail(G2048_ERR_ARG, "g2048_step_many: null pointer");
    if (!aligned16(boards_in) || !aligned16(boards_out)) return fail(G2048_ERR_ARG, "g2048_step_many: board arrays must be 16-byte aligned");
    const bool f64 = (opts & G2048_STEP_REWARD_F64) != 0u, ar = (opts & G2048_STEP_AUTO_RESET) != 0u;
    const bool random_actions = (opts & G2048_STEP_RANDOM_ACTIONS) != 0u;
    if (!aligned4(score_inout) || (episodes_out_or_null && !aligned4(episodes_out_or_null)) ||
        (reward_stream_out_or_null && (reinterpret_cast<uintptr_t>(reward_stream_out_or_null) & (f64 ? 7u : 3u))))
        return fail(G2048_ERR_ARG, "g2048_step_many: score / reward / episode arrays misaligned");
    if (!random_actions && !actions_stream_or_null)
        return fail(G2048_ERR_ARG, "g2048_step_many: no policy: pass an actions stream or set G2048_STEP_RANDOM_ACTIONS");
    if (opts & ~(G2048_STEP_REWARD_F64 | G2048_STEP_AUTO_RESET | G2048_STEP_RANDOM_ACTIONS)) return fail(G2048_ERR_ARG, "g2048_step_many: unknown opts 0x%x", opts);
    if (steps == 0) return fail(G2048_ERR_ARG, "g2048_step_many: steps must be at least 1");
    hipStream_t s = static_cast<hipStream_t>(stream);
#define G2048_LAUNCH_MANY(F, A, R) hipLaunchKernelGGL((step_many_kernel<F, A, kBlock, R>), dim3(blocks_for(n)), dim3(kBlock), 0, s, \
                           static_cast<const uint4 *>(boards_in), actions_stream_or_null, static_cast<uint4 *>(boards_out), score_inout, \
                           reward_stream_out_or_null, flags_stream_out_or_null, flags_last_out, episodes_out_or_null, seed, step_index0, \
                           steps, board_id_base, n)
#define G2048_LAUNCH_MANY_R(F, A) do { if (random_actions) G2048_LAUNCH_MANY(F, A, true); else G2048_LAUNCH_MANY(F, A, false); } while (0)
    if (f64 && ar) G2048_LAUNCH_MANY_R(true, true);
    else if (f64) G2048_LAUNCH_MANY_R(true, false);
    else if (ar) G2048_LAUNCH_MANY_R(false, true);
    else G2048_LAUNCH_MANY_R(false, false);
#undef G2048_LAUNCH_MANY_R
#undef G2048_LAUNCH_MANY
    return check_launch("g2048_step_many");
}

int g2048_env_step(void *board_inout, uint32_t *score_inout, uint32_t action, uint32_t op, void *record_out, uint64_t seed,
                   uint64_t index, uint64_t board_id, void *stream)
{
    if (!board_inout || !score_inout || !record_out) return fail(G2048_ERR_ARG, "g2048_env_step: null pointer");
    if (!aligned16(board_inout) || !aligned4(score_inout) || !aligned16(record_out)) return fail(G2048_ERR_ARG, "g2048_env_step: misaligned pointer");
    const uint32_t token = (op >> G2048_ENV_TOKEN_SHIFT) & 0xffffu;
    if (op >> (G2048_ENV_TOKEN_SHIFT + 16)) return fail(G2048_ERR_ARG, "g2048_env_step: unknown op bits 0x%x", op);
    op &= (1u << G2048_ENV_TOKEN_SHIFT) - 1u;
    if (op > G2048_ENV_OP_MOVE_AGENT) return fail(G2048_ERR_ARG, "g2048_env_step: unknown op %u", op);
    const Keys k = rng_keys(seed, op == G2048_ENV_OP_RESET ? DOM_RESET : DOM_STEP, index);
    hipLaunchKernelGGL(env_step_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<uint4 *>(board_inout),
                       score_inout, action, op, static_cast<EnvRecord *>(record_out), k.k0, k.k1, board_id, token);
    return check_launch("g2048_env_step");
}

int g2048_replay_games(const void *boards0, const uint32_t *score0_or_null, const uint64_t *game_ids_or_null, uint64_t game_id_base,
                       const uint8_t *actions, size_t actions_stride, const int32_t *n_moves, void *boards_hist_out,
                       uint32_t *score_hist_out_or_null, uint8_t *flags_hist_out_or_null, size_t hist_stride, uint64_t seed,
                       size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards0 || !actions || !n_moves || !boards_hist_out) return fail(G2048_ERR_ARG, "g2048_replay_games: null pointer");
    if (!aligned16(boards0) || !aligned16(boards_hist_out) || !aligned4(n_moves) || (score0_or_null && !aligned4(score0_or_null)) ||
        (score_hist_out_or_null && !aligned4(score_hist_out_or_null)) || (game_ids_or_null && (reinterpret_cast<uintptr_t>(game_ids_or_null) & 7u)))
        return fail(G2048_ERR_ARG, "g2048_replay_games: misaligned array");
    if (hist_stride == 0) return fail(G2048_ERR_ARG, "g2048_replay_games: hist_stride must be at least 1 (max moves + 1)");
    if (actions_stride == 0) return fail(G2048_ERR_ARG, "g2048_replay_games: actions_stride must be at least 1 (bytes per game row)");
    hipLaunchKernelGGL(replay_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(boards0), score0_or_null, reinterpret_cast<const unsigned long long *>(game_ids_or_null),
                       game_id_base, actions, actions_stride, n_moves, static_cast<uint4 *>(boards_hist_out), score_hist_out_or_null,
                       flags_hist_out_or_null, hist_stride, seed, n);
    return check_launch("g2048_replay_games");
}

int g2048_reset(void *boards_out, uint32_t *score_out, uint64_t seed, uint64_t epoch, uint64_t board_id_base,
                size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards_out) return fail(G2048_ERR_ARG, "g2048_reset: null pointer");
    if (!aligned16(boards_out)) return fail(G2048_ERR_ARG, "g2048_reset: board array must be 16-byte aligned");
    const Keys k = rng_keys(seed, DOM_RESET, epoch);
    hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint4 *>(boards_out), score_out, k.k0, k.k1, board_id_base, n);
    return check_launch("g2048_reset");
}

int g2048_valid_moves(const void *boards, uint8_t *mask4_out, size_t n, uint32_t opts, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards || !mask4_out) return fail(G2048_ERR_ARG, "g2048_valid_moves: null pointer");
    if (!aligned16(boards)) return fail(G2048_ERR_ARG, "g2048_valid_moves: board array must be 16-byte aligned");
    if (opts > G2048_VALID_AGENT) return fail(G2048_ERR_ARG, "g2048_valid_moves: unknown opts 0x%x", opts);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint4 *b = static_cast<const uint4 *>(boards);
    if (opts == G2048_VALID_AGENT) hipLaunchKernelGGL(valid_kernel<true>, dim3(blocks_for(n)), dim3(kBlock), 0, s, b, mask4_out, n);
    else hipLaunchKernelGGL(valid_kernel<false>, dim3(blocks_for(n)), dim3(kBlock), 0, s, b, mask4_out, n);
    return check_launch("g2048_valid_moves");
}

static int track_impl(const uint8_t *flags, const uint32_t *expanded_or_null, uint8_t *alive_inout, int32_t *moves_inout,
                      int32_t *valid_inout, int32_t *invalid_inout, int32_t *milestone_move_inout,
                      unsigned long long *expanded_sum_inout_or_null, int32_t move_index, size_t n, void *stream,
                      const uint32_t *keyblock)
{
    if (n == 0) return G2048_OK;
    if (!flags || !alive_inout || !moves_inout || !valid_inout || !invalid_inout || !milestone_move_inout)
        return fail(G2048_ERR_ARG, "g2048_track_episodes: null pointer");
    if (!aligned16(milestone_move_inout)) return fail(G2048_ERR_ARG, "g2048_track_episodes: milestone array must be 16-byte aligned");
    hipLaunchKernelGGL(track_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), flags,
                       expanded_or_null, alive_inout, moves_inout, valid_inout, invalid_inout,
                       reinterpret_cast<int4 *>(milestone_move_inout), expanded_sum_inout_or_null, move_index, n, keyblock);
    return check_launch("g2048_track_episodes");
}

int g2048_track_episodes(const uint8_t *flags, const uint32_t *expanded_or_null, uint8_t *alive_inout, int32_t *moves_inout,
                         int32_t *valid_inout, int32_t *invalid_inout, int32_t *milestone_move_inout,
                         unsigned long long *expanded_sum_inout_or_null, int32_t move_index, size_t n, void *stream)
{
    return track_impl(flags, expanded_or_null, alive_inout, moves_inout, valid_inout, invalid_inout, milestone_move_inout,
                      expanded_sum_inout_or_null, move_index, n, stream, nullptr);
}

int g2048_track_episodes_dyn(const uint8_t *flags, const uint32_t *expanded_or_null, uint8_t *alive_inout, int32_t *moves_inout,
                             int32_t *valid_inout, int32_t *invalid_inout, int32_t *milestone_move_inout,
                             unsigned long long *expanded_sum_inout_or_null, const uint32_t *keyblock, size_t n, void *stream)
{
    if (!keyblock) return fail(G2048_ERR_ARG, "g2048_track_episodes_dyn: null key block");
    return track_impl(flags, expanded_or_null, alive_inout, moves_inout, valid_inout, invalid_inout, milestone_move_inout,
                      expanded_sum_inout_or_null, 0, n, stream, keyblock);
}

static int sample_impl(const float *probs, const uint8_t *mask4_or_null, uint8_t *actions_out, float *prob_out,
                       uint64_t seed, uint64_t step_index, uint64_t env_id_base, size_t n, void *stream, const uint32_t *keyblock)
{
    if (n == 0) return G2048_OK;
    if (!probs || !actions_out || !prob_out) return fail(G2048_ERR_ARG, "g2048_sample_actions: null pointer");
    if (!aligned16(probs) || !aligned4(prob_out)) return fail(G2048_ERR_ARG, "g2048_sample_actions: misaligned array");
    const Keys k = rng_keys(seed, DOM_POLICY, step_index);
    hipLaunchKernelGGL(sample_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4 *>(probs), mask4_or_null, actions_out, prob_out, k.k0, k.k1, env_id_base, n,
                       keyblock);
    return check_launch("g2048_sample_actions");
}

int g2048_sample_actions(const float *probs, const uint8_t *mask4_or_null, uint8_t *actions_out, float *prob_out,
                         uint64_t seed, uint64_t step_index, uint64_t env_id_base, size_t n, void *stream)
{
    return sample_impl(probs, mask4_or_null, actions_out, prob_out, seed, step_index, env_id_base, n, stream, nullptr);
}

int g2048_sample_actions_dyn(const float *probs, const uint8_t *mask4_or_null, uint8_t *actions_out, float *prob_out,
                             const uint32_t *keyblock, uint64_t env_id_base, size_t n, void *stream)
{
    if (!keyblock) return fail(G2048_ERR_ARG, "g2048_sample_actions_dyn: null key block");
    return sample_impl(probs, mask4_or_null, actions_out, prob_out, 0, 0, env_id_base, n, stream, keyblock);
}

int g2048_simulate_move(const void *boards, const uint8_t *actions, const uint8_t *highest_code_or_null, void *succ_boards_out,
                        double *reward_out, uint8_t *done_out, uint8_t *count_out, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards || !actions || !succ_boards_out || !reward_out || !done_out || !count_out)
        return fail(G2048_ERR_ARG, "g2048_simulate_move: null pointer");
    if (!aligned16(boards) || !aligned16(succ_boards_out) || (reinterpret_cast<uintptr_t>(reward_out) & 7u))
        return fail(G2048_ERR_ARG, "g2048_simulate_move: misaligned array");
    hipLaunchKernelGGL(simulate_kernel, dim3(blocks_for(n * 32)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(boards), actions, highest_code_or_null, static_cast<uint4 *>(succ_boards_out),
                       reward_out, done_out, count_out, n);
    return check_launch("g2048_simulate_move");
}

int g2048_simulate_move_sampled(const void *boards, const uint8_t *actions, void *succ_boards_out, double *reward_out,
                                uint8_t *done_out, uint8_t *count_out, uint64_t seed, uint64_t step_index, uint64_t state_id_base,
                                size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards || !actions || !succ_boards_out || !reward_out || !done_out || !count_out)
        return fail(G2048_ERR_ARG, "g2048_simulate_move_sampled: null pointer");
    if (!aligned16(boards) || !aligned16(succ_boards_out) || (reinterpret_cast<uintptr_t>(reward_out) & 7u))
        return fail(G2048_ERR_ARG, "g2048_simulate_move_sampled: misaligned array");
    const Keys k = rng_keys(seed, DOM_SIMULATE, step_index);
    hipLaunchKernelGGL(simulate_sampled_kernel, dim3(blocks_for(n * 8)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(boards), actions, static_cast<uint4 *>(succ_boards_out), reward_out, done_out,
                       count_out, k.k0, k.k1, state_id_base, n);
    return check_launch("g2048_simulate_move_sampled");
}

int g2048_eval(const void *boards, int kind, const uint8_t *phase_or_null, double *out, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards || !out) return fail(G2048_ERR_ARG, "g2048_eval: null pointer");
    if (!aligned16(boards) || (reinterpret_cast<uintptr_t>(out) & 7u)) return fail(G2048_ERR_ARG, "g2048_eval: misaligned array");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint4 *b = static_cast<const uint4 *>(boards);
    const dim3 grid(blocks_for(n)), block(kBlock);
    switch (kind) {
#define G2048_EVAL_CASE(K) case K: hipLaunchKernelGGL(eval_kernel<K>, grid, block, 0, s, b, phase_or_null, out, n); break;
        G2048_EVAL_CASE(G2048_EVAL_FAST)
        G2048_EVAL_CASE(G2048_EVAL_FULL)
        G2048_EVAL_CASE(G2048_EVAL_PPO_HEURISTIC)
        G2048_EVAL_CASE(G2048_EVAL_MONO_PP)
        G2048_EVAL_CASE(G2048_EVAL_MONO_PM)
        G2048_EVAL_CASE(G2048_EVAL_MONO_MP)
        G2048_EVAL_CASE(G2048_EVAL_MONO_MM)
        G2048_EVAL_CASE(G2048_EVAL_PPO_SHAPING)
        G2048_EVAL_CASE(G2048_EVAL_PATTERN)
        G2048_EVAL_CASE(G2048_EVAL_CORNER_BONUS)
        G2048_EVAL_CASE(G2048_EVAL_MERGE_POTENTIAL)
#undef G2048_EVAL_CASE
        default: return fail(G2048_ERR_ARG, "g2048_eval: unknown kind %d", kind);
    }
    return check_launch("g2048_eval");
}

int g2048_obs_f32(const void *boards, float *obs_out, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards || !obs_out) return fail(G2048_ERR_ARG, "g2048_obs_f32: null pointer");
    if (!aligned16(boards) || !aligned16(obs_out)) return fail(G2048_ERR_ARG, "g2048_obs_f32: arrays must be 16-byte aligned");
    hipLaunchKernelGGL(obs_kernel, dim3(blocks_for(n * 4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint32_t *>(boards), reinterpret_cast<float4 *>(obs_out), n * 4);
    return check_launch("g2048_obs_f32");
}

int g2048_obs_16(const void *boards, void *obs_out, int bf16, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards || !obs_out) return fail(G2048_ERR_ARG, "g2048_obs_16: null pointer");
    if (!aligned16(boards) || (reinterpret_cast<uintptr_t>(obs_out) & 7u)) return fail(G2048_ERR_ARG, "g2048_obs_16: misaligned array");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bf16) hipLaunchKernelGGL(obs16_kernel<true>, dim3(blocks_for(n * 4)), dim3(kBlock), 0, s, static_cast<const uint32_t *>(boards),
                                 static_cast<uint2 *>(obs_out), n * 4);
    else hipLaunchKernelGGL(obs16_kernel<false>, dim3(blocks_for(n * 4)), dim3(kBlock), 0, s, static_cast<const uint32_t *>(boards),
                            static_cast<uint2 *>(obs_out), n * 4);
    return check_launch("g2048_obs_16");
}

int g2048_pack_i32(const int32_t *tiles, void *boards_out, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!tiles || !boards_out) return fail(G2048_ERR_ARG, "g2048_pack_i32: null pointer");
    if (!aligned16(tiles) || !aligned16(boards_out)) return fail(G2048_ERR_ARG, "g2048_pack_i32: arrays must be 16-byte aligned");
    hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(n * 4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const int4 *>(tiles), static_cast<uint32_t *>(boards_out), n * 4);
    return check_launch("g2048_pack_i32");
}

int g2048_unpack_i32(const void *boards, int32_t *tiles_out, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards || !tiles_out) return fail(G2048_ERR_ARG, "g2048_unpack_i32: null pointer");
    if (!aligned16(boards) || !aligned16(tiles_out)) return fail(G2048_ERR_ARG, "g2048_unpack_i32: arrays must be 16-byte aligned");
    hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for(n * 4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint32_t *>(boards), reinterpret_cast<int4 *>(tiles_out), n * 4);
    return check_launch("g2048_unpack_i32");
}

int g2048_synth_boards(void *boards_out, uint64_t seed, uint64_t board_id_base, size_t n, uint32_t p_empty_u16,
                       uint32_t max_code_, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards_out) return fail(G2048_ERR_ARG, "g2048_synth_boards: null pointer");
    if (!aligned16(boards_out)) return fail(G2048_ERR_ARG, "g2048_synth_boards: board array must be 16-byte aligned");
    if (max_code_ < 1 || max_code_ > 17 || p_empty_u16 > 65536) return fail(G2048_ERR_ARG, "g2048_synth_boards: bad distribution");
    const Keys k = rng_keys(seed, DOM_SYNTH_BOARD, 0);
    hipLaunchKernelGGL(synth_boards_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint4 *>(boards_out), k.k0, k.k1, board_id_base, n, p_empty_u16, max_code_);
    return check_launch("g2048_synth_boards");
}

int g2048_synth_actions(uint8_t *actions_out, uint64_t seed, uint64_t step_index, uint64_t board_id_base, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!actions_out) return fail(G2048_ERR_ARG, "g2048_synth_actions: null pointer");
    const Keys k = rng_keys(seed, DOM_SYNTH_ACTION, step_index);
    hipLaunchKernelGGL(synth_actions_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       actions_out, k.k0, k.k1, board_id_base, n);
    return check_launch("g2048_synth_actions");
}

int g2048_metrics(const void *boards, const uint32_t *score, const uint8_t *flags_or_null, const uint32_t *expanded_or_null,
                  unsigned long long *out24, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards || !out24) return fail(G2048_ERR_ARG, "g2048_metrics: null pointer");
    if (!aligned16(boards)) return fail(G2048_ERR_ARG, "g2048_metrics: board array must be 16-byte aligned");
    unsigned grid = blocks_for(n);
    if (grid > 2048u) grid = 2048u;
    hipLaunchKernelGGL(metrics_kernel, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(boards), score, flags_or_null, expanded_or_null, out24, n);
    return check_launch("g2048_metrics");
}

int g2048_keys_advance(uint32_t *keyblock_out, unsigned long long *counter_inout, uint64_t seed, void *stream)
{
    if (!keyblock_out || !counter_inout) return fail(G2048_ERR_ARG, "g2048_keys_advance: null pointer");
    if ((reinterpret_cast<uintptr_t>(counter_inout) & 7u) || !aligned4(keyblock_out))
        return fail(G2048_ERR_ARG, "g2048_keys_advance: misaligned pointer");
    hipLaunchKernelGGL(keys_advance_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), keyblock_out, counter_inout, seed);
    return check_launch("g2048_keys_advance");
}

int g2048_selftest(uint32_t *result_out, void *stream)
{
    if (!result_out) return fail(G2048_ERR_ARG, "g2048_selftest: null pointer");
    volatile double x = 0.1, y = 3.0;
    const double p = x * y;        // rounded product, computed on the host
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), result_out,
                       0xA3A2A1A0u, 0xB3B2B1B0u, (double)x, (double)y, -p);
    return check_launch("g2048_selftest");
}

}  // extern "C"
