// g2048_beam.hip -- BeamSearchAgent.get_action (reference agents/beam_search_agent.py:71-181) on gfx950.
//
// One wavefront owns one game. The beam (<= 128 boards, 16 B each) lives in LDS. Each level: all the moves (stage A), the
// children that changed the board packed into LDS in the reference's generation order, then spawn + heuristic (stage B)
// and the ranking, 64 children per pass; the first `width` write themselves to the beam. Details at beam_decide.
// Scores are computed in the reference's operation order (bit-exact with the oracle); no MFMA, no global memory
// traffic inside the search (root in, action out). beam_decide() is the search as a device function; beam_kernel
// runs it once per game (g2048_beam_get_action); play_kernel / play_spec_kernel loop it with the env step
// (g2048_play_games), the latter with helper wavefronts that search the next moves' possible roots ahead of time.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>

#include "../../include/g2048.h"
#include "../../include/g2048_testing.h"
#include "g2048_board.h"
#include "g2048_instrument.h"
#include "g2048_rng.h"

using namespace g2048;

namespace {

constexpr int kMaxWidth = G2048_BEAM_MAX_WIDTH;
constexpr size_t kOrderMaxGames = 1u << 20;
constexpr int kSimdsPerCu = 4, kFallbackCus = 256;     // CDNA: four SIMDs per compute unit; MI355X's CU count if the device cannot be asked

// Launch arithmetic that depends on the chip's size, all of it from the compute-unit count of the device the call runs on
// (hipDeviceGetAttribute, asked per call: a partitioned or CU-masked device simply reports fewer):
//   order_row      blocks b, b + row, b + 2 row ... of a one-wavefront-per-block launch share a SIMD when row = the number of
//                  SIMDs (measured on MI355X, tools/ubench/placement.hip): beam_order_kernel deals the games in rows of it;
//   order_min      the balanced order pays from four searches per SIMD on (at two it costs more than it gains);
//   helper_cap     helper wavefronts never take more than a quarter of the wavefronts the chip holds at once (resident blocks
//                  per CU x CUs / 4), so that owners find room whatever the dispatch order.
// Internal flag of beam_decide (bits 0-1 are G2048_BEAM_FIXED_DOWN / RANK_BY_COUNTING): s_setprio by remaining levels.
// Set by g2048_beam_get_action when every block of the launch is resident at once; with more blocks than the chip holds,
// wavefronts that end together leave the rest of the grid to a burst dispatch on an emptied chip (8192 games: -10 %).
constexpr uint32_t kFlagPrioByRemaining = 4u;
constexpr int kPrioT1 = 2, kPrioT2 = 5, kPrioT3 = 10;     // s_setprio 1 / 2 / 3 above this many remaining levels (any set that separates the last ~10 measures the same)
struct LaunchPlan { uint32_t order_row, order_min, helper_cap; };
constexpr uint32_t kHelperDiv = 4u;      // helpers take at most 1 / kHelperDiv of the resident wavefronts (1/2 and 1/1 measured slower)
constexpr LaunchPlan launch_plan(int cus, int resident_blocks_per_cu)
{
    const uint32_t c = cus > 0 ? (uint32_t)cus : (uint32_t)kFallbackCus;
    const uint32_t b = resident_blocks_per_cu > 0 ? (uint32_t)resident_blocks_per_cu : 32u;
    return LaunchPlan{c * kSimdsPerCu, 4u * c * kSimdsPerCu, c * b / kHelperDiv > 0u ? c * b / kHelperDiv : 1u};
}
static_assert(launch_plan(256, 32).order_row == 1024 && launch_plan(256, 32).order_min == 4096 &&
              launch_plan(256, 32).helper_cap == 8192u / kHelperDiv, "MI355X: the values rounds 1-2 had hard-coded");

constexpr uint32_t kSpecSlotsPerGame = 8u;       // = kSpec below (a power of two; 4 and 16 measured slower, profiles/r03_eval_helpers.txt)
constexpr uint32_t default_helpers(uint32_t n_games, uint32_t helper_cap)
{
    const uint32_t want = n_games / 2u > 1024u ? n_games / 2u : 1024u;      // round 3: half the games (a quarter before)
    const uint32_t most = kSpecSlotsPerGame * n_games < want ? kSpecSlotsPerGame * n_games : want;
    return most < helper_cap ? most : helper_cap;
}

int device_cus()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        return kFallbackCus;
    }
    return cus;
}

__device__ __forceinline__ uint32_t prefix_count(unsigned long long ballot)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// ---------------------------------------------------------------- ranking by sorting network
// The fast levels rank unique 32-bit keys (score * 512 + 511 - generation index). All-pairs counting costs ~1.5 VALU
// instructions per (key, lane) pair -- 190 to 240 per level at width 20 and two fifths of a decision's time. A bitonic
// network sorts the 64 keys a wavefront holds, one per lane, in 21 compare-exchange steps of 3 to 5 instructions: the
// partner of lane e at distance j is e ^ j, reached through DPP for j = 1, 2 (quad_perm), 4 (quad reverse, then half-row
// mirror), 8 (row rotate by 8) -- folded into the v_max / v_min that consume it -- and through gfx950's
// v_permlane16_swap / v_permlane32_swap for j = 16, 32; which of the pair keeps the larger key is a compile-time lane mask
// (an SGPR pair feeding v_cndmask).
constexpr uint64_t cx_mask(int k, int j)         // lanes that keep the LARGER key in step (k, j) of a descending sort
{
    uint64_t m = 0;
    for (int e = 0; e < 64; ++e) {
        const bool desc = k >= 64 || (e & k) == 0, lower = (e & j) == 0;
        if (desc == lower) m |= 1ull << e;
    }
    return m;
}

__device__ __forceinline__ uint32_t pick_by_mask(uint32_t if0, uint32_t if1, uint64_t mask)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
    return r;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_of(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }

// A keep-the-larger mask that is a union of whole 16-lane rows, or of whole 4-lane banks with the
// same banks in every row, can be applied by the DPP row_mask / bank_mask of the v_max that follows an unconditional v_min -- no
// v_cndmask and no 64-bit mask in SGPRs (two s_mov per step otherwise): -1 if the mask is not of that shape.
constexpr int rows_of_mask(uint64_t m)
{
    int r = 0;
    for (int row = 0; row < 4; ++row) {
        const uint64_t bits = (m >> (16 * row)) & 0xffffull;
        if (bits == 0xffffull) r |= 1 << row;
        else if (bits != 0) return -1;
    }
    return r;
}
constexpr int banks_of_mask(uint64_t m)
{
    int b = 0;
    for (int bank = 0; bank < 4; ++bank) {
        const uint64_t bits = (m >> (4 * bank)) & 0xfull;
        if (bits == 0xfull) b |= 1 << bank;
        else if (bits != 0) return -1;
    }
    uint64_t all = 0;
    for (int row = 0; row < 4; ++row)
        for (int bank = 0; bank < 4; ++bank)
            if ((b >> bank) & 1) all |= 0xfull << (16 * row + 4 * bank);
    return all == m ? b : -1;
}

// One compare-exchange step of the 32-bit network. The asm forms are only instantiated for masks of the shape their DPP
// row_mask / bank_mask immediates can express (if constexpr: an untaken form is never emitted, whatever the optimisation level).
// (A two-instruction form -- borrow of a DPP subtraction, s_xor with the mask, VOP2 DPP select -- measured 2.7 % slower at 4096
// games: it lengthens the dependent chain; profiles/r03_beam_latency.txt. Git history has it.)
template <uint64_t KEEP_MAX, int J>
__device__ __forceinline__ uint32_t cx_step_m(uint32_t key)
{
    constexpr uint64_t keep_max = KEEP_MAX;
    constexpr int rows = rows_of_mask(KEEP_MAX), banks = banks_of_mask(KEEP_MAX);
    if constexpr (J >= 16 && rows > 0) {         // lane swaps: both lanes of a pair hold both keys; the larger on the rows of the mask
        uint32_t a, b, out;
        asm("s_nop 1" : "+v"(key));              // (a written-out step may have produced key: wait states before the lane swap reads it)
        if constexpr (J == 16) { const auto r = __builtin_amdgcn_permlane16_swap(key, key, false, false); a = r[0]; b = r[1]; }
        else { const auto r = __builtin_amdgcn_permlane32_swap(key, key, false, false); a = r[0]; b = r[1]; }
        asm("v_min_u32 %0, %1, %2\n\t"
            "s_nop 0\n\t"                                   // (with the v_min: the wait states between the swap's write and the DPP read)
            "v_max_u32_dpp %0, %1, %2 quad_perm:[0,1,2,3] row_mask:%3 bank_mask:0xf"
            : "=&v"(out) : "v"(a), "v"(b), "i"(rows));
        return out;
    } else if constexpr (J == 8 && banks > 0) {
        uint32_t out;
        asm("s_nop 1\n\t"
            "v_min_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_max_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:%2"
            : "=&v"(out) : "v"(key), "i"(banks));
        return out;
    } else if constexpr (J == 4 && banks > 0) {  // e ^ 3 (quad_perm [3,2,1,0]) into a scratch register, e ^ 7 on the operand = e ^ 4
        uint32_t out, tmp;
        asm("s_nop 1\n\t"
            "v_mov_b32_dpp %1, %2 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_max_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:%3"
            : "=&v"(out), "=&v"(tmp) : "v"(key), "i"(banks));
        return out;
    } else {                                     // v_max_dpp + v_min_dpp + v_cndmask by the step's lane mask
        uint32_t hi, lo;
        if constexpr (J == 1) { hi = max(dpp_of<0xB1>(key), key); lo = min(dpp_of<0xB1>(key), key); }
        else if constexpr (J == 2) { hi = max(dpp_of<0x4E>(key), key); lo = min(dpp_of<0x4E>(key), key); }
        else if constexpr (J == 4) { const uint32_t r = dpp_of<0x1B>(key); hi = max(dpp_of<0x141>(r), key); lo = min(dpp_of<0x141>(r), key); }
        else if constexpr (J == 8) { hi = max(dpp_of<0x128>(key), key); lo = min(dpp_of<0x128>(key), key); }
        else if constexpr (J == 16) { const auto r = __builtin_amdgcn_permlane16_swap(key, key, false, false); hi = max(r[0], r[1]); lo = min(r[0], r[1]); }
        else { const auto r = __builtin_amdgcn_permlane32_swap(key, key, false, false); hi = max(r[0], r[1]); lo = min(r[0], r[1]); }
        return pick_by_mask(lo, hi, keep_max);
    }
}

template <int K, int J>
__device__ __forceinline__ uint32_t cx_step(uint32_t key) { return cx_step_m<cx_mask(K, J), J>(key); }

// stages k = 2 .. KMAX of the 64-lane descending network: KMAX = 64 sorts the wavefront's keys, descending by lane; KMAX = 16
// leaves every 16-lane row sorted, rows 0 and 2 descending, rows 1 and 3 ASCENDING
template <int KMAX>
__device__ __forceinline__ uint32_t sort_stages(uint32_t key)
{
    key = cx_step<2, 1>(key);
    key = cx_step<4, 2>(key); key = cx_step<4, 1>(key);
    key = cx_step<8, 4>(key); key = cx_step<8, 2>(key); key = cx_step<8, 1>(key);
    key = cx_step<16, 8>(key); key = cx_step<16, 4>(key); key = cx_step<16, 2>(key); key = cx_step<16, 1>(key);
    if (KMAX >= 32) {
        key = cx_step<32, 16>(key); key = cx_step<32, 8>(key); key = cx_step<32, 4>(key); key = cx_step<32, 2>(key);
        key = cx_step<32, 1>(key);
    }
    if (KMAX >= 64) {
        key = cx_step<64, 32>(key); key = cx_step<64, 16>(key); key = cx_step<64, 8>(key); key = cx_step<64, 4>(key);
        key = cx_step<64, 2>(key); key = cx_step<64, 1>(key);
    }
    return key;
}

// the last stage alone: sorts a sequence that descends and then ascends (bitonic) into descending order
__device__ __forceinline__ uint32_t merge64_desc(uint32_t key)
{
    key = cx_step<64, 32>(key); key = cx_step<64, 16>(key); key = cx_step<64, 8>(key); key = cx_step<64, 4>(key);
    key = cx_step<64, 2>(key); key = cx_step<64, 1>(key);
    return key;
}

// The 64 largest of up to 80 keys, descending by lane: a = one key per lane, b = up to 16 more in lanes 48..63 (0 elsewhere
// and for "no key"; real keys are > 0). with_b is wave-uniform. The 16 smallest of a cannot be among the first 48 of the
// union, which is all a beam of width <= 32 takes.
__device__ __forceinline__ uint32_t top64_desc(uint32_t a, uint32_t b, bool with_b)
{
    if (!with_b) return sort_stages<64>(a);
    // the two sorts are independent: their first ten steps are written alternately, so that one register's compare-exchange
    // fills the wait states and the latency of the other's (a wavefront's own latency is what bounds a search at four
    // wavefronts per SIMD)
#define G2048_STEP_AB(K, J) a = cx_step<K, J>(a); b = cx_step<K, J>(b);
    G2048_STEP_AB(2, 1)
    G2048_STEP_AB(4, 2) G2048_STEP_AB(4, 1)
    G2048_STEP_AB(8, 4) G2048_STEP_AB(8, 2) G2048_STEP_AB(8, 1)
    G2048_STEP_AB(16, 8) G2048_STEP_AB(16, 4) G2048_STEP_AB(16, 2) G2048_STEP_AB(16, 1)         // b: row 3 ascending
#undef G2048_STEP_AB
    a = cx_step<32, 16>(a); a = cx_step<32, 8>(a); a = cx_step<32, 4>(a); a = cx_step<32, 2>(a); a = cx_step<32, 1>(a);
    a = merge64_desc(a);
    return merge64_desc(pick_by_mask(a, b, 0xffff000000000000ull));
}

// The same network on 64-bit keys (hi, lo) for the levels whose scores are f64 (1..3): the partner's two words come through
// DPP moves or the permlane swaps, one v_cmp_gt_u64 decides, and which side a lane takes is the compare mask XNOR the step's
// keep-the-larger mask (scalar unit).
struct Key64 { uint32_t hi, lo; };

__device__ __forceinline__ Key64 take_by_compare(const Key64 &x, const Key64 &y, uint64_t keep_max)
{
    // both lanes of a pair hold the same (x, y) up to order; the lane that keeps the larger takes x iff x > y
    const uint64_t kx = ((uint64_t)x.hi << 32) | x.lo, ky = ((uint64_t)y.hi << 32) | y.lo;
    const uint64_t x_gt = __ballot(kx > ky);
    const uint64_t take_x = ~(x_gt ^ keep_max);
    return Key64{pick_by_mask(y.hi, x.hi, take_x), pick_by_mask(y.lo, x.lo, take_x)};
}

template <int K, int J>
__device__ __forceinline__ Key64 cx_step64(const Key64 &key)
{
    constexpr uint64_t keep_max = cx_mask(K, J);
    if (J == 16) {
        const auto h = __builtin_amdgcn_permlane16_swap(key.hi, key.hi, false, false);
        const auto l = __builtin_amdgcn_permlane16_swap(key.lo, key.lo, false, false);
        return take_by_compare(Key64{h[0], l[0]}, Key64{h[1], l[1]}, keep_max);
    }
    if (J == 32) {
        const auto h = __builtin_amdgcn_permlane32_swap(key.hi, key.hi, false, false);
        const auto l = __builtin_amdgcn_permlane32_swap(key.lo, key.lo, false, false);
        return take_by_compare(Key64{h[0], l[0]}, Key64{h[1], l[1]}, keep_max);
    }
    Key64 o;
    if (J == 1) o = Key64{dpp_of<0xB1>(key.hi), dpp_of<0xB1>(key.lo)};
    else if (J == 2) o = Key64{dpp_of<0x4E>(key.hi), dpp_of<0x4E>(key.lo)};
    else if (J == 4) o = Key64{dpp_of<0x141>(dpp_of<0x1B>(key.hi)), dpp_of<0x141>(dpp_of<0x1B>(key.lo))};
    else o = Key64{dpp_of<0x128>(key.hi), dpp_of<0x128>(key.lo)};
    // here x = the partner's key, y = this lane's: take the partner's iff (partner > mine) == (this lane keeps the larger)
    return take_by_compare(o, key, keep_max);
}

template <int KMAX>
__device__ __forceinline__ Key64 sort_stages64(Key64 key)
{
    key = cx_step64<2, 1>(key);
    key = cx_step64<4, 2>(key); key = cx_step64<4, 1>(key);
    key = cx_step64<8, 4>(key); key = cx_step64<8, 2>(key); key = cx_step64<8, 1>(key);
    key = cx_step64<16, 8>(key); key = cx_step64<16, 4>(key); key = cx_step64<16, 2>(key); key = cx_step64<16, 1>(key);
    if (KMAX >= 32) {
        key = cx_step64<32, 16>(key); key = cx_step64<32, 8>(key); key = cx_step64<32, 4>(key); key = cx_step64<32, 2>(key);
        key = cx_step64<32, 1>(key);
    }
    if (KMAX >= 64) {
        key = cx_step64<64, 32>(key); key = cx_step64<64, 16>(key); key = cx_step64<64, 8>(key); key = cx_step64<64, 4>(key);
        key = cx_step64<64, 2>(key); key = cx_step64<64, 1>(key);
    }
    return key;
}

__device__ __forceinline__ Key64 top64_desc64(Key64 a, Key64 b, bool with_b)
{
    a = sort_stages64<64>(a);
    if (!with_b) return a;
    b = sort_stages64<16>(b);
    Key64 c = {pick_by_mask(a.hi, b.hi, 0xffff000000000000ull), pick_by_mask(a.lo, b.lo, 0xffff000000000000ull)};
    c = cx_step64<64, 32>(c); c = cx_step64<64, 16>(c); c = cx_step64<64, 8>(c); c = cx_step64<64, 4>(c);
    c = cx_step64<64, 2>(c); c = cx_step64<64, 1>(c);
    return c;
}

__global__ __launch_bounds__(64) void sort_selftest_kernel(uint32_t *keys, const uint32_t *extra, int with_extra, int wide)
{
    if (wide) {                                  // 64-bit keys: (lo, hi) word pairs
        const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
        const Key64 a = {keys[2 * i + 1], keys[2 * i]};
        Key64 b = {0u, 0u};
        if (with_extra && threadIdx.x >= 48) {
            const size_t e = (size_t)blockIdx.x * 16 + threadIdx.x - 48;
            b = Key64{extra[2 * e + 1], extra[2 * e]};
        }
        const Key64 r = top64_desc64(a, b, with_extra != 0);
        keys[2 * i] = r.lo; keys[2 * i + 1] = r.hi;
        return;
    }
    const uint32_t a = keys[blockIdx.x * 64 + threadIdx.x];
    const uint32_t b = (with_extra && threadIdx.x >= 48) ? extra[blockIdx.x * 16 + threadIdx.x - 48] : 0u;
    keys[blockIdx.x * 64 + threadIdx.x] = top64_desc(a, b, with_extra != 0);
}

// LDS of one search (one wavefront = one game). A level can have 4 * width valid children; they are scored and ranked 64 at a
// time, so PASSES = ceil(4 * width / 64): 1 for width <= 16, 2 up to 32 (the reference's evaluation width is 20), 4 up to 64,
// 8 up to 128.
template <int PASSES>
struct BeamShared {
    static constexpr int kWidth = 16 * PASSES;      // widest beam this instance can hold
    uint4 board[kWidth];                            // the beam, rank order
    uint32_t root[kWidth];                          // per beam entry: root action | max code << 8
    uint4 cboard[64 * PASSES];                      // moved (pre-spawn) boards of the VALID children, compacted in generation
    uint32_t croot[64 * PASSES];                    //   order; croot: root action | parent max << 8
    alignas(16) double score[64 * PASSES + 16];     // f64 scores (levels 1..3) or, reinterpreted, u32 keys
};

struct Decision { uint32_t action; float prob; uint32_t expanded; };

// BeamSearchAgent.get_action for the game this wavefront owns. mask_in < 0: no caller mask. Every lane returns the same
// Decision. Must be called by all 64 lanes (it contains workgroup barriers).
//   stage A: lane 2p + axis makes BOTH moves of one axis of parent p (g2048_board.h move_axis), 32 parents per round; the
//     children that changed the board go to LDS at their ballot-prefix index = the reference's generation order (parent
//     rank, action);
//   stage B, 64 children per pass: the spawn -- child j of the decision takes draw j, exactly as the Python loop consumes
//     its RNG -- and the heuristic score, fed the empty count and max code the kernel already knows; key / score to LDS;
//   ranking: the order of Python's stable sorted(reverse=True) (score descending, generation order ascending).
//     _fast_evaluate is integer-valued, so on its levels (0 and >= 4) the pair (score, order) is the single unique integer
//     score * 512 + (511 - index); beams up to 32 wide sort those keys with the network above (top64_desc) and lane r copies
//     the r-th child into the beam. Everywhere else (levels 1..3 with their f64 scores, very small or very large levels, wider
//     beams) every child counts the children that sort before it, with broadcast LDS reads, sixteen keys per trip so that the
//     reads of a trip are in flight together, and the first `width` write themselves to the beam at their rank.
// Tried and measured slower on MI355X (profiles/r02_beam_*.txt): four games per 256-thread block sharing one pass for
// everybody's leftover children (17 % fewer VALU instructions, but twice the barrier wait); a radix select of the top-k keys
// with wave ballots instead of the all-pairs count (8 % fewer VALU, 2x the SALU, slower); one wavefront per 64-child group
// (no fewer instructions, three real barriers per level); handing the empty counts / draw indices from stage A to stage B.
template <int PASSES>
__device__ __forceinline__ Decision beam_decide(BeamShared<PASSES> &sh, const Board &root, int mask_in, int width, int depth,
                                                uint32_t early_thr, uint32_t mid_thr, uint32_t k0, uint32_t k1, uint64_t gid,
                                                uint32_t flags)
{
    uint4 *const s_board = sh.board;
    uint32_t *const s_root = sh.root;
    uint4 *const s_cboard = sh.cboard;
    uint32_t *const s_croot = sh.croot;
    double *const s_score = sh.score;
    const uint32_t lane = threadIdx.x;
    const bool fixed_down = (flags & 1u) != 0u;          // G2048_BEAM_FIXED_DOWN
    const bool count_rank = (flags & 2u) != 0u;          // G2048_BEAM_RANK_BY_COUNTING
    const bool by_remaining = (flags & kFlagPrioByRemaining) != 0u;

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
    // After a level ranked by the sorting network the beam is not copied out: lane r keeps `pick`, the children-array slot of
    // the rank-r child, and the next level's lanes fetch their parents from there (from_pick, wave-uniform) -- one LDS
    // round trip and two barriers fewer per level. A wavefront's LDS operations execute in order, and a level reads all its
    // parents (one instruction) before it writes any child.
    bool from_pick = false;
    uint32_t pick = 0u;
    // lane constants of the row-per-lane tail stream (row r = lane & 3 of a quad's child): masks for "odd row", "rows 2..3",
    // "row 0 or 3", "not row 3", the flag "row 3", and the lane a tail key is fetched from (lane 48 + k <- quad k)
    struct { uint32_t odd, low2, edge, notlast, fetch; bool last; } tq;
    {
        const uint32_t r = lane & 3u;
        tq.odd = (r & 1u) ? ~0u : 0u; tq.low2 = r >= 2u ? ~0u : 0u; tq.edge = (r == 0u || r == 3u) ? ~0u : 0u;
        tq.notlast = r != 3u ? ~0u : 0u; tq.last = r == 3u;
        tq.fetch = lane >= 48u ? (lane - 48u) * 16u : lane * 4u;
        asm volatile("" : "+v"(tq.odd), "+v"(tq.low2), "+v"(tq.edge), "+v"(tq.notlast), "+v"(tq.fetch));
    }
    // stage A's lane 2p + axis keeps its axis for the whole search: the direction network's selector words (rows -> lines,
    // forward lines -> rows, reversed lines -> rows with the agent's DOWN quirk folded in) are loop-invariant registers
    AxisSel asel = axis_sel((lane & 1u) != 0u, fixed_down);
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(asel.in[k]), "+v"(asel.fwd[k]), "+v"(asel.rev[k]));   // (not re-derived per level)

    for (int level = 0; level == 0 || level < actual_depth; ++level) {
        const bool fast = level == 0 || level > 3;             // :122, :139
        if (by_remaining) {
            // Issue priority by the levels this search still has to run. The SIMD's arbiter serves the oldest wavefront first:
            // of four searches that share a SIMD the first runs at a lone wavefront's pace and is done after 60 % of the launch,
            // the last ones then finish on a SIMD that holds one or two wavefronts (tools/beam_timeline.py). Longest-remaining-
            // first keeps all of them on the SIMD until the last levels (profiles/r03_beam_priority.txt).
            const int rem = actual_depth - level;
            if (rem > kPrioT3) __builtin_amdgcn_s_setprio(3);
            else if (rem > kPrioT2) __builtin_amdgcn_s_setprio(2);
            else if (rem > kPrioT1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // ---- stage A
        uint32_t total_valid = 0;
        const int n_parents = level == 0 ? 1 : nb;
        for (int round = 0; round * 32 < n_parents; ++round) {                   // one round unless the beam is wider than 32
            const uint32_t par = (uint32_t)round * 32u + (lane >> 1);
            const bool vertical = (lane & 1u) != 0u;
            const bool on = (int)par < n_parents;
            Board P = root;
            uint32_t ra_f = (vertical ? 1u : 0u) | (root_max << 8), ra_r = (vertical ? 3u : 2u) | (root_max << 8);
            bool en_f = on, en_r = on;
            if (level == 0) {
                en_f = on && ((mask >> (vertical ? 1 : 0)) & 1u);
                en_r = on && ((mask >> (vertical ? 3 : 2)) & 1u);
            } else if (from_pick) {                                              // (beams up to 32 wide: one round)
                const uint32_t qs = on ? (uint32_t)__shfl((int)pick, (int)(lane >> 1), 64) : 0u;
                uint32_t q4 = qs * 4u;
                asm volatile("" : "+v"(q4));
                const uint4 pv = s_cboard[qs];
                P = Board{{pv.x, pv.y, pv.z, pv.w}};
                ra_f = ra_r = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_croot) + q4);
            } else if (on) {
                const uint4 pv = s_board[par];
                P = Board{{pv.x, pv.y, pv.z, pv.w}};
                ra_f = ra_r = s_root[par];
            }
            Board cf, cr;
            move_axis_sel(P, asel, cf, cr);                                      // :115 / :152 (the agent's DOWN = rot180(true DOWN))
            const bool vf = en_f & !same(cf, P), vr = en_r & !same(cr, P);       // (no short circuit: one straight block, no exec region)
            const unsigned long long bf = __ballot(vf), br = __ballot(vr);
            const uint32_t before = total_valid + prefix_count(bf) + prefix_count(br);    // valid children generated earlier
            // partner lane (same parent, other axis) through a DPP quad swap
            const uint32_t mine = (vf ? 1u : 0u) | (vr ? 2u : 0u);
            const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
            // order inside a parent: LEFT(h,f) UP(v,f) RIGHT(h,r) DOWN(v,r)
            const uint32_t idx_f = vertical ? before - ((other >> 1) & 1u) : before;
            const uint32_t idx_r = vertical ? before + (mine & 1u) : before + (mine & 1u) + (other & 1u);
            total_valid += (uint32_t)__popcll(bf) + (uint32_t)__popcll(br);
            // (the 4-byte slot address is formed on its own: left to itself the compiler derives it from the 16-byte one with a
            // 64-bit multiply-add)
            uint32_t of = idx_f * 4u, orr = idx_r * 4u;
            asm volatile("" : "+v"(of), "+v"(orr));
            if (vf) { s_cboard[idx_f] = make_uint4(cf.w[0], cf.w[1], cf.w[2], cf.w[3]); *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_croot) + of) = ra_f; }
            if (vr) { s_cboard[idx_r] = make_uint4(cr.w[0], cr.w[1], cr.w[2], cr.w[3]); *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_croot) + orr) = ra_r; }
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
        // ---- stage B: spawn + score of the compacted children, one per lane and pass (a pass only runs if it has children)
        // Ranking by sorting network (above) for beams up to 32 wide, when the level has 17 .. 80 children (wave-uniform): then
        // the up to 16 children beyond the first 64 sit in lanes 48..63 of the second pass, where the network wants their keys.
        // The common level of a width-17..32 search: _fast_evaluate scores and 65..80 children. Both passes -- 64 children, then
        // the up to 16 more in lanes 48..63 -- the write-back and the sorting network run as ONE straight-line block, so that the
        // second pass's spawn + score (independent work) fills the issue slots the first pass's dependent chains and the network's
        // serial compare-exchange steps leave open: with four wavefronts per SIMD (4096 games) the search is bound by each
        // wavefront's own latency, not by instruction issue (profiles/r03_beam_latency.txt).
        if (PASSES == 2 && fast && !count_rank && total_valid > 64u && total_valid <= 80u) {
            // first stream: children 0..63, one per lane
            const uint4 cv0 = s_cboard[lane];
            const uint32_t cr0 = s_croot[lane];
            Board c0 = {{cv0.x, cv0.y, cv0.z, cv0.w}};
            const uint32_t nm0 = count_empty(c0);
            const unsigned long long b0 = __ballot(nm0 != 0u);
            const uint32_t j0 = draws + prefix_count(b0);
            // second stream: children 64..79, one ROW per lane -- lane 4k + r holds row r of child 64 + k, so the 16 children
            // fill the wavefront instead of a quarter of it; what spans the board (empty count, rank of the spawn cell, max
            // code, corners, vertical neighbours) moves between the four lanes of a quad through DPP quad permutes. 84 vector
            // instructions instead of 147, and independent of the first stream until the network.
            const uint32_t ci1 = 64u + (lane >> 2);
            const bool live1 = ci1 < total_valid;
            uint32_t w1 = reinterpret_cast<const uint32_t *>(s_cboard)[256u + lane];
            const uint32_t cr1 = s_croot[ci1];                                   // (slots up to 127 exist; dead ones hold stale data)
            const uint32_t z1 = zflag(w1), cnt1 = popc(z1);
            const uint32_t pair1 = cnt1 + dpp_of<0xB1>(cnt1);                     // rows {0,1} / {2,3}
            const uint32_t nm1 = pair1 + dpp_of<0x4E>(pair1);                     // empty cells of the child, in all four lanes
            const uint32_t above1 = (dpp_of<0xA0>(cnt1) & tq.odd) + (dpp_of<0x00>(pair1) & tq.low2);      // empty cells in the rows above
            const unsigned long long b1 = __ballot(live1 && nm1 != 0u && tq.last);
            const uint32_t j1 = draws + (uint32_t)__popcll(b0) + prefix_count(b1);  // (lanes of a quad count the quads before theirs)
            const uint32_t h1 = rng_draw(k0, k1, gid, j1);
            const uint32_t k1_ = ((((h1 >> 16) * nm1) >> 16) - above1);           // rank of the spawn cell inside this row, if it is here
            {
                const uint32_t ones = 0x01010101u;
                const uint32_t target = k1_ < cnt1 ? (k1_ + 1u) * ones : 0x7f7f7f7fu;
                const uint32_t hit = zflag(((z1 >> 7) * ones) ^ target) & z1;
                w1 |= hit >> (((h1 & 0xffffu) >= 58982u) ? 6u : 7u);
            }
            const uint32_t pm1 = cr1 >> 8;
            uint32_t f1 = zflag(w1 ^ ((pm1 + 1u) * 0x01010101u));                 // some cell holds parent max + 1?
            f1 |= dpp_of<0xB1>(f1); f1 |= dpp_of<0x4E>(f1);
            const uint32_t cm1 = pm1 + (f1 ? 1u : 0u);
            uint32_t cc1 = max(w1 & 0xffu, w1 >> 24) & tq.edge;                   // corners live in rows 0 and 3
            cc1 = max(cc1, dpp_of<0xB1>(cc1)); cc1 = max(cc1, dpp_of<0x4E>(cc1));
            const uint32_t n1 = w1 + B7F;
            uint32_t pr1 = popc(n1 & ~((w1 ^ (w1 >> 8)) + B7F) & B80);           // equal non-zero neighbours in the row ...
            pr1 += popc(n1 & ~((w1 ^ dpp_of<0xF9>(w1)) + B7F) & B80 & tq.notlast);  // ... and towards the row below
            pr1 += dpp_of<0xB1>(pr1); pr1 += dpp_of<0x4E>(pr1);
            const uint32_t sc1 = (nm1 - (nm1 ? 1u : 0u)) * 10u + cm1 * 2u + (cc1 ? (2u << cc1) : 0u) + pr1 * 2u;
            const uint32_t e1 = live1 ? (sc1 << 9) + (511u - ci1) : 0u;
            const uint32_t key1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)tq.fetch, (int)e1);   // lane 48 + k <- quad k
            if (live1) {
                reinterpret_cast<uint32_t *>(s_cboard)[256u + lane] = w1;
                if (tq.last) s_croot[ci1] = (cr1 & 0xffu) | (cm1 << 8);
            }
            spawn_rowprefix(c0, rng_draw(k0, k1, gid, j0));                      // :155 (a no-op on a full board)
            draws += (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1);
            const uint32_t pm0 = cr0 >> 8;
            const uint32_t cm0 = pm0 + (has_code(c0, pm0 + 1u) ? 1u : 0u);
            const uint32_t key0 = (eval_fast_u32_known(c0, nm0 - (nm0 ? 1u : 0u), cm0) << 9) + (511u - lane);
            s_cboard[lane] = make_uint4(c0.w[0], c0.w[1], c0.w[2], c0.w[3]);
            s_croot[lane] = (cr0 & 0xffu) | (cm0 << 8);
            pick = 511u - (top64_desc(key0, key1, true) & 511u);
            nb = width;                                                          // more than 64 children, width <= 32
            from_pick = true;
            continue;
        }
        // the same for a fast level with up to 64 children (level 0 included): one stream, the 64-key network, hand-over by slots
        if (PASSES <= 2 && fast && !count_rank && total_valid <= 64u) {
            const bool live0 = lane < total_valid;
            const uint4 cv0 = s_cboard[live0 ? lane : 0u];
            const uint32_t cr0 = s_croot[live0 ? lane : 0u];
            Board c0 = {{cv0.x, cv0.y, cv0.z, cv0.w}};
            const uint32_t nm0 = count_empty(c0);
            const unsigned long long b0 = __ballot(live0 && nm0 != 0u);
            spawn_rowprefix(c0, rng_draw(k0, k1, gid, draws + prefix_count(b0)));    // :155 (a no-op on a full board)
            draws += (uint32_t)__popcll(b0);
            const uint32_t pm0 = cr0 >> 8;
            const uint32_t cm0 = pm0 + (has_code(c0, pm0 + 1u) ? 1u : 0u);
            const uint32_t e0 = (eval_fast_u32_known(c0, nm0 - (nm0 ? 1u : 0u), cm0) << 9) + (511u - lane);
            if (live0) {
                s_cboard[lane] = make_uint4(c0.w[0], c0.w[1], c0.w[2], c0.w[3]);
                s_croot[lane] = (cr0 & 0xffu) | (cm0 << 8);
            }
            pick = 511u - (sort_stages<64>(live0 ? e0 : 0u) & 511u);
            nb = (int)min(total_valid, (uint32_t)width);
            from_pick = true;
            continue;
        }
        from_pick = false;
        const bool net = PASSES <= 2 && total_valid > 16u && total_valid <= (PASSES == 2 ? 80u : 64u) && !(count_rank && fast);
        Board child[PASSES];
        double score[PASSES];
        uint32_t ikey[PASSES], cinfo[PASSES];          // cinfo: root action | max code << 8 of the spawned child
        uint32_t cidx[PASSES];                         // the child this lane holds in pass p ...
        bool livep[PASSES];                            // ... if any
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            score[p] = -INFINITY; ikey[p] = 0u; cinfo[p] = 0u; child[p] = root;
            const bool tail_row = net && p == 1;
            cidx[p] = tail_row ? lane + 16u : (uint32_t)(p * 64) + lane;
            livep[p] = cidx[p] < total_valid && (!tail_row || lane >= 48u);
            if ((uint32_t)(p * 64) < total_valid) {                              // wave-uniform
                const uint32_t ci = cidx[p];
                const bool live = livep[p];
                const uint4 cv = s_cboard[live ? ci : 0u];
                Board c = {{cv.x, cv.y, cv.z, cv.w}};
                const uint32_t cr = s_croot[live ? ci : 0u];
                const uint32_t pmax = cr >> 8;
                // a child with an empty cell consumes the next draw, in generation order (:262-269); only the reference's
                // rot180-DOWN quirk can produce a "changed" board without one
                const uint32_t n_moved = count_empty(c);
                const unsigned long long bc = __ballot(live && n_moved != 0u);
                const uint32_t j = draws + prefix_count(bc);
                draws += (uint32_t)__popcll(bc);
                // (row select + byte-wise prefix inside the row; the all-rows prefix form is 50 issue cycles cheaper per pass and
                // measured 2.7 % slower at 4096 games -- a longer dependent chain; profiles/r03_beam_latency.txt)
                spawn_rowprefix(c, rng_draw(k0, k1, gid, j));                    // :118 / :155; a no-op on a full board
                // :122 / :158-161. What the evaluators need is already known: the empty count (one fewer after a spawn)
                // and the max code -- a move raises the parent's max by at most one, exactly when some cell now holds
                // parent max + 1 (two max tiles merged, or a 2/4 spawned onto a board whose max was lower)
                const uint32_t n_child = n_moved - (n_moved ? 1u : 0u);
                const uint32_t cmax = pmax + (has_code(c, pmax + 1u) ? 1u : 0u);
                cinfo[p] = (cr & 0xffu) | (cmax << 8);
                if (fast) {
                    ikey[p] = live ? (eval_fast_u32_known(c, n_child, cmax) << 9) + (511u - ci) : 0u;
                    if (!net) reinterpret_cast<uint32_t *>(s_score)[ci] = ikey[p];       // ci < 64 * PASSES always
                } else {
                    score[p] = live ? eval_full_known(c, phase, n_child, cmax) : -INFINITY;
                    if (!tail_row || lane >= 48u) s_score[ci] = score[p];        // (the tail row's idle lanes own no slot)
                }
                child[p] = c;
            }
        }
        if (net) {
            // the spawned children go back to their LDS slots, the keys are sorted in registers, and lane r takes the child
            // whose key came r-th
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                if (livep[p]) {
                    s_cboard[cidx[p]] = make_uint4(child[p].w[0], child[p].w[1], child[p].w[2], child[p].w[3]);
                    s_croot[cidx[p]] = cinfo[p];
                }
            const bool with_tail = PASSES == 2 && total_valid > 64u;
            nb = (int)min(total_valid, (uint32_t)width);
            bool exact = !count_rank;                  // (the test switch: laid out for the network, ranked by counting)
            if (fast) {
                // the keys are unique, so their descending order is the stable order of :131 / :174
                pick = 511u - (top64_desc(ikey[0], PASSES == 2 ? ikey[PASSES - 1] : 0u, with_tail) & 511u);
            } else {
                // f64 scores as order-preserving unsigned keys (sign bit flipped, negative values complemented; the score
                // is never -0.0 or NaN), the low seven bits replaced by 127 - generation index: equal scores then sort in
                // generation order, as Python's stable sort leaves them. Two scores that differ ONLY in those seven bits
                // would be misordered: if sorted neighbours agree in everything else, their scores are read back, and unless
                // they are equal the level goes to the counting loop below instead.
                Key64 k[PASSES];
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(score[p]);
                    uint32_t hi = (uint32_t)(bits >> 32), lo = (uint32_t)bits;
                    const uint32_t neg = (uint32_t)((int32_t)hi >> 31);
                    hi ^= neg | 0x80000000u; lo ^= neg;
                    k[p] = livep[p] ? Key64{hi, (lo & ~127u) | (127u - cidx[p])} : Key64{0u, 0u};
                }
                const Key64 sorted = top64_desc64(k[0], k[PASSES - 1], with_tail);
                const uint32_t nhi = (uint32_t)__shfl_down((int)sorted.hi, 1, 64), nlo = (uint32_t)__shfl_down((int)sorted.lo, 1, 64);
                // (every run of such neighbours matters, not only those inside the beam: a later member of a run that reaches
                // into the beam may belong before an earlier one. With the tail merged in only the first 48 places are sorted;
                // a run that reaches place 47 is not trusted.)
                const uint32_t last = with_tail ? 46u : 62u;
                const bool close = lane <= last && sorted.hi == nhi && ((sorted.lo ^ nlo) < 128u) && (nhi | nlo) != 0u;
                pick = 127u - (sorted.lo & 127u);
                if (exact && __ballot(close)) {         // equal scores (common: transpositions) or scores a few ulp apart?
                    const double mine = s_score[pick], next = __shfl_down(mine, 1, 64);
                    exact = __ballot(close && (mine != next || (with_tail && lane == last))) == 0ull;
                }
            }
            if (exact) {
                from_pick = true;
                continue;
            }
        }
        __syncthreads();
        // ---- stable descending rank (:131, :174) among the valid children. Every pass that ran wrote all its 64 slots (zeros
        // / -inf beyond the last child), so reading up to the next multiple of 16 (8) past total_valid sees only those.
        uint32_t rank[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) rank[p] = 0u;
        if (fast) {
            const uint4 *keys = reinterpret_cast<const uint4 *>(s_score);
            // the next trip's sixteen keys are requested before this trip's are compared (the last request reads slots
            // nobody uses; they are inside s_score)
            uint4 n0 = keys[0], n1 = keys[1], n2 = keys[2], n3 = keys[3];
            for (uint32_t j = 0; j < total_valid; j += 16) {
                const uint4 q0 = n0, q1 = n1, q2 = n2, q3 = n3;
                n0 = keys[(j >> 2) + 4]; n1 = keys[(j >> 2) + 5]; n2 = keys[(j >> 2) + 6]; n3 = keys[(j >> 2) + 7];
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const uint32_t me = ikey[p];
                    rank[p] += (q0.x > me ? 1u : 0u) + (q0.y > me ? 1u : 0u) + (q0.z > me ? 1u : 0u) + (q0.w > me ? 1u : 0u) +
                               (q1.x > me ? 1u : 0u) + (q1.y > me ? 1u : 0u) + (q1.z > me ? 1u : 0u) + (q1.w > me ? 1u : 0u) +
                               (q2.x > me ? 1u : 0u) + (q2.y > me ? 1u : 0u) + (q2.z > me ? 1u : 0u) + (q2.w > me ? 1u : 0u) +
                               (q3.x > me ? 1u : 0u) + (q3.y > me ? 1u : 0u) + (q3.z > me ? 1u : 0u) + (q3.w > me ? 1u : 0u);
                }
            }
        } else {
            for (uint32_t j = 0; j < total_valid; j += 2) {
                const double2 sj = *reinterpret_cast<const double2 *>(&s_score[j]);
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const uint32_t ci = cidx[p];
                    rank[p] += (sj.x > score[p] || (sj.x == score[p] && j < ci)) ? 1u : 0u;
                    rank[p] += (sj.y > score[p] || (sj.y == score[p] && j + 1 < ci)) ? 1u : 0u;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if (livep[p] && rank[p] < (uint32_t)width) {                          // :132 / :175
                s_board[rank[p]] = make_uint4(child[p].w[0], child[p].w[1], child[p].w[2], child[p].w[3]);
                s_root[rank[p]] = cinfo[p];
            }
        }
        nb = (int)min(total_valid, (uint32_t)width);
        __syncthreads();
    }
    const Decision d = {(from_pick ? s_croot[uniform(pick)] : s_root[0]) & 0xffu, 1.0f, expanded};      // :178-181
    __syncthreads();                                                            // read before any reuse of the LDS
    return d;
}

// Blocks b, b + row_len, b + 2 row_len, ... of a one-wavefront-per-block launch share a SIMD when row_len is the chip's SIMD
// count (MI355X: 256 CUs x 4 = 1024; measured, tools/ubench/placement.hip; the caller passes launch_plan().order_row), and a launch lasts as long as its most loaded SIMD: with the games in caller order the
// heaviest SIMD of the benchmark batch carries 17 % more children than the mean. A decision's cost is its depth, which
// takes one of three values fixed by the root's empty cells (beam_decide :96-106); order[b] = the game block b searches:
// the games sorted by that class, deepest first, dealt to the SIMDs in rows of row_len that alternate direction, so that every
// SIMD gets deep and shallow searches (launch 145 -> 128 us when the order is free, profiles/r02_beam_balance.txt).
// Deepest-first is also the order that keeps the tail short when the batch is larger than the chip holds at once.
// One block; a counting sort whose per-class counts are wave ballots (two LDS atomics per wavefront and class in all).
// Which game a block takes never changes a result.
__device__ __forceinline__ uint32_t depth_class(const uint4 &rv)         // 0 deepest search .. 2 shallowest
{
    const uint32_t root_empty = count_empty(Board{{rv.x, rv.y, rv.z, rv.w}});
    return root_empty >= 10u ? 2u : root_empty <= 4u ? 1u : 0u;
}

__global__ __launch_bounds__(1024) void beam_order_kernel(const uint4 *__restrict__ roots, uint32_t *__restrict__ order, uint32_t n,
                                                         int depth, uint32_t row_len)
{
    __shared__ uint32_t s_count[3], s_cursor[3];
    constexpr int kHeld = 8;                                 // classes of the first 8192 games stay in registers
    // cost order of the classes: depth for 5..9 empty cells, min(depth + 5, 25) for <= 4, min(depth - 5, 10) for >= 10
    const int d0 = depth, d1 = min(depth + 5, 25), d2 = min(depth - 5, 10);
    // rank of each class when sorted by cost, heaviest first (ties keep class order)
    const uint32_t r0 = (d1 > d0 ? 1u : 0u) + (d2 > d0 ? 1u : 0u), r1 = (d0 >= d1 ? 1u : 0u) + (d2 > d1 ? 1u : 0u);
    const uint32_t r2 = (d0 >= d2 ? 1u : 0u) + (d1 >= d2 ? 1u : 0u);
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < 3) { s_count[threadIdx.x] = 0u; s_cursor[threadIdx.x] = 0u; }
    uint32_t held[kHeld];
#pragma unroll
    for (int k = 0; k < kHeld; ++k) {                        // independent loads: one memory round trip
        const uint32_t i = (uint32_t)k * 1024u + threadIdx.x;
        held[k] = i < n ? depth_class(roots[i]) : 3u;
    }
    // per-wavefront counts of the three classes, by ballots
    uint32_t wc0 = 0, wc1 = 0, wc2 = 0;
    auto tally = [&](uint32_t cls) {
        wc0 += (uint32_t)__popcll(__ballot(cls == 0u)); wc1 += (uint32_t)__popcll(__ballot(cls == 1u));
        wc2 += (uint32_t)__popcll(__ballot(cls == 2u));
    };
#pragma unroll
    for (int k = 0; k < kHeld; ++k) if ((uint32_t)k * 1024u < n) tally(held[k]);
    for (uint32_t i0 = kHeld * 1024u; i0 < n; i0 += 1024u) {
        const uint32_t i = i0 + threadIdx.x;
        tally(i < n ? depth_class(roots[i]) : 3u);
    }
    __syncthreads();
    const uint32_t mine = lane == 0 ? wc0 : lane == 1 ? wc1 : wc2;
    if (lane < 3) atomicAdd(&s_count[lane], mine);
    __syncthreads();
    // first rank of each class = the games in heavier classes; this wavefront's range inside each class
    const uint32_t c0 = s_count[0], c1 = s_count[1], c2 = s_count[2];
    const uint32_t start0 = (r1 < r0 ? c1 : 0u) + (r2 < r0 ? c2 : 0u), start1 = (r0 < r1 ? c0 : 0u) + (r2 < r1 ? c2 : 0u);
    const uint32_t start2 = (r0 < r2 ? c0 : 0u) + (r1 < r2 ? c1 : 0u);
    uint32_t base = 0u;
    if (lane < 3) base = atomicAdd(&s_cursor[lane], mine);
    uint32_t at0 = start0 + (uint32_t)__builtin_amdgcn_readlane((int)base, 0), at1 = start1 + (uint32_t)__builtin_amdgcn_readlane((int)base, 1);
    uint32_t at2 = start2 + (uint32_t)__builtin_amdgcn_readlane((int)base, 2);
    const uint32_t rows = (n + row_len - 1u) / row_len;
    auto place = [&](uint32_t i, uint32_t cls) {
        const unsigned long long m0 = __ballot(cls == 0u), m1 = __ballot(cls == 1u), m2 = __ballot(cls == 2u);
        const uint32_t r = cls == 0u ? at0 + prefix_count(m0) : cls == 1u ? at1 + prefix_count(m1) : at2 + prefix_count(m2);
        at0 += (uint32_t)__popcll(m0); at1 += (uint32_t)__popcll(m1); at2 += (uint32_t)__popcll(m2);
        const uint32_t row = r / row_len, q = r - row * row_len, len = row + 1u == rows ? n - row * row_len : row_len;
        if (cls < 3u) order[row * row_len + ((row & 1u) ? len - 1u - q : q)] = i;
    };
#pragma unroll
    for (int k = 0; k < kHeld; ++k) if ((uint32_t)k * 1024u < n) place((uint32_t)k * 1024u + threadIdx.x, held[k]);
    for (uint32_t i0 = kHeld * 1024u; i0 < n; i0 += 1024u) {
        const uint32_t i = i0 + threadIdx.x;
        place(i, i < n ? depth_class(roots[i]) : 3u);
    }
}

// ---------------------------------------------------------------- the order of the previous call
// beam_order_kernel costs a launch of its own in front of every batch (4.9 us of a 104 us call at 4096 games). Callers that
// search batch after batch (an evaluation loop, the benchmark) can do without it: every block of call e drops its game into a
// per-class list (one atomic ticket at its start, one store at its end), and call e + 1 deals the games from those lists --
// the order of the PREVIOUS batch's roots. Any permutation gives the same results; this one is balanced to the extent that a
// game's depth class survives one move (always, in the benchmark; nearly always, in an evaluation loop). The caller passes
// the history buffer (zero-filled before its first use and whenever a call is not the successor of the last one that used it)
// and a call index that counts up from 1.
// Tickets: 4096 blocks asking one counter for a ticket at the same moment queue for ~10 ns each -- and the other memory
// requests of their compute unit queue behind them (a quarter of the blocks started up to 45 us late, measured). So every
// class has kHistSubs counters, chosen by the block index, each in a 128-byte line of its own (~100 tickets per counter),
// and a class's list is kHistSubs segments of ceil(n / kHistSubs) entries.
// Layout (32-bit words): two headers {magic, n, call} that alternate (a line each); three counter sets of 3 x kHistSubs
// lines that rotate (this call counts in one, reads the previous one, clears the next); two sets of lists.
constexpr uint32_t kHistMagic = 0x32303438u, kHistLine = 32u, kHistSubs = 16u, kHistHdr = kHistLine;
constexpr uint32_t kHistCnt = 2u * kHistLine, kHistSet = 3u * kHistSubs * kHistLine, kHistLists = kHistCnt + 3u * kHistSet;
__host__ __device__ constexpr uint32_t hist_seg(uint32_t n) { return (n + kHistSubs - 1u) >> 4; }
constexpr size_t hist_words(size_t n) { return kHistLists + 2u * 3u * kHistSubs * (size_t)hist_seg((uint32_t)n); }

__device__ __forceinline__ size_t hist_lookup(const uint32_t *hist, uint32_t call_index, uint32_t n, int depth, uint32_t row_len,
                                              uint32_t b)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t *hp = hist + ((call_index - 1u) & 1u) * kHistHdr;                 // written by the previous call, not by this one
    const uint32_t *cp = hist + kHistCnt + ((call_index - 1u) % 3u) * kHistSet;
    const uint32_t magic = uniform(hp[0]), hn = uniform(hp[1]), hc = uniform(hp[2]);
    // lane 16 c + s holds the tickets given out by counter s of class c; inclusive sums along each row of 16 lanes
    uint32_t v = lane < 3u * kHistSubs ? cp[lane * kHistLine] : 0u;
    v = v > n ? n + 1u : v;                                                           // (garbage cannot wrap the sums)
    uint32_t incl = v;                                                                // (row_shr shifts zeros in at a row's start)
    incl += dpp_of<0x111>(incl); incl += dpp_of<0x112>(incl); incl += dpp_of<0x114>(incl); incl += dpp_of<0x118>(incl);
    static_assert(kHistSubs == 16u, "one DPP row per class");
    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15), c1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
    const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 47);
    if (magic != kHistMagic || hn != n || hc != call_index - 1u || c0 > n || c1 > n || c2 > n || c0 + c1 + c2 != n) return b;
    // cost order of the classes as in beam_order_kernel: heaviest first, ties keep class order
    const int d0 = depth, d1 = min(depth + 5, 25), d2 = min(depth - 5, 10);
    const uint32_t r0 = (d1 > d0 ? 1u : 0u) + (d2 > d0 ? 1u : 0u), r1 = (d0 >= d1 ? 1u : 0u) + (d2 > d1 ? 1u : 0u);
    const uint32_t r2 = (d0 >= d2 ? 1u : 0u) + (d1 >= d2 ? 1u : 0u);
    const uint32_t start0 = (r1 < r0 ? c1 : 0u) + (r2 < r0 ? c2 : 0u), start1 = (r0 < r1 ? c0 : 0u) + (r2 < r1 ? c2 : 0u);
    const uint32_t start2 = (r0 < r2 ? c0 : 0u) + (r1 < r2 ? c1 : 0u);
    // block b sits in row b / row_len of the deal; odd rows run backwards
    const bool pow2 = (row_len & (row_len - 1u)) == 0u;                                // (4 x CUs: a shift on every current part)
    const uint32_t sh = 31u - (uint32_t)__builtin_clz(row_len | 1u);
    const uint32_t rows = pow2 ? (n + row_len - 1u) >> sh : (n + row_len - 1u) / row_len, row = pow2 ? b >> sh : b / row_len;
    const uint32_t q1 = b - row * row_len;
    const uint32_t len = row + 1u == rows ? n - row * row_len : row_len;
    const uint32_t r = row * row_len + ((row & 1u) ? len - 1u - q1 : q1);
    uint32_t cls, idx;
    if (r >= start0 && r - start0 < c0) { cls = 0u; idx = r - start0; }
    else if (r >= start1 && r - start1 < c1) { cls = 1u; idx = r - start1; }
    else { cls = 2u; idx = r - start2; }
    // the segment of class cls that holds its idx-th game: the first counter whose inclusive sum exceeds idx
    const unsigned long long over = __ballot((lane >> 4) == cls && lane < 3u * kHistSubs && incl > idx);
    if (!over) return b;
    const uint32_t at = (uint32_t)__builtin_ctzll(over);                              // lane 16 cls + sub
    const uint32_t before = uniform((uint32_t)__shfl((int)(incl - v), (int)at, 64));   // (uniform: the search must not see a per-lane game id)
    const uint32_t k = idx - before, seg = hist_seg(n);
    if (k >= seg) return b;
    const uint32_t g = uniform(hist[kHistLists + ((size_t)((call_index - 1u) & 1u) * 3u * kHistSubs + at) * seg + k]);
    return g < n ? g : b;
}

template <int PASSES>
__global__ __launch_bounds__(64) void beam_kernel(const uint4 *__restrict__ roots, const uint8_t *__restrict__ mask_in,
                                                 uint8_t *__restrict__ action_out, float *__restrict__ prob_out,
                                                 uint32_t *__restrict__ expanded_out, int width, int depth,
                                                 uint32_t early_thr, uint32_t mid_thr, uint32_t k0, uint32_t k1,
                                                 uint64_t id_base, uint32_t flags, const uint32_t *__restrict__ keyblock,
                                                 const uint32_t *__restrict__ order, uint32_t *hist, uint32_t call_index,
                                                 uint32_t row_len)
{
    if (keyblock) { k0 = keyblock[4]; k1 = keyblock[5]; }            // KB_BEAM of the device key block
    __shared__ BeamShared<PASSES> sh;
    size_t g = order ? (size_t)order[blockIdx.x] : (size_t)blockIdx.x;
    const uint32_t n = gridDim.x;
    if (hist) g = uniform((uint32_t)hist_lookup(hist, call_index, n, depth, row_len, blockIdx.x));
    const uint4 rv = roots[g];
    const Board root = {{rv.x, rv.y, rv.z, rv.w}};
    // this game's place in the order of the NEXT call: a ticket inside its cost class now, the list entry when the search is done
    const int mask_arg = mask_in ? (int)(mask_in[g] & 15u) : -1;                // (loaded before the ticket is asked for: vmcnt counts in order)
    const uint32_t hist_cls = hist ? uniform(depth_class(rv)) : 0u;             // (scalar: nothing but the ticket is held in a vector register)
    uint32_t hist_idx = 0u;
    if (hist && threadIdx.x == 0) {
        // The ticket is not needed before the search is over. The word offset goes through an opaque vector register: with an
        // address it can prove uniform the compiler's atomic optimizer rewrites this into count-the-lanes + one atomic +
        // s_waitcnt + v_readfirstlane, i.e. it waits at once -- 4096 blocks then queue on three counters for 48 us.
        uint32_t woff = kHistCnt + (call_index % 3u) * kHistSet + (hist_cls * kHistSubs + (blockIdx.x & (kHistSubs - 1u))) * kHistLine;
        asm volatile("" : "+v"(woff));
        hist_idx = atomicAdd(&hist[woff], 1u);
    }
    if (hist && blockIdx.x == 0 && threadIdx.x < 3u * kHistSubs)     // the counters the call after the next one will count in
        hist[kHistCnt + ((call_index + 1u) % 3u) * kHistSet + threadIdx.x * kHistLine] = 0u;
    // what the end of the block needs of all this waits in LDS, not in scalar registers that would be spilled inside the level loop
    __shared__ unsigned long long s_hist_ptr[2];
    __shared__ uint32_t s_hist[4];
    if (hist && threadIdx.x == 0) {
        const uint32_t seg = hist_seg(n);
        const size_t at = kHistLists + ((size_t)(call_index & 1u) * 3u * kHistSubs + hist_cls * kHistSubs + (blockIdx.x & (kHistSubs - 1u))) * seg;
        s_hist_ptr[0] = (unsigned long long)(uintptr_t)(hist + at);
        s_hist_ptr[1] = (unsigned long long)(uintptr_t)(hist + (call_index & 1u) * kHistHdr);
        s_hist[0] = blockIdx.x == 0 ? 1u : 0u; s_hist[1] = seg; s_hist[2] = call_index; s_hist[3] = n;
    }
    const bool with_hist = hist != nullptr;
    asm volatile("" ::: "memory");
    [[maybe_unused]] const unsigned long long tick0 = kBeamTiming ? wall_clock64() : 0ull;       // (measurement builds only: when and where each search ran)
    const Decision d = beam_decide<PASSES>(sh, root, mask_arg, width, depth, early_thr, mid_thr, k0, k1, id_base + g, flags);
    if (with_hist && threadIdx.x == 0) {
        const uint32_t seg = s_hist[1];
        if (hist_idx < seg) reinterpret_cast<uint32_t *>((uintptr_t)s_hist_ptr[0])[hist_idx] = (uint32_t)g;
        if (s_hist[0]) {                                             // block 0 (only read by the next call, i.e. after this kernel)
            uint32_t *h = reinterpret_cast<uint32_t *>((uintptr_t)s_hist_ptr[1]);
            h[0] = kHistMagic; h[1] = s_hist[3]; h[2] = s_hist[2];
        }
    }
    if (threadIdx.x == 0) {
        action_out[g] = (uint8_t)d.action;
        if constexpr (kBeamTiming) {
            const uint32_t dur = (uint32_t)(wall_clock64() - tick0);
            prob_out[g] = __uint_as_float((uint32_t)tick0);
            if (expanded_out) expanded_out[g] = (dur < 0x3ffffu ? dur : 0x3ffffu) | (simd_id() << 18);
        } else {
            prob_out[g] = d.prob;
            if (expanded_out) expanded_out[g] = d.expanded;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's evaluation loop (run_evaluation.py:48-69, evaluate_beam_search.py:16-98) fused per game: the
// wavefront that owns a game alternates get_action (above) and Game2048Env.step until the game is over or the move cap
// is reached, with the per-move bookkeeping (milestones, valid / invalid counters) in registers. No launch, no host,
// no other game is involved between two moves; draws are the ones the step-by-step driver uses -- move t of game g
// takes (seed, BEAM, t, g, j) for the search and (seed, STEP, t, g) for the spawn -- so the results are identical.
// Per-game bookkeeping of the evaluation loop (evaluate_beam_search.py:48-69), identical in every lane and kept in vector
// registers. Two other homes for it were built and measured (profiles/r03_beam_priority.txt, section 8): scalar registers
// (every step output through v_readfirstlane: play_spec_kernel<2> 140 -> 102 vector registers, four wavefronts per SIMD
// instead of three, but 126 spilled scalars) and LDS (lane 0 updates a 52-byte record once per move: 106 registers) -- both
// 0.200 s for the 4096-game evaluation against 0.1975 s like this, so the plain form stays.
struct GameState {
    Board b;
    uint32_t sc;
    int32_t ms[8];
    int32_t nvalid, ninvalid, t;
    unsigned long long expanded;
    bool alive;
};

// actions: this game's row of the caller's action stream (one byte per move, written as the move is applied -- by the owner,
// whoever searched it), or nullptr
__device__ __forceinline__ void game_apply(GameState &st, const StepOut &o, uint32_t expanded, uint32_t action, uint8_t *actions)
{
    if (actions && threadIdx.x == 0) actions[st.t] = (uint8_t)action;
    st.b = o.board;
    st.sc += o.gain;
    st.expanded += expanded;
    const int32_t maxcode = (int32_t)(o.flags >> G2048_FLAG_MAXCODE_SHIFT);
#pragma unroll
    for (int k = 0; k < 8; ++k) if (st.ms[k] < 0 && maxcode >= 6 + k) st.ms[k] = st.t;     // evaluate_beam_search.py:60-64
    if (o.flags & G2048_FLAG_VALID) ++st.nvalid; else ++st.ninvalid;
    st.alive = !(o.flags & G2048_FLAG_DONE);
    ++st.t;
}

__device__ __forceinline__ void game_store(const GameState &st, size_t g, uint4 *boards, uint32_t *score, int32_t *moves_out,
                                           int32_t *valid_out, int32_t *invalid_out, int4 *milestone_out,
                                           unsigned long long *expanded_out, uint8_t *alive_out)
{
    boards[g] = make_uint4(st.b.w[0], st.b.w[1], st.b.w[2], st.b.w[3]);
    score[g] = st.sc;
    moves_out[g] = st.t; valid_out[g] = st.nvalid; invalid_out[g] = st.ninvalid;
    milestone_out[2 * g] = make_int4(st.ms[0], st.ms[1], st.ms[2], st.ms[3]);
    milestone_out[2 * g + 1] = make_int4(st.ms[4], st.ms[5], st.ms[6], st.ms[7]);
    if (expanded_out) expanded_out[g] = st.expanded;
    alive_out[g] = st.alive ? 1 : 0;
}

// One wavefront plays one game from start to finish.
template <int PASSES>
__global__ __launch_bounds__(64) void play_kernel(uint4 *__restrict__ boards, uint32_t *__restrict__ score,
                                                 int32_t *__restrict__ moves_out, int32_t *__restrict__ valid_out,
                                                 int32_t *__restrict__ invalid_out, int4 *__restrict__ milestone_out,
                                                 unsigned long long *__restrict__ expanded_out, uint8_t *__restrict__ alive_out,
                                                 int width, int depth, uint32_t early_thr, uint32_t mid_thr, int max_moves,
                                                 uint64_t seed, uint64_t id_base, uint32_t flags, uint8_t *__restrict__ actions_out)
{
    __shared__ BeamShared<PASSES> sh;
    const size_t g = blockIdx.x;
    const uint64_t gid = id_base + g;
    uint8_t *const my_actions = actions_out ? actions_out + g * (size_t)max_moves : nullptr;
    const uint4 rv = boards[g];
    GameState st = {Board{{rv.x, rv.y, rv.z, rv.w}}, score[g], {-1, -1, -1, -1, -1, -1, -1, -1}, 0, 0, 0, 0ull, true};
    while (st.t < max_moves && st.alive) {
        const Keys kb = rng_keys(seed, DOM_BEAM, (uint64_t)st.t), ks = rng_keys(seed, DOM_STEP, (uint64_t)st.t);
        const Decision d = beam_decide<PASSES>(sh, st.b, -1, width, depth, early_thr, mid_thr, kb.k0, kb.k1, gid, flags);
        game_apply(st, step_board(st.b, d.action, rng_draw(ks.k0, ks.k1, gid, 0u)), d.expanded, d.action, my_actions);
    }
    if (threadIdx.x == 0) game_store(st, g, boards, score, moves_out, valid_out, invalid_out, milestone_out, expanded_out, alive_out);
}

// ---------------------------------------------------------------- speculative helpers
// A game is a chain of decisions, ~80 us each on a lone wavefront, and the evaluation ends with a long tail of few games
// (the ones the reference's DOWN quirk keeps stuck until the move cap, and the long good ones) on an almost idle chip.
// The env's draws are counter-based, so the owner of a game knows, before it searches move t, every board move t + 1
// can start from: the successor of each valid action (with the spawn of move t) or, after an invalid move, the same
// board again -- and then again at t + 2, ... The same launch therefore carries helper wavefronts (the blocks after the
// last game; they are dispatched last, i.e. when games have ended and SIMDs are free). A game whose owner has registered
// it gets kSpec request slots, each served by one helper: the owner posts (root, move index) pairs, searches move t
// itself, steps, and if the board it now has is one it posted for t + 1 it takes that helper's decision instead of
// searching again, and so on down the chain. A helper's decision is beam_decide on exactly the root and draws the owner
// would have used, so the games are the same with or without helpers (tests/test_gpu_evaluate.py); only the time changes.
// Owners never wait for helpers beyond a bounded poll of a posted result, helpers leave when every game is resolved.
// Occupancy is left to the compiler: forced to five or six wavefronts per SIMD the helper wavefronts are resident from the
// start and their speculation competes with 4096 running games -- 0.22 s / 0.25 s against 0.20 s; capping the resident blocks
// with unused LDS 0.21 - 0.23 s (profiles/r03_beam_priority.txt, section 8).
constexpr int kHelperPollSleep = 16;     // x 64 cycles between two looks of a helper at its slot (4, 8, 32: no effect measured)
constexpr int kPrioOwner = 3, kPrioHelper = 3;       // s_setprio of a registered game's owner and of the helpers (ordinary games: 0)
constexpr int kSpec = (int)kSpecSlotsPerGame;
constexpr size_t kSpecMaxGames = 1u << 16;           // beyond this the workspace is not worth it: one wavefront per game
constexpr uint32_t kNone = 0xffffffffu;

struct SpecSlot {                 // 64 bytes per (game, slot), in the workspace g2048_play_games allocates
    uint32_t board[4], t;         // request: search this root with the draws of move t ...
    uint32_t seq;                 // ... request number (the owner's round), stored last (release); kNone: the game is over
    unsigned long long res;       // the helper's answer in ONE word: request number << 32 | children generated << 2 | action -- one store,
                                  // one load per poll (three words before: a second round trip for every result an owner took)
    uint32_t bound;               // a helper serves this slot
    uint32_t pad[7];
};
static_assert(sizeof(SpecSlot) == 64 && offsetof(SpecSlot, res) % 8 == 0, "one slot per 64-byte line, the result word aligned");
struct SpecCtl { uint32_t resolved, registered, next_unit, started; };     // started: owner blocks that have begun

__device__ __forceinline__ uint32_t ld_acquire(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_release(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_relaxed(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int PASSES>
__device__ void spec_helper(BeamShared<PASSES> &sh, SpecCtl *ctl, const uint32_t *reg_list, SpecSlot *slots, uint32_t n_games,
                            int width, int depth, uint32_t early_thr, uint32_t mid_thr, uint64_t seed, uint64_t id_base,
                            uint32_t flags)
{
    for (;;) {
        uint32_t u = 0;
        if (threadIdx.x == 0) u = atomicAdd(&ctl->next_unit, 1u);
        u = uniform(u);
        const uint32_t ticket = u / kSpec, k = u % kSpec;
        if (ticket >= n_games) return;
        uint32_t g;
        // until the ticket's game exists, or no game is left. A helper only idles here while every owner block is on the chip:
        // should the dispatcher have placed helpers while owners still wait for a slot (a partitioned or smaller device, an
        // out-of-order dispatch), it gives its slot back after a bounded wait -- owners never depend on helpers.
        for (uint32_t idle = 0;; ++idle) {
            g = uniform(ld_acquire(&reg_list[ticket]));
            if (g != kNone) break;
            if (uniform(ld_relaxed(&ctl->resolved)) >= n_games) return;
            if (idle >= 64u && uniform(ld_relaxed(&ctl->started)) < n_games) return;       // ~64 x 3.4 us
            for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);
        }
        SpecSlot *slot = slots + (size_t)g * kSpec + k;
        if (threadIdx.x == 0) st_relaxed(&slot->bound, 1u);
        uint32_t last = 0u;
        for (;;) {
            const uint32_t q = uniform(ld_acquire(&slot->seq));
            if (q == kNone) break;                                   // every owner closes its slots when its game ends
            if (q == last) { __builtin_amdgcn_s_sleep(kHelperPollSleep); continue; }
            const Board root = {{uniform(ld_relaxed(&slot->board[0])), uniform(ld_relaxed(&slot->board[1])),
                                 uniform(ld_relaxed(&slot->board[2])), uniform(ld_relaxed(&slot->board[3]))}};
            const uint32_t t = uniform(ld_relaxed(&slot->t));
            // (a payload torn by the owner moving on carries the old q, which the owner no longer accepts)
            const Keys kb = rng_keys(seed, DOM_BEAM, (uint64_t)t);
            const Decision d = beam_decide<PASSES>(sh, root, -1, width, depth, early_thr, mid_thr, kb.k0, kb.k1, id_base + g,
                                                   flags);
            if (threadIdx.x == 0)
                __hip_atomic_store(&slot->res, ((unsigned long long)q << 32) | ((unsigned long long)d.expanded << 2) | (d.action & 3u),
                                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            last = q;
        }
    }
}

template <int PASSES>
__global__ __launch_bounds__(64) void play_spec_kernel(uint4 *__restrict__ boards, uint32_t *__restrict__ score,
                                                      int32_t *__restrict__ moves_out, int32_t *__restrict__ valid_out,
                                                      int32_t *__restrict__ invalid_out, int4 *__restrict__ milestone_out,
                                                      unsigned long long *__restrict__ expanded_out,
                                                      uint8_t *__restrict__ alive_out, int width, int depth, uint32_t early_thr,
                                                      uint32_t mid_thr, int max_moves, uint64_t seed, uint64_t id_base,
                                                      uint32_t flags, uint8_t *__restrict__ actions_out, SpecCtl *ctl, uint32_t *reg_list,
                                                      SpecSlot *slots, uint32_t n_games, int stuck_thr, uint32_t reg_resolved,
                                                      uint32_t wait_ticks)
{
    __shared__ BeamShared<PASSES> sh;
    __shared__ uint4 s_req_board[kSpec];
    __shared__ uint32_t s_req_t[kSpec];
    if (blockIdx.x >= n_games) {
        __builtin_amdgcn_s_setprio(kPrioHelper);
        spec_helper<PASSES>(sh, ctl, reg_list, slots, n_games, width, depth, early_thr, mid_thr, seed, id_base, flags);
        return;
    }
    const uint32_t lane = threadIdx.x;
    const size_t g = blockIdx.x;
    const uint64_t gid = id_base + g;
    const uint4 rv = boards[g];
    GameState st = {Board{{rv.x, rv.y, rv.z, rv.w}}, score[g], {-1, -1, -1, -1, -1, -1, -1, -1}, 0, 0, 0, 0ull, true};
    SpecSlot *const my = slots + g * kSpec;
    uint8_t *const my_actions = actions_out ? actions_out + g * (size_t)max_moves : nullptr;
    if (lane == 0) atomicAdd(&ctl->started, 1u);
    // (measurement builds only, tools/play_timeline.py: the milestone record then carries when / where the game ran)
    [[maybe_unused]] const uint32_t tm_start = kPlayTiming ? (uint32_t)wall_clock64() : 0u;
    [[maybe_unused]] int32_t tm_reg_t = -1, tm_searches = 0, tm_hits = 0, tm_late = 0;
    [[maybe_unused]] uint32_t tm_reg_tick = 0u;
    bool registered = false;
    uint32_t seq = 0u;
    int stuck = 0;                                                   // invalid moves minus valid ones, floored at 0
    while (st.t < max_moves && st.alive) {
        if (!registered && (stuck >= stuck_thr || ((st.t & 31) == 0 && uniform(ld_relaxed(&ctl->resolved)) >= reg_resolved))) {
            if (lane == 0) st_release(&reg_list[atomicAdd(&ctl->registered, 1u)], (uint32_t)g);
            registered = true;
            __builtin_amdgcn_s_setprio(kPrioOwner);         // a registered game is on the run's critical path
            if constexpr (kPlayTiming) { tm_reg_t = st.t; tm_reg_tick = (uint32_t)wall_clock64(); }
        }
        const Keys ks = rng_keys(seed, DOM_STEP, (uint64_t)st.t);
        const uint32_t draw = rng_draw(ks.k0, ks.k1, gid, 0u);
        uint32_t on = 0u;                                            // slots that hold a request of this round
        if (registered) {
            const uint32_t bound = (uint32_t)__ballot(lane < (uint32_t)kSpec && ld_relaxed(&my[lane & (kSpec - 1)].bound) != 0u);
            if (bound) {
                ++seq;
                // lanes 0..3: the env's successor of action `lane`; the valid ones that do not end the game come first,
                // then, if some action is invalid, the unchanged board at t + 1, t + 2, ...
                const StepOut c = step_board(st.b, lane & 3u, draw);
                const bool cv = lane < 4u && (c.flags & G2048_FLAG_VALID) && !(c.flags & G2048_FLAG_DONE);
                const unsigned long long vb = __ballot(cv);
                const uint32_t nv = (uint32_t)__popcll(vb), r = prefix_count(vb);
                const bool any_invalid = (uint32_t)__popcll(__ballot(lane < 4u && !(c.flags & G2048_FLAG_VALID))) != 0u;
                if (cv && r < (uint32_t)kSpec) {
                    s_req_board[r] = make_uint4(c.board.w[0], c.board.w[1], c.board.w[2], c.board.w[3]);
                    s_req_t[r] = (uint32_t)st.t + 1u;
                }
                if (lane >= nv && lane < (uint32_t)kSpec) {
                    s_req_board[lane] = make_uint4(st.b.w[0], st.b.w[1], st.b.w[2], st.b.w[3]);
                    s_req_t[lane] = any_invalid ? (uint32_t)st.t + 1u + (lane - nv) : kNone;
                }
                __syncthreads();
                const bool post = lane < (uint32_t)kSpec && ((bound >> lane) & 1u) && s_req_t[lane & (kSpec - 1)] < (uint32_t)max_moves;
                if (post) {
                    const uint4 q = s_req_board[lane];
                    SpecSlot *sl = my + lane;
                    st_relaxed(&sl->board[0], q.x); st_relaxed(&sl->board[1], q.y);
                    st_relaxed(&sl->board[2], q.z); st_relaxed(&sl->board[3], q.w);
                    st_relaxed(&sl->t, s_req_t[lane]);
                    st_release(&sl->seq, seq);
                }
                on = (uint32_t)__ballot(post);
            }
        }
        {
            const Keys kb = rng_keys(seed, DOM_BEAM, (uint64_t)st.t);
            const Decision d = beam_decide<PASSES>(sh, st.b, -1, width, depth, early_thr, mid_thr, kb.k0, kb.k1, gid, flags);
            const StepOut o = step_board(st.b, d.action, draw);
            game_apply(st, o, d.expanded, d.action, my_actions);
            stuck = (o.flags & G2048_FLAG_VALID) ? max(stuck - 1, 0) : stuck + 1;
            if constexpr (kPlayTiming) { if (registered) ++tm_searches; }
        }
        // every slot's answer word, fetched once for the round (lane k: slot k); a slot whose answer had not arrived yet is polled again below
        unsigned long long res_v = 0ull;
        if (on && lane < (uint32_t)kSpec) res_v = __hip_atomic_load(&my[lane].res, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        while (on && st.alive && st.t < max_moves) {                 // decisions the helpers have made for where we are now
            const uint4 q = s_req_board[lane & (kSpec - 1)];
            const bool hit = lane < (uint32_t)kSpec && ((on >> lane) & 1u) && s_req_t[lane & (kSpec - 1)] == (uint32_t)st.t &&
                             q.x == st.b.w[0] && q.y == st.b.w[1] && q.z == st.b.w[2] && q.w == st.b.w[3];
            const unsigned long long hb = __ballot(hit);
            if (!hb) break;
            const uint32_t k = (uint32_t)__builtin_ctzll(hb);
            const unsigned long long t0 = wall_clock64();
            uint32_t r_lo = (uint32_t)__shfl((int)(uint32_t)res_v, (int)k, 64), r_hi = (uint32_t)__shfl((int)(uint32_t)(res_v >> 32), (int)k, 64);
            r_lo = uniform(r_lo); r_hi = uniform(r_hi);
            bool ready = r_hi == seq;
            while (!ready && wall_clock64() - t0 < wait_ticks) {
                __builtin_amdgcn_s_sleep(8);
                const unsigned long long r = __hip_atomic_load(&my[k].res, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                r_lo = uniform((uint32_t)r); r_hi = uniform((uint32_t)(r >> 32));
                ready = r_hi == seq;
            }
            if constexpr (kPlayTiming) { if (!ready) ++tm_late; else ++tm_hits; }
            if (!ready) break;                                       // a late helper: search this move ourselves
            const uint32_t action = r_lo & 3u, ex = r_lo >> 2;
            const Keys k2 = rng_keys(seed, DOM_STEP, (uint64_t)st.t);
            const StepOut o = step_board(st.b, action, rng_draw(k2.k0, k2.k1, gid, 0u));
            game_apply(st, o, ex, action, my_actions);
            stuck = (o.flags & G2048_FLAG_VALID) ? max(stuck - 1, 0) : stuck + 1;
            on &= ~(1u << k);
        }
        __syncthreads();                                             // s_req_* are rewritten next round
    }
    if (lane < (uint32_t)kSpec) st_release(&my[lane].seq, kNone);
    if (lane == 0) {
        game_store(st, g, boards, score, moves_out, valid_out, invalid_out, milestone_out, expanded_out, alive_out);
        if constexpr (kPlayTiming) {
            milestone_out[2 * g] = make_int4((int)tm_start, (int)(uint32_t)wall_clock64(), tm_reg_t, (int)tm_reg_tick);
            milestone_out[2 * g + 1] = make_int4(tm_searches, tm_hits, tm_late, 0);
        }
        atomicAdd(&ctl->resolved, 1u);
    }
}

}  // namespace

extern "C" {

// defined in g2048_kernels.hip; the beam entry point reports through the same thread-local string
const char *g2048_last_error(void);
void g2048_set_last_error_(const char *msg);

// Blocks of beam_kernel<passes> the current device holds at once (occupancy x CUs), asked once per device and kernel.
static size_t beam_resident_blocks(int passes)
{
    constexpr int kDevs = 64;
    static std::atomic<uint32_t> cache[kDevs][4];                    // 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevs) { (void)hipGetLastError(); return 0; }
    const int pi = passes == 1 ? 0 : passes == 2 ? 1 : passes == 4 ? 2 : 3;
    uint32_t v = cache[dev][pi].load(std::memory_order_relaxed);
    if (v == 0u) {
        int per_cu = 0;
        const hipError_t e = pi == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, beam_kernel<1>, 64, 0)
                           : pi == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, beam_kernel<2>, 64, 0)
                           : pi == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, beam_kernel<4>, 64, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, beam_kernel<8>, 64, 0);
        if (e != hipSuccess || per_cu <= 0) { (void)hipGetLastError(); return 0; }
        v = (uint32_t)per_cu * (uint32_t)device_cus();
        cache[dev][pi].store(v, std::memory_order_relaxed);
    }
    return v;
}

static int beam_impl(const void *root_boards, const uint8_t *valid_mask_or_null, uint8_t *action_out,
                     float *prob_out, uint32_t *expanded_out_or_null, int width, int depth,
                     int early_threshold, int mid_threshold, uint64_t seed, uint64_t step_index,
                     uint64_t game_id_base, size_t n_games, uint32_t opts, void *stream, const uint32_t *keyblock,
                     uint32_t *order_ws = nullptr, uint32_t *hist = nullptr, uint32_t call_index = 0)
{
    if (n_games == 0) return G2048_OK;
    if (!root_boards || !action_out || !prob_out) { g2048_set_last_error_("g2048_beam_get_action: null pointer"); return G2048_ERR_ARG; }
    if (reinterpret_cast<uintptr_t>(root_boards) & 15u) { g2048_set_last_error_("g2048_beam_get_action: root array must be 16-byte aligned"); return G2048_ERR_ARG; }
    if (width < 1 || width > kMaxWidth) { g2048_set_last_error_("g2048_beam_get_action: width must be in 1..128"); return G2048_ERR_ARG; }
    if (opts & ~(G2048_BEAM_FIXED_DOWN | G2048_BEAM_RANK_BY_COUNTING)) { g2048_set_last_error_("g2048_beam_get_action: unknown opts"); return G2048_ERR_ARG; }
    if (n_games > 0x7fffffffu) { g2048_set_last_error_("g2048_beam_get_action: too many games for one launch"); return G2048_ERR_ARG; }
    if (early_threshold < 0 || mid_threshold < 0) { g2048_set_last_error_("g2048_beam_get_action: negative threshold"); return G2048_ERR_ARG; }
    const Keys k = rng_keys(seed, DOM_BEAM, step_index);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)n_games);
    const uint4 *roots = static_cast<const uint4 *>(root_boards);
    uint32_t fd = ((opts & G2048_BEAM_FIXED_DOWN) ? 1u : 0u) | ((opts & G2048_BEAM_RANK_BY_COUNTING) ? 2u : 0u);
    // every block resident at once (4096 games on MI355X: four of the six a SIMD holds): issue priority by remaining levels
    if (n_games <= beam_resident_blocks(width <= 16 ? 1 : width <= 32 ? 2 : width <= 64 ? 4 : 8)) fd |= kFlagPrioByRemaining;
    // with scratch for it, and a batch of at least four searches per SIMD, the blocks take the games in a depth-balanced order
    uint32_t *order = nullptr;
    const LaunchPlan plan = launch_plan((order_ws || hist) ? device_cus() : 0, 0);
    if (hist && !(n_games >= plan.order_min && n_games <= kOrderMaxGames)) hist = nullptr;
    if (order_ws && !hist && n_games >= plan.order_min && n_games <= kOrderMaxGames) {
        order = order_ws;
        hipLaunchKernelGGL(beam_order_kernel, dim3(1), dim3(1024), 0, s, roots, order, (uint32_t)n_games, depth, plan.order_row);
    }
    {
#define G2048_LAUNCH_BEAM(P) hipLaunchKernelGGL(beam_kernel<P>, grid, dim3(64), 0, s, roots, valid_mask_or_null, action_out, prob_out, \
                           expanded_out_or_null, width, depth, (uint32_t)early_threshold, (uint32_t)mid_threshold, \
                           k.k0, k.k1, game_id_base, fd, keyblock, order, hist, call_index, plan.order_row)
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

static int play_resident_per_cu(int passes)          // blocks of play_spec_kernel<passes> a compute unit holds (0: could not ask)
{
    int resident = 0;
    const hipError_t oe = passes == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, play_spec_kernel<1>, 64, 0)
                        : passes == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, play_spec_kernel<2>, 64, 0)
                        : passes == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, play_spec_kernel<4>, 64, 0)
                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, play_spec_kernel<8>, 64, 0);
    if (oe != hipSuccess) { (void)hipGetLastError(); return 0; }
    return resident;
}

static size_t play_workspace_bytes(size_t n_games)
{
    const size_t list_bytes = (n_games * sizeof(uint32_t) + 63u) & ~(size_t)63u;
    return 64u + list_bytes + n_games * kSpec * sizeof(SpecSlot);
}

// caller_ws: nullptr = allocate the helper workspace from the stream-ordered allocator (g2048_play_games)
static int play_impl(void *boards_inout, uint32_t *score_inout, int32_t *moves_out, int32_t *valid_out, int32_t *invalid_out,
                     int32_t *milestone_move_out, unsigned long long *expanded_sum_out_or_null, uint8_t *alive_out,
                     uint8_t *actions_out_or_null, int width, int depth, int early_threshold, int mid_threshold, int max_moves,
                     uint64_t seed, uint64_t game_id_base, size_t n_games, uint32_t opts, void *stream, void *caller_ws,
                     const uint32_t *tuning = nullptr)
{
    if (n_games == 0) return G2048_OK;
    if (!boards_inout || !score_inout || !moves_out || !valid_out || !invalid_out || !milestone_move_out || !alive_out) {
        g2048_set_last_error_("g2048_play_games: null pointer"); return G2048_ERR_ARG;
    }
    if ((reinterpret_cast<uintptr_t>(boards_inout) & 15u) || (reinterpret_cast<uintptr_t>(milestone_move_out) & 15u)) {
        g2048_set_last_error_("g2048_play_games: board / milestone arrays must be 16-byte aligned"); return G2048_ERR_ARG;
    }
    if (width < 1 || width > kMaxWidth || max_moves < 0 || early_threshold < 0 || mid_threshold < 0 || n_games > 0x7fffffffu ||
        (opts & ~(G2048_BEAM_FIXED_DOWN | G2048_PLAY_ONE_PHASE | G2048_BEAM_RANK_BY_COUNTING))) {
        g2048_set_last_error_("g2048_play_games: bad width / max_moves / thresholds / opts / n_games"); return G2048_ERR_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t fd = ((opts & G2048_BEAM_FIXED_DOWN) ? 1u : 0u) | ((opts & G2048_BEAM_RANK_BY_COUNTING) ? 2u : 0u);
    const int passes = width <= 16 ? 1 : width <= 32 ? 2 : width <= 64 ? 4 : 8;
    if (actions_out_or_null && max_moves > 0) {         // 0xFF = "no move": the owners overwrite one byte per move they apply
        const hipError_t me = hipMemsetAsync(actions_out_or_null, 0xff, n_games * (size_t)max_moves, s);
        if (me != hipSuccess) { g2048_set_last_error_(hipGetErrorString(me)); return G2048_ERR_HIP; }
    }
#define G2048_PLAY_ARGS static_cast<uint4 *>(boards_inout), score_inout, moves_out, valid_out, invalid_out, \
                        reinterpret_cast<int4 *>(milestone_move_out), expanded_sum_out_or_null, alive_out, width, depth, \
                        (uint32_t)early_threshold, (uint32_t)mid_threshold, max_moves, seed, game_id_base, fd, actions_out_or_null
    if ((opts & G2048_PLAY_ONE_PHASE) || n_games > kSpecMaxGames) {
        const dim3 grid((unsigned)n_games);
#define G2048_LAUNCH_PLAY(P) hipLaunchKernelGGL(play_kernel<P>, grid, dim3(64), 0, s, G2048_PLAY_ARGS)
        if (passes == 1) G2048_LAUNCH_PLAY(1);
        else if (passes == 2) G2048_LAUNCH_PLAY(2);
        else if (passes == 4) G2048_LAUNCH_PLAY(4);
        else G2048_LAUNCH_PLAY(8);
#undef G2048_LAUNCH_PLAY
    } else {
        // Helpers: eight per game for a small batch, half the games for a large one, never more than a quarter of the
        // wavefronts the device holds of this kernel at once (launch_plan: CUs x resident blocks per CU / 4 -- 768 on a whole
        // MI355X at width 20, where the kernel's 140 vector registers allow 12 blocks per CU; g2048_device_plan reports it), so
        // owners always find room whatever the dispatch order. A game registers for them once it is stuck (16 more invalid
        // than valid moves lately) or once an eighth of the games (at least 256) is left; an owner polls at most 150 us for a
        // posted result. Measured flat around these values (profiles/r02_eval_helpers.txt); g2048_play_games_tuned overrides
        // them for measurements and tests (every field clamped).
        const uint32_t n = (uint32_t)n_games;
        const LaunchPlan plan = launch_plan(device_cus(), play_resident_per_cu(passes));
        uint32_t helpers = default_helpers(n, plan.helper_cap);
        uint32_t games_left = std::max<uint32_t>(n / 8u, 256u);
        int stuck_thr = 16;
        uint32_t wait_us = 150;             // round 3 (60 before; profiles/r03_eval_helpers.txt)
        if (tuning) {
            helpers = std::min<uint32_t>(std::min<uint32_t>(tuning[0], (uint32_t)kSpec * n), plan.helper_cap);
            games_left = tuning[1];
            stuck_thr = (int)std::min<uint32_t>(std::max<uint32_t>(tuning[2], 1u), 1u << 20);
            wait_us = std::min<uint32_t>(tuning[3], 1000u);
        }
        const uint32_t reg_resolved = games_left >= n ? 0u : n - games_left;
        const size_t list_bytes = (n_games * sizeof(uint32_t) + 63u) & ~(size_t)63u;
        const size_t bytes = play_workspace_bytes(n_games);
        char *ws = static_cast<char *>(caller_ws);
        if (!ws) {
            hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&ws), bytes, s);
            if (e != hipSuccess) { g2048_set_last_error_(hipGetErrorString(e)); return G2048_ERR_HIP; }
        }
        SpecCtl *ctl = reinterpret_cast<SpecCtl *>(ws);
        uint32_t *reg_list = reinterpret_cast<uint32_t *>(ws + 64);
        SpecSlot *slots = reinterpret_cast<SpecSlot *>(ws + 64 + list_bytes);
        hipError_t me = hipMemsetAsync(ws, 0, bytes, s);
        if (me == hipSuccess) me = hipMemsetAsync(reg_list, 0xff, list_bytes, s);
        if (me != hipSuccess) {
            if (!caller_ws) (void)hipFreeAsync(ws, s);
            g2048_set_last_error_(hipGetErrorString(me)); return G2048_ERR_HIP;
        }
        const dim3 grid((unsigned)(n + helpers));
#define G2048_LAUNCH_PLAY(P) hipLaunchKernelGGL(play_spec_kernel<P>, grid, dim3(64), 0, s, G2048_PLAY_ARGS, ctl, reg_list, slots, n, \
                                                stuck_thr, reg_resolved, wait_us * 100u)
        if (passes == 1) G2048_LAUNCH_PLAY(1);
        else if (passes == 2) G2048_LAUNCH_PLAY(2);
        else if (passes == 4) G2048_LAUNCH_PLAY(4);
        else G2048_LAUNCH_PLAY(8);
#undef G2048_LAUNCH_PLAY
        if (!caller_ws) (void)hipFreeAsync(ws, s);
    }
#undef G2048_PLAY_ARGS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g2048_set_last_error_(hipGetErrorString(e)); return G2048_ERR_HIP; }
    return G2048_OK;
}

int g2048_play_games(void *boards_inout, uint32_t *score_inout, int32_t *moves_out, int32_t *valid_out, int32_t *invalid_out,
                     int32_t *milestone_move_out, unsigned long long *expanded_sum_out_or_null, uint8_t *alive_out, uint8_t *actions_out_or_null,
                     int width, int depth, int early_threshold, int mid_threshold, int max_moves, uint64_t seed,
                     uint64_t game_id_base, size_t n_games, uint32_t opts, void *stream)
{
    return play_impl(boards_inout, score_inout, moves_out, valid_out, invalid_out, milestone_move_out, expanded_sum_out_or_null,
                     alive_out, actions_out_or_null, width, depth, early_threshold, mid_threshold, max_moves, seed, game_id_base, n_games, opts, stream,
                     nullptr);
}

size_t g2048_play_games_workspace(size_t n_games)
{
    return (n_games == 0 || n_games > kSpecMaxGames) ? 0 : play_workspace_bytes(n_games);
}

int g2048_play_games_ws(void *boards_inout, uint32_t *score_inout, int32_t *moves_out, int32_t *valid_out, int32_t *invalid_out,
                        int32_t *milestone_move_out, unsigned long long *expanded_sum_out_or_null, uint8_t *alive_out, uint8_t *actions_out_or_null,
                        int width, int depth, int early_threshold, int mid_threshold, int max_moves, uint64_t seed,
                        uint64_t game_id_base, size_t n_games, uint32_t opts, void *workspace, size_t workspace_bytes,
                        void *stream)
{
    const size_t need = g2048_play_games_workspace(n_games);
    if (!workspace || need == 0) opts |= G2048_PLAY_ONE_PHASE;                  // no scratch: every game on its one wavefront
    else if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 63u)) {
        g2048_set_last_error_("g2048_play_games_ws: workspace smaller than g2048_play_games_workspace(n_games) or not 64-byte aligned");
        return G2048_ERR_ARG;
    }
    return play_impl(boards_inout, score_inout, moves_out, valid_out, invalid_out, milestone_move_out, expanded_sum_out_or_null,
                     alive_out, actions_out_or_null, width, depth, early_threshold, mid_threshold, max_moves, seed, game_id_base, n_games, opts, stream,
                     (opts & G2048_PLAY_ONE_PHASE) ? nullptr : workspace);
}

int g2048_play_games_tuned(void *boards_inout, uint32_t *score_inout, int32_t *moves_out, int32_t *valid_out, int32_t *invalid_out,
                           int32_t *milestone_move_out, unsigned long long *expanded_sum_out_or_null, uint8_t *alive_out, uint8_t *actions_out_or_null,
                           int width, int depth, int early_threshold, int mid_threshold, int max_moves, uint64_t seed,
                           uint64_t game_id_base, size_t n_games, uint32_t opts, void *workspace, size_t workspace_bytes,
                           const uint32_t *tuning4, void *stream)
{
    if (!tuning4) { g2048_set_last_error_("g2048_play_games_tuned: null tuning"); return G2048_ERR_ARG; }
    const size_t need = g2048_play_games_workspace(n_games);
    if (!workspace || need == 0) opts |= G2048_PLAY_ONE_PHASE;
    else if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 63u)) {
        g2048_set_last_error_("g2048_play_games_tuned: workspace smaller than g2048_play_games_workspace(n_games) or not 64-byte aligned");
        return G2048_ERR_ARG;
    }
    return play_impl(boards_inout, score_inout, moves_out, valid_out, invalid_out, milestone_move_out, expanded_sum_out_or_null,
                     alive_out, actions_out_or_null, width, depth, early_threshold, mid_threshold, max_moves, seed, game_id_base, n_games, opts, stream,
                     (opts & G2048_PLAY_ONE_PHASE) ? nullptr : workspace, tuning4);
}

int g2048_launch_plan(int compute_units, int resident_blocks_per_cu, size_t n_games, uint32_t *out4)
{
    if (!out4 || compute_units < 0 || resident_blocks_per_cu < 0 || n_games > 0xffffffffu) {
        g2048_set_last_error_("g2048_launch_plan: bad arguments"); return G2048_ERR_ARG;
    }
    const LaunchPlan p = launch_plan(compute_units ? compute_units : device_cus(), resident_blocks_per_cu);
    out4[0] = p.order_row; out4[1] = p.order_min; out4[2] = p.helper_cap; out4[3] = default_helpers((uint32_t)n_games, p.helper_cap);
    return G2048_OK;
}

int g2048_device_plan(int width, size_t n_games, uint32_t *out6)
{
    if (!out6 || width < 1 || width > kMaxWidth || n_games > 0xffffffffu) {
        g2048_set_last_error_("g2048_device_plan: bad arguments"); return G2048_ERR_ARG;
    }
    const int passes = width <= 16 ? 1 : width <= 32 ? 2 : width <= 64 ? 4 : 8;
    const int cus = device_cus(), resident = play_resident_per_cu(passes);
    const LaunchPlan p = launch_plan(cus, resident);
    const size_t beam_blocks = beam_resident_blocks(passes);
    out6[0] = (uint32_t)cus; out6[1] = (uint32_t)resident; out6[2] = p.helper_cap;
    out6[3] = default_helpers((uint32_t)n_games, p.helper_cap); out6[4] = (uint32_t)beam_blocks;
    out6[5] = n_games <= beam_blocks ? 1u : 0u;
    return G2048_OK;
}

int g2048_sort_selftest(uint32_t *keys_inout, const uint32_t *extra_or_null, size_t n_waves, int key_bits, void *stream)
{
    if (n_waves == 0) return G2048_OK;
    if (!keys_inout || n_waves > 0x7fffffffu || (key_bits != 32 && key_bits != 64)) {
        g2048_set_last_error_("g2048_sort_selftest: bad arguments"); return G2048_ERR_ARG;
    }
    hipLaunchKernelGGL(sort_selftest_kernel, dim3((unsigned)n_waves), dim3(64), 0, static_cast<hipStream_t>(stream), keys_inout,
                       extra_or_null, extra_or_null ? 1 : 0, key_bits == 64 ? 1 : 0);
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

size_t g2048_beam_workspace_bytes(size_t n_games)
{
    return (n_games >= launch_plan(device_cus(), 0).order_min && n_games <= kOrderMaxGames) ? n_games * sizeof(uint32_t) : 0;
}

int g2048_beam_get_action_ws(const void *root_boards, const uint8_t *valid_mask_or_null, uint8_t *action_out,
                             float *prob_out, uint32_t *expanded_out_or_null, int width, int depth,
                             int early_threshold, int mid_threshold, uint64_t seed, uint64_t step_index,
                             uint64_t game_id_base, size_t n_games, uint32_t opts, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    const size_t need = g2048_beam_workspace_bytes(n_games);
    if (workspace && need && (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 3u))) {
        g2048_set_last_error_("g2048_beam_get_action_ws: workspace smaller than g2048_beam_workspace_bytes(n_games) or misaligned");
        return G2048_ERR_ARG;
    }
    return beam_impl(root_boards, valid_mask_or_null, action_out, prob_out, expanded_out_or_null, width, depth, early_threshold,
                     mid_threshold, seed, step_index, game_id_base, n_games, opts, stream, nullptr,
                     need ? static_cast<uint32_t *>(workspace) : nullptr);
}

size_t g2048_beam_history_bytes(size_t n_games)
{
    return (n_games >= launch_plan(device_cus(), 0).order_min && n_games <= kOrderMaxGames) ? hist_words(n_games) * sizeof(uint32_t) : 0;
}

int g2048_beam_get_action_hist(const void *root_boards, const uint8_t *valid_mask_or_null, uint8_t *action_out,
                               float *prob_out, uint32_t *expanded_out_or_null, int width, int depth,
                               int early_threshold, int mid_threshold, uint64_t seed, uint64_t step_index,
                               uint64_t game_id_base, size_t n_games, uint32_t opts, void *history, size_t history_bytes,
                               uint32_t call_index, void *stream)
{
    const size_t need = g2048_beam_history_bytes(n_games);
    if (history && need && (history_bytes < need || (reinterpret_cast<uintptr_t>(history) & 3u) || call_index == 0u)) {
        g2048_set_last_error_("g2048_beam_get_action_hist: history smaller than g2048_beam_history_bytes(n_games), misaligned, or call_index 0");
        return G2048_ERR_ARG;
    }
    return beam_impl(root_boards, valid_mask_or_null, action_out, prob_out, expanded_out_or_null, width, depth, early_threshold,
                     mid_threshold, seed, step_index, game_id_base, n_games, opts, stream, nullptr, nullptr,
                     (history && need) ? static_cast<uint32_t *>(history) : nullptr, call_index);
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
