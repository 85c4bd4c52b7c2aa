// g2048_rollout.hip -- the PPO data path around the env step, for gfx950 (C-ABI: include/g2048.h).
//
//   rollout_step_kernel   one PPO rollout step of every env in ONE launch: masked sampling of the action from the policy's
//                         probabilities (agents/ppo_agent.py:211-221), Game2048Env.step (environment/game_2048.py:170-210),
//                         then -- from the board the lane still holds in VGPRs -- the NEXT observation
//                         (PPOAgent.normalize_state, agents/ppo_agent.py:184-195) and the NEXT valid-move mask
//                         (environment/game_2048.py:69-95). One lane per env, everything the next policy call needs is
//                         written by this kernel, nothing is re-read.
//   PPOAgent.remember reward shaping (agents/ppo_agent.py:234-269) for an ORDERED batch of transitions, including its two
//   stateful terms:
//     shaping_scan_*      exclusive running maximum of "highest tile seen" (:241-246): three-kernel max-scan;
//     seen_insert_kernel  the seen_states set (:259-262) as an open-addressing device hash set keyed by the 16-byte board,
//                         each key keeping the MINIMUM transition index that presented it = its first occurrence in order;
//     shaping_apply_kernel every term of remember() in the reference's f64 order.
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>

#include "../../include/g2048.h"
#include "../../include/g2048_testing.h"
#include "g2048_board.h"
#include "g2048_rng.h"

using namespace g2048;

extern "C" void g2048_set_last_error_(const char *msg);

namespace {

int fail(int code, const char *msg)
{
    g2048_set_last_error_(msg);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[200];
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        g2048_set_last_error_(buf);
        return G2048_ERR_HIP;
    }
    return G2048_OK;
}

inline bool aligned(const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1u)) == 0; }

#include "g2048_step_table.h"

constexpr int kRolloutBlock = 128;

// ---------------------------------------------------------------- rollout step ------
// LDS: the direction selectors (g2048_board.h "direction by table") and the 18 possible observation values
// float32(code) / float32(15); every wave fills both itself (same values from every wave, own write before own read).
__global__ __launch_bounds__(kRolloutBlock) void rollout_step_kernel(
    // (the first 14 dwords -- what a wavefront needs before its loads can go out -- arrive preloaded in SGPRs: g2048/_build.py)
    size_t n, const uint4 *boards_in, const float4 *__restrict__ probs, uint32_t *__restrict__ score,
    const uint8_t *__restrict__ mask_in, const unsigned long long *__restrict__ step_counter, uint64_t seed,
    uint64_t step_index, uint64_t id_base, uint4 *boards_out, uint8_t *__restrict__ actions_out, float *__restrict__ prob_out,
    void *__restrict__ reward_out, uint8_t *__restrict__ flags_out, void *__restrict__ obs_next,
    uint8_t *__restrict__ mask_next, uint4 *__restrict__ next_boards_out, uint8_t *__restrict__ state_max_out, uint32_t opts)
{
    __shared__ uint4 s_dir[kStepTableWords / 4];             // selector words + the reward's f64 constants + the observation values
    const float *const s_obs = reinterpret_cast<const StepTable *>(s_dir)->obs;
    const TenthFromLds tenth{reinterpret_cast<const StepTable *>(s_dir)};
    // At 65,536 envs a SIMD holds ONE wavefront: whatever it waits for, nothing else runs meanwhile. So everything the wavefront
    // reads goes out at once -- the direction table's word, the lane's board / probabilities / score / mask (every lane loads:
    // the lanes past the end a clamped, valid index whose data they drop, so that no exec region separates the loads from
    // the table's LDS write below) and the step counter -- and the head of the wavefront is one memory round trip, not three
    // (table word -> LDS write; counter -> keys; lane data).
    const uint2 table_word = step_table_word();
    const size_t i = (size_t)blockIdx.x * kRolloutBlock + threadIdx.x;
    const size_t ic = i < n ? i : n - 1u;
    const uint4 pv = boards_in[ic];
    const float4 p = probs[ic];
    const uint32_t sc = score[ic];
    const uint32_t mask_loaded = mask_in ? (uint32_t)mask_in[ic] : 0u;
    // keys of this step: uniform, derived on the scalar unit from (seed, domain, step index); the index may come from a
    // device counter so that a captured hipGraph of a whole rollout can be replayed
    const uint64_t index = step_index + (step_counter ? (uint64_t)*step_counter : 0ull);
    step_table_store(s_dir, table_word);
    const Keys kp = rng_keys(seed, DOM_POLICY, index), ks = rng_keys(seed, DOM_STEP, index), ke = rng_keys(seed, DOM_EPISODE, index);
    const uint64_t id = id_base + i;
    const Board prev = {{pv.x, pv.y, pv.z, pv.w}};
    const uint32_t mask = mask_in ? mask_loaded : valid_mask_env(prev);
    float pa;
    const uint32_t a = sample_action(p.x, p.y, p.z, p.w, mask, rng_draw(kp.k0, kp.k1, id, 0u), pa);
    const uint4 si = s_dir[2u * a], so = s_dir[2u * a + 1u];
    const StepOut o = step_board_sel(prev, DirSel{si.x, si.y, si.z, si.w, so.x, so.y, so.z, so.w}, rng_draw(ks.k0, ks.k1, id, 0u), tenth);
    if (i >= n) return;                                      // (the lanes past the end computed on their clamped loads; they store nothing)
    if (next_boards_out) next_boards_out[i] = make_uint4(o.board.w[0], o.board.w[1], o.board.w[2], o.board.w[3]);
    if (state_max_out) state_max_out[i] = (uint8_t)max_code(prev);
    Board cur = o.board;
    uint32_t s = sc + o.gain;
    if ((opts & G2048_STEP_AUTO_RESET) && (o.flags & G2048_FLAG_DONE)) {
        cur = fresh_board(rng_draw(ke.k0, ke.k1, id, 0u), rng_draw(ke.k0, ke.k1, id, 1u));
        s = 0u;
    }
    boards_out[i] = make_uint4(cur.w[0], cur.w[1], cur.w[2], cur.w[3]);
    score[i] = s;
    actions_out[i] = (uint8_t)a;
    prob_out[i] = pa;
    if (opts & G2048_STEP_REWARD_F64) static_cast<double *>(reward_out)[i] = o.reward;
    else static_cast<float *>(reward_out)[i] = (float)o.reward;
    flags_out[i] = (uint8_t)o.flags;
    if (mask_next) mask_next[i] = (uint8_t)valid_mask_env(cur);
    if (obs_next) {
        const uint32_t kind = (opts >> G2048_ROLLOUT_OBS_SHIFT) & 3u;
        float v[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[4 * r + c] = s_obs[(cur.w[r] >> (8 * c)) & 31u];
        }
        if (kind == G2048_OBS_F32) {
            float4 *dst = static_cast<float4 *>(obs_next) + 4 * i;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = make_float4(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);
        } else {
            uint32_t h[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t lo, hi;
                if (kind == G2048_OBS_BF16) {       // finite non-negative values: round to nearest even by integer add
                    const uint32_t u0 = __float_as_uint(v[2 * k]), u1 = __float_as_uint(v[2 * k + 1]);
                    lo = (u0 + 0x7fffu + ((u0 >> 16) & 1u)) >> 16;
                    hi = (u1 + 0x7fffu + ((u1 >> 16) & 1u)) >> 16;
                } else {
                    lo = (uint32_t)__half_as_ushort(__float2half_rn(v[2 * k]));
                    hi = (uint32_t)__half_as_ushort(__float2half_rn(v[2 * k + 1]));
                }
                h[k] = lo | (hi << 16);
            }
            uint4 *dst = static_cast<uint4 *>(obs_next) + 2 * i;
            dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
            dst[1] = make_uint4(h[4], h[5], h[6], h[7]);
        }
    }
}

// ---------------------------------------------------------------- minibatch ---------
// PPOMemory.sample (agents/ppo_agent.py:21-50: batch_size distinct transitions, np.random.choice(replace=False)) and the first
// lines of PPOAgent.update (:342-354: normalize states and next states, tensors on the device) for a trajectory buffer that
// already lives in HBM. Sample j of the batch is transition P(j), P a keyed bijection of 0 .. n-1: a four-round Feistel
// network on 2h bits (2^2h >= n) walked until it lands below n -- cycle walking; the walk ends because P permutes the 2^2h
// values and j itself is below n --, so the batch is a sample WITHOUT replacement whatever its size, with no table, no sort and
// no host round trip (minibatch_index: g2048_rng.h, where tests/hostsim checks on the CPU that it is a bijection for every n).
// Four lanes per sample (one board row = one float4 of each observation per lane).
template <int OBS_KIND, bool REWARD_F64>
__global__ __launch_bounds__(256) void minibatch_kernel(const void *__restrict__ obs, const uint8_t *__restrict__ actions,
                                                       const float *__restrict__ logp, const void *__restrict__ rewards,
                                                       const uint32_t *__restrict__ next_rows, const uint8_t *__restrict__ flags,
                                                       uint64_t n, uint64_t batch, uint32_t half_bits, uint32_t k0, uint32_t k1,
                                                       float4 *__restrict__ states_out, long long *__restrict__ actions_out,
                                                       float *__restrict__ logp_out, float *__restrict__ rewards_out,
                                                       float4 *__restrict__ next_states_out, float *__restrict__ dones_out,
                                                       long long *__restrict__ indices_out)
{
    const uint64_t gidx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t j = gidx >> 2;
    const uint32_t r = (uint32_t)gidx & 3u;
    if (j >= batch) return;
    const uint64_t i = minibatch_index(j, n, half_bits, k0, k1);
    float4 st;
    if (OBS_KIND == G2048_OBS_F32) {
        st = static_cast<const float4 *>(obs)[4u * i + r];
    } else {
        const uint2 h = static_cast<const uint2 *>(obs)[4u * i + r];
        if (OBS_KIND == G2048_OBS_BF16) {
            st = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16),
                             __uint_as_float(h.y & 0xffff0000u));
        } else {
            st = make_float4(__half2float(__ushort_as_half((unsigned short)(h.x & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(h.x >> 16))),
                             __half2float(__ushort_as_half((unsigned short)(h.y & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(h.y >> 16))));
        }
    }
    states_out[gidx] = st;
    const uint32_t x = next_rows[4u * i + r];                         // PPOAgent.normalize_state (:184-195): float32(code) / float32(15)
    next_states_out[gidx] = make_float4((float)(x & 0xffu) / 15.0f, (float)((x >> 8) & 0xffu) / 15.0f,
                                        (float)((x >> 16) & 0xffu) / 15.0f, (float)(x >> 24) / 15.0f);
    if (r != 0u) return;
    actions_out[j] = (long long)actions[i];
    logp_out[j] = logp[i];
    rewards_out[j] = REWARD_F64 ? (float)static_cast<const double *>(rewards)[i] : static_cast<const float *>(rewards)[i];
    dones_out[j] = (flags[i] & G2048_FLAG_DONE) ? 1.0f : 0.0f;
    if (indices_out) indices_out[j] = (long long)i;
}

// ---------------------------------------------------------------- running maximum ---
// prev_highest[i] = max(carry, maxcode[0..i)), maxcode[i] = flags[i] >> 3 (the max log2 code after transition i).
// 4096 transitions per block: 256 threads x 16 consecutive flag bytes (one 16-byte load).
constexpr int kScanBlock = 256, kScanPerThread = 16, kScanTile = kScanBlock * kScanPerThread;

__device__ __forceinline__ uint32_t bytes_max_code(uint32_t w)
{
    const uint32_t a = (w >> 3) & 0x1fu, b = (w >> 11) & 0x1fu, c = (w >> 19) & 0x1fu, d = w >> 27;
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ uint4 load_flags16(const uint8_t *flags, size_t base, size_t n)
{
    if (base + 16 <= n) return *reinterpret_cast<const uint4 *>(flags + base);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (size_t k = 0; k < 16 && base + k < n; ++k) w[k >> 2] |= (uint32_t)flags[base + k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t block_max(uint32_t v, uint32_t *s_red)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    if ((threadIdx.x & 63u) == 0u) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    v = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    __syncthreads();
    return v;
}

__global__ __launch_bounds__(kScanBlock) void scan_tile_max_kernel(const uint8_t *__restrict__ flags, uint32_t *__restrict__ tile_max, size_t n)
{
    __shared__ uint32_t s_red[4];
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanPerThread;
    uint32_t m = 0u;
    if (base < n) {
        const uint4 f = load_flags16(flags, base, n);
        m = max(max(bytes_max_code(f.x), bytes_max_code(f.y)), max(bytes_max_code(f.z), bytes_max_code(f.w)));
    }
    m = block_max(m, s_red);
    if (threadIdx.x == 0) tile_max[blockIdx.x] = m;
}

// one block: tile_max[t] <- max(carry, tile_max[0..t)) in place, *highest <- max(carry, all)
__global__ __launch_bounds__(kScanBlock) void scan_tiles_kernel(uint32_t *tile_max, uint32_t n_tiles, uint32_t *highest_inout)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = *highest_inout;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += kScanBlock) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = t < n_tiles ? tile_max[t] : 0u;
        uint32_t inc = v;                                                  // inclusive max-scan inside the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
            if ((threadIdx.x & 63u) >= (uint32_t)off) inc = max(inc, o);
        }
        if ((threadIdx.x & 63u) == 63u) s_wave[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t before = s_carry;                                          // everything before this wave
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before = max(before, s_wave[w]);
        uint32_t excl = (uint32_t)__shfl_up((int)inc, 1);
        excl = (threadIdx.x & 63u) == 0u ? before : max(before, excl);
        if (t < n_tiles) tile_max[t] = excl;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) s_carry = max(before, inc);
        __syncthreads();
    }
    if (threadIdx.x == 0) *highest_inout = s_carry;
}

__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ tile_before,
                                                               uint8_t *__restrict__ prev_highest, size_t n)
{
    __shared__ uint32_t s_wave[4];
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanPerThread;
    uint4 f = make_uint4(0u, 0u, 0u, 0u);
    if (base < n) f = load_flags16(flags, base, n);
    const uint32_t w[4] = {f.x, f.y, f.z, f.w};
    uint32_t mine = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) mine = max(mine, bytes_max_code(w[k]));
    uint32_t inc = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
        if ((threadIdx.x & 63u) >= (uint32_t)off) inc = max(inc, o);
    }
    if ((threadIdx.x & 63u) == 63u) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = tile_before[blockIdx.x];
    for (uint32_t q = 0; q < (threadIdx.x >> 6); ++q) run = max(run, s_wave[q]);
    const uint32_t up = (uint32_t)__shfl_up((int)inc, 1);
    if ((threadIdx.x & 63u) != 0u) run = max(run, up);
    if (base >= n) return;
    uint32_t out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t o = 0u;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            o |= run << (8 * b);
            run = max(run, (w[k] >> (8 * b + 3)) & 0x1fu);
        }
        out[k] = o;
    }
    if (base + 16 <= n) {
        *reinterpret_cast<uint4 *>(prev_highest + base) = make_uint4(out[0], out[1], out[2], out[3]);
    } else {
        for (size_t k = 0; k < 16 && base + k < n; ++k) prev_highest[base + k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
    }
}

// ---------------------------------------------------------------- seen-states set ----
// Slot = 32 bytes: key (the 16-byte board), first = smallest global transition index that presented the key, state
// (0 empty, 1 being written, 2 full). Key halves are written once, by the lane that won the 0 -> 1 exchange, and are
// performed before state becomes 2.
struct SeenSlot { unsigned long long key_lo, key_hi, first; uint32_t state, pad; };
static_assert(sizeof(SeenSlot) == G2048_SEEN_SLOT_BYTES, "slot layout is part of the ABI");

__device__ __forceinline__ uint32_t board_hash(const uint4 &b)
{
    uint32_t h = b.x * 0x9E3779B1u;
    h = (h ^ (h >> 15) ^ b.y) * 0x85EBCA77u;
    h = (h ^ (h >> 13) ^ b.z) * 0xC2B2AE3Du;
    h = (h ^ (h >> 16) ^ b.w) * 0x27D4EB2Fu;
    return h ^ (h >> 15);
}

// returns the slot of `key`, inserting it if absent; in both cases first = min(first, idx)
__device__ __forceinline__ uint32_t seen_upsert(SeenSlot *table, uint32_t mask, const uint4 &key, unsigned long long idx, bool &inserted,
                                                uint32_t *overflow)
{
    const unsigned long long klo = (unsigned long long)key.x | ((unsigned long long)key.y << 32);
    const unsigned long long khi = (unsigned long long)key.z | ((unsigned long long)key.w << 32);
    uint32_t slot = board_hash(key) & mask, result = 0xffffffffu;
    inserted = false;
    // The loop condition is WAVE-UNIFORM (a ballot): a lane that has won a slot must execute its stores inside the
    // iteration it won in, because other lanes of the same wave may be waiting for exactly that slot to open. With a
    // per-lane exit the compiler is free to treat "store, then leave" as a loop-exit path, which only runs once the whole
    // wave has left the loop -- the waiting lanes then never see the slot open (observed: rehash lost 3 % of its keys to
    // the spin bound). `spins` bounds the wait on a slot another lane is still writing, so that no input can hang a launch.
    // Inside a launch every access to the table is a returning read-modify-write atomic: those execute at the device's
    // coherence point. (Loads, agent-scope sc1 loads included, are L2-served, and the eight XCDs' L2s are not coherent with
    // each other; rather than depend on which hand-off forms happen to be safe, no load is used here. Not a hot path.)
    bool done = false;
    uint32_t probes = 0, spins = 0;
    while (__ballot(!done) != 0ull) {               // wave-uniform: nobody leaves before everybody is done (see above)
        if (done) continue;
        SeenSlot *s = table + slot;
        const uint32_t st = atomicCAS(&s->state, 0u, 1u);         // 0: this lane owns the slot now
        if (st == 0u) {
            const unsigned long long o0 = atomicExch(&s->key_lo, klo), o1 = atomicExch(&s->key_hi, khi), o2 = atomicExch(&s->first, idx);
            asm volatile("s_waitcnt vmcnt(0)" :: "v"(o0), "v"(o1), "v"(o2) : "memory");   // all three performed ...
            atomicExch(&s->state, 2u);                                                    // ... before the slot opens
            inserted = true;
            result = slot;
            done = true;
        } else if (st == 2u) {
            const unsigned long long a = atomicAdd(&s->key_lo, 0ull), b = atomicAdd(&s->key_hi, 0ull);
            if (a == klo && b == khi) {
                atomicMin(&s->first, idx);
                result = slot;
                done = true;
            } else {
                slot = (slot + 1u) & mask;
                spins = 0u;
                if (++probes > mask) { atomicOr(overflow, 1u); done = true; }       // table full: reported, never hangs
            }
        } else if (++spins > (1u << 22)) {                         // st == 1 for far too long: report instead of hanging
            atomicOr(overflow, 2u);
            done = true;
        }
    }
    return result;
}

__global__ __launch_bounds__(256) void seen_insert_kernel(const uint4 *__restrict__ boards, unsigned long long index_base, SeenSlot *table,
                                                         uint32_t mask, unsigned long long *count, uint32_t *overflow,
                                                         uint32_t *__restrict__ slot_out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    bool inserted = false;
    if (i < n) {
        const uint4 key = boards[i];
        slot_out[i] = seen_upsert(table, mask, key, index_base + i, inserted, overflow);
    }
    const unsigned long long bal = __ballot(inserted);
    if ((threadIdx.x & 63u) == 0u && bal) atomicAdd(count, (unsigned long long)__popcll(bal));
}

// re-insert every full slot of an old table into a (larger, zeroed) new one, keeping its first index
__global__ __launch_bounds__(256) void seen_rehash_kernel(const SeenSlot *__restrict__ old_table, size_t old_slots, SeenSlot *new_table,
                                                         uint32_t new_mask, uint32_t *overflow)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= old_slots) return;
    const SeenSlot s = old_table[i];
    if (s.state != 2u) return;
    const uint4 key = make_uint4((uint32_t)s.key_lo, (uint32_t)(s.key_lo >> 32), (uint32_t)s.key_hi, (uint32_t)(s.key_hi >> 32));
    bool inserted;
    (void)seen_upsert(new_table, new_mask, key, s.first, inserted, overflow);
}

// ---------------------------------------------------------------- remember() ---------
// PPOAgent.remember (agents/ppo_agent.py:234-269) for transition i of the ordered batch, terms in the reference's order:
//   :241-246  new highest tile: + 5.0 * (log2(next_max) - log2(highest_tile_seen))     [running value before i]
//   :249-251  regression:       + -2.0 * (log2(current_max) - log2(next_max))            if next_max < current_max
//   :254-256  top tiles:        + 0.1 * sum(log2 of the four largest tiles of next_state)
//   :259-262  novelty:          + 0.2 if next_state was never presented before            [first occurrence in order]
//   :265-266  heuristic:        + 0.3 * evaluate_heuristic(next_state)
__global__ __launch_bounds__(256) void shaping_apply_kernel(const uint4 *__restrict__ next_boards, const uint8_t *__restrict__ state_max,
                                                           const uint8_t *__restrict__ flags, const double *__restrict__ env_reward,
                                                           const uint8_t *__restrict__ prev_highest, const SeenSlot *__restrict__ table,
                                                           const uint32_t *__restrict__ slots, unsigned long long index_base,
                                                           double *__restrict__ shaped, uint8_t *__restrict__ novel_out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 nv = next_boards[i];
    const Board nb = {{nv.x, nv.y, nv.z, nv.w}};
    const uint32_t nm = (uint32_t)flags[i] >> G2048_FLAG_MAXCODE_SHIFT, ph = prev_highest[i], cm = state_max[i];
    double r = env_reward[i];
    if (nm > ph) r += 5.0 * ((double)nm - (double)ph);
    if (nm < cm) r += -2.0 * ((double)cm - (double)nm);
    r += 0.1 * (double)top4_code_sum(nb);
    const uint32_t slot = slots[i];
    const bool novel = slot != 0xffffffffu && table[slot].first == index_base + i;
    if (novel) r += 0.2;
    r += 0.3 * eval_ppo_heuristic(nb);
    shaped[i] = r;
    if (novel_out) novel_out[i] = novel ? 1 : 0;
}

}  // namespace

// ==================================================================== C-ABI ====
extern "C" {

int g2048_rollout_step(const void *boards_in, const float *probs, const uint8_t *mask4_in_or_null, void *boards_out,
                       uint32_t *score_inout, uint8_t *actions_out, float *prob_out, void *reward_out, uint8_t *flags_out,
                       void *obs_next_out_or_null, uint8_t *mask4_next_out_or_null, void *next_boards_out_or_null,
                       uint8_t *state_maxcode_out_or_null, uint64_t seed, uint64_t step_index,
                       const unsigned long long *step_counter_or_null, uint64_t env_id_base, size_t n, uint32_t opts, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!boards_in || !probs || !boards_out || !score_inout || !actions_out || !prob_out || !reward_out || !flags_out)
        return fail(G2048_ERR_ARG, "g2048_rollout_step: null pointer");
    if (!aligned(boards_in, 16) || !aligned(boards_out, 16) || !aligned(probs, 16) || (next_boards_out_or_null && !aligned(next_boards_out_or_null, 16)))
        return fail(G2048_ERR_ARG, "g2048_rollout_step: board / probability arrays must be 16-byte aligned");
    const uint32_t kind = (opts >> G2048_ROLLOUT_OBS_SHIFT) & 3u;
    if ((opts & ~(G2048_STEP_REWARD_F64 | G2048_STEP_AUTO_RESET | (3u << G2048_ROLLOUT_OBS_SHIFT))) || kind > G2048_OBS_BF16)
        return fail(G2048_ERR_ARG, "g2048_rollout_step: unknown opts");
    if (!aligned(score_inout, 4) || !aligned(prob_out, 4) || !aligned(reward_out, (opts & G2048_STEP_REWARD_F64) ? 8 : 4) ||
        (obs_next_out_or_null && !aligned(obs_next_out_or_null, 16)) || (step_counter_or_null && !aligned(step_counter_or_null, 8)))
        return fail(G2048_ERR_ARG, "g2048_rollout_step: misaligned array");
    hipLaunchKernelGGL(rollout_step_kernel, dim3((unsigned)((n + kRolloutBlock - 1) / kRolloutBlock)), dim3(kRolloutBlock), 0,
                       static_cast<hipStream_t>(stream), n, static_cast<const uint4 *>(boards_in), reinterpret_cast<const float4 *>(probs),
                       score_inout, mask4_in_or_null, step_counter_or_null, seed, step_index, env_id_base,
                       static_cast<uint4 *>(boards_out), actions_out, prob_out, reward_out, flags_out,
                       obs_next_out_or_null, mask4_next_out_or_null, static_cast<uint4 *>(next_boards_out_or_null),
                       state_maxcode_out_or_null, opts);
    return check_launch("g2048_rollout_step");
}

int g2048_minibatch_gather(const void *obs, uint32_t obs_kind, const uint8_t *actions, const float *log_probs, const void *rewards,
                           uint32_t rewards_f64, const void *next_boards, const uint8_t *flags, size_t n_transitions, size_t batch,
                           uint64_t seed, uint64_t sample_index, float *states_out, int64_t *actions_out, float *old_log_probs_out,
                           float *rewards_out, float *next_states_out, float *dones_out, int64_t *indices_out_or_null, void *stream)
{
    if (batch == 0) return G2048_OK;
    if (!obs || !actions || !log_probs || !rewards || !next_boards || !flags || !states_out || !actions_out || !old_log_probs_out ||
        !rewards_out || !next_states_out || !dones_out)
        return fail(G2048_ERR_ARG, "g2048_minibatch_gather: null pointer");
    if (batch > n_transitions) return fail(G2048_ERR_ARG, "g2048_minibatch_gather: batch larger than the buffer (sampling is without replacement)");
    if (obs_kind > G2048_OBS_BF16) return fail(G2048_ERR_ARG, "g2048_minibatch_gather: unknown observation dtype");
    if (!aligned(obs, 16) || !aligned(next_boards, 16) || !aligned(states_out, 16) || !aligned(next_states_out, 16) ||
        !aligned(log_probs, 4) || !aligned(rewards, rewards_f64 ? 8 : 4) || !aligned(actions_out, 8) || !aligned(old_log_probs_out, 4) ||
        !aligned(rewards_out, 4) || !aligned(dones_out, 4) || (indices_out_or_null && !aligned(indices_out_or_null, 8)))
        return fail(G2048_ERR_ARG, "g2048_minibatch_gather: misaligned array");
    const uint32_t half_bits = minibatch_half_bits((uint64_t)n_transitions);
    const Keys k = rng_keys(seed, DOM_MINIBATCH, sample_index);
    if (batch > ((size_t)1 << 36)) return fail(G2048_ERR_ARG, "g2048_minibatch_gather: batch too large for one launch");
    const dim3 grid((unsigned)((batch * 4u + 255u) / 256u));
#define G2048_LAUNCH_MB(K, F) hipLaunchKernelGGL((minibatch_kernel<K, F>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), obs, actions, \
                              log_probs, rewards, static_cast<const uint32_t *>(next_boards), flags, (uint64_t)n_transitions, (uint64_t)batch, \
                              half_bits, k.k0, k.k1, reinterpret_cast<float4 *>(states_out), reinterpret_cast<long long *>(actions_out), \
                              old_log_probs_out, rewards_out, reinterpret_cast<float4 *>(next_states_out), dones_out, \
                              reinterpret_cast<long long *>(indices_out_or_null))
    if (obs_kind == G2048_OBS_F32) { if (rewards_f64) G2048_LAUNCH_MB(0, true); else G2048_LAUNCH_MB(0, false); }
    else if (obs_kind == G2048_OBS_F16) { if (rewards_f64) G2048_LAUNCH_MB(1, true); else G2048_LAUNCH_MB(1, false); }
    else { if (rewards_f64) G2048_LAUNCH_MB(2, true); else G2048_LAUNCH_MB(2, false); }
#undef G2048_LAUNCH_MB
    return check_launch("g2048_minibatch_gather");
}

size_t g2048_shaping_scan_workspace(size_t n) { return ((n + kScanTile - 1) / kScanTile + 1) * sizeof(uint32_t); }

int g2048_shaping_scan(const uint8_t *flags, uint8_t *prev_highest_out, uint32_t *highest_code_inout, void *workspace, size_t n,
                       void *stream)
{
    if (n == 0) return G2048_OK;
    if (!flags || !prev_highest_out || !highest_code_inout || !workspace) return fail(G2048_ERR_ARG, "g2048_shaping_scan: null pointer");
    if (!aligned(flags, 16) || !aligned(prev_highest_out, 16) || !aligned(workspace, 4) || !aligned(highest_code_inout, 4))
        return fail(G2048_ERR_ARG, "g2048_shaping_scan: flag arrays must be 16-byte aligned");
    const size_t tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles > 0x7fffffffu) return fail(G2048_ERR_ARG, "g2048_shaping_scan: too many transitions for one call");
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t *tile_max = static_cast<uint32_t *>(workspace);
    hipLaunchKernelGGL(scan_tile_max_kernel, dim3((unsigned)tiles), dim3(kScanBlock), 0, s, flags, tile_max, n);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kScanBlock), 0, s, tile_max, (uint32_t)tiles, highest_code_inout);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)tiles), dim3(kScanBlock), 0, s, flags, tile_max, prev_highest_out, n);
    return check_launch("g2048_shaping_scan");
}

int g2048_seen_insert(const void *next_boards, uint64_t index_base, void *table, uint32_t capacity_log2,
                      unsigned long long *count_inout, uint32_t *overflow_flag, uint32_t *slot_out, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!next_boards || !table || !count_inout || !overflow_flag || !slot_out) return fail(G2048_ERR_ARG, "g2048_seen_insert: null pointer");
    if (!aligned(next_boards, 16) || !aligned(table, 16) || !aligned(count_inout, 8) || !aligned(slot_out, 4) || !aligned(overflow_flag, 4))
        return fail(G2048_ERR_ARG, "g2048_seen_insert: misaligned array");
    if (capacity_log2 < 4 || capacity_log2 > 31) return fail(G2048_ERR_ARG, "g2048_seen_insert: capacity_log2 must be in 4..31");
    hipLaunchKernelGGL(seen_insert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(next_boards), (unsigned long long)index_base, static_cast<SeenSlot *>(table),
                       (uint32_t)((1ull << capacity_log2) - 1ull), count_inout, overflow_flag, slot_out, n);
    return check_launch("g2048_seen_insert");
}

int g2048_seen_rehash(const void *old_table, uint32_t old_capacity_log2, void *new_table, uint32_t new_capacity_log2,
                      uint32_t *overflow_flag, void *stream)
{
    if (!old_table || !new_table || !overflow_flag) return fail(G2048_ERR_ARG, "g2048_seen_rehash: null pointer");
    if (old_capacity_log2 < 4 || old_capacity_log2 > 31 || new_capacity_log2 < old_capacity_log2 || new_capacity_log2 > 31)
        return fail(G2048_ERR_ARG, "g2048_seen_rehash: bad capacities");
    if (!aligned(old_table, 16) || !aligned(new_table, 16)) return fail(G2048_ERR_ARG, "g2048_seen_rehash: misaligned table");
    const size_t slots = (size_t)1 << old_capacity_log2;
    hipLaunchKernelGGL(seen_rehash_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const SeenSlot *>(old_table), slots, static_cast<SeenSlot *>(new_table),
                       (uint32_t)((1ull << new_capacity_log2) - 1ull), overflow_flag);
    return check_launch("g2048_seen_rehash");
}

int g2048_shaping_apply(const void *next_boards, const uint8_t *state_maxcode, const uint8_t *flags, const double *env_reward,
                        const uint8_t *prev_highest, const void *table, const uint32_t *slots, uint64_t index_base,
                        double *shaped_out, uint8_t *novel_out_or_null, size_t n, void *stream)
{
    if (n == 0) return G2048_OK;
    if (!next_boards || !state_maxcode || !flags || !env_reward || !prev_highest || !table || !slots || !shaped_out)
        return fail(G2048_ERR_ARG, "g2048_shaping_apply: null pointer");
    if (!aligned(next_boards, 16) || !aligned(table, 16) || !aligned(env_reward, 8) || !aligned(shaped_out, 8) || !aligned(slots, 4))
        return fail(G2048_ERR_ARG, "g2048_shaping_apply: misaligned array");
    hipLaunchKernelGGL(shaping_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(next_boards), state_maxcode, flags, env_reward, prev_highest,
                       static_cast<const SeenSlot *>(table), slots, (unsigned long long)index_base, shaped_out, novel_out_or_null, n);
    return check_launch("g2048_shaping_apply");
}

}  // extern "C"
