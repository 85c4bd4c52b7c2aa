// g2048_rng.h -- host-side key derivation of the counter RNG (DESIGN.md "RNG").
// (k0, k1) = f(seed, domain, index) is uniform over a launch, so it is computed
// once on the host and passed to the kernel as two scalars (SGPRs) -- or, for graph-replayed loops, once per move by
// the one-thread keys_advance kernel into a device key block the *_dyn entry points read; the per-lane
// part, rng_draw(k0, k1, id, ctr), lives in g2048_board.h.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define G2048_RNG_HD __host__ __device__ inline
#else
#define G2048_RNG_HD inline
#endif

namespace g2048 {

G2048_RNG_HD uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct Keys { uint32_t k0, k1; };

G2048_RNG_HD Keys rng_keys(uint64_t seed, uint32_t domain, uint64_t index)
{
    const uint64_t a = splitmix64(seed ^ ((uint64_t)domain * 0xD1B54A32D192ED03ull));
    const uint64_t b = splitmix64(a ^ splitmix64(index + 0x2048204820482048ull));
    return Keys{(uint32_t)b, (uint32_t)(b >> 32)};
}

// the term the high word of a board / game id contributes to a draw (g2048_board.h rng_draw: h += k1 + id_hi * 0x9E3779B1 + ...)
G2048_RNG_HD uint32_t rng_hi_term(uint64_t id) { return (uint32_t)(id >> 32) * 0x9E3779B1u; }

// ---- sampling without replacement (g2048_minibatch_gather; PPOMemory.sample, agents/ppo_agent.py:21-50) ----------------
// Sample j of a batch is transition P(j), P a keyed bijection of 0 .. n-1: a four-round Feistel network on 2h bits
// (2^2h >= n; a Feistel network is a bijection of its 2h-bit domain whatever the round function) walked until it lands
// below n -- cycle walking: the walk from j < n follows the cycle of j under that bijection, so it reaches a value below n
// (j itself at the latest) and distinct j end on distinct values. Host and device compile this one definition.
G2048_RNG_HD uint32_t minibatch_half_bits(uint64_t n)
{
    uint32_t bits = 1;
    while (bits < 64u && ((uint64_t)1 << bits) < n) ++bits;
    return (bits + 1u) / 2u;
}

G2048_RNG_HD uint32_t minibatch_round(uint32_t x, uint32_t key)
{
    uint32_t h = x ^ key;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}

G2048_RNG_HD uint64_t minibatch_index(uint64_t j, uint64_t n, uint32_t half_bits, uint32_t k0, uint32_t k1)
{
    const uint32_t mask = half_bits >= 32u ? 0xffffffffu : (1u << half_bits) - 1u;
    uint64_t x = j;
    do {
        uint32_t l = (uint32_t)(x >> half_bits) & mask, r = (uint32_t)x & mask;
        uint32_t t;
        t = l ^ (minibatch_round(r, k0) & mask); l = r; r = t;
        t = l ^ (minibatch_round(r, k1) & mask); l = r; r = t;
        t = l ^ (minibatch_round(r, k0 * 0x9E3779B1u + 1u) & mask); l = r; r = t;
        t = l ^ (minibatch_round(r, k1 * 0x85EBCA77u + 2u) & mask); l = r; r = t;
        x = ((uint64_t)l << half_bits) | r;
    } while (x >= n);
    return x;
}

}  // namespace g2048
