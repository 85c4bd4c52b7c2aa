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

}  // namespace g2048
