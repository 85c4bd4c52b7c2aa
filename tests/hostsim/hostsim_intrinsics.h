// hostsim_intrinsics.h -- TEST HARNESS ONLY: portable C++ stand-ins for the two gfx950 builtins the kernels' board
// arithmetic (csrc/g2048_board.h) uses, so that header can be compiled for the host CPU by the tests. The device
// build uses the real instructions; g2048_selftest checks on the GPU that they behave as modelled here.
#pragma once
#include <stdint.h>

// v_perm_b32: the eight bytes {s0:s1} (s1 = bytes 0..3, s0 = bytes 4..7) picked by the selector bytes; selector
// 0x0c yields 0x00, 0x0d and above 0xff
static inline uint32_t hostsim_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const uint64_t v = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = (sel >> (8 * k)) & 0xffu;
        const uint32_t byte = b < 8 ? (uint32_t)(v >> (8 * b)) & 0xffu : (b == 12 ? 0u : 0xffu);
        r |= byte << (8 * k);
    }
    return r;
}

// v_dot4_u32_u8: c + sum over the four bytes of a[i] * b[i]
static inline uint32_t hostsim_udot4(uint32_t a, uint32_t b, uint32_t c)
{
    for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
    return c;
}

#define G2048_PERM(s0, s1, sel) hostsim_perm((s0), (s1), (sel))
#define G2048_UDOT4(a, b, c) hostsim_udot4((a), (b), (c))
