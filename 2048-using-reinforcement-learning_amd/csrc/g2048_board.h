// g2048_board.h -- per-board 2048 arithmetic for CDNA4 (gfx950), one board per lane.
//
// A board is 16 x uint8 log2 codes (0 = empty, k = tile 2^k, k <= 17), row-major,
// held in four 32-bit VGPRs: w[r] = row r, cell (r,c) in bits [8c, 8c+8). One
// global_load_dwordx4 brings a board in; everything below is SWAR over those
// four registers -- byte lanes never carry into each other because codes stay
// < 0x40.
//
// The slide is done for all four lines of a direction at once: with the four
// row words w[0..3], byte lane c of the words IS column c, so "slide UP" is a
// byte-lane-parallel compaction + merge across the four words. LEFT/RIGHT run
// on the 4x4 byte transpose (8 v_perm_b32), DOWN/RIGHT on the reversed word
// order, so one code path serves all four actions and lanes of a wave never
// diverge on the action.
//
// What each routine implements (reference file:line) is stated at the routine.
// Under hipcc everything here is device code. The header is also compilable by a
// plain host compiler -- tests/hostsim does that to unit-test this exact arithmetic
// on a CPU against the oracle -- in which case the includer supplies stand-ins for
// the two gfx950 builtins through G2048_PERM / G2048_UDOT4; nothing in the product
// does.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define G2048_HD __device__ __forceinline__
#define G2048_PERM(s0, s1, sel) __builtin_amdgcn_perm((s0), (s1), (sel))
#define G2048_UDOT4(a, b, c) __builtin_amdgcn_udot4((a), (b), (c), false)
#else
#define G2048_HD inline
#if !defined(G2048_PERM) || !defined(G2048_UDOT4)
#error "host builds of g2048_board.h (test harness only) must define G2048_PERM and G2048_UDOT4"
#endif
#endif

namespace g2048 {

struct Board { uint32_t w[4]; };

constexpr uint32_t B80 = 0x80808080u;
constexpr uint32_t B7F = 0x7f7f7f7fu;

// RNG domains (DESIGN.md "RNG")
enum : uint32_t { DOM_STEP = 1, DOM_RESET = 2, DOM_BEAM = 3, DOM_SYNTH_BOARD = 4, DOM_SYNTH_ACTION = 5, DOM_EPISODE = 6, DOM_POLICY = 7,
                  DOM_SIMULATE = 8, DOM_MINIBATCH = 9 };

// ---------------------------------------------------------------- intrinsics --
// v_perm_b32: bytes of {s0:s1} (s1 = bytes 0..3, s0 = bytes 4..7) picked by the
// selector bytes; selector 0x0c yields 0x00.
G2048_HD uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return G2048_PERM(s0, s1, sel); }

G2048_HD uint32_t popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }      // v_bcnt_u32_b32

// sum over the 4 bytes of a[i]*b[i] + c  (v_dot4_u32_u8)
G2048_HD uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) { return G2048_UDOT4(a, b, c); }

// ------------------------------------------------------------------- flags ----
// 0x80 in every byte lane that is non-zero / zero.
G2048_HD uint32_t nzflag(uint32_t x) { return (x + B7F) & B80; }
G2048_HD uint32_t zflag(uint32_t x) { return ~(x + B7F) & B80; }
// 0x80 where x == y and x != 0
G2048_HD uint32_t eqnzflag(uint32_t x, uint32_t y) { return (x + B7F) & ~((x ^ y) + B7F) & B80; }
// 0x80 where x >= y (bytes < 0x80)
G2048_HD uint32_t geflag(uint32_t x, uint32_t y) { return ((x | B80) - y) & B80; }
// v_perm selector: byte lanes with the flag take s0's byte, the others s1's
G2048_HD uint32_t selof(uint32_t flag) { return 0x03020100u + (flag >> 5); }
G2048_HD uint32_t pick(uint32_t if_flag, uint32_t otherwise, uint32_t sel) { return perm(if_flag, otherwise, sel); }

G2048_HD uint32_t count_empty(const Board &b)
{
    return popc(zflag(b.w[0])) + popc(zflag(b.w[1])) + popc(zflag(b.w[2])) + popc(zflag(b.w[3]));
}

G2048_HD bool same(const Board &a, const Board &b)
{
    uint32_t d = (a.w[0] ^ b.w[0]) | (a.w[1] ^ b.w[1]) | (a.w[2] ^ b.w[2]) | (a.w[3] ^ b.w[3]);
#if defined(__HIPCC__)
    // keep this ONE integer test: left to itself the compiler sometimes turns it into four compares whose results it
    // then packs bit by bit with 16-bit shifts and ors (11 VALU instructions instead of 6)
    asm volatile("" : "+v"(d));
#endif
    return d == 0;
}

// 4x4 byte transpose: 2 x 4 v_perm_b32
G2048_HD Board transpose(const Board &b)
{
    uint32_t t0 = perm(b.w[1], b.w[0], 0x05010400u);   // r0.0 r1.0 r0.1 r1.1
    uint32_t t1 = perm(b.w[1], b.w[0], 0x07030602u);   // r0.2 r1.2 r0.3 r1.3
    uint32_t t2 = perm(b.w[3], b.w[2], 0x05010400u);   // r2.0 r3.0 r2.1 r3.1
    uint32_t t3 = perm(b.w[3], b.w[2], 0x07030602u);   // r2.2 r3.2 r2.3 r3.3
    Board o;
    o.w[0] = perm(t2, t0, 0x05040100u);                // r0.0 r1.0 r2.0 r3.0
    o.w[1] = perm(t2, t0, 0x07060302u);                // r0.1 r1.1 r2.1 r3.1
    o.w[2] = perm(t3, t1, 0x05040100u);
    o.w[3] = perm(t3, t1, 0x07060302u);
    return o;
}

G2048_HD Board rot180(const Board &b)
{
    Board o;
    o.w[0] = perm(0u, b.w[3], 0x00010203u);
    o.w[1] = perm(0u, b.w[2], 0x00010203u);
    o.w[2] = perm(0u, b.w[1], 0x00010203u);
    o.w[3] = perm(0u, b.w[0], 0x00010203u);
    return o;
}

// ------------------------------------------------------------------- slide ----
// The reference's row rule (environment/game_2048.py:116-168, identically
// agents/beam_search_agent.py:213-242): drop zeros, scan toward the far end,
// equal neighbours merge once into code+1 (score += 2^(code+1)), pad with zeros.
// Here: L[k] holds position k of four independent lines (one per byte lane),
// sliding toward k = 0. Returns the score gained by all four lines; merges = number of merge events.
G2048_HD uint32_t slide_lines(uint32_t L[4], uint32_t &merges)
{
    // compaction: three stages, stage k closes a hole at position k
    {
        uint32_t s = selof(nzflag(L[2]));
        L[2] = pick(L[2], L[3], s);
        L[3] = pick(L[3], 0u, s);
        s = selof(nzflag(L[1]));
        L[1] = pick(L[1], L[2], s);
        L[2] = pick(L[2], L[3], s);
        L[3] = pick(L[3], 0u, s);
        s = selof(nzflag(L[0]));
        L[0] = pick(L[0], L[1], s);
        L[1] = pick(L[1], L[2], s);
        L[2] = pick(L[2], L[3], s);
        L[3] = pick(L[3], 0u, s);
    }
    const uint32_t A = L[0], B = L[1], C = L[2], D = L[3];
    // merge decisions, scanning from position 0: a tile merges at most once
    const uint32_t m01 = eqnzflag(A, B);
    const uint32_t m12 = eqnzflag(B, C) & ~m01;
    const uint32_t m23 = eqnzflag(C, D) & ~m12;
    const uint32_t s01 = selof(m01), s12 = selof(m12), s23 = selof(m23);
    const uint32_t Cp = C + (m23 >> 7);
    const uint32_t o0 = A + (m01 >> 7);
    const uint32_t o1 = pick(Cp, B + (m12 >> 7), s01);
    const uint32_t o2 = pick(pick(0u, D, s23), pick(D, Cp, s12), s01);
    const uint32_t o3 = pick(0u, D, selof(m01 | m12 | m23));
    L[0] = o0; L[1] = o1; L[2] = o2; L[3] = o3;
    // score: merged tiles sit in o0 (m01) and in o1 (m01&m23 | m12) or o2 (m23 & ~m01); the last two
    // never share a byte lane (m12 excludes m23, m01&m23 excludes ~m01), so they fold into one word.
    const uint32_t g1 = (m01 & m23) | m12, g2 = m23 & ~m01;
    merges = popc(m01) + popc(g1 | g2);             // every merge frees exactly one cell
    // no branch on "any merge": without one both words are zero, the eight terms are 1 << 0 each and the result is 0 anyway
    const uint32_t G0 = pick(o0, 0u, s01), G12 = pick(o1, 0u, selof(g1)) | pick(o2, 0u, selof(g2));
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += (1u << ((G0 >> (8 * k)) & 0xffu)) + (1u << ((G12 >> (8 * k)) & 0xffu));
    return acc - (8u - merges);                     // every non-merged byte contributed 1 << 0
}

// One move on the whole board. action: 0 LEFT, 1 UP, 2 RIGHT, 3 DOWN
// (environment/game_2048.py:11-16). Env semantics (:97-114).
G2048_HD Board move_env(const Board &b, uint32_t action, uint32_t &gain, uint32_t &merges)
{
    const bool horiz = (action & 1u) == 0u;
    const bool rev = (action & 2u) != 0u;
    const Board t = transpose(b);
    uint32_t L[4];
    {
        const uint32_t x0 = horiz ? t.w[0] : b.w[0], x1 = horiz ? t.w[1] : b.w[1];
        const uint32_t x2 = horiz ? t.w[2] : b.w[2], x3 = horiz ? t.w[3] : b.w[3];
        L[0] = rev ? x3 : x0; L[1] = rev ? x2 : x1; L[2] = rev ? x1 : x2; L[3] = rev ? x0 : x3;
    }
    gain = slide_lines(L, merges);
    Board v;
    v.w[0] = rev ? L[3] : L[0]; v.w[1] = rev ? L[2] : L[1]; v.w[2] = rev ? L[1] : L[2]; v.w[3] = rev ? L[0] : L[3];
    const Board vt = transpose(v);
    Board o;
    o.w[0] = horiz ? vt.w[0] : v.w[0]; o.w[1] = horiz ? vt.w[1] : v.w[1];
    o.w[2] = horiz ? vt.w[2] : v.w[2]; o.w[3] = horiz ? vt.w[3] : v.w[3];
    return o;
}

G2048_HD Board move_env(const Board &b, uint32_t action, uint32_t &gain)
{
    uint32_t merges;
    return move_env(b, action, gain, merges);
}

// ---- direction by table -----------------------------------------------------------------------------------------
// The same move with the direction handled by ONE two-stage v_perm network whose four byte selectors are per-lane
// data instead of per-lane selects between a transposed and an untransposed, a reversed and an unreversed copy:
//   stage 1 pairs rows (0,2) and (1,3):  u0 = perm(x2,x0,a)  u1 = perm(x2,x0,b)  u2 = perm(x3,x1,a)  u3 = perm(x3,x1,b)
//   stage 2 pairs (u0,u2) and (u1,u3):   y0 = perm(u2,u0,c)  y1 = perm(u2,u0,d)  y2 = perm(u3,u1,c)  y3 = perm(u3,u1,d)
// With (a,b,c,d) = (lo,hi,lo,hi) the network is the identity, with (hi,lo,hi,lo) it reverses the word order, with the
// interleaving selectors it transposes (this pairing, unlike transpose() above, can also pass words through), and
// RIGHT folds its byte reversal into stage 1. Eight words per action: rows -> lines, then lines -> rows (the same
// four for LEFT / UP / DOWN, whose maps are involutions). 16 v_perm per move, no v_cndmask; the kernels keep the
// table in LDS and fetch a lane's eight words with two ds_read_b128 (derivation: tools/dirnet.py).
struct DirSel { uint32_t a, b, c, d, oa, ob, oc, od; };

#define G2048_DIR_TABLE_WORDS 32
#define G2048_DIR_TABLE_INIT { \
    /* 0 LEFT  */ 0x05040100u, 0x07060302u, 0x06020400u, 0x07030501u,   0x05040100u, 0x07060302u, 0x06020400u, 0x07030501u, \
    /* 1 UP    */ 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u,   0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, \
    /* 2 RIGHT */ 0x07060302u, 0x05040100u, 0x07030501u, 0x06020400u,   0x05040100u, 0x07060302u, 0x00040206u, 0x01050307u, \
    /* 3 DOWN  */ 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u,   0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u }

G2048_HD void dir_net(const uint32_t x[4], uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t y[4])
{
    const uint32_t u0 = perm(x[2], x[0], a), u1 = perm(x[2], x[0], b), u2 = perm(x[3], x[1], a), u3 = perm(x[3], x[1], b);
    y[0] = perm(u2, u0, c); y[1] = perm(u2, u0, d); y[2] = perm(u3, u1, c); y[3] = perm(u3, u1, d);
}

G2048_HD Board move_env_sel(const Board &b, const DirSel &s, uint32_t &gain, uint32_t &merges)
{
    uint32_t L[4];
    dir_net(b.w, s.a, s.b, s.c, s.d, L);
    gain = slide_lines(L, merges);
    Board o;
    dir_net(L, s.oa, s.ob, s.oc, s.od, o.w);
    return o;
}

// Both moves of one axis at once (beam kernel: vertical = (UP, DOWN), horizontal = (LEFT, RIGHT)) through the direction network:
// a lane's axis never changes during a search, so its three selector sets -- rows -> lines, forward lines -> rows, reversed
// lines -> rows -- are loop-invariant per-lane data and both moves cost 24 v_perm in all: no transpose whose result half the
// lanes discard, no v_cndmask (rounds 1-2 paired two select-based moves: 28 v_perm + 16 v_cndmask with the agent's DOWN quirk
// applied by the caller). The reversed slide runs on the line order
// {3,2,1,0}; its results R[k] are position 3-k, which is exactly the line order RIGHT's output selectors expect
// (G2048_DIR_TABLE_INIT: RIGHT's input net yields LEFT's lines reversed). For the vertical axis the reverse-output
// selectors also carry BeamSearchAgent._make_move's DOWN quirk (agents/beam_search_agent.py:209-210 vs :251-253, result =
// rot180(true DOWN)): rot180 of the rows {R3,R2,R1,R0} is the byte reversal of {R0,R1,R2,R3} -- identity word routing in
// stage 1, byte reversal in stage 2; `fixed_down` routes the words back in reversed order instead (DOWN's table entry).
struct AxisSel { uint32_t in[4], fwd[4], rev[4]; };

G2048_HD AxisSel axis_sel(bool vertical, bool fixed_down)
{
    AxisSel s;
    const uint32_t lo = 0x03020100u, hi = 0x07060504u;
    // horizontal: LEFT's rows <-> lines transpose (an involution), RIGHT's lines -> rows
    s.in[0] = vertical ? lo : 0x05040100u; s.in[1] = vertical ? hi : 0x07060302u;
    s.in[2] = vertical ? lo : 0x06020400u; s.in[3] = vertical ? hi : 0x07030501u;
    s.fwd[0] = s.in[0]; s.fwd[1] = s.in[1]; s.fwd[2] = s.in[2]; s.fwd[3] = s.in[3];
    s.rev[0] = vertical ? (fixed_down ? hi : lo) : 0x05040100u;
    s.rev[1] = vertical ? (fixed_down ? lo : hi) : 0x07060302u;
    s.rev[2] = vertical ? (fixed_down ? hi : 0x00010203u) : 0x00040206u;
    s.rev[3] = vertical ? (fixed_down ? lo : 0x04050607u) : 0x01050307u;
    return s;
}

G2048_HD void move_axis_sel(const Board &b, const AxisSel &s, Board &fwd, Board &rev)
{
    uint32_t F[4], mf, mr;
    dir_net(b.w, s.in[0], s.in[1], s.in[2], s.in[3], F);
    uint32_t R[4] = {F[3], F[2], F[1], F[0]};
    (void)slide_lines(F, mf);
    (void)slide_lines(R, mr);
    dir_net(F, s.fwd[0], s.fwd[1], s.fwd[2], s.fwd[3], fwd.w);
    dir_net(R, s.rev[0], s.rev[1], s.rev[2], s.rev[3], rev.w);
}

// BeamSearchAgent._make_move (agents/beam_search_agent.py:194-258): LEFT/UP/RIGHT
// as the env; DOWN returns rot180 of the true result because the post-transform
// (:252-253) is not the inverse of the pre-transform (:210). `fixed_down` turns
// the quirk off (non-parity option).
G2048_HD Board move_agent(const Board &b, uint32_t action, uint32_t &gain, bool fixed_down)
{
    Board o = move_env(b, action, gain);
    const bool quirk = (action == 3u) && !fixed_down;
    const Board r = rot180(o);
    o.w[0] = quirk ? r.w[0] : o.w[0]; o.w[1] = quirk ? r.w[1] : o.w[1];
    o.w[2] = quirk ? r.w[2] : o.w[2]; o.w[3] = quirk ? r.w[3] : o.w[3];
    return o;
}

// Game2048Env.get_valid_moves (environment/game_2048.py:69-95) without moving:
// a direction is valid iff some line has a tile with an empty cell further along
// the direction, or two equal neighbours. Bit a = action a.
G2048_HD uint32_t valid_mask_env(const Board &b)
{
    const uint32_t n0 = nzflag(b.w[0]), n1 = nzflag(b.w[1]), n2 = nzflag(b.w[2]), n3 = nzflag(b.w[3]);
    const uint32_t z0 = n0 ^ B80, z1 = n1 ^ B80, z2 = n2 ^ B80, z3 = n3 ^ B80;
    // vertical: byte lane = column
    const uint32_t up = (z0 & (n1 | n2 | n3)) | (z1 & (n2 | n3)) | (z2 & n3);
    const uint32_t down = (z3 & (n2 | n1 | n0)) | (z2 & (n1 | n0)) | (z1 & n0);
    const uint32_t pv = eqnzflag(b.w[0], b.w[1]) | eqnzflag(b.w[1], b.w[2]) | eqnzflag(b.w[2], b.w[3]);
    // horizontal: within a word; cell c at byte c
    uint32_t left = 0, right = 0, ph = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t n = r == 0 ? n0 : r == 1 ? n1 : r == 2 ? n2 : n3;
        const uint32_t z = n ^ B80;
        left |= z & ((n >> 8) | (n >> 16) | (n >> 24));      // empty with a tile at a higher column
        right |= z & ((n << 8) | (n << 16) | (n << 24));     // empty with a tile at a lower column
        ph |= eqnzflag(b.w[r], b.w[r] >> 8);                 // (byte 3 compares with the 0 shifted in: never "equal and non-zero")
    }
    return ((left | ph) ? 1u : 0u) | ((up | pv) ? 2u : 0u) | ((right | ph) ? 4u : 0u) | ((down | pv) ? 8u : 0u);
}

// BeamSearchAgent._check_valid_moves (agents/beam_search_agent.py:183-192):
// LEFT/UP/RIGHT agree with the env; DOWN compares rot180(true DOWN) with the board.
G2048_HD uint32_t valid_mask_agent(const Board &b, bool fixed_down)
{
    uint32_t m = valid_mask_env(b);
    if (!fixed_down) {
        uint32_t g;
        const Board d = move_agent(b, 3u, g, false);
        m = (m & 7u) | (same(d, b) ? 0u : 8u);
    }
    return m;
}

// true iff no move changes the board (environment/game_2048.py:279-288)
G2048_HD bool game_over(const Board &b) { return valid_mask_env(b) == 0u; }

// ------------------------------------------------------------------- spawn ----
// add_new_tile (environment/game_2048.py:59-67) / _add_random_tile
// (agents/beam_search_agent.py:260-269): the idx-th empty cell in row-major
// order gets a 2 (code 1) or a 4 (code 2). h is one 32-bit draw:
// idx = ((h >> 16) * n_empty) >> 16, four iff (h & 0xffff) >= 58982.
// No-op on a full board. Returns n_empty before the spawn.
// Two formulations, both exact (tests/hostsim checks them against each other and the oracle); which is faster depends on
// the surrounding kernel: the row-select form in the beam kernel (a shorter dependent chain), the all-rows prefix form in the
// step kernel, where its by-product -- the post-spawn zero flags -- also feeds the tile sums (13.33 vs 13.51 us). A third form
// (row select + clearing k flags one by one) lost to both and is gone (profiles/r03_beam_latency.txt).
// (1) row select, then the k-th empty cell of that row by a byte-wise prefix count of its zero indicators instead
// of clearing k flags one by one (round 3): (z >> 7) * 0x01010101 has, in byte c, the number of empty cells in columns 0..c;
// the cell whose count equals k + 1 is the one. A full board needs no special case: its z is 0.
G2048_HD uint32_t spawn_rowprefix(Board &b, uint32_t h, bool enable = true, uint32_t *zf_out = nullptr)
{
    const uint32_t ones = 0x01010101u;
    const uint32_t z0 = zflag(b.w[0]), z1 = zflag(b.w[1]), z2 = zflag(b.w[2]), z3 = zflag(b.w[3]);
    const uint32_t c0 = popc(z0), c1 = c0 + popc(z1), c2 = c1 + popc(z2), n = c2 + popc(z3);
    const uint32_t idx = ((h >> 16) * n) >> 16;
    const uint32_t row = (idx >= c0 ? 1u : 0u) + (idx >= c1 ? 1u : 0u) + (idx >= c2 ? 1u : 0u);
    const uint32_t z = row == 0 ? z0 : row == 1 ? z1 : row == 2 ? z2 : z3;
    const uint32_t k1 = idx + 1u - (row == 0 ? 0u : row == 1 ? c0 : row == 2 ? c1 : c2);      // 1-based rank inside the row
    const uint32_t hit = zflag(((z >> 7) * ones) ^ (enable ? k1 * ones : 0x7f7f7f7fu)) & z;       // 0x7f: never a rank
    const uint32_t add = hit >> (((h & 0xffffu) >= 58982u) ? 6 : 7);
    b.w[0] |= row == 0 ? add : 0u; b.w[1] |= row == 1 ? add : 0u;
    b.w[2] |= row == 2 ? add : 0u; b.w[3] |= row == 3 ? add : 0u;
    if (zf_out) {               // zero flags of the board AFTER the spawn
        zf_out[0] = z0 ^ (row == 0 ? hit : 0u); zf_out[1] = z1 ^ (row == 1 ? hit : 0u);
        zf_out[2] = z2 ^ (row == 2 ? hit : 0u); zf_out[3] = z3 ^ (row == 3 ? hit : 0u);
    }
    return n;
}

// (2) prefix sums: the zero indicators (0/1 per byte) times 0x01010101 give, in byte c of row r, the number of
// empty cells in columns 0..c of that row; adding the broadcast count of the rows above turns it into the 1-based
// row-major rank of every empty cell. The chosen cell is the one whose rank equals idx + 1 -- one flag in one of
// the four words, with no row selection and no "k-th set bit" loop.
// zf_out (optional): the zero flags of the board AFTER the spawn.
G2048_HD uint32_t spawn_prefix(Board &b, uint32_t h, bool enable = true, uint32_t *zf_out = nullptr)
{
    const uint32_t ones = 0x01010101u;
    const uint32_t z0 = zflag(b.w[0]), z1 = zflag(b.w[1]), z2 = zflag(b.w[2]), z3 = zflag(b.w[3]);
    const uint32_t p0 = (z0 >> 7) * ones, p1 = (z1 >> 7) * ones, p2 = (z2 >> 7) * ones, p3 = (z3 >> 7) * ones;
    const uint32_t c0 = p0 >> 24, c1 = c0 + (p1 >> 24), c2 = c1 + (p2 >> 24), n = c2 + (p3 >> 24);
    const uint32_t idx = ((h >> 16) * n) >> 16;
    // rank to match, broadcast to every byte; 0x7f (never a rank, and still < 0x80 for the flag arithmetic)
    // switches the spawn off
    uint32_t t_on = (idx + 1u) * ones;
#if defined(__HIPCC__)
    // computed for every lane and then SELECTED: left to itself the compiler wraps these four instructions in an exec-mask region
    // (s_and_saveexec .. s_or exec) in the middle of the kernel, which is a scheduling barrier for everything around it
    asm volatile("" : "+v"(t_on));
#endif
    const uint32_t target = enable ? t_on : 0x7f7f7f7fu;
    const uint32_t h0 = zflag(p0 ^ target) & z0;
    const uint32_t h1 = zflag((p1 + c0 * ones) ^ target) & z1;
    const uint32_t h2 = zflag((p2 + c1 * ones) ^ target) & z2;
    const uint32_t h3 = zflag((p3 + c2 * ones) ^ target) & z3;
    const uint32_t sh = ((h & 0xffffu) >= 58982u) ? 6u : 7u;          // 0x80 >> 7 = code 1 (tile 2), >> 6 = code 2 (tile 4)
    b.w[0] |= h0 >> sh; b.w[1] |= h1 >> sh; b.w[2] |= h2 >> sh; b.w[3] |= h3 >> sh;
    if (zf_out) { zf_out[0] = z0 ^ h0; zf_out[1] = z1 ^ h1; zf_out[2] = z2 ^ h2; zf_out[3] = z3 ^ h3; }
    return n;
}

// reset (environment/game_2048.py:29-48): empty board, two spawns
// The two spawns start from a known board, so no empty-cell ranking is needed: the first draw indexes 16 cells directly
// (((h >> 16) * 16) >> 16 = h >> 28), the second indexes the 15 that are left, i.e. skips the first one's cell.
G2048_HD Board fresh_board(uint32_t h0, uint32_t h1)
{
    const uint32_t cell0 = h0 >> 28;
    uint32_t cell1 = ((h1 >> 16) * 15u) >> 16;
    cell1 += cell1 >= cell0 ? 1u : 0u;
    const uint32_t code0 = (h0 & 0xffffu) >= 58982u ? 2u : 1u, code1 = (h1 & 0xffffu) >= 58982u ? 2u : 1u;
    const uint32_t t0 = code0 << (8u * (cell0 & 3u)), t1 = code1 << (8u * (cell1 & 3u));
    const uint32_t r0 = cell0 >> 2, r1 = cell1 >> 2;
    Board b;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) b.w[r] = (r0 == r ? t0 : 0u) | (r1 == r ? t1 : 0u);
    return b;
}

// ----------------------------------------------------------------- RNG --------
// draw(k0, k1, id, ctr): two xorshift-multiply finalizers; (k0, k1) are derived
// on the host from (seed, domain, index) (g2048_rng.h). Restated independently
// in oracle/g2048_oracle.c.
G2048_HD uint32_t rng_draw(uint32_t k0, uint32_t k1, uint64_t id, uint32_t ctr)
{
    uint32_t h = (uint32_t)id ^ k0;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += k1 + (uint32_t)(id >> 32) * 0x9E3779B1u + ctr * 0x85EBCA77u;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}

// The same draw for a caller whose ids share their high word (a launch that does not cross a multiple of 2^32: the host splits
// one that would): k1h = k1 + rng_hi_term(id) (g2048_rng.h) is formed once, on the host or the scalar unit, and the lane hashes 32 bits.
G2048_HD uint32_t rng_draw_lo(uint32_t k0, uint32_t k1h, uint32_t id_lo, uint32_t ctr)
{
    uint32_t h = id_lo ^ k0;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += k1h + ctr * 0x85EBCA77u;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}

// --------------------------------------------------------------- tile sums ----
struct TileStats {
    uint32_t total;      // sum of tile values
    uint32_t edge;       // row0 + row3 + col0 + col3 (corners twice)  (game_2048.py:254-256)
    uint32_t orbits;     // OR of (1 << code) over all cells (bit 0 set iff any empty)
};

G2048_HD TileStats tile_stats_z(const Board &b, uint32_t n_empty, uint32_t z0, uint32_t z1, uint32_t z2, uint32_t z3)
{
    uint32_t rs[4], outer = 0, orb = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t x = b.w[r];
        const uint32_t t0 = 1u << (x & 0xffu), t1 = 1u << ((x >> 8) & 0xffu);
        const uint32_t t2 = 1u << ((x >> 16) & 0xffu), t3 = 1u << (x >> 24);
        rs[r] = t0 + t1 + t2 + t3;
        outer += t0 + t3;
        orb |= t0 | t1 | t2 | t3;
    }
    // empty cells contributed 1 each: subtract their counts
    const uint32_t zall = n_empty;
    const uint32_t zcols = popc((z0 | (z1 >> 1) | (z2 >> 2) | (z3 >> 3)) & 0xf00000f0u);  // col 0 and col 3 flags of 4 rows
    TileStats s;
    s.total = rs[0] + rs[1] + rs[2] + rs[3] - zall;
    s.edge = rs[0] + rs[3] + outer - popc(z0) - popc(z3) - zcols;
    s.orbits = orb;
    return s;
}

G2048_HD TileStats tile_stats(const Board &b, uint32_t n_empty)
{
    return tile_stats_z(b, n_empty, zflag(b.w[0]), zflag(b.w[1]), zflag(b.w[2]), zflag(b.w[3]));
}

G2048_HD TileStats tile_stats(const Board &b) { return tile_stats(b, count_empty(b)); }

G2048_HD uint32_t max_code(const Board &b)
{
    const TileStats s = tile_stats(b, 0u);          // only .orbits is used
    return 31u - (uint32_t)__builtin_clz(s.orbits | 1u);
}

// ----------------------------------------------------------------- reward -----
// Game2048Env._calculate_reward (environment/game_2048.py:212-277) in the
// reference's f64 operation order. The milestone branch (:229-241) compares the
// env's highest_tile -- not yet updated when the reward is computed (:195 vs
// :200-203) -- with max(prev_board); the two are always equal inside step(), so
// the branch is dead there and is not generated. `cur` is the post-spawn board.
// Must be compiled with -ffp-contract=off (the 0.1 terms are mul THEN add).
// edge / total of the reward as an IEEE-754 correctly rounded f64 quotient without the generic division's range
// handling: both operands are integers below 2^23 (exact in f64, no overflow / underflow / denormal is reachable), so
// the scaling (v_div_scale_f64 x2) and the special-case fix-up (v_div_fixup_f64) of the compiler's expansion are
// no-ops here; and of that expansion's two Newton steps on the reciprocal ONE is enough for these operands: v_rcp_f64, one
// Newton step, quotient, residual, final fma give the bits of a / b for every reachable pair -- all 1.76e13 of them were
// compared on the device (tools/ubench/div_check.hip, profiles/r05_div_check.txt; without any Newton step half of them differ).
// 0 / 0 (the degenerate all-empty board) gives NaN as a / b does: rcp(0) = inf, fma(-0, inf, 1) = NaN.
G2048_HD double div_small_ints(double a, double b)
{
#if defined(__HIPCC__)
    double y = __builtin_amdgcn_rcp(b);
    const double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
#else
    return a / b;
#endif
}

G2048_HD double reward_env_from(double r, const Board &cur, const TileStats &st, bool valid,
                                uint32_t empty_before, uint32_t empty_after)
{
    if (!valid) r -= 2.0;
    r += (double)((int)empty_after - (int)empty_before) * 0.5;
    r += ((double)st.edge / (double)st.total) * 1.0;
    if (empty_after <= 2u) r -= 2.0;
    // ordered pairs (:267-275): row i pairs (j-1, j) with both > 0 and x[j] >= x[j-1]; same for col i
    const uint32_t n0 = nzflag(cur.w[0]), n1 = nzflag(cur.w[1]), n2 = nzflag(cur.w[2]), n3 = nzflag(cur.w[3]);
    const uint32_t v01 = geflag(cur.w[1], cur.w[0]) & n0 & n1;
    const uint32_t v12 = geflag(cur.w[2], cur.w[1]) & n1 & n2;
    const uint32_t v23 = geflag(cur.w[3], cur.w[2]) & n2 & n3;
    const uint32_t colcnt = (v01 >> 7) + (v12 >> 7) + (v23 >> 7);        // byte i = col_ordered_i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = cur.w[i], n = i == 0 ? n0 : i == 1 ? n1 : i == 2 ? n2 : n3;
        const uint32_t h = geflag(x >> 8, x) & n & (n >> 8);                  // (n >> 8 has no flag in byte 3)
        const uint32_t c = popc(h) + ((colcnt >> (8 * i)) & 0xffu);
        r += (double)c * 0.1;
    }
    return r;
}

G2048_HD double reward_env(const Board &cur, const TileStats &st, uint32_t gain, bool valid,
                           uint32_t empty_before, uint32_t empty_after)
{
    return reward_env_from((double)gain / 4.0, cur, st, valid, empty_before, empty_after);
}

// The same value with the first three terms folded: gain/4, -2.0 for an invalid move and (after - before) * 0.5 are
// all exact multiples of 0.25 of small magnitude, so the reference's three roundings are no-ops and the partial
// sum equals (gain - 8*[invalid] + 2*(after - before)) * 0.25 exactly -- one conversion and one multiply.
// (Round 3 tried the four products (double)c * 0.1 from an 8-entry LDS table: the table fill and the address arithmetic cost
// more than the conversions and multiplications they replace: 386 against 374 static instructions. Not kept.)
// `tenth(c)` supplies (double)c * 0.1 for the ordered-pair counts c = 0 .. 6 of :267-275: the product itself by default, or -- in
// the kernels that keep the direction table in LDS anyway -- a read of the eight products from the same table (an address shift and
// an LDS read instead of a conversion and an f64 multiply on the VALU, which is what bounds the step; the values are the same
// IEEE products, computed by the host compiler: G2048_TENTHS_INIT).
// Interface: kColShift = the right shift that turns a 0x80 byte flag into a column count's unit (7: counts; 4: counts * 8, a byte
// offset into a table of doubles), operator()(row pairs, column pairs in that unit) -> (double)(row + col) * 0.1.
// crowded(r, empty_after) applies :263-264 (`if empty_after <= 2: reward -= 2.0`): the subtraction by default, or r + T[empty_after]
// with T = {-2, -2, -2, 0, 0, ...} from the same table (r is never -0.0 at that point -- a sum whose last term is the non-negative
// quotient -- so adding +0.0 leaves every bit alone).
struct TenthByProduct {
    static constexpr uint32_t kColShift = 7u;
    G2048_HD double operator()(uint32_t rows, uint32_t cols) const { return (double)(rows + cols) * 0.1; }
    G2048_HD double crowded(double r, uint32_t empty_after) const { return empty_after <= 2u ? r - 2.0 : r; }
};

#define G2048_TENTHS 8
#define G2048_TENTHS_INIT { 0.0 * 0.1, 1.0 * 0.1, 2.0 * 0.1, 3.0 * 0.1, 4.0 * 0.1, 5.0 * 0.1, 6.0 * 0.1, 7.0 * 0.1 }
#define G2048_CROWDED 17
#define G2048_CROWDED_INIT { -2.0, -2.0, -2.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 }

template <class TENTH = TenthByProduct>
G2048_HD double reward_env_folded(const Board &cur, const TileStats &st, uint32_t gain, bool valid,
                                  uint32_t empty_before, uint32_t empty_after, TENTH tenth = TENTH())
{
    const int32_t q = (int32_t)gain - (valid ? 0 : 8) + 2 * ((int32_t)empty_after - (int32_t)empty_before);
    double r = (double)q * 0.25;
    r += div_small_ints((double)st.edge, (double)st.total) * 1.0;
    r = tenth.crowded(r, empty_after);
    const uint32_t n0 = nzflag(cur.w[0]), n1 = nzflag(cur.w[1]), n2 = nzflag(cur.w[2]), n3 = nzflag(cur.w[3]);
    const uint32_t v01 = geflag(cur.w[1], cur.w[0]) & n0 & n1;
    const uint32_t v12 = geflag(cur.w[2], cur.w[1]) & n1 & n2;
    const uint32_t v23 = geflag(cur.w[3], cur.w[2]) & n2 & n3;
    const uint32_t colcnt = (v01 >> TENTH::kColShift) + (v12 >> TENTH::kColShift) + (v23 >> TENTH::kColShift);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = cur.w[i], n = i == 0 ? n0 : i == 1 ? n1 : i == 2 ? n2 : n3;
        const uint32_t h = geflag(x >> 8, x) & n & (n >> 8);                  // (n >> 8 has no flag in byte 3)
        r += tenth(popc(h), (colcnt >> (8 * i)) & 0xffu);
    }
    return r;
}

// done <=> no direction changes the board (environment/game_2048.py:279-288): a full board
// without equal neighbours, or (degenerate) an all-empty board. pair_count is defined below.
G2048_HD uint32_t pair_count(const Board &b);
G2048_HD bool game_over_counted(const Board &b, uint32_t n_empty)
{
#if defined(__HIPCC__)
    // ONE wave-uniform branch around the pair count (taken when some lane of the wavefront holds a full board) instead of the
    // nest of exec-mask regions the two early returns compile to: a full board is rare, and every exec region in the middle of
    // the step is a scheduling barrier for the instructions around it
    const bool full = n_empty == 0u;
    uint32_t pairs = 1u;
    if (__builtin_amdgcn_ballot_w64(full) != 0ull) pairs = pair_count(b);
    return n_empty == 16u || (full && pairs == 0u);
#else
    if (n_empty == 16u) return true;
    if (n_empty != 0u) return false;
    return pair_count(b) == 0u;
#endif
}

// Game2048Env.step (environment/game_2048.py:170-210) for one board: move (:185), valid (:188),
// spawn iff valid (:191-192), reward on the post-spawn board (:195), done (:198), highest tile
// (:201-203, always the current max inside step()). h is the board's 32-bit draw for this step.
struct StepOut { Board board; uint32_t gain; double reward; uint32_t flags; };

// MOVE: a callable (const Board &, uint32_t &gain, uint32_t &merges) -> Board doing the env move
template <class MOVE, class TENTH = TenthByProduct>
G2048_HD StepOut step_board_with(const Board &prev, MOVE move, uint32_t h, TENTH tenth = TENTH())
{
    StepOut o;
    uint32_t merges;
    Board cur = move(prev, o.gain, merges);
    const bool valid = !same(cur, prev);
    uint32_t empty_mid;                 // empties of the moved board, before the spawn
    uint32_t zf[4];                     // zero flags of the post-spawn board
    empty_mid = spawn_prefix(cur, h, valid, zf);
    // a slide moves tiles and every merge frees one cell; a valid move then fills one (a valid move always
    // leaves an empty cell: either a tile slid into a gap or a merge freed a cell)
    const uint32_t empty_before = empty_mid - merges;
    const uint32_t empty_after = empty_mid - (valid ? 1u : 0u);
    const TileStats st = tile_stats_z(cur, empty_after, zf[0], zf[1], zf[2], zf[3]);
    o.reward = reward_env_folded(cur, st, o.gain, valid, empty_before, empty_after, tenth);
    const bool done = game_over_counted(cur, empty_after);
    const uint32_t maxcode = 31u - (uint32_t)__builtin_clz(st.orbits | 1u);
    o.flags = (done ? 1u : 0u) | (valid ? 2u : 0u) | (maxcode << 3);
    o.board = cur;
    return o;
}

G2048_HD StepOut step_board(const Board &prev, uint32_t action, uint32_t h)
{
    return step_board_with(prev, [action](const Board &b, uint32_t &g, uint32_t &m) { return move_env(b, action, g, m); }, h);
}

// An action outside 0..3 moves nothing in the reference (_execute_move, environment/game_2048.py:97-114, has no branch for
// it), so the step is an invalid move: no spawn, the -2.0 of :226-227. noop = "this lane's action is such a value".
template <class TENTH = TenthByProduct>
G2048_HD StepOut step_board_sel_noop(const Board &prev, const DirSel &sel, bool noop, uint32_t h, TENTH tenth = TENTH())
{
    return step_board_with(prev, [&sel, noop](const Board &b, uint32_t &g, uint32_t &m) {
        const Board r = move_env_sel(b, sel, g, m);
        g = noop ? 0u : g; m = noop ? 0u : m;
        return Board{{noop ? b.w[0] : r.w[0], noop ? b.w[1] : r.w[1], noop ? b.w[2] : r.w[2], noop ? b.w[3] : r.w[3]}};
    }, h, tenth);
}

// the same step with the direction given as its selector words (see "direction by table")
template <class TENTH = TenthByProduct>
G2048_HD StepOut step_board_sel(const Board &prev, const DirSel &sel, uint32_t h, TENTH tenth = TENTH())
{
    return step_board_with(prev, [&sel](const Board &b, uint32_t &g, uint32_t &m) { return move_env_sel(b, sel, g, m); }, h, tenth);
}

// ---------------------------------------------------------------- policy ------
// Masked categorical sampling of PPOAgent.get_action (agents/ppo_agent.py:211-221) for one env: the reference
// samples from softmax(log(p + 1e-10) + mask), i.e. from weights w_a = p_a + 1e-10 on the valid actions. Here:
// inverse CDF in f32 with one 32-bit draw h (u = (h >> 8) * 2^-24), sums taken left to right; returns the action
// and its probability w_a / sum (the caller takes the log). A mask with no valid action samples unmasked.
G2048_HD uint32_t sample_action(float p0, float p1, float p2, float p3, uint32_t mask4, uint32_t h, float &prob)
{
    const uint32_t m = (mask4 & 15u) ? (mask4 & 15u) : 15u;
    const float w0 = (m & 1u) ? p0 + 1e-10f : 0.0f, w1 = (m & 2u) ? p1 + 1e-10f : 0.0f;
    const float w2 = (m & 4u) ? p2 + 1e-10f : 0.0f, w3 = (m & 8u) ? p3 + 1e-10f : 0.0f;
    const float c0 = w0, c1 = c0 + w1, c2 = c1 + w2, sum = c2 + w3;
    const float t = ((float)(h >> 8) * 5.9604644775390625e-08f) * sum;
    uint32_t a = 3u;
    if (t < c2) a = 2u;
    if (t < c1) a = 1u;
    if (t < c0) a = 0u;
    if (!((m >> a) & 1u)) a = 31u - (uint32_t)__builtin_clz(m);      // rounding fell past the last valid action
    const float wa = a == 0u ? w0 : a == 1u ? w1 : a == 2u ? w2 : w3;
    prob = wa / sum;
    return a;
}

// --------------------------------------------------------- simulate_move ------
// The first `full` empty cells of m (row-major) get a 4 (code 2), the next one gets code `last` (0 = leave
// it). Ranks of the empty cells come from a byte-wise inclusive prefix sum of the zero indicators.
G2048_HD Board fill_empties(const Board &m, uint32_t full, uint32_t last)
{
    const uint32_t ones = 0x01010101u;
    uint32_t z[4], p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { z[r] = zflag(m.w[r]); p[r] = (z[r] >> 7) * ones; }
    const uint32_t o1 = p[0] >> 24, o2 = o1 + (p[1] >> 24), o3 = o2 + (p[2] >> 24);
    const uint32_t fullb = full * ones, nextb = (full + 1u) * ones;
    Board o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t rank = p[r] + (r == 0 ? 0u : r == 1 ? o1 : r == 2 ? o2 : o3) * ones;     // 1-based, <= 16
        const uint32_t le = geflag(fullb, rank) & z[r];
        const uint32_t eq = zflag(rank ^ nextb) & z[r];
        o.w[r] = m.w[r] | (le >> 6) | ((eq >> 7) * last);
    }
    return o;
}

// Game2048Env.simulate_move (environment/game_2048.py:341-387), successor k of (state, action) as the reference
// actually produces it: `empty` is listed once from the moved board M; successor k = 2*i + t sets empty[i] to
// 2 (t = 0) or 4 (t = 1) ON TOP OF the previous successor (self.board = new_state.copy(), :378), so empty[0..i)
// already hold 4; its reward is _calculate_reward(True, state, score) evaluated while self.board is still the
// PREVIOUS successor (M for k = 0) and with the env's own highest_tile attribute (so the milestone branch
// :229-241 is live here); done is is_game_over() of the successor itself.
struct SimOut { Board board; double reward; bool done; };

G2048_HD uint32_t simulate_count(const Board &state, uint32_t action, Board &moved, uint32_t &gain)
{
    moved = move_env(state, action, gain);
    return same(moved, state) ? 0u : 2u * count_empty(moved);
}

G2048_HD SimOut simulate_successor(const Board &state, const Board &moved, uint32_t gain, uint32_t k, uint32_t highest_code)
{
    SimOut o;
    o.board = fill_empties(moved, k >> 1, (k & 1u) ? 2u : 1u);
    const Board seen = k == 0u ? moved : fill_empties(moved, (k - 1u) >> 1, ((k - 1u) & 1u) ? 2u : 1u);
    double r = (double)gain / 4.0;
    if (highest_code > max_code(state)) {          // :229-241; every term is exact at these magnitudes
        r += 2.0 * (double)highest_code;
        if (highest_code >= 8u) r += 50.0;
        if (highest_code >= 9u) r += 100.0;
        if (highest_code >= 10u) r += 200.0;
        if (highest_code >= 11u) r += 500.0;
    }
    const uint32_t seen_empty = count_empty(seen);
    o.reward = reward_env_from(r, seen, tile_stats(seen, seen_empty), true, count_empty(state), seen_empty);
    o.done = game_over_counted(o.board, count_empty(o.board));
    return o;
}

// The hybrid agent's simulate_move (agents/hybrid.py:578-629, monkey-patched onto its private copy of the env, :694-697):
// the move (no quirk here: rot90-based, all four directions true), then up to three DISTINCT empty cells of the moved
// board -- random.sample(empty_cells, min(3, n)) -- each as a 2-successor and a 4-successor whose reward
// (_calculate_simulation_reward, :671-692: sum gained + new max if it rose + 0.1 per empty cell) is weighted by 0.9 / 0.1.
// Successor k = 2 * pick + (0: tile 2, 1: tile 4). The picks are sampling without replacement: pick i is the
// idx(h_i, n - i)-th empty cell, row-major, among those not picked before.
struct SampledOut { Board board; double reward; };

G2048_HD uint32_t sampled_pick_rank(uint32_t pick, uint32_t n_empty, uint32_t h0, uint32_t h1, uint32_t h2)
{
    const uint32_t r0 = ((h0 >> 16) * n_empty) >> 16;
    if (pick == 0u) return r0;
    uint32_t r1 = ((h1 >> 16) * (n_empty - 1u)) >> 16;
    if (r1 >= r0) r1 += 1u;
    if (pick == 1u) return r1;
    uint32_t r2 = ((h2 >> 16) * (n_empty - 2u)) >> 16;
    const uint32_t lo = r0 < r1 ? r0 : r1, hi = r0 < r1 ? r1 : r0;
    if (r2 >= lo) r2 += 1u;
    if (r2 >= hi) r2 += 1u;
    return r2;
}

G2048_HD SampledOut simulate_sampled_successor(const Board &state, const Board &moved, uint32_t k, uint32_t n_empty,
                                               uint32_t h0, uint32_t h1, uint32_t h2)
{
    SampledOut o;
    const uint32_t rank = sampled_pick_rank(k >> 1, n_empty, h0, h1, h2);
    const uint32_t code = (k & 1u) ? 2u : 1u;
    {           // only the empty cell of row-major rank `rank` gets the tile (ranks by a byte-wise prefix sum of the zero flags)
        const uint32_t ones = 0x01010101u;
        uint32_t z[4], p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { z[r] = zflag(moved.w[r]); p[r] = (z[r] >> 7) * ones; }
        const uint32_t o1 = p[0] >> 24, o2 = o1 + (p[1] >> 24), o3 = o2 + (p[2] >> 24);
        const uint32_t want = (rank + 1u) * ones;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t rk = p[r] + (r == 0 ? 0u : r == 1 ? o1 : r == 2 ? o2 : o3) * ones;      // 1-based rank of every empty cell
            const uint32_t eq = zflag(rk ^ want) & z[r];
            o.board.w[r] = moved.w[r] | ((eq >> 7) * code);
        }
    }
    const uint32_t old_max = max_code(state), mv_max = max_code(moved);
    const uint32_t new_max = mv_max > code ? mv_max : code;
    const uint32_t gained = 1u << code;                                           // a move conserves the tile sum (:673-675)
    const uint32_t bonus = new_max > old_max ? (1u << new_max) : 0u;             // :678-682
    double r = (double)(gained + bonus);
    r = r + (double)(n_empty - 1u) * 0.1;                                         // :685, :688
    o.reward = r * ((k & 1u) ? 0.1 : 0.9);                                        // :622, :628
    return o;
}

// ------------------------------------------------------------- heuristics -----
G2048_HD uint32_t pair_count(const Board &b)      // adjacent equal non-zero pairs, h + v
{
    // seven flag words (bit 7 of a byte each); shifted right by 0..6 they occupy seven different bits of the
    // byte, so one OR-reduction and one popcount count them all
    uint32_t w = eqnzflag(b.w[0], b.w[1]) | (eqnzflag(b.w[1], b.w[2]) >> 1) | (eqnzflag(b.w[2], b.w[3]) >> 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) w |= eqnzflag(b.w[r], b.w[r] >> 8) >> (3 + r);       // byte 3 meets the 0 shifted in: no flag there
    return popc(w);
}

G2048_HD uint32_t max_corner_code(const Board &b)
{
    const uint32_t a = b.w[0] & 0xffu, c = b.w[0] >> 24, d = b.w[3] & 0xffu, e = b.w[3] >> 24;
    const uint32_t m0 = a > c ? a : c, m1 = d > e ? d : e;
    return m0 > m1 ? m0 : m1;
}

// BeamSearchAgent._fast_evaluate (agents/beam_search_agent.py:280-314):
// empty*10 + log2(max)*2 + max non-empty corner*2 + 2*(#adjacent equal pairs). Integer-valued.
// Every term is a small integer (< 2^19 in total), so the value is exact in any of u32 / f32 / f64.
G2048_HD uint32_t eval_fast_u32(const Board &b)
{
    const uint32_t cc = max_corner_code(b);
    return count_empty(b) * 10u + max_code(b) * 2u + (cc ? (2u << cc) : 0u) + pair_count(b) * 2u;
}

G2048_HD double eval_fast(const Board &b) { return (double)eval_fast_u32(b); }

// the same with the empty count and the max code supplied by a caller that already knows them (beam kernel)
G2048_HD uint32_t eval_fast_u32_known(const Board &b, uint32_t n_empty, uint32_t maxcode)
{
    const uint32_t cc = max_corner_code(b);
    return n_empty * 10u + maxcode * 2u + (cc ? (2u << cc) : 0u) + pair_count(b) * 2u;
}

// true iff some cell holds `code` (code >= 1)
G2048_HD bool has_code(const Board &b, uint32_t code)
{
    const uint32_t c = code * 0x01010101u;
    return (zflag(b.w[0] ^ c) | zflag(b.w[1] ^ c) | zflag(b.w[2] ^ c) | zflag(b.w[3] ^ c)) != 0u;
}

// sum of codes over adjacent equal non-zero pairs (_calculate_merge_potential, :387-403)
G2048_HD uint32_t merge_potential(const Board &b)
{
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t f = eqnzflag(b.w[r], b.w[r + 1]);
        acc = dot4(b.w[r], f >> 7, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t f = eqnzflag(b.w[r], b.w[r] >> 8);
        acc = dot4(b.w[r], f >> 7, acc);
    }
    return acc;
}

// BeamSearchAgent._evaluate_state (agents/beam_search_agent.py:316-373), phase 0/1/2 =
// early/mid/late (:271-278), f64, same operation order. snake_patterns[0] (:37-42).
G2048_HD double eval_full_known(const Board &b, uint32_t phase, uint32_t e, uint32_t mc);

G2048_HD double eval_full(const Board &b, uint32_t phase) { return eval_full_known(b, phase, count_empty(b), max_code(b)); }

G2048_HD double eval_full_known(const Board &b, uint32_t phase, uint32_t e, uint32_t mc)
{
    const double we = phase == 0 ? 15.0 : phase == 1 ? 10.0 : 8.0;
    const double wm = phase == 0 ? 1.0 : phase == 1 ? 1.5 : 2.0;
    const double wc = phase == 0 ? 2.0 : phase == 1 ? 2.5 : 3.0;
    const double wg = phase == 0 ? 2.0 : phase == 1 ? 1.5 : 1.0;
    double empty_score = (double)e * we;
    if (e <= 2u) empty_score -= 10.0;
    double max_score = (double)mc * wm;
    if (mc >= 9u) max_score *= 1.2;
    if (mc >= 10u) max_score *= 1.5;
    if (mc >= 11u) max_score *= 2.0;
    const double corner_bonus = ((double)max_corner_code(b) * 2.0) * wc;
    const double mp = (double)merge_potential(b) * wg;
    uint32_t sn = dot4(b.w[0], 0x0c0d0e0fu, 0u);      // 15 14 13 12
    sn = dot4(b.w[1], 0x0b0a0908u, sn);               //  8  9 10 11
    sn = dot4(b.w[2], 0x04050607u, sn);               //  7  6  5  4
    sn = dot4(b.w[3], 0x03020100u, sn);               //  0  1  2  3
    const double snake = (double)sn / 100.0;
    return (((empty_score + max_score) + corner_bonus) + mp) + snake;
}

// Game2048Env._evaluate_pattern (environment/game_2048.py:313-339; no caller in the reference): max of two weighted sums of
// the real tile values, each / 100.0 -- the snake weights 16..1 and the corner weights 16, 8, 4, ... 0.25. Every product and
// every partial sum is an exact binary fraction far below 2^53 whatever order numpy adds them in, so the corner sum is taken
// four times as large in integers (weights 64 .. 1) and scaled by 0.25; one rounding each: the division by 100.0.
G2048_HD double eval_pattern(const Board &b)
{
    uint32_t snake = 0, corner4 = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t code = (b.w[r] >> (8 * c)) & 0xffu;
            const uint32_t tile = code ? (1u << code) : 0u;
            const uint32_t ws = r == 0 ? 16u - c : r == 1 ? 9u + c : r == 2 ? 8u - c : 1u + c;        // :319-324
            snake += tile * ws;
            corner4 += tile << (6 - r - c);                                                          // :327-332, times 4
        }
    }
    const double s = (double)snake / 100.0, k = ((double)corner4 * 0.25) / 100.0;                   // :335-336
    return s > k ? s : k;                                                                            // :339
}

// agents/beam_search_agent.py:271-278 with thresholds as tile values
G2048_HD uint32_t phase_of(uint32_t maxcode, uint32_t early_thr, uint32_t mid_thr)
{
    const uint32_t tile = maxcode ? (1u << maxcode) : 0u;
    return tile < early_thr ? 0u : (tile < mid_thr ? 1u : 2u);
}

// PPOAgent.monotonicity counts (agents/ppo_agent.py:300-333): pairs with both > 0;
// rle/rge over the 12 horizontal pairs (a <= b / a >= b), cle/cge over the 12 vertical ones.
struct MonoCounts { uint32_t rle, rge, cle, cge; };

G2048_HD MonoCounts mono_counts(const Board &b)
{
    MonoCounts m = {0, 0, 0, 0};
    const uint32_t n0 = nzflag(b.w[0]), n1 = nzflag(b.w[1]), n2 = nzflag(b.w[2]), n3 = nzflag(b.w[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t x = b.w[r], n = r == 0 ? n0 : r == 1 ? n1 : r == 2 ? n2 : n3;
        const uint32_t both = n & (n >> 8);
        m.rle += popc(geflag(x >> 8, x) & both);
        m.rge += popc(geflag(x, x >> 8) & both);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t na = r == 0 ? n0 : r == 1 ? n1 : n2, nb = r == 0 ? n1 : r == 1 ? n2 : n3;
        m.cle += popc(geflag(b.w[r + 1], b.w[r]) & na & nb);
        m.cge += popc(geflag(b.w[r], b.w[r + 1]) & na & nb);
    }
    return m;
}

// PPOAgent.evaluate_heuristic (agents/ppo_agent.py:271-298)
G2048_HD double eval_ppo_heuristic(const Board &b)
{
    const MonoCounts m = mono_counts(b);
    const uint32_t best = (m.rle > m.rge ? m.rle : m.rge) + (m.cle > m.cge ? m.cle : m.cge);
    double score = 0.0;
    score += 2.0 * ((double)best / 24.0);
    if (max_corner_code(b) == max_code(b)) score += 1.0;
    // tiles >= 8  <=>  code >= 3
    const uint32_t c3 = 0x03030303u;
    const uint32_t high = popc(geflag(b.w[0], c3)) + popc(geflag(b.w[1], c3)) + popc(geflag(b.w[2], c3)) + popc(geflag(b.w[3], c3));
    if (high > 0u) score += -0.1 * (double)high;
    return score;
}

// sum of the four largest codes = sum over thresholds t >= 1 of min(4, #cells with code >= t).
// Cell c contributes the unary mask (1 << c) - 1 (bit t-1 set <=> t <= c); the 16 masks are added into a
// bit-sliced counter (ones / twos / saturating fours), so every threshold is counted at once.
G2048_HD uint32_t top4_code_sum(const Board &b)
{
    uint32_t c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x = (1u << ((b.w[r] >> (8 * k)) & 0xffu)) - 1u;
            const uint32_t carry0 = c0 & x;
            c0 ^= x;
            c2 |= c1 & carry0;
            c1 ^= carry0;
        }
    }
    return 4u * popc(c2) + 2u * popc(c1 & ~c2) + popc(c0 & ~c2);
}

// The two pure per-transition shaping terms of PPOAgent.remember (agents/ppo_agent.py:253-266), added to
// reward_in in the reference's order: + 0.1 * sum(log2 of the top-4 tiles) then + 0.3 * evaluate_heuristic.
G2048_HD double eval_ppo_shaping(const Board &b, double reward_in)
{
    double reward = reward_in;
    reward += 0.1 * (double)top4_code_sum(b);
    reward += 0.3 * eval_ppo_heuristic(b);
    return reward;
}

G2048_HD double eval_monotonicity(const Board &b, int kind /*0 ++,1 +-,2 -+,3 --*/)
{
    const MonoCounts m = mono_counts(b);
    const uint32_t r = (kind & 2) ? m.rge : m.rle;
    const uint32_t c = (kind & 1) ? m.cge : m.cle;
    return (double)(r + c) / 24.0;
}

}  // namespace g2048
