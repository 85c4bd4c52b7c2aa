#!/usr/bin/env python3
"""Derives the per-action selector words of the two-stage v_perm direction network (csrc/g2048_board.h, "direction by
table") by search over a symbolic model of v_perm_b32, for rows -> lines and lines -> rows, and prints them."""
# derive per-action selectors of the unified 2-stage v_perm network (in: rows -> lines, out: lines -> rows)
import itertools
def perm(s0, s1, sel):   # lists of 4 symbols; sel list of 4 ints
    src = s1 + s0
    return [src[i] for i in sel]
def selword(sel): return sum(b << (8*i) for i, b in enumerate(sel))
B = [[(r, c) for c in range(4)] for r in range(4)]       # board rows: word r byte c
def transpose(w): return [[w[r][k] for r in range(4)] for k in range(4)]
def target_in(a):
    horiz, rev = (a & 1) == 0, (a & 2) != 0
    x = transpose(B) if horiz else B
    return [x[3-k] if rev else x[k] for k in range(4)]
def solve(inp, tgt):
    # stage1: u0,u1 from (inp2, inp0); u2,u3 from (inp3, inp1), shared selectors sA (u0,u2) sB (u1,u3)
    # stage2: o0 = perm(u2,u0,sC) o1 = perm(u2,u0,sD) o2 = perm(u3,u1,sC) o3 = perm(u3,u1,sD)
    sols = []
    for sA in itertools.product(range(8), repeat=4):
        u0, u2 = perm(inp[2], inp[0], sA), perm(inp[3], inp[1], sA)
        # o0 must be buildable from u2,u0
        def find(t, hi, lo):
            src = lo + hi
            try: return [src.index(s) for s in t]
            except ValueError: return None
        c0 = find(tgt[0], u2, u0); d1 = find(tgt[1], u2, u0)
        if c0 is None or d1 is None: continue
        for sB in itertools.product(range(8), repeat=4):
            u1, u3 = perm(inp[2], inp[0], sB), perm(inp[3], inp[1], sB)
            c2 = find(tgt[2], u3, u1); d3 = find(tgt[3], u3, u1)
            if c2 is None or d3 is None: continue
            if c2 == c0 and d3 == d1:
                sols.append((sA, sB, tuple(c0), tuple(d1)))
    return sols
for a in range(4):
    tin = target_in(a)
    s = solve(B, tin)
    print("action", a, "IN  nsol", len(s), [hex(selword(x)) for x in s[0]])
    # out: lines (symbols = tin) -> rows B
    s2 = solve(tin, B)
    print("action", a, "OUT nsol", len(s2), [hex(selword(x)) for x in s2[0]])
    allin = s; allout = s2
    # prefer solutions with out == in
    common = [x for x in s if x in s2]
    print("   common", len(common), [hex(selword(y)) for y in common[0]] if common else None)
