#!/usr/bin/env python3
"""Estimate VALU issue cycles of a kernel from a hipcc -S listing using the measured gfx950 rates
(tools/ubench: ~2.5 cycles for the simple VOP2 integer ops, ~4.3 for the rest, v_rcp_f64 16).
usage: isa_cost.py file.s name-substring [--hist]"""
import collections
import re
import sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32",
        "v_lshrrev_b32", "v_add_co_u32", "v_addc_co_u32"}
s = open(sys.argv[1]).read()
flt = sys.argv[2]
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    ins = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    valu = [x for x in ins if x.startswith('v_')]
    cyc = 0.0
    for x in valu:
        base = x.replace("_e32", "").replace("_e64", "")
        cyc += 16.0 if base.startswith("v_rcp_f64") else (2.5 if base in FAST else 4.3)
    print(name[:80], "valu", len(valu), "fast", sum(x.replace("_e32", "") in FAST for x in valu), "est cycles/wave %.0f" % cyc)
    if "--hist" in sys.argv:
        for k, v in collections.Counter(valu).most_common(60):
            print("   %4d %s" % (v, k))
