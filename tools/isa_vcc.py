#!/usr/bin/env python3
"""Finds back-to-back v_cndmask_b32 in VOP2 form (implicit VCC): on gfx950 every VOP2 v_cndmask whose previous VALU
instruction is also a VOP2 v_cndmask (scalar instructions in between do not help) costs ~22 cycles instead of ~4.5; the
VOP3 encoding reading the same vcc does not (tools/ubench/vcc_forms.hip, profiles/r02_vcc_forms.txt).
usage: isa_vcc.py file.s [kernel-substring]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    prev_valu = None
    n_e32 = n_slow = 0
    for ln, l in enumerate(body.split('\n')):
        t = l.strip()
        if not l.startswith('\t') or t.startswith(('.', ';')):
            continue
        op = t.split()[0]
        if not op.startswith('v_'):
            continue
        vop2 = op.startswith('v_cndmask_b32_e32') or op == 'v_cndmask_b32'
        if vop2:
            n_e32 += 1
            if prev_valu:
                n_slow += 1
        prev_valu = vop2
    print("%-90s VOP2 v_cndmask %3d, of which right after another one %3d" % (name[:90], n_e32, n_slow))
