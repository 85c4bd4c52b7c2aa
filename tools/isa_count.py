#!/usr/bin/env python3
"""Count instructions per kernel in a hipcc -S listing (VALU / SALU / VMEM / LDS, f64, perm ...).
usage: isa_count.py file.s [name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    ins = [l.strip().split()[0] for l in body.split('\n')
           if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter()
    for x in ins:
        k = ('valu' if x.startswith('v_') else 'salu' if x.startswith('s_') else
             'vmem' if x.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else
             'lds' if x.startswith('ds_') else 'other')
        c[k] += 1
    extra = dict(f64=sum('f64' in x for x in ins), perm=ins.count('v_perm_b32'),
                 cndmask=sum(x.startswith('v_cndmask') for x in ins),
                 mul_lo=sum(x.startswith('v_mul_lo') for x in ins),
                 sdwa=sum(x.endswith('_sdwa') for x in ins),
                 branch=sum(x.startswith('s_cbranch') for x in ins))
    print(name[:70], dict(c), extra)
