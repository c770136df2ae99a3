"""Instruction mix per kernel from `make asm` output (/tmp/ocean_api.s)."""
import re, sys
from collections import Counter
pat = sys.argv[1] if len(sys.argv) > 1 else "2048"
s = open('/tmp/ocean_api.s').read()
parts = re.split(r'\n(_ZN5ocean\S*):\s*;[^\n]*\n', s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    if pat not in name: continue
    body = body.split('.Lfunc_end')[0]
    c = Counter(); n = 0
    for l in body.split('\n'):
        if not l.startswith('\t'): continue
        t = l.strip()
        if not t or t[0] in '.;': continue
        op = t.split()[0]; n += 1
        if op.startswith('v_'):
            c['valu_f64' if 'f64' in op else ('valu_trans' if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', op) else 'valu')] += 1
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): c['branch'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith(('ds_', 'global_', 'buffer_', 'scratch_', 'flat_')): c[op] += 1
        else: c['other'] += 1
    print(name[9:60], n, dict(sorted(c.items())))
