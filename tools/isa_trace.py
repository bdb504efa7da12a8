"""Compact instruction-class trace of one kernel of a `hipcc -S` listing: M = MFMA, v = VALU, r / w = LDS read / write, g / G = global
load / store, s = SALU, W(..) = s_waitcnt, BAR = s_barrier, BR = branch; run lengths behind the letter.
usage: isa_trace.py <listing.s> <kernel-name regex> [label to start at]"""
import re, sys
s = open(sys.argv[1]).read()
name = [n for n in re.findall(r'^(_Z\S*):', s, re.M) if re.search(sys.argv[2], n)][0]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
def cls(l):
    p = l.split(None, 1); op = p[0]; arg = p[1] if len(p) > 1 else ''
    if op.startswith('.LBB'): return '\n' + op + '\n'
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('v_'): return 'v'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'r'
    if op.startswith('ds_'): return 'w'
    if op.startswith('s_waitcnt'): return 'W(' + arg.replace(' ', '') + ')'
    if op.startswith('s_barrier'): return 'BAR'
    if op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('s_load'): return 'g'
    if op.startswith('global_store') or op.startswith('global_atomic'): return 'G'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'BR'
    if op.startswith('scratch'): return 'SCRATCH'
    if op.startswith('s_'): return 's'
    return '?' + op
res = []; prev = None; cnt = 0
for l in s[i:j].split('\n')[1:]:
    l = l.split(';')[0].strip()
    if not l or (l.startswith('.') and not l.startswith('.LBB')): continue
    c = cls(l)
    if c == prev and len(c) == 1: cnt += 1
    else:
        if prev is not None: res.append(prev + (str(cnt) if cnt > 1 else ''))
        prev = c; cnt = 1
res.append(prev + (str(cnt) if cnt > 1 else ''))
t = ' '.join(res)
if len(sys.argv) > 3: t = t[t.index(sys.argv[3]):]
print(t)
