"""Instruction census of every loop of one kernel in a gfx950 assembly file (hipcc -S --cuda-device-only):
python tools/loop_census.py file.s <mangled-name fragment>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
frag = sys.argv[2]
start = re.search(r'^\S*%s\S*:' % re.escape(frag), s, re.M).start()
lines = s[start:s.index('.Lfunc_end', start)].split('\n')
labels = {}
for n, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = n
loops = []
for n, l in enumerate(lines):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
        loops.append((labels[mm.group(1)], n))
for a, b in loops:
    c = collections.Counter()
    for l in lines[a:b + 1]:
        t = l.strip().split()
        if not t or t[0].endswith(':') or t[0].startswith(';') or t[0].startswith('.'):
            continue
        op = t[0]
        key = ('mfma' if 'mfma' in op else 'accvgpr' if 'accvgpr' in op else 'pk_fma' if 'pk_fma' in op else 'pk_mul' if 'pk_mul' in op
               else 'pk_add' if 'pk_add' in op else 'ds_read' if op.startswith('ds_read') else 'ds_write' if op.startswith('ds_write')
               else 'dpp' if 'dpp' in l else 'exp' if 'v_exp' in op else 's_waitcnt' if op == 's_waitcnt' else 's_nop' if op == 's_nop'
               else 'scratch' if op.startswith('scratch') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_')
               else 'vmem' if op.startswith('global') or op.startswith('buffer') else 'other')
        c[key] += 1
    print('loop of %d lines: %s' % (b - a, dict(c)))
