"""Count the instructions between consecutive MFMAs of a kernel's main loop in hipcc -S output.
usage: python3 tools/slot_census.py file.s kernel_substring"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(key) + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
# main loop = the region between the labels that holds the most MFMAs: take everything from the first loop
# header to the last backward branch
mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
cnt, cur, kinds = [], 0, []
ck = {}
for l in body[mf[0]:mf[-1] + 1]:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'):
        cnt.append(cur); kinds.append(ck); cur = 0; ck = {}
    else:
        cur += 1
        k = 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else 'salu' if op.startswith('s_') else 'valu'
        ck[k] = ck.get(k, 0) + 1
print('mfma count', len(mf))
for i, (c, k) in enumerate(zip(cnt, kinds)):
    print('%3d: %3d  %s' % (i, c, ' '.join('%s=%d' % kv for kv in sorted(k.items()))))
