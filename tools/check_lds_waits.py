"""Static check of hand-counted `s_waitcnt lgkmcnt(N)` against the ISA hipcc emitted.

Several kernels read LDS with inline asm the compiler does not track and wait with a counted `s_waitcnt lgkmcnt(N)`
(csrc/winograd.hip forward / weight gradient (both variants), csrc/winograd_s2.hip forward / input gradient / weight
gradient, csrc/conv_bf16.hip, csrc/routing_rows.hip).  LDS operations complete
in order, so a read's result is ready at its first use iff SOME wait between the read and the use allows at most as many
outstanding LDS operations as were issued after the read up to that wait.  A compiler that merges, drops or reorders LDS
instructions would break a hand-counted N silently; this script disassembles each kernel, takes the innermost loop that
holds the expected number of LDS reads and verifies every LDS read of it (one iteration feeding the next included).

    python3 tools/check_lds_waits.py            # exit code 0 = every read is covered in every kernel below

Run by __graft_entry__.build() and tests/test_host_logic.py (CPU: hipcc cross-compiles)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'cs231-capsule-yolo-traffic-sign-detection_amd', 'csrc')
LDS = re.compile(r'^\s+(ds_read|ds_write|ds_bpermute|ds_swizzle|ds_permute)')

# (source file, extra flags, mangled-name fragments that select the kernel, minimum LDS reads of the loop to check)
TARGETS = [
    ('winograd.hip', [], ['wino_conv_kernelILi1', 'WinoArgsE'], 40),
    ('winograd.hip', [], ['wino_wgrad_kernelILi0E'], 10),
    ('winograd.hip', [], ['wino_wgrad_kernelILi1E'], 10),
    ('winograd.hip', [], ['wino_wgrad_kernelILi2E'], 10),
    ('winograd_s2.hip', [], ['wino2_conv_kernelILi0ELb1'], 10),
    ('winograd_s2.hip', [], ['wino2_conv_kernelILi1ELb0'], 10),
    ('winograd_s2.hip', [], ['wino2_wgrad_kernelILb1'], 10),
    ('conv_bf16.hip', [], ['conv_bf16_kernelILi256ELi256ELi2ELi4ELb0'], 24),
    ('conv_bf16.hip', [], ['conv_bf16_kernelILi512ELi128ELi4ELi2ELb0'], 24),
    ('routing_rows.hip', ['-fno-slp-vectorize'], ['caps_rows_kernelILi21ELi16ELi3ELi0ELi2'], 120),
    ('routing_rows.hip', ['-fno-slp-vectorize'], ['caps_rows_kernelILi16ELi16ELi3ELi1ELi2'], 90),
]


def vregs(tok):                                                # 'v[2:5]' -> {2,3,4,5}; 'v17' -> {17}
    tok = tok.strip()
    m = re.match(r'v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def check_kernel(lines, frags, min_reads):
    start = next((i for i, l in enumerate(lines) if l.startswith('_ZN') and l.split(':')[0].endswith(tuple(['E', 't'])) and all(f in l for f in frags)
                  and re.match(r'^_ZN[^ ]*:', l)), None)
    if start is None:
        return 'kernel not found', 0, 1
    body = []
    for l in lines[start:]:
        if l.startswith('.Lfunc_end'):
            break
        body.append(l)
    labels = dict((m.group(1), i) for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m)
    loop = None
    for i, l in enumerate(body):
        m = re.match(r'\s+s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a, b = labels[m.group(1)], i
            n = sum(1 for k in range(a, b) if re.match(r'^\s+ds_read', body[k]))
            if n >= min_reads and (loop is None or b - a < loop[1] - loop[0]):
                loop = (a, b)
    if loop is None:
        return 'no loop with >= %d LDS reads' % min_reads, 0, 1
    seq = body[loop[0]:loop[1]]
    stream = []                                                # (kind, mnemonic, dst regs, src regs, lgkmcnt, text) of two iterations
    for it in range(2):
        for l in seq:
            m = re.match(r'^\s+([a-z_0-9]+)\s*(.*)$', l)
            if not m:
                continue
            mn, rest = m.group(1), m.group(2).split(';')[0]
            ops = [o.split(' ')[0] for o in re.split(r',\s*', rest.strip()) if o] if rest.strip() else []
            if mn == 's_waitcnt':
                w = re.search(r'lgkmcnt\((\d+)\)', rest)
                stream.append(('wait', mn, set(), set(), int(w.group(1)) if w else None, l))
                continue
            if mn == 's_barrier':
                continue
            has_dst = mn.startswith('v_') or mn.startswith('ds_read') or mn.startswith('global_load_dword') or mn.startswith('ds_bpermute')
            dst = vregs(ops[0]) if (has_dst and ops) else set()
            src = set()
            for o in (ops[1:] if has_dst else ops):
                src |= vregs(o)
            if mn.startswith('v_mfma') or 'fmac' in mn or mn.endswith('_dpp') or mn.startswith('v_pk_fma') and len(ops) == 4 and ops[0] == ops[3]:
                src |= dst                                     # read-modify-write destinations
            stream.append(('lds' if LDS.match(l) else 'op', mn, dst, src, None, l))
    half = len(stream) // 2
    bad = checked = 0
    for k in range(half):
        kind, mn, dst, src, _, text = stream[k]
        if kind != 'lds' or not mn.startswith('ds_read') or not dst:
            continue
        use, live = None, set(dst)
        for q in range(k + 1, min(len(stream), k + half)):
            if stream[q][3] & live:
                use = q
                break
            live -= stream[q][2]                               # overwritten before any use: not our value any more
            if not live:
                break
        if use is None:
            continue
        ok, n_lds = False, 0
        for q in range(k + 1, use):
            if stream[q][0] == 'lds':
                n_lds += 1
            elif stream[q][0] == 'wait' and stream[q][4] is not None and n_lds >= stream[q][4]:
                ok = True
                break
        checked += 1
        if not ok:
            bad += 1
            print('    NOT COVERED: %s ... first used by %s' % (text.strip(), stream[use][5].strip()))
    return None, checked, bad


def main():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    asm, total_bad = {}, 0
    with tempfile.TemporaryDirectory() as td:
        for src, flags, frags, min_reads in TARGETS:
            key = (src, tuple(flags))
            if key not in asm:
                out = os.path.join(td, src + '.s')
                subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                                '-munsafe-fp-atomics'] + flags + ['-S', '--cuda-device-only', os.path.join(CSRC, src), '-o', out],
                               check=True, stderr=subprocess.DEVNULL)
                asm[key] = open(out).read().split('\n')
            err, checked, bad = check_kernel(asm[key], frags, min_reads)
            name = '%s: %s' % (src, ' '.join(frags))
            if err:
                print('%s: %s' % (name, err))
                total_bad += 1
            else:
                print('%s: %d LDS reads of the loop checked against the waits in front of their first use, %d not covered'
                      % (name, checked, bad))
                total_bad += bad
    print('check_lds_waits: %s' % ('ok' if total_bad == 0 else '%d problem(s)' % total_bad))
    return 1 if total_bad else 0


if __name__ == '__main__':
    sys.exit(main())
