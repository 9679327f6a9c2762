"""Static check of the hand-counted vector-memory waits of winograd4.hip (csrc): its loop loads the B-operand ring and the input
patch with inline-asm `global_load_dwordx4` that hipcc does not count, and waits for them with `s_waitcnt vmcnt(N)` statements
whose N is a LOWER bound of the operations younger than the load that is needed.

The script compiles the file to ISA and, per kernel, walks EVERY control-flow path of the emitted code (both sides of each
conditional branch; a loop is re-entered until the state at its labels repeats) with the queue of the outstanding vector-memory
operations (loads, stores, atomics: they retire in order) as the state.  Every
`s_waitcnt vmcnt(N)` (hand-written or hipcc's) retires all but the N youngest.  An instruction that reads or writes a destination
register of a load still in the queue is an error: the wait in front of it was too weak (or hipcc moved / copied the register
between the load and its wait: cdna_hip_programming.md section 5.7 item 1).

    python3 tools/check_vmcnt.py        # exit code 0 = every use of a loaded register stands behind a sufficient wait

Run by __graft_entry__.build() and tests/test_host_logic.py (CPU only: hipcc cross-compiles)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'cs231-capsule-yolo-traffic-sign-detection_amd', 'csrc')
SOURCES = ['winograd4.hip', 'winograd4_wgrad.hip', 'winograd4_s2.hip']
MIN_MFMA = {}
ONLY = {}
# Kernels with hand-waited asm loads that are NOT path-walked (the bf16 weight gradient: its two-bodies-per-iteration loop has a path in
# the emitted code -- first body, no second body, loop again -- that the loop condition excludes and the walker cannot know) are held to
# ZERO scratch instead: a spilled asm destination is stored before its data has arrived, and scratch operations count in vmcnt.
NO_SCRATCH = {'conv_bf16.hip': 'wgrad_bf16_kernel'}


def regs(tok):
    tok = tok.strip()
    m = re.match(r'([va])\[(\d+):(\d+)\]$', tok)
    if m:
        return set((m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r'([va])(\d+)$', tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def parse(line):
    m = re.match(r'^\s+([a-z_0-9]+)\s*(.*)$', line)
    if not m:
        return None
    mn, rest = m.group(1), m.group(2).split(';')[0].strip()
    ops = [o.strip().split(' ')[0] for o in re.split(r',\s*', rest) if o.strip()] if rest else []
    return mn, ops, rest


def kernels(asm):
    lines = asm.split('\n')
    out, cur, name = [], None, None
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m and 'kernel' in m.group(1):
            name, cur = m.group(1), []
        elif l.startswith('.Lfunc_end') and cur is not None:
            out.append((name, cur)); cur = None
        elif cur is not None:
            cur.append(l)
    return out


def check_kernel(name, body, max_steps=20000000, min_mfma=144):
    """Walks every control-flow path of the kernel (both sides of each conditional branch, loops until the state repeats) with the
    in-order queue of outstanding vector-memory operations as the state."""
    if sum(1 for l in body if 'v_mfma' in l) < min_mfma:
        return None
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.L\w+):', l)
        if m:
            labels[m.group(1)] = i
    instrs = [parse(l) for l in body]
    errors, seen, nloads, nwaits = {}, {}, set(), set()
    stack = [(0, ())]
    steps = 0
    while stack:
        pc, queue = stack.pop()
        queue = list(queue)
        while pc < len(body):
            if body[pc].startswith('.L'):
                # state at a label: the loads in flight, and how many stores stand behind each of them.  A state whose loads are
                # the same and whose store counts are all >= those of a state already walked is covered by it (every wait retires
                # at least as much there), so paths that differ only in which conditional stores ran do not multiply.
                while queue and not queue[0][0]:
                    queue.pop(0)              # stores older than every load in flight never matter again
                loads = tuple(e for e in queue if e[0])
                gaps, n = [], 0
                for e in reversed(queue):
                    if e[0]:
                        gaps.append(n); n = 0
                    else:
                        n += 1
                gaps = tuple(reversed(gaps))
                known = seen.setdefault((pc, loads), [])
                if any(all(k <= g for k, g in zip(kg, gaps)) for kg in known):
                    break
                known.append(gaps)
            p = instrs[pc]
            pc += 1
            if p is None:
                continue
            steps += 1
            if steps > max_steps:
                raise RuntimeError('check_vmcnt: path walk does not converge')
            mn, ops, rest = p
            if mn == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', rest)
                if m:
                    nwaits.add(pc)
                    n = int(m.group(1))
                    while len(queue) > n:
                        queue.pop(0)
                continue
            if mn == 's_endpgm':
                break
            touched = set()
            for o in ops:
                touched |= regs(o)
            for dst, at in queue:
                if dst & touched:
                    errors[(pc - 1, at)] = '%s: line %d `%s` touches %s of the load at line %d, still in flight' % (
                        name[:60], pc - 1, body[pc - 1].strip(), sorted(dst & touched)[:2], at)
            if mn.startswith('global_load') or mn.startswith('buffer_load') or mn.startswith('scratch_load'):
                # (an LDS-DMA load -- `... offen lds` -- has no destination register: its first operand is the address)
                dst = frozenset() if re.search(r'\blds\b', rest) else frozenset(regs(ops[0]))
                queue.append((dst, pc - 1 if dst else -1)); nloads.add(pc)
            elif mn.startswith('global_store') or mn.startswith('global_atomic') or mn.startswith('buffer_store') or mn.startswith('scratch_store'):
                queue.append((frozenset(), -1))
            if len(queue) > 63:
                queue.pop(0)                  # (the counter saturates; older operations have long retired)
            if mn == 's_branch':
                pc = labels[ops[0]]
            elif mn.startswith('s_cbranch'):
                stack.append((labels[ops[0]], tuple(queue)))
    return len(nloads), len(nwaits), list(errors.values())


def main():
    bad = 0
    for src in SOURCES:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, 'k.s')
            subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                            '-munsafe-fp-atomics', '-Wno-unused-result', '-S', '--cuda-device-only', '-o', out, os.path.join(CSRC, src)],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            asm = open(out).read()
        for name, body in kernels(asm):
            if src in ONLY and ONLY[src] not in name:
                continue
            r = check_kernel(name, body, min_mfma=MIN_MFMA.get(src, 144))
            if r is None:
                continue
            nloads, nwaits, errors = r
            print('%s %s: %d loads and %d vmcnt waits on the walked paths, %d uses of a register whose load is still in flight'
                  % (src, name[:70], nloads, nwaits, len(errors)))
            for e in errors[:10]:
                print('   ', e)
            bad += len(errors)
    for src, frag in NO_SCRATCH.items():
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, 'k.s')
            subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                            '-munsafe-fp-atomics', '-Wno-unused-result', '-S', '--cuda-device-only', '-o', out, os.path.join(CSRC, src)],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            asm = open(out).read()
        for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', asm, re.S):
            if frag not in m.group(1):
                continue
            sz = int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(2)).group(1))
            print('%s %s: %d bytes of scratch per lane (hand-waited loads: must be 0)' % (src, m.group(1)[:70], sz))
            bad += 1 if sz else 0
    print('check_vmcnt: ok' if bad == 0 else 'check_vmcnt: FAILED')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
