"""Static check of the software wait states behind every MFMA in the ISA hipcc emitted for csrc/*.hip.

gfx950 does NOT interlock an MFMA's destination against the next non-accumulating access: after `v_mfma D, A, B, C` any
instruction that reads or writes a register of D other than an MFMA that takes D whole as its C operand (the accumulate
chain: 0 states) must stand at least  passes + 4  wait states behind it (cdna_hip_programming.md section 5.7 item 2: "8-pass
XDL: 12 states"; 4-pass 8, 16-pass 20 -- one state per issued instruction, `s_nop N` = N + 1).  hipcc pads these pairs
itself for MFMAs it generates -- but an `asm volatile` MFMA is ONE opaque statement to it: neither the hazard recognizer nor
the scheduler looks inside, so a compiler-placed `v_accvgpr_read` / `v_mov` / VALU consumer right behind the statement reads
the accumulator before the last pass has landed (lanes 48-63 are written by the last of the four 16-lane passes: the
run-to-run differences in exactly those lanes that parked the fused bf16 weight gradient in round 2).
The kernels in winograd.hip and winograd_s2.hip issue their MFMAs from inline asm with explicit register classes (conv_bf16.hip
and conv.hip keep them next to hand-placed waits), so this script checks EVERY v_mfma of every kernel of those sources:

  * D -> next reader / writer of any register of D that is not an accumulating MFMA on the same D:   >= passes + 4 states
    (stricter than what hipcc pads its own MFMAs with -- passes + 2 for the f32 forms, e.g. 10 states behind a
    v_mfma_f32_16x16x4_f32 in routing_mfma.hip -- so sources whose MFMAs are all compiler-generated builtins are only listed
    with --all, for information: the compiler's hazard recognizer owns those)

The walk follows the straight-line code behind the MFMA, takes both sides of a branch conservatively (fall-through and the
target label) and stops a path once enough states have passed.

    python3 tools/check_mfma_hazards.py [--all | file.hip ...]      # exit code 0 = no pair closer than required

Run by __graft_entry__.build() and tests/test_host_logic.py (CPU only: hipcc cross-compiles)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'cs231-capsule-yolo-traffic-sign-detection_amd', 'csrc')
FLAGS = {'routing_rows.hip': ['-fno-slp-vectorize'], 'routing_caps.hip': ['-fno-slp-vectorize']}

ASM_MFMA_SOURCES = ['winograd.hip', 'winograd4.hip', 'winograd4_wgrad.hip', 'winograd4_s2.hip', 'winograd_s2.hip', 'conv_bf16.hip', 'conv.hip']    # checked by default (and by build())

# passes of the MFMA forms these sources use (MI355X_MICROARCH.md, matrix-core cycle table: cycles per SIMD / 4)
PASSES = [(r'v_mfma_f32_32x32x2_?f32', 16), (r'v_mfma_f32_16x16x4_?f32', 8), (r'v_mfma_f32_32x32x16_bf16', 8),
          (r'v_mfma_f32_16x16x32_bf16', 4), (r'v_mfma_f32_32x32x8_?bf16', 16), (r'v_mfma_f32_16x16x16_?bf16', 8)]


def regs(tok):
    """'v[2:5]' -> {('v',2)..('v',5)}, 'a7' -> {('a',7)}; anything else -> empty."""
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


def passes_of(mn):
    for pat, n in PASSES:
        if re.match(pat + r'$', mn):
            return n
    return 16 if mn.startswith('v_mfma') else None          # an unknown form: assume the longest


def states_of(mn, rest):
    if mn == 's_nop':
        return int(rest.strip() or 0, 0) + 1
    return 1


def touches(mn, ops, dset):
    """Does this instruction read or write a register of dset?  (Every register operand counts: a store's data, an address, a
    VALU source or destination, a load's destination.)"""
    hit = set()
    for o in ops:
        hit |= regs(o) & dset
    return hit


def check_body(body, name):
    labels = dict((m.group(1), i) for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m)
    bad = []
    n_mfma = 0
    for i, l in enumerate(body):
        p = parse(l)
        if not p or not p[0].startswith('v_mfma'):
            continue
        mn, ops, _ = p
        n_mfma += 1
        need = passes_of(mn) + 4
        D = regs(ops[0])
        # walk: (index, states so far); both sides of every branch
        work, seen = [(i + 1, 0)], set()
        while work:
            k, st = work.pop()
            while k < len(body) and st < need:
                if (k, st) in seen:
                    break
                seen.add((k, st))
                q = parse(body[k])
                if not q:
                    k += 1
                    continue
                qmn, qops, qrest = q
                if qmn.startswith('v_mfma'):
                    qD, qC = regs(qops[0]), regs(qops[3]) if len(qops) > 3 else set()
                    if qD == D and qC == D:
                        break                                   # the accumulate chain continues: this MFMA now owns D (checked on its own)
                    if (qD | qC | regs(qops[1]) | regs(qops[2])) & D:
                        bad.append((name, l.strip(), body[k].strip(), st, need))
                        break
                    st += 1                                     # (an independent MFMA: one state, as LLVM's hazard recognizer counts it)
                    k += 1
                    continue
                if touches(qmn, qops, D):
                    bad.append((name, l.strip(), body[k].strip(), st, need))
                    break
                if qmn in ('s_endpgm',):
                    break
                mb = re.match(r's_c?branch\w*$', qmn)
                if mb:
                    tgt = qops[-1] if qops else None
                    if tgt in labels:
                        work.append((labels[tgt], st + 1))
                    if qmn == 's_branch':
                        break
                st += states_of(qmn, qrest)
                k += 1
    return n_mfma, bad


def kernels_of(lines):
    """(name, body lines) of every function in a hipcc -S device listing."""
    out, cur, name = [], None, None
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if l.startswith('.Lfunc_end') and cur is not None:
            out.append((name, cur))
            cur = None
            continue
        if cur is not None:
            cur.append(l)
    return out


def main():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    args = [a for a in sys.argv[1:] if a != '--all']
    srcs = args or (sorted(f for f in os.listdir(CSRC) if f.endswith('.hip')) if '--all' in sys.argv else ASM_MFMA_SOURCES)
    total_bad = 0
    with tempfile.TemporaryDirectory() as td:
        for src in srcs:
            path = src if os.path.isabs(src) else os.path.join(CSRC, src)
            out = os.path.join(td, os.path.basename(src) + '.s')
            subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                            '-munsafe-fp-atomics'] + FLAGS.get(os.path.basename(src), []) + ['-S', '--cuda-device-only', path, '-o', out],
                           check=True, stderr=subprocess.DEVNULL)
            n_all, bad_all = 0, []
            for name, body in kernels_of(open(out).read().split('\n')):
                n, bad = check_body(body, name)
                n_all += n
                bad_all += bad
            print('%s: %d MFMAs checked, %d closer to an access of their destination than passes + 4 states'
                  % (os.path.basename(src), n_all, len(bad_all)))
            for name, mf, cons, st, need in bad_all[:12]:
                print('    %s\n      %s\n      -> %s   (%d states, %d needed)' % (name[:90], mf, cons, st, need))
            if os.path.basename(src) in ASM_MFMA_SOURCES or args:
                total_bad += len(bad_all)
    print('check_mfma_hazards: %s' % ('ok' if total_bad == 0 else '%d violation(s)' % total_bad))
    return 1 if total_bad else 0


if __name__ == '__main__':
    sys.exit(main())
