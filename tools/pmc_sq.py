"""MFMA-busy and LDS-conflict fractions of the hot kernels from two rocprofv3 --pmc passes of `tools/run_kernels.py all 32 1`
(pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES, pass 2: SQ_LDS_BANK_CONFLICT
SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS; each with --kernel-trace only).
usage: python3 tools/pmc_sq.py m_counter_collection.csv l_counter_collection.csv > profiles/rNN_pmc_sq.json"""
import collections, csv, json, re, sys


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'\(.*', '', k).replace('void ', '')
        d.setdefault((k, int(r['Dispatch_Id'])), {})[r['Counter_Name']] = float(r['Counter_Value'])
    return d


m, l = load(sys.argv[1]), load(sys.argv[2])
want = ('wino_conv_kernel', 'wino4_conv_kernel', 'wino4_wgrad_kernel', 'wino_wgrad_kernel', 'wino2_conv_kernel', 'wino4s2_conv_kernel', 'wino2_wgrad_kernel', 'conv1_', 'caps1_')
out = {'method': 'rocprofv3 --pmc (two passes, --kernel-trace only) of tools/run_kernels.py all 32 1; last launch of each kernel; '
                 'mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): the MFMA counter ticks per SIMD, four per CU '
                 '(cross-check: SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = the flops the kernel issues); '
                 'lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE', 'kernels': {}}
last = {}
for (k, disp), v in m.items():
    if k.startswith(want):
        last[k] = (disp, v)
lastl = {}
for (k, disp), v in l.items():
    if k.startswith(want):
        lastl[k] = v
for k, (disp, v) in last.items():
    e = {'counters': {a: round(b) for a, b in v.items()}}
    if v.get('SQ_BUSY_CU_CYCLES'):
        e['mfma_busy'] = round(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (4.0 * v['SQ_BUSY_CU_CYCLES']), 4)
    w = lastl.get(k)
    if w:
        e['counters'].update({a: round(b) for a, b in w.items()})
        if w.get('SQ_LDS_IDX_ACTIVE'):
            e['lds_conflict'] = round(w.get('SQ_LDS_BANK_CONFLICT', 0.0) / w['SQ_LDS_IDX_ACTIVE'], 4)
    out['kernels'][k] = e
print(json.dumps(out, indent=1))
