"""HBM traffic and SQ counters of the bf16 GEMM / weight-gradient kernels (last launch of each instantiation) from four rocprofv3 --pmc
passes of `tools/run_kernels.py bf16 32 1`:   python3 tools/pmc_bf16.py <fetch.csv> <write.csv> <mfma.csv> <lds.csv> [head]
bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 tallies 128-byte requests at 64 bytes: tools/pmc_traffic.py)."""
import csv, json, re, sys


def load(path):
    d = {}
    for r in csv.DictReader(open(path)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'\(.*', '', k).replace('void ', '')
        d.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])   # later launches overwrite earlier ones: the last one stays
    return d


f, w, m, l = (load(p) for p in sys.argv[1:5])
out = {'head': sys.argv[5] if len(sys.argv) > 5 else None,
       'method': 'rocprofv3 --kernel-trace --pmc (four separate passes) of tools/run_kernels.py bf16 32 1 (608 x 608, batch 32); last launch of '
                 'each kernel instantiation; bytes = 2 x FETCH_SIZE + WRITE_SIZE; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); '
                 'lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE', 'kernels': {}}
for k in sorted(f):
    if not (k.startswith('conv_bf16_kernel') or k.startswith('wgrad_bf16_kernel')):
        continue
    e = {}
    if k in w:
        e['read_bytes'] = round(2 * f[k]['FETCH_SIZE'] * 1024); e['write_bytes'] = round(w[k]['WRITE_SIZE'] * 1024)
        e['bytes'] = e['read_bytes'] + e['write_bytes']
    if k in m and m[k].get('SQ_BUSY_CU_CYCLES'):
        e['mfma_busy'] = round(m[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m[k]['SQ_BUSY_CU_CYCLES']), 4)
        e['wait_inst_any_per_wave_cycle'] = round(m[k]['SQ_WAIT_INST_ANY'] / m[k]['SQ_WAVE_CYCLES'], 4)
    if k in l and l[k].get('SQ_LDS_IDX_ACTIVE'):
        e['lds_conflict'] = round(l[k]['SQ_LDS_BANK_CONFLICT'] / l[k]['SQ_LDS_IDX_ACTIVE'], 4)
        e['lds_active_per_busy_cycle'] = round(l[k]['SQ_LDS_IDX_ACTIVE'] / m[k]['SQ_BUSY_CU_CYCLES'], 4) if k in m else None
    out['kernels'][k] = e
print(json.dumps(out, indent=1))
