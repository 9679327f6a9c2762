"""HBM traffic and SQ counters of the bf16 GEMM / weight-gradient kernels (last launch of each instantiation) from four rocprofv3 --pmc
passes of `tools/run_kernels.py bf16 32 1`:   python3 tools/pmc_bf16.py <fetch.csv> <write.csv> <mfma.csv> <lds.csv> [head]
bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 tallies 128-byte requests at 64 bytes: tools/pmc_traffic.py)."""
import csv, json, re, sys


def load(path):
    """kernel -> launches in dispatch order, each a dict counter -> value"""
    d = {}
    for r in csv.DictReader(open(path)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'\(.*', '', k).replace('void ', '')
        d.setdefault(k, {}).setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    return dict((k, [v[i] for i in sorted(v)]) for k, v in d.items())


f, w, m, l = (load(p) for p in sys.argv[1:5])
# tools/run_kernels.py bf16 <B> 1 launches, in this order, twice each: conv_2 forward + statistics (N = 256), conv_2 input gradient with
# fp32 output (N = 128), conv_2 weight gradient, conv_3 input gradient (4 classes in one launch, N = 256) with the BatchNorm sums, the same
# without them: so <256, 256, ...> without BNF is launch 2 = conv_2 forward and launch 4 = conv_3 input gradient
LABELS = {('conv_bf16_kernel<256, 256, 2, 4, false, false, false>', 1): 'conv_2 forward + statistics',
          ('conv_bf16_kernel<256, 256, 2, 4, false, false, false>', 3): 'conv_3 input gradient, 4 classes in one launch',
          ('conv_bf16_kernel<256, 256, 2, 4, false, true, false>', 1): 'conv_3 input gradient, 4 classes in one launch, with the BatchNorm sums',
          ('conv_bf16_kernel<512, 128, 4, 2, true, false, true>', 1): 'conv_2 input gradient (fp32 output)',
          ('wgrad_bf16_kernel<3, 1, 4, 8, 1, false>', 1): 'conv_2 weight gradient',
          ('wgrad_bf16_kernel<3, 1, 4, 8, 1, true>', 1): 'conv_2 weight gradient with the BatchNorm-backward apply on the way in'}
out = {'head': sys.argv[5] if len(sys.argv) > 5 else None,
       'method': 'rocprofv3 --kernel-trace --pmc (four separate passes) of tools/run_kernels.py bf16 32 1 (608 x 608, batch 32); the second '
                 'launch of each operation; bytes = 2 x FETCH_SIZE + WRITE_SIZE; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); '
                 'lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE', 'kernels': {}}
for (k, idx), label in LABELS.items():
    if k not in f or idx >= len(f[k]):
        continue
    e = {'kernel': k}
    if k in w and idx < len(w[k]):
        e['read_bytes'] = round(2 * f[k][idx]['FETCH_SIZE'] * 1024); e['write_bytes'] = round(w[k][idx]['WRITE_SIZE'] * 1024)
        e['bytes'] = e['read_bytes'] + e['write_bytes']
    if k in m and idx < len(m[k]) and m[k][idx].get('SQ_BUSY_CU_CYCLES'):
        c = m[k][idx]
        e['mfma_busy'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']), 4)
        e['wait_inst_any_per_wave_cycle'] = round(c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], 4)
    if k in l and idx < len(l[k]) and l[k][idx].get('SQ_LDS_IDX_ACTIVE'):
        c = l[k][idx]
        e['lds_conflict'] = round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 4)
    out['kernels'][label] = e
print(json.dumps(out, indent=1))
