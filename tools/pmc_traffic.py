"""HBM traffic per launch of the hot kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each in its own
run, --kernel-trace only) of `python3 tools/run_kernels.py all 32 1`.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests as 64 bytes for wide
coalesced reads, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16-byte-per-lane stores.  Both counters are
in KB.  Calibration inside this data set: the routing kernel streams 88.8 MB algorithmically and reads
2 x 43.6 MB = 87.3 MB.

usage: python3 tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv [git head] > profiles/rNN_pmc_traffic.json"""
import collections, csv, json, re, sys


def load(path, name):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != name:
            continue
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'\(.*', '', k).replace('void ', '')
        d.setdefault(k, []).append(float(r['Counter_Value']) * 1024.0)
    return d


f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
# launch order inside tools/run_kernels.py: warm-up + 1 rep of each op, forward before input gradient
pick = {
    'conv_wino_fwd/conv_2': ('wino_conv_kernel<1>', -1),      # (with the BatchNorm statistics, as in the training step)
    'conv_wino_dgrad/conv_2': ('wino_conv_kernel<0>', -1),
    'conv_wino4_fwd/conv_2': ('wino4_conv_kernel<1>', -1),
    'conv_wino4_dgrad/conv_2': ('wino4_conv_kernel<0>', -1),
    'conv_wino4_wgrad/conv_2': ('wino4_wgrad_kernel<0>', -1),
    'conv_wino4_wgrad_bn/conv_2': ('wino4_wgrad_kernel<4>', -1),
    'conv_wino_wgrad/conv_2': ('wino_wgrad_kernel<0>', 1),
    'conv_wino_wgrad_bn/conv_2': ('wino_wgrad_kernel<2>', -1),    # (premasked gradient, as in the training step)
    'routing_fwd': ('caps1_fwd_kernel<5, true>', -1),
    'routing_bwd': ('caps1_bwd_kernel<5, true>', -1),
    'conv_wino42_fwd/conv_3': ('wino4s2_conv_kernel<1, false, 0>', -1),      # (with the BatchNorm statistics)
    'conv_wino42_dgrad/conv_3': ('wino4s2_conv_kernel<0, false, 2>', -1),    # (with conv_2's BatchNorm-backward sums: z read, premasked store)
    'conv_wino42_dgrad_plain/conv_3': ('wino4s2_conv_kernel<0, false, 1>', -1),
    'conv_wino2_fwd/conv_3': ('wino2_conv_kernel<0, false>', -1),
    'conv_wino2_dgrad/conv_3': ('wino2_conv_kernel<1, false>', -1),
    'conv_wino2_wgrad/conv_3': ('wino2_wgrad_kernel<false>', -1),
    'conv_gemm_fwd/conv_3': ('conv_gemm_kernel<1, true>', -1),
    'conv_wgrad/conv_3': ('conv_wgrad_kernel<2, 1, 2, 2, true>', -1),
}
out = {'head': sys.argv[3] if len(sys.argv) > 3 else None,
       'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/run_kernels.py all 32 1; '
                 'bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes)',
       'kernels': {}}
for key, (kern, idx) in pick.items():
    if kern in f and kern in w:
        rd, wr = 2.0 * f[kern][idx], w[kern][idx]
        out['kernels'][key] = {'kernel': kern, 'read_bytes': round(rd), 'write_bytes': round(wr), 'bytes': round(rd + wr)}
# the general routing kernels (C = 43 heads): last launch of every instantiation that ran
for kern in f:
    if (kern.startswith('caps_rows_kernel') or kern.startswith('caps_bwd_kernel')) and kern in w:
        rd, wr = 2.0 * f[kern][-1], w[kern][-1]
        out['kernels']['routing_c43/' + kern] = {'kernel': kern, 'read_bytes': round(rd), 'write_bytes': round(wr), 'bytes': round(rd + wr)}
print(json.dumps(out, indent=1))
