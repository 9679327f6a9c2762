#!/usr/bin/env python
"""Headline benchmark: train images/sec of darkcapsule (GTSDB-shaped 416x416, n_grid 13, 3 routing
iterations, batch 32 per GPU, fp32, recon off) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + darkcapsule_loss + backward + (N>1: one RCCL all-reduce of the flat gradient bucket)
+ fused Adam.  `value` is measured with the batch resident in HBM before the timed region (the bench contract); the
same step fed from host memory through the device-side input pipeline is `pcie_inclusive`.  Rank 0 prints ONE JSON
line.  Extra objects on that line:
  roofline      dominant kernel (conv_2's fused Winograd kernel, fp32 MFMA): the MFMA FLOPs the kernel ISSUES per launch
                (direct-convolution FLOPs / 2.25; / 4 for the F(4x4,3x3) forward / input gradient) divided by the launch's mean duration (HIP events on the launch stream
                inside the timed steps) and by the fp32 MFMA peak; `effective_vs_direct` prices the same time against the
                direct-convolution FLOPs (can exceed 1); `traffic` = HBM bytes from the newest PMC passes in profiles/
  roofline_routing       the C = 1 routing kernel of this model (HBM-bound): algorithmic bytes / duration
  roofline_routing_c43   the general routing kernels on the C = 43 heads (CapsuleNet, DarkCapsuleNet3), timed in this run
  loss_curve_parity      20 Adam steps from the reference's default initialisation on the headline's kernels, held PER STEP to the
                envelope of the reference's own 17-run ensemble (tests/golden/curves_ens.npz): `steps_within_envelope`, the
                per-step deviations and bounds as fractions of the curve's range (dw64: well-conditioned, every step strict)
  cpu_baseline  the CPU oracle (PyTorch-CPU restatement of the reference) timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA dense peak (~2.5 PF)
PEAK_HBM_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pmc_traffic():
    """HBM bytes per launch measured with rocprofv3 PMC passes (tools/pmc_traffic.py -> profiles/*_pmc_traffic.json);
    the newest file wins.  bench.py cannot collect counters itself: it only reports what was measured."""
    import glob
    best = {}
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):
        try:
            with open(f) as fh:
                d = json.load(fh)
            for k, v in d.get('kernels', {}).items():
                best[k] = dict(v, source=os.path.basename(f), head=d.get('head'))
        except Exception:
            pass
    return best


def rocprof_launch_us(kernel_substr):
    """Average duration (us) of a kernel at its LARGEST problem size in the newest committed rocprofv3 kernel-trace summary
    (profiles/r*_bench_kernel_stats.csv, written by tools/collect_profiles.sh from the same bench command): the figure the
    HIP-event bracket inside this run is to be compared with (the bracket adds ~3 us to a 18 us kernel)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_kernel_stats.csv')))
    for f in reversed(files):
        try:
            rows = [r for r in csv.DictReader(open(f)) if kernel_substr in r['Name']]
            if rows:
                r = max(rows, key=lambda q: float(q['TotalDurationNs']))
                return {'file': 'profiles/' + os.path.basename(f), 'avg_us': round(float(r['AverageNs']) / 1e3, 3),
                        'calls': int(r['Calls']), 'entry': r['Name'][-60:]}
        except Exception:
            pass
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (BASELINE configs[2]: 32)')
    ap.add_argument('--input', type=int, default=416, help='image side; n_grid = input/32')
    ap.add_argument('--n_iter', type=int, default=3)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help="bf16: the backbone behind the first layer on bf16 MFMA kernels (BASELINE configs[4] with --input 608 --n_iter 5)")
    ap.add_argument('--sync-bn', action='store_true', help='N>1: BatchNorm statistics over the global batch (2 small all-reduces per BN layer)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=0, help='batch of the CPU baseline step (0 = --batch: the identical step)')
    ap.add_argument('--cpu-steps', type=int, default=3, help='timed CPU-baseline steps after the warm-up step (SURVEY 8d: >= 3)')
    ap.add_argument('--no-extras', action='store_true', help='skip loss_curve_parity, roofline_routing_c43 and secondary')
    ap.add_argument('--backend', default=os.environ.get('CAPSYOLO_DP_BACKEND', 'nccl'), choices=['nccl', 'gloo'],
                    help="torch.distributed backend for N>1: 'nccl' (= RCCL over xGMI, one GPU per rank; the default) or 'gloo' "
                         "(rehearsal: the ranks may share one GPU, collectives go through host memory)")
    return ap.parse_args()


def host_cores():
    """Threads the CPU baseline may use: the cgroup CPU quota if there is one, else the affinity mask,
    never more than 16 (a one-GPU box's CPU share)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(args, g):
    """Oracle (checker code, CPU) timed on the host cores: a reported baseline, never the thing shipped."""
    import torch
    from capsyolo_amd import synth
    from oracle import loss_fns as OL
    from oracle import models as OM
    cores = host_cores()
    torch.set_num_threads(cores)
    p = types.SimpleNamespace(n_classes=43, n_grid=g, n_boxes=2, dropout=0.0, recon=False, recon_coef=5e-4,
                              darknet_input=args.input, device='cpu')
    torch.manual_seed(0)
    net = OM.DarkCapsuleNet(p, n_iter=args.n_iter).train()
    opt = torch.optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
    B = args.cpu_batch or args.batch
    x = torch.from_numpy(synth.images(B, args.input)).permute(0, 3, 1, 2).contiguous()
    y = torch.from_numpy(synth.gtsdb_labels(B, g, 43))

    def step(n):
        out = net(x[:n])
        loss = OL.darkcapsule_loss(out, y[:n], p)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.item()
    nw = min(B, 4)
    step(nw)                                 # warm-up on a slice (thread pool, allocator, oneDNN primitive caches)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        step(B)
    dt = time.perf_counter() - t0
    return {'value': round(B * args.cpu_steps / dt, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': 'oracle DarkCapsuleNet %dx%d train step (forward + loss + backward + Adam), batch %d%s, %d timed step(s) after a '
                      '%d-image warm-up step, torch %s CPU, %d threads'
                      % (args.input, args.input, B, ' = the identical step' if B == args.batch else ' (of the %d-image batch)' % args.batch,
                         args.cpu_steps, nw, torch.__version__, cores)}


def git_head():
    """Short commit id of the tree being measured: from git, or from .bench_head (the GPU box receives a snapshot without
    .git; tools/gpu.sh writes the file before sending)."""
    try:
        with open(os.path.join(ROOT, '.bench_head')) as f:
            return f.read().strip() or None
    except Exception:
        pass
    try:
        import subprocess
        return subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True,
                              timeout=5).stdout.strip() or None
    except Exception:
        return None


def loss_curve_parity(dev):
    """Loss-curve parity PER STEP against the envelope of the reference's own ensemble (tests/golden/curves_ens.npz, written by
    tests/golden/make_golden.py from /root/reference: the unperturbed run, 8 one-ulp and 8 sixteen-ulp input perturbations, the run
    in double; the rule is tests/helpers.py: curve_envelope / envelope_verdict -- the one the GPU tests assert).  20 Adam steps of
    DarkCapsuleNet from the reference's DEFAULT initialisation (torch.manual_seed(1234) + the constructor: the product draws the same
    weights, checked against the digests in the fixture) on two recipes whose 2^18 first-layer pixels open every kernel gate of the
    416 x 416 headline step without a switch (first block: patch-moment statistics + one-pass backward; conv_2: F(4x4,3x3) /
    F(3x3,4x4); conv_3: F(4x4,2x2); conv_4 / conv_5 are small here and take F(2x2,2x2)):
      dw64   64 x 64, n_grid 2, batch 64: WELL-CONDITIONED (the reference's one-ulp twin stays within 1.7e-5 of the range on all 20
             steps): every step is held to 3 x that band = 5.2e-5 of the curve's range (1e-4 relative on step 0)
      di256  256 x 256, n_grid 8, batch 4: chaotic from step 3 on (Adam's sign-like first steps): strict on steps 0 .. 2, the rest
             counted against 4 sigma of the reference's own spread."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import curve_envelope, envelope_verdict, hip_curve_default_init, load_golden
    from capsyolo_amd import ops
    g = load_golden('curves_ens')
    out = {'against': 'tests/golden/curves_ens.npz (reference runs: unperturbed, 8 x one-ulp, 8 x sixteen-ulp, fp64); rule: |c_k - mean_k| <= '
                      'max(floor_k, 4 max_{j<=k} sigma_j), floor_0 = 1e-4 |mean_0|, floor_k = floor_frac x range; strict on the steps whose '
                      'envelope is <= 2 % of the range', 'steps': 20}
    for tag, floor_frac in (('dw64', None), ('di256', 2e-4)):      # None: 3 x the reference's own one-ulp band (5.2e-5 of the range)
        H, gg, B = (int(v) for v in g[tag + '_cfg'][:3])
        env = curve_envelope(g, tag, floor_frac=floor_frac)
        c = hip_curve_default_init(g, tag)
        v = envelope_verdict(c, env)
        # every kernel class of the headline step runs here too (conv_4 / conv_5 are below the F(4x4,2x2) threshold and take F(2x2,2x2))
        gate = bool(ops.USE_CONV1_MOMENTS and ops.USE_CONV1_ONEPASS and ops.USE_WINOGRAD and B * H * H >= ops.CONV1_MOMENTS_MIN_PIXELS
                    and ops.USE_WINOGRAD4 and ops.USE_WINOGRAD4_WGRAD and B * H * H >= ops.WINOGRAD4_MIN_PIXELS
                    and ops.USE_WINOGRAD4_S2 and ops.USE_WINOGRAD4_S2_DGRAD and B * (H // 2) ** 2 >= ops.WINOGRAD4_S2_MIN_PIXELS)
        out[tag] = {'config': 'DarkCapsuleNet %dx%d, n_grid %d, batch %d, the reference\'s default initialisation (seed 1234), Adam lr 1e-3, default kernels' % (H, H, gg, B),
                    'same_kernels_as_value': gate, 'ok': v['ok'],
                    'steps_within_envelope': v['steps_within_envelope'], 'strict_steps': v['strict_steps'], 'strict_steps_inside': v['strict_ok'],
                    'chaotic_steps': v['chaotic_steps'], 'chaotic_steps_inside': v['chaotic_inside'],
                    'dev_frac_of_range_per_step': [float('%.3g' % d) for d in v['dev'] / env['span']],
                    'bound_frac_of_range_per_step': [float('%.3g' % d) for d in env['bound'] / env['span']],
                    'max_dev_frac_of_range': float('%.3g' % (v['dev'] / env['span']).max()),
                    'first_step_rel_dev': float('%.3g' % (v['dev'][0] / abs(env['mean'][0]))),
                    'reference_one_ulp_band_frac_of_range': float('%.3g' % (env['one_ulp_band'] / env['span']).max()),
                    'max_dev_from_fp64_reference_frac_of_range': float('%.3g' % (np.abs(c - env['curve64']) / env['span']).max()),
                    'fp32_reference_from_fp64_reference_frac_of_range': float('%.3g' % (np.abs(env['base'] - env['curve64']) / env['span']).max()),
                    'final_loss': round(float(c[-1]), 6), 'reference_final_loss': round(float(env['base'][-1]), 6)}
    out['steps_within_envelope'] = out['dw64']['steps_within_envelope']
    return out


def routing_c43(dev, B, reps=5):
    """The general (C > 1) routing kernels on the two C = 43 heads, forward and backward, HIP events around `reps`
    back-to-back calls on the launch stream.  Both heads are bound by fp32 vector work, not by HBM: u_hat = u W is
    recomputed in every iteration (it cannot be kept: R*N*C*Dout floats), so the HBM fraction of these kernels is small
    by construction; `f32_frac` prices the FLOPs the iteration structure needs against the fp32 vector / MFMA peak."""
    import torch
    from capsyolo_amd import ops
    out = {}
    g = 13
    for name, R, N, C, Dout, gather in (('capsule_head', B, 1296, 43, 16, 0), ('darkcapsule3_head', g * g * B, 512, 43, 21, g)):
        u = torch.randn(B, 4 * g, 4 * g, 256, device=dev) if gather else torch.randn(R, N, 8, device=dev)
        W = 0.1 * torch.randn(1, N, C, 8, Dout, device=dev)
        u.requires_grad_(True)
        W.requires_grad_(True)
        v = ops.routing(u, W, 3, gather, B if gather else 0)
        gv = torch.randn_like(v)
        torch.autograd.grad(v, (u, W), gv)

        def timed(fn):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps
        ud, Wd = u.detach(), W.detach()
        f_ms = timed(lambda: ops.routing(ud, Wd, 3, gather, B if gather else 0))
        fb_ms = timed(lambda: torch.autograd.grad(ops.routing(u, W, 3, gather, B if gather else 0), (u, W), gv))
        fwd_b = 4.0 * (R * N * 8 + N * C * 8 * Dout + R * C * Dout)
        bwd_b = 4.0 * (2 * R * N * 8 + 2 * N * C * 8 * Dout + R * C * Dout)
        trip = float(R) * N * C
        fwd_f = trip * (3 * 2 * 8 * Dout + 2 * 2 * Dout + 3 * 2 * Dout + 16)      # issued: u_hat recomputed in each of the 3 iterations
        fwd_f_alg = trip * (2 * 8 * Dout + 3 * 2 * Dout + 2 * 2 * Dout + 5 * 3)    # R N C (2 Din Dout + r 2 Dout + (r-1) 2 Dout + ~5 r)
        out[name] = {'shape': {'R': R, 'N': N, 'C': C, 'Din': 8, 'Dout': Dout, 'n_iter': 3},
                     'fwd_ms': round(f_ms, 4), 'bwd_ms': round(fb_ms - f_ms, 4),
                     'algorithmic_bytes_fwd': int(fwd_b), 'algorithmic_bytes_bwd': int(bwd_b),
                     'hbm_frac_fwd': round(fwd_b / (f_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 5),
                     'hbm_frac_bwd': round(bwd_b / ((fb_ms - f_ms) * 1e-3) / 1e9 / PEAK_HBM_GBPS, 5),
                     'flops_fwd': fwd_f, 'f32_frac_fwd': round(fwd_f / (f_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                     'flops_fwd_algorithmic': fwd_f_alg,
                     'f32_frac_algorithmic_fwd': round(fwd_f_alg / (f_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                     'bound': 'fp32 vector FLOPs (arithmetic intensity %.0f flop/B against a ridge of %.0f)'
                              % (fwd_f / fwd_b, PEAK_FP32_MATRIX_TFLOPS * 1e3 / PEAK_HBM_GBPS)}
        del u, W
    out['note'] = ('launch_ms are HIP-event means over %d back-to-back calls (the CapsuleNet head is 6 short launches per '
                   'forward: the bracket includes their gaps); per-kernel rocprof durations and PMC traffic: the newest set under profiles/' % reps)
    return out


def _conv_layer_flops(net, x):
    """name -> direct-convolution FLOPs (2*M*N*K) of every conv layer of `net` for input x (NCHW), from the layer geometry."""
    from capsyolo_amd.models import HipConv2d, HipMaxPool2
    out = {}
    B, _, H, W = x.shape
    for name, m in net.named_modules():
        if isinstance(m, HipMaxPool2):
            H, W = H // 2, W // 2
        if isinstance(m, HipConv2d):
            Ho, Wo = (H + 2 * m.padding - m.k) // m.stride + 1, (W + 2 * m.padding - m.k) // m.stride + 1
            out[name.split('.')[-1]] = 2.0 * B * Ho * Wo * m.weight.shape[0] * m.weight.shape[1] * m.k * m.k
            H, W = Ho, Wo
    return out


def _timed_steps(step, warm, steps):
    import torch
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, float(loss.item())


def _dominant(ops, flops, steps, peak_by_prefix):
    """The longest conv launch class of the timer (mean ms x launches per step) with the MFMA FLOPs it issues against its peak."""
    best = None
    for key, (cnt, ms) in ops.timer.summary().items():
        kind, _, layer = key.partition('/')
        if layer not in flops or kind not in peak_by_prefix:
            continue
        issued_frac, peak = peak_by_prefix[kind]
        per_step = ms * cnt / steps
        if best is None or per_step > best['ms_per_step']:
            ach = flops[layer] * issued_frac / (ms * 1e-3) / 1e12
            best = {'kernel': key, 'launch_ms': round(ms, 4), 'ms_per_step': round(per_step, 4), 'bound': 'mfma',
                    'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                    'issued_flops_per_launch': flops[layer] * issued_frac, 'direct_conv_flops_per_launch': flops[layer]}
    return best


_KIND_PEAKS = {   # timer key prefix -> (issued / direct-convolution FLOPs, MFMA peak of the arithmetic type)
    'conv_wino_fwd': (1 / 2.25, PEAK_FP32_MATRIX_TFLOPS), 'conv_wino_dgrad': (1 / 2.25, PEAK_FP32_MATRIX_TFLOPS),
    'conv_wino4_fwd': (1 / 4.0, PEAK_FP32_MATRIX_TFLOPS), 'conv_wino4_dgrad': (1 / 4.0, PEAK_FP32_MATRIX_TFLOPS),
    'conv_wino_wgrad': (1 / 2.25, PEAK_FP32_MATRIX_TFLOPS), 'conv_wino_wgrad_bn': (1 / 2.25, PEAK_FP32_MATRIX_TFLOPS),
    'conv_wino4_wgrad': (1 / 4.0, PEAK_FP32_MATRIX_TFLOPS), 'conv_wino4_wgrad_bn': (1 / 4.0, PEAK_FP32_MATRIX_TFLOPS),
    'conv_wino2_fwd': (9 / 16, PEAK_FP32_MATRIX_TFLOPS), 'conv_wino2_dgrad': (9 / 16, PEAK_FP32_MATRIX_TFLOPS),
    'conv_wino42_fwd': (25 / 64, PEAK_FP32_MATRIX_TFLOPS), 'conv_wino42_dgrad': (25 / 64, PEAK_FP32_MATRIX_TFLOPS),
    'conv_wino2_wgrad': (9 / 16, PEAK_FP32_MATRIX_TFLOPS), 'conv_gemm_fwd': (1.0, PEAK_FP32_MATRIX_TFLOPS),
    'conv_wgrad': (1.0, PEAK_FP32_MATRIX_TFLOPS),
    'conv_bf16_fwd': (1.0, PEAK_BF16_MATRIX_TFLOPS), 'conv_bf16_wgrad': (1.0, PEAK_BF16_MATRIX_TFLOPS),
}


def secondary(dev, steps=5):
    """BASELINE configs[1], configs[4] (one GPU's share) and configs[0] (on the GPU) on the same driver-timed line, bounded: 5 timed steps each after
    3 / 2 warm-up steps with the batch resident in HBM, then 3 more steps under the per-launch HIP-event timer for the
    dominant conv kernel's fraction of its MFMA peak (the brackets slow a 235-launch step by ~20 %: they are NOT in `value`)."""
    import torch
    from capsyolo_amd import loss_fns, models, ops, optim, synth
    out = {}
    # --- configs[1]: experiments/darknet_d GTSDB 416x416 detection, batch 16, fp32 (conv backbone kernels only)
    B, H, g = 16, 416, 13
    p = types.SimpleNamespace(n_classes=0, n_grid=g, n_boxes=2, dropout=0.0, darknet_input=H, device='cuda', model='darknet_d',
                              l_coord=5.0, l_noobj=0.5)
    torch.manual_seed(0)
    net = models.DarkNet(p).to(dev).train()
    opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
    x = torch.from_numpy(synth.images(B, H)).permute(0, 3, 1, 2).contiguous().to(dev)
    y = torch.from_numpy(synth.gtsdb_labels(B, g, 0)).to(dev)

    def step():
        loss = loss_fns.dark_loss(net(x), y, p)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    dt, final = _timed_steps(step, 3, steps)
    flops = _conv_layer_flops(net, x)
    ops.timer.reset()
    ops.timer.enabled = True
    _timed_steps(step, 0, 3)
    ops.timer.enabled = False
    out['darknet_d_416_b16_f32'] = {
        'config': {'workload': 'experiments/darknet_d GTSDB-shaped 416x416, n_grid 13, 2 boxes, batch 16, fp32 (BASELINE configs[1])'},
        'value': round(B / dt, 2), 'unit': 'images/s', 'ms_per_step': round(1e3 * dt, 3), 'steps': steps, 'warmup': 3,
        'dtype': 'f32', 'final_loss': round(final, 6), 'roofline': _dominant(ops, flops, 3, _KIND_PEAKS),
        'train_gflop_per_image': round(3 * sum(flops.values()) / B / 1e9, 2)}
    del net, opt, x, y
    # --- configs[4], one GPU's share: darkcapsule 608x608, 5 routing iterations, bf16 MFMA path, batch 32
    B, H, g = 32, 608, 19
    p = types.SimpleNamespace(n_classes=43, n_grid=g, n_boxes=2, dropout=0.0, recon=False, recon_coef=5e-4, darknet_input=H,
                              device='cuda', n_iter=5, model='darkcapsule', precision='bf16')
    torch.manual_seed(0)
    net = models.DarkCapsuleNet(p).to(dev).train()
    opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
    x = torch.from_numpy(synth.images(B, H)).permute(0, 3, 1, 2).contiguous().to(dev)
    y = torch.from_numpy(synth.gtsdb_labels(B, g, 43)).to(dev)

    def step5():
        loss = loss_fns.darkcapsule_loss(net(x), y, p)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    dt, final = _timed_steps(step5, 2, steps)
    flops = _conv_layer_flops(net, x)
    ops.timer.reset()
    ops.timer.enabled = True
    _timed_steps(step5, 0, 3)
    ops.timer.enabled = False
    out['darkcapsule_608_r5_b32_bf16'] = {
        'config': {'workload': 'darkcapsule GTSDB-shaped 608x608, n_grid 19, 5 routing iters, batch 32, bf16 MFMA path '
                               '(BASELINE configs[4], one GPU\'s share)'},
        'value': round(B / dt, 2), 'unit': 'images/s', 'ms_per_step': round(1e3 * dt, 3), 'steps': steps, 'warmup': 2,
        'dtype': 'bf16', 'final_loss': round(final, 6), 'roofline': _dominant(ops, flops, 3, _KIND_PEAKS),
        'train_gflop_per_image': round(3 * sum(flops.values()) / B / 1e9, 2)}
    del net, opt, x, y
    torch.cuda.empty_cache()
    # --- configs[0] on the GPU: experiments/capsule, GTSRB-shaped 32x32, 43 classes, 3 routing iterations, batch 32, reconstruction on
    B = 32
    p = types.SimpleNamespace(n_classes=43, dropout=0.0, recon=True, recon_coef=5e-4, device='cuda', model='capsule')
    torch.manual_seed(0)
    net = models.CapsuleNet(p).to(dev).train()
    opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
    x = torch.from_numpy(synth.images(B, 32)).permute(0, 3, 1, 2).contiguous().to(dev)
    y = torch.from_numpy(synth.gtsrb_labels(B, 43)).to(dev)

    def step0():
        scores, recon = net(x, y, True)
        loss = loss_fns.capsule_loss(scores, y, p, x, recon)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    dt, final = _timed_steps(step0, 5, 4 * steps)
    ops.timer.reset()
    ops.timer.enabled = True
    _timed_steps(step0, 0, 3)
    ops.timer.enabled = False
    kt = ops.timer.summary()
    # the same step captured once in a HIP graph and replayed (capsyolo_amd/graph_step.py; main.py --graph): the model is launch-bound
    from capsyolo_amd.graph_step import GraphedStep

    def fwd0(m, xb, yb):
        sc, rec = m(xb, yb, True)
        return sc, loss_fns.capsule_loss(sc, yb, p, xb, rec)
    graph_err = None
    try:
        gstep = GraphedStep(net, fwd0, opt, (x, y))
        dtg, finalg = _timed_steps(lambda: gstep(x, y)[1], 5, 4 * steps)
    except Exception as e:      # (an extra must never cost the line its headline)
        graph_err, dtg, finalg = '%s: %s' % (type(e).__name__, e), float('nan'), float('nan')
    out['capsule_32_b32_f32'] = {
        'config': {'workload': 'experiments/capsule GTSRB-shaped 32x32, 43 classes, 3 routing iters, batch 32, recon on, fp32 '
                               '(BASELINE configs[0] on the GPU)'},
        'value': round(B / dt, 1), 'unit': 'images/s', 'ms_per_step': round(1e3 * dt, 3), 'steps': 4 * steps, 'warmup': 5,
        'dtype': 'f32', 'final_loss': round(final, 6),
        'hip_graph_replay': {'error': graph_err} if graph_err else
                            {'value': round(B / dtg, 1), 'unit': 'images/s', 'ms_per_step': round(1e3 * dtg, 3), 'steps': 4 * steps,
                             'final_loss': round(finalg, 6), 'note': 'forward + loss + backward + Adam captured once (torch.cuda.graph over '
                             'the C-ABI launches), replayed per step; batch and optimizer scalars through device memory'},
        'kernel_ms_per_step': dict((k, round(ms * cnt / 3, 4)) for k, (cnt, ms) in sorted(kt.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:8])}
    del net, opt, x, y
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import capsyolo_amd  # noqa: F401
    from capsyolo_amd import dp, loss_fns, models, ops, optim, synth

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    rank, world, local_rank = dp.init_from_env(args.backend)
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)' % (args.gpus, world))
    local_rank = dp.local_device_index(local_rank, args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ops.SYNC_BN = bool(args.sync_bn) and world > 1
    g = args.input // 32
    B = args.batch
    p = types.SimpleNamespace(n_classes=43, n_grid=g, n_boxes=2, dropout=0.0, recon=False, recon_coef=5e-4,
                              darknet_input=args.input, device='cuda', n_iter=args.n_iter, model='darkcapsule',
                              precision=args.precision)
    torch.manual_seed(0)
    net = models.DarkCapsuleNet(p).to(dev).train()
    dp.broadcast_parameters(net)
    opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
    bucket = dp.GradBucket(net)
    # this rank's shard of the global synthetic batch, resident on the device before timing
    lo = rank * B
    x = torch.from_numpy(synth.images(B, args.input, first=lo)).permute(0, 3, 1, 2).contiguous().to(dev)
    y = torch.from_numpy(synth.gtsdb_labels(B, g, 43, first=lo)).to(dev)

    def step():
        out = net(x)
        loss = loss_fns.darkcapsule_loss(out, y, p)
        opt.zero_grad()
        loss.backward()
        bucket.allreduce_mean()
        opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    ops.timer.reset()
    ops.timer.enabled = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.timer.enabled = False
    final_loss = float(loss.item())
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kt = ops.timer.summary()

    # PCIe-inclusive rate (never `value`): the same step fed from host memory through the device-side input pipeline
    # (uint8 over PCIe, three pinned buffers, converted on the GPU) -- what main.py's loop does per batch
    from capsyolo_amd.input_pipeline import DeviceFeeder, quantize_if_exact
    n_h2d = min(args.steps, 10)
    x_host = quantize_if_exact(synth.images(B, args.input, first=lo))      # once per data set, as main.py does
    y_host = synth.gtsdb_labels(B, g, 43, first=lo)
    feeder = DeviceFeeder([(x_host, y_host)] * (n_h2d + 1), dev)
    it = iter(feeder)
    xb, yb = next(it)

    def step_on(xb, yb):
        out = net(xb)
        l = loss_fns.darkcapsule_loss(out, yb, p)
        opt.zero_grad()
        l.backward()
        bucket.allreduce_mean()
        opt.step()
    step_on(xb, yb)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for xb, yb in it:
        step_on(xb, yb)
    torch.cuda.synchronize()
    h2d_elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([h2d_elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        h2d_elapsed = float(t.item())

    # Every collective of this run is behind us: all ranks leave the process group TOGETHER, here, and what follows
    # (rank 0's roofline extras, the parity curve, the CPU baseline -- seconds of work) runs on a process that no other
    # rank waits for and that no longer owns a communicator.
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        ops.SYNC_BN = False

    if rank == 0:
        M = B * args.input * args.input
        conv2_flops = 2.0 * M * 256 * (9 * 128)          # algorithmic: 2*Cin*k^2*Cout*Ho*Wo per image
        # conv_2's three kernels (77.5 % of the model's FLOPs); the dominant one by time carries `roofline`
        pmc = pmc_traffic()
        head = git_head()
        cands = []
        for key, kname, executed, peak in ((k_, n_, e_, PEAK_FP32_MATRIX_TFLOPS) for k_, n_, e_ in (
                ('conv_wino_fwd/conv_2', 'wino_conv_kernel (conv_2 forward, fused Winograd F(2x2,3x3), fp32 MFMA)', 1 / 2.25),
                ('conv_wino4_fwd/conv_2', 'wino4_conv_kernel<1> (conv_2 forward, fused Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32, with the BatchNorm statistics)', 1 / 4.0),
                ('conv_wino4_dgrad/conv_2', 'wino4_conv_kernel<0> (conv_2 input gradient, fused Winograd F(4x4,3x3))', 1 / 4.0),
                ('conv_gemm_fwd/conv_2', 'conv_gemm_kernel<2,true> (conv_2 forward, implicit GEMM, fp32 MFMA)', 1.0),
                ('conv_wino_dgrad/conv_2', 'wino_conv_kernel (conv_2 input gradient, fused Winograd F(2x2,3x3))', 1 / 2.25),
                ('conv_gemm_dgrad/conv_2', 'conv_gemm_kernel<2,true> (conv_2 input gradient, implicit GEMM)', 1.0),
                ('conv_wino4_wgrad/conv_2', 'wino4_wgrad_kernel<0> + finish (conv_2 weight gradient, fused Winograd F(3x3,4x4) on v_mfma_f32_16x16x4_f32)', 1 / 4.0),
                ('conv_wino4_wgrad_bn/conv_2', 'wino4_wgrad_kernel<4> + finish (conv_2 weight gradient, fused Winograd F(3x3,4x4), with the block\'s BatchNorm backward pass 2 applied to the premasked gradient on the way in and dz written for the input-gradient kernel)', 1 / 4.0),
                ('conv_wino_wgrad/conv_2', 'wino_wgrad_kernel + finish (conv_2 weight gradient, fused Winograd F(3x3,2x2))', 1 / 2.25),
                ('conv_wino_wgrad_bn/conv_2', 'wino_wgrad_kernel<2> + finish (conv_2 weight gradient, fused Winograd F(3x3,2x2), with the block\'s BatchNorm backward pass 2 applied to the (premasked) gradient on the way in and dz written for the input-gradient kernel)', 1 / 2.25),
                ('conv_wgrad/conv_2', 'conv_wgrad_kernel<2,2,2,2,true> + wgrad_reduce_kernel (conv_2 weight gradient, fp32 MFMA)', 1.0),
                ('conv_bf16_fwd/conv_2', 'conv_bf16_kernel<256,256,2,4> (conv_2 forward, persistent implicit GEMM, LDS-DMA staged, v_mfma_f32_32x32x16_bf16)', 1.0),
                ('conv_bf16_dgrad/conv_2', 'conv_bf16_kernel<512,128,4,2> (conv_2 input gradient, bf16 MFMA)', 1.0),
                ('conv_bf16_wgrad/conv_2', 'wgrad_bf16_kernel<3,1,4,8> + reduce (conv_2 weight gradient, bf16 MFMA, transposing LDS reads, 8 waves)', 1.0))):
            if 'bf16' in key:
                peak = PEAK_BF16_MATRIX_TFLOPS
            if key in kt:
                n, ms = kt[key]
                direct = conv2_flops / (ms * 1e-3) / 1e12
                ach = direct * executed                               # the MFMA FLOPs this kernel issues per second
                tr = pmc.get(key)
                cands.append({'kernel': kname, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak,
                              'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                              'effective_vs_direct': round(direct / peak, 4),
                              'issued_flops_per_launch': conv2_flops * executed,
                              'direct_conv_flops_per_launch': conv2_flops,
                              'traffic': tr['bytes'] if tr else None,
                              'traffic_source': ({'file': 'profiles/' + tr['source'], 'measured_at_head': tr.get('head'),
                                                  'this_run_head': head, 'method': '2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes',
                                                  'algorithmic_bytes': int(M * (128 + (3 * 256 if key.endswith('_bn/conv_2') else 256)) * 4 + 1152 * 256 * 4)}
                                                 if tr else None),
                              'launch_ms': round(ms, 4), 'launches_timed': n,
                              'note': 'achieved = issued MFMA FLOPs (direct-convolution 2*M*N*K = %.3f TFLOP with M=%d, N=256, '
                                      'K=1152, / 2.25 for the F(2x2,3x3) / F(3x3,2x2) Winograd kernels, / 4 for F(4x4,3x3) / F(3x3,4x4)) / mean launch time; '
                                      'effective_vs_direct = direct-convolution FLOPs / time / peak' % (conv2_flops / 1e12, M)})
        cands.sort(key=lambda d: -d['launch_ms'])
        R = g * g * B
        rt_bytes = 4.0 * (R * 512 * 8 + 512 * 1 * 8 * 5 + R * 1 * 5)
        nr, msr = kt.get('routing_fwd', (0, float('nan')))
        rt_gbps = rt_bytes / (msr * 1e-3) / 1e9
        line = {
            'metric': 'train images/sec darkcapsule GTSDB 416x416 @1/2/4/8 GPU; loss-curve parity',
            'value': round(world * B * args.steps / elapsed, 3), 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.precision == 'fp32' else 'bf16', 'data': 'synthetic',
            'value_definition': 'whole-job images/s with each rank\'s batch resident in HBM before the timed region (bench contract); '
                                'pcie_inclusive = the same step with the batch fed from host memory every step',
            'config': {'workload': 'experiments/darkcapsule GTSDB-shaped %dx%d, n_grid %d, %d routing iters, batch %d per GPU, '
                                   'recon off, %s (%s)' % (args.input, args.input, g, args.n_iter, B,
                                                           'fp32' if args.precision == 'fp32' else 'bf16 MFMA path (bf16 activations, fp32 accumulation / statistics / master weights; first block and routing head fp32)',
                                                           'BASELINE configs[2]' if (args.input, args.n_iter, B, args.precision) == (416, 3, 32, 'fp32')
                                                           else 'BASELINE configs[4], one GPU\'s share' if (args.input, args.n_iter, B, args.precision) == (608, 5, 32, 'bf16')
                                                           else 'not a BASELINE configuration'),
                       'global_batch': world * B, 'parallelism': 'dp%d%s' % (world, '+syncbn' if (args.sync_bn and world > 1) else ''), 'final_loss': round(final_loss, 6)},
            'roofline': cands[0] if cands else None,
            'roofline_other_conv2': cands[1:],
            'roofline_routing': {'kernel': 'caps1_fwd_kernel<5,true> (fused routing, C=1, cell gather folded into the load; '
                                           'launch_ms includes the HIP-event bracket, rocprof: profiles/)',
                                 'bound': 'hbm', 'achieved': round(rt_gbps, 1), 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                                 'frac': round(rt_gbps / PEAK_HBM_GBPS, 4),
                                 'traffic': pmc['routing_fwd']['bytes'] if 'routing_fwd' in pmc else None,
                                 'traffic_source': ('profiles/' + pmc['routing_fwd']['source']) if 'routing_fwd' in pmc else None,
                                 'algorithmic_bytes': int(rt_bytes),
                                 'launch_ms': round(msr, 5), 'launches_timed': nr,
                                 'rocprof': (lambda rp: dict(rp, frac=round(rt_bytes / (rp['avg_us'] * 1e-6) / 1e9 / PEAK_HBM_GBPS, 4))
                                             if rp else None)(rocprof_launch_us('caps1_fwd_kernel'))},
            'pcie_inclusive': {'value': round(world * B * n_h2d / h2d_elapsed, 3), 'unit': 'images/s (all ranks, slowest rank\'s time)',
                               'ms_per_step': round(1e3 * h2d_elapsed / n_h2d, 3), 'steps': n_h2d,
                               'h2d_bytes_per_step': int(x_host.size + y_host.nbytes),
                               'host_ms_per_batch': dict((k, round(v / max(1, feeder.host_ms['batches']), 3))
                                                         for k, v in feeder.host_ms.items() if k != 'batches'),
                               'note': 'batch fed from host memory every step: uint8 over PCIe, three pinned buffers, '
                                       'centring + NHWC->NCHW on the device (capsyolo_amd/input_pipeline.py)'},
            'kernel_ms': dict((k, round(v[1], 4)) for k, v in sorted(kt.items())),
        }
        def extra(name, fn):        # an extra that fails is reported as such: it must never cost the line its headline
            try:
                line[name] = fn()
            except Exception as e:
                line[name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        if not args.no_extras:
            extra('roofline_routing_c43', lambda: routing_c43(dev, B))
            extra('loss_curve_parity', lambda: loss_curve_parity(dev))
            if world == 1:
                torch.cuda.empty_cache()
                extra('secondary', lambda: secondary(dev))
        if world == 1 and not args.no_cpu_baseline:
            extra('cpu_baseline', lambda: cpu_baseline(args, g))
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
