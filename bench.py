#!/usr/bin/env python
"""Headline benchmark: train images/sec of darkcapsule (GTSDB-shaped 416x416, n_grid 13, 3 routing
iterations, batch 32 per GPU, fp32, recon off) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + darkcapsule_loss + backward + (N>1: one RCCL all-reduce of the flat gradient bucket)
+ fused Adam, on a synthetic batch that is resident in HBM before the timed region.  Rank 0 prints ONE JSON
line.  Extra objects on that line:
  roofline      dominant kernel (conv_2 forward implicit GEMM, fp32 MFMA): algorithmic FLOPs per launch divided
                by the launch's mean duration, measured with HIP events on the launch stream inside the timed steps
  roofline_routing   the fused routing kernel (HBM-bound): algorithmic bytes / duration
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
                best[k] = dict(v, source=os.path.basename(f))
        except Exception:
            pass
    return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (BASELINE configs[2]: 32)')
    ap.add_argument('--input', type=int, default=416, help='image side; n_grid = input/32')
    ap.add_argument('--n_iter', type=int, default=3)
    ap.add_argument('--sync-bn', action='store_true', help='N>1: BatchNorm statistics over the global batch (2 small all-reduces per BN layer)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=2, help='sample batch for the CPU baseline')
    ap.add_argument('--cpu-steps', type=int, default=2)
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
    B = args.cpu_batch
    x = torch.from_numpy(synth.images(B, args.input)).permute(0, 3, 1, 2).contiguous()
    y = torch.from_numpy(synth.gtsdb_labels(B, g, 43))

    def step():
        out = net(x)
        loss = OL.darkcapsule_loss(out, y, p)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.item()
    step()                                   # warm-up
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        step()
    dt = time.perf_counter() - t0
    return {'value': round(B * args.cpu_steps / dt, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': 'oracle DarkCapsuleNet %dx%d train step, batch %d (of the %d-image batch), %d timed steps after 1 warm-up, '
                      'torch %s CPU, %d threads' % (args.input, args.input, B, args.batch, args.cpu_steps,
                                                    torch.__version__, cores)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import capsyolo_amd  # noqa: F401
    from capsyolo_amd import dp, loss_fns, models, ops, optim, synth

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    rank, world, local_rank = dp.init_from_env('nccl')
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ops.SYNC_BN = bool(args.sync_bn) and world > 1
    g = args.input // 32
    B = args.batch
    p = types.SimpleNamespace(n_classes=43, n_grid=g, n_boxes=2, dropout=0.0, recon=False, recon_coef=5e-4,
                              darknet_input=args.input, device='cuda', n_iter=args.n_iter, model='darkcapsule')
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
    # (uint8 over PCIe, double-buffered, converted on the GPU) -- what main.py's loop does per batch
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

    if rank == 0:
        M = B * args.input * args.input
        conv2_flops = 2.0 * M * 256 * (9 * 128)          # algorithmic: 2*Cin*k^2*Cout*Ho*Wo per image
        # conv_2's three kernels (77.5 % of the model's FLOPs); the dominant one by time carries `roofline`
        pmc = pmc_traffic()
        cands = []
        for key, kname, executed in (
                ('conv_wino_fwd/conv_2', 'wino_conv_kernel (conv_2 forward, fused Winograd F(2x2,3x3), fp32 MFMA)', 1 / 2.25),
                ('conv_gemm_fwd/conv_2', 'conv_gemm_kernel<2,true> (conv_2 forward, implicit GEMM, fp32 MFMA)', 1.0),
                ('conv_wino_dgrad/conv_2', 'wino_conv_kernel (conv_2 input gradient, fused Winograd F(2x2,3x3))', 1 / 2.25),
                ('conv_gemm_dgrad/conv_2', 'conv_gemm_kernel<2,true> (conv_2 input gradient, implicit GEMM)', 1.0),
                ('conv_wino_wgrad/conv_2', 'wino_wgrad_kernel + finish (conv_2 weight gradient, fused Winograd F(3x3,2x2))', 1 / 2.25),
                ('conv_wgrad/conv_2', 'conv_wgrad_kernel<2,2,2,2,true> + wgrad_reduce_kernel (conv_2 weight gradient, fp32 MFMA)', 1.0)):
            if key in kt:
                n, ms = kt[key]
                ach = conv2_flops / (ms * 1e-3) / 1e12
                cands.append({'kernel': kname, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_FP32_MATRIX_TFLOPS,
                              'unit': 'TFLOP/s', 'frac': round(ach / PEAK_FP32_MATRIX_TFLOPS, 4),
                              'traffic': pmc[key]['bytes'] if key in pmc else None,
                              'traffic_source': (pmc[key]['source'] + ': 2 x FETCH_SIZE + WRITE_SIZE bytes per launch; '
                                                 'algorithmic input+output+weights = %.2f GB' % ((M * (128 + 256) * 4 + 1152 * 256 * 4) / 1e9))
                              if key in pmc else None,
                              'launch_ms': round(ms, 4), 'launches_timed': n,
                              'executed_frac': round(ach * executed / PEAK_FP32_MATRIX_TFLOPS, 4),
                              'note': 'achieved = direct-convolution FLOPs (M=%d, N=256, K=1152: %.3f TFLOP) / launch time; '
                                      'executed_frac = MFMA FLOPs actually issued / peak (Winograd issues 1/2.25 of them, '
                                      'so frac can exceed 1)' % (M, conv2_flops / 1e12)})
        cands.sort(key=lambda d: -d['launch_ms'])
        R = g * g * B
        rt_bytes = 4.0 * (R * 512 * 8 + 512 * 1 * 8 * 5 + R * 1 * 5)
        nr, msr = kt.get('routing_fwd', (0, float('nan')))
        rt_gbps = rt_bytes / (msr * 1e-3) / 1e9
        line = {
            'metric': 'train images/sec darkcapsule GTSDB 416x416 @1/2/4/8 GPU; loss-curve parity',
            'value': round(world * B * args.steps / elapsed, 3), 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'experiments/darkcapsule GTSDB-shaped %dx%d, n_grid %d, %d routing iters, batch %d per GPU, '
                                   'recon off, fp32 (%s)' % (args.input, args.input, g, args.n_iter, B,
                                                             'BASELINE configs[2]' if (args.input, args.n_iter, B) == (416, 3, 32)
                                                             else 'not a BASELINE configuration'),
                       'global_batch': world * B, 'parallelism': 'dp%d%s' % (world, '+syncbn' if (args.sync_bn and world > 1) else ''), 'final_loss': round(final_loss, 6)},
            'roofline': cands[0] if cands else None,
            'roofline_other_conv2': cands[1:],
            'roofline_routing': {'kernel': 'caps1_fwd_kernel<5,true> (fused routing, C=1, cell gather folded into the load; '
                                           'launch_ms includes the HIP-event bracket, rocprof: profiles/)',
                                 'bound': 'hbm', 'achieved': round(rt_gbps, 1), 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                                 'frac': round(rt_gbps / PEAK_HBM_GBPS, 4),
                                 'traffic': pmc['routing_fwd']['bytes'] if 'routing_fwd' in pmc else None,
                                 'algorithmic_bytes': int(rt_bytes),
                                 'launch_ms': round(msr, 5), 'launches_timed': nr},
            'pcie_inclusive': {'value': round(world * B * n_h2d / h2d_elapsed, 3), 'unit': 'images/s (this rank x world)',
                               'ms_per_step': round(1e3 * h2d_elapsed / n_h2d, 3), 'steps': n_h2d,
                               'h2d_bytes_per_step': int(x_host.size + y_host.nbytes),
                               'note': 'batch fed from host memory every step: uint8 over PCIe, pinned double buffer, '
                                       'centring + NHWC->NCHW on the device (capsyolo_amd/input_pipeline.py)'},
            'kernel_ms': dict((k, round(v[1], 4)) for k, v in sorted(kt.items())),
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args, g)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
