"""Data parallel on real kernels: two ranks share cuda:0 (gloo: RCCL wants one GPU per rank) and train DarkCapsuleNet
on the two halves of a global batch.  With SYNC_BN the averaged gradients, the loss and the BatchNorm running statistics
must equal one process on the whole batch (losses divide by the local batch, shards are equal)."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import make_params, synth_gtsdb_labels, synth_images

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    import capsyolo_amd  # noqa: F401
    from capsyolo_amd import dp, loss_fns, models, ops
    dp.init_from_env('gloo')
    torch.cuda.set_device(0)
    p = make_params(model='darkcapsule', n_grid=2, darknet_input=64, recon=False, device='cuda')
    torch.manual_seed(100 + rank)
    net = models.DarkCapsuleNet(p).cuda().train()
    dp.broadcast_parameters(net)
    ref = copy.deepcopy(net)
    B = 8
    x = torch.from_numpy(synth_images(B, 64, seed=31)).cuda()
    y = torch.from_numpy(synth_gtsdb_labels(B, 2, 43, seed=32)).cuda()
    lo, hi = dp.shard_range(B, rank, world)
    ops.SYNC_BN = True
    loss = loss_fns.darkcapsule_loss(net(x[lo:hi]), y[lo:hi], p)
    loss.backward()
    n = dp.GradBucket(net).allreduce_mean()
    assert n > 0
    lsum = loss.detach().double().clone()
    dist.all_reduce(lsum)
    ops.SYNC_BN = False                                 # one process, whole batch, same initial weights
    loss1 = loss_fns.darkcapsule_loss(ref(x), y, p)
    loss1.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(lsum.item() / world, loss1.item(), rtol=2e-5)
    for (name, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
        if b.grad is None:
            assert a.grad is None, name
            continue
        ga, gb = a.grad.double(), b.grad.double()
        err = (ga - gb).norm().item() / max(gb.norm().item(), 1e-12)
        assert err < 2e-3 or gb.norm().item() < 1e-6, (name, err, gb.norm().item())
    for (name, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
        np.testing.assert_allclose(a.double().cpu().numpy(), b.double().cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=name)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmpdir, 'ok%d' % rank), 'w').write('ok')


def test_two_ranks_with_sync_bn_equal_one_process_on_the_global_batch(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()


def _nccl_worker(rank, world, port, tmpdir):
    """backend 'nccl' (= RCCL) on the one GPU there is: world size 1, but the REAL device collective, the bucket's
    pack -> all-reduce -> re-point path (forced) and a bench.py-shaped step."""
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    assert dist.get_backend() == 'nccl'
    t = torch.arange(1000, dtype=torch.float32, device='cuda')
    dist.all_reduce(t)                                   # RCCL all-reduce on the device
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))
    import capsyolo_amd  # noqa: F401
    from capsyolo_amd import dp, loss_fns, models, optim
    p = make_params(model='darkcapsule', n_grid=2, darknet_input=64, recon=False, device='cuda')
    torch.manual_seed(0)
    net = models.DarkCapsuleNet(p).cuda().train()
    dp.broadcast_parameters(net)                          # world 1: no-op
    ref = copy.deepcopy(net)
    x = torch.from_numpy(synth_images(4, 64, seed=31)).cuda()
    y = torch.from_numpy(synth_gtsdb_labels(4, 2, 43, seed=32)).cuda()
    opt, opt_ref = (optim.Adam([q for q in m.parameters() if q.requires_grad], lr=1e-3) for m in (net, ref))
    bucket = dp.GradBucket(net)
    for _ in range(2):
        loss = loss_fns.darkcapsule_loss(net(x), y, p)
        opt.zero_grad()
        loss.backward()
        before = dict((n, q.grad.clone()) for n, q in net.named_parameters() if q.grad is not None)
        n = bucket.allreduce_mean(force=True)
        assert n == sum(g.numel() for g in before.values())
        for name, q in net.named_parameters():
            if q.grad is not None:
                assert torch.equal(q.grad, before[name]), name          # mean over one rank: unchanged ...
                assert q.grad.data_ptr() >= bucket.flat.data_ptr() and \
                    q.grad.data_ptr() < bucket.flat.data_ptr() + bucket.flat.numel() * 4, name   # ... and now a view of the bucket
        opt.step()
        loss_r = loss_fns.darkcapsule_loss(ref(x), y, p)
        opt_ref.zero_grad()
        loss_r.backward()
        opt_ref.step()
    torch.cuda.synchronize()
    for (name, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(a, b), name                                    # Adam from the bucket == Adam from the gradients
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmpdir, 'nccl_ok'), 'w').write('ok')


def test_nccl_backend_world1_bucket_and_step(tmp_path):
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_nccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / 'nccl_ok').exists()


def test_bench_two_ranks_over_gloo_on_one_gpu(tmp_path):
    """bench.py's N > 1 code path, executed: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` exactly
    as the driver launches it, except that the two ranks share the one GPU and the collectives go over gloo
    (--backend gloo: RCCL wants one GPU per rank).  Covers init_from_env, broadcast_parameters, the barrier-bracketed
    timed region, the MAX-reduce of the elapsed time, the bucket all-reduce inside the step, the all-rank pcie_inclusive
    loop, every rank leaving the process group together BEFORE rank 0 computes its extras, and the one JSON line."""
    import json
    import subprocess
    import sys
    from helpers import REPO
    port = 25500 + (os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONUNBUFFERED='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--batch', '2', '--input', '64',
           '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--backend', 'gloo']
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4 and d['scaling'] == 'weak' and d['steps'] == 2
    assert np.isfinite(d['value']) and d['value'] > 0
    assert np.isfinite(d['pcie_inclusive']['value']) and d['pcie_inclusive']['value'] > 0
    assert np.isfinite(d['config']['final_loss'])
    assert 'cpu_baseline' not in d                                # N = 1 only
    # rank 0's extras ran after the group was gone (they would hang or raise inside a live, half-destroyed group)
    assert 'loss_curve_parity' in d and 'roofline_routing_c43' in d


def test_bench_two_ranks_sync_bn_over_gloo(tmp_path):
    """The same launch with --sync-bn: the ten small statistics all-reduces per step run inside the timed region."""
    import json
    import subprocess
    import sys
    from helpers import REPO
    port = 23500 + (os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONUNBUFFERED='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--batch', '2', '--input', '64',
           '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extras', '--sync-bn', '--backend', 'gloo']
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['config']['parallelism'] == 'dp2+syncbn' and np.isfinite(d['config']['final_loss'])
