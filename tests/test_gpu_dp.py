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
