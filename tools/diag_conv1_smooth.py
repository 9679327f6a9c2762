"""First block (conv 3->Cout 3x3 -> BatchNorm -> LeakyReLU) on spatially correlated, offset images with zero-sum edge filters
(ADVICE round 2): errors of the batch variance, dgamma, dbeta and dW of the two-pass (recompute) and the one-pass (patch-moment)
paths against torch fp64.  The iid-randn images of the tests have a near-diagonal moment matrix; natural images do not."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
import capsyolo_amd
from capsyolo_amd import models, ops


def images(kind, B, H, W, seed):
    rng = np.random.default_rng(seed)
    if kind == 'iid':
        u8 = rng.integers(0, 256, (B, 3, H, W))
    else:                                   # smooth: blurred noise, std ~0.2, around +0.7 (bright image), quantised to k/128 like the data sets
        z = torch.from_numpy(rng.standard_normal((B, 3, H, W)))
        k = torch.ones(1, 1, 9, 9, dtype=torch.float64) / 81
        for _ in range(2):
            z = F.conv2d(z.reshape(B * 3, 1, H, W), k, padding=4).reshape(B, 3, H, W)
        z = z / z.std() * 0.2 + 0.7
        u8 = np.clip(np.round(z.numpy() * 128 + 128), 0, 255)
    return torch.from_numpy(((u8.astype(np.float32)) - 128.0) / 128.0)


def run(kind, cout, B, H, W, onepass, edge):
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(3, cout, 3, 1, 1).double()
    if edge:                                 # zero-sum (high-pass) filters: sum over the 27 taps = 0
        w = conv.weight.data
        conv.weight.data = w - w.mean(dim=(1, 2, 3), keepdim=True)
    conv.weight.data = conv.weight.data.float().double()
    conv.bias.data = conv.bias.data.float().double()
    bn = torch.nn.BatchNorm2d(cout).double()
    bn.weight.data = (1 + 0.2 * torch.randn(cout)).double()
    bn.bias.data = (0.1 * torch.randn(cout)).double()
    ref = torch.nn.Sequential(conv, bn, torch.nn.LeakyReLU(0.1)).train()
    x = images(kind, B, H, W, 141)
    g = torch.randn(B, cout, H, W, generator=torch.Generator().manual_seed(142))
    yr = ref(x.double())
    yr.backward(g.double())
    z = conv(x.double())
    var_ref = z.var(dim=(0, 2, 3), unbiased=True)
    ops.CONV1_MOMENTS_MIN_PIXELS = 0 if onepass else (1 << 62)
    seq = models.FusedBackbone()
    seq.add_module('conv_1', models.HipConv2d(3, cout, 3, 1, 1))
    seq.add_module('bn_1', models.HipBatchNorm2d(cout))
    seq.add_module('relu_1', models.HipLeakyReLU(0.1))
    seq.conv_1.load_state_dict({k: v.float() for k, v in conv.state_dict().items()})
    sd = torch.nn.BatchNorm2d(cout).state_dict()
    sd['weight'], sd['bias'] = bn.weight.data.float(), bn.bias.data.float()
    seq.bn_1.load_state_dict(sd)
    seq.cuda().train()
    seq.bn_1.momentum = 1.0                  # running_var = the batch's unbiased variance
    yh = seq(x.cuda(), nchw_in=True)
    yh.backward(g.permute(0, 2, 3, 1).contiguous().cuda())
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).abs().max() / b.abs().max())

    def relel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float(((a - b).abs() / b.abs().clamp(min=1e-300)).max())
    return dict(out=rel(yh.permute(0, 3, 1, 2), yr), var_elem=relel(seq.bn_1.running_var, var_ref),
                dW=rel(seq.conv_1.weight.grad, conv.weight.grad), dgamma=rel(seq.bn_1.weight.grad, bn.weight.grad),
                dbeta=rel(seq.bn_1.bias.grad, bn.bias.grad))


if __name__ == '__main__':
    for kind, edge in (('iid', False), ('smooth', False), ('smooth', True)):
        for shape in ((128, 4, 64, 64), (128, 8, 256, 256)):
            for onepass in (False, True):
                r = run(kind, *shape, onepass, edge)
                print('%-6s edge=%d Cout=%d B=%d %dx%d %-8s ' % ((kind, edge) + shape + ('onepass' if onepass else 'twopass',))
                      + ' '.join('%s=%.2e' % kv for kv in r.items()), flush=True)
