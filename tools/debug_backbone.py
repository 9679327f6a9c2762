"""Layer-by-layer activation/gradient comparison of the DarkCapsuleNet backbone (HIP vs fp64 oracle)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import closed_form_state, make_params, synth_gtsdb_labels, synth_images
import capsyolo_amd
from capsyolo_amd import loss_fns, models, ops
from oracle import loss_fns as OL, models as OM
T = torch.from_numpy
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())
p = make_params(model='darkcapsule', n_grid=2, darknet_input=64, recon=False, device='cuda')
x = T(synth_images(4, 64, seed=22)); y = T(synth_gtsdb_labels(4, 2, 43, seed=23))
o = OM.DarkCapsuleNet(p); o.load_state_dict(closed_form_state(o)); o.double().train()
acts = {}
h = x.double()
for name, m in o.conv.named_children():
    h = m(h)
    if name.startswith('relu') or name.startswith('conv') or name.startswith('bn'):
        h.retain_grad(); acts[name] = h
u = OM.cell_gather(h.contiguous(), 2)
v = o.traffic_sign_capsules(u)[:, 0, 0, 0, :]
out = v.view(2, 2, 4, 5).permute(2, 0, 1, 3)
OL.darkcapsule_loss(out, y, p).backward()

n = models.DarkCapsuleNet(p); n.load_state_dict(closed_form_state(n)); n.cuda().train()
hacts = {}
hh = x.cuda(); nchw = True
mods = dict(n.conv.named_children())
for k in range(1, 6):
    c, b, r = mods['conv_%d' % k], mods['bn_%d' % k], mods['relu_%d' % k]
    cfg = ops.ConvBlockCfg(c.k, c.stride, c.padding, nchw, b, r.slope)
    hh = ops.conv_block(hh, c.weight, c.bias, b.weight, b.bias, cfg); nchw = False
    hh.retain_grad(); hacts['relu_%d' % k] = hh
vv = n.traffic_sign_capsules(hh, gather_g=2, gather_B=4)
loss_fns.darkcapsule_loss(vv.view(4, 2, 2, 5), y.cuda(), p).backward()
for k in range(5, 0, -1):
    a, ah = acts['relu_%d' % k], hacts['relu_%d' % k]
    print('relu_%d act err %.2e  grad err %.2e   (grad absmax %.3e, grad mean %.3e)' % (
        k, rel(ah.permute(0, 3, 1, 2), a), rel(ah.grad.permute(0, 3, 1, 2), a.grad), a.grad.abs().max(), a.grad.mean()))
    d = (ah.grad.permute(0, 3, 1, 2).double().cpu() - a.grad)
    print('     grad diff: mean %.3e  absmax %.3e; per-channel mean of diff (first 4): %s' % (d.mean(), d.abs().max(), d.mean(dim=(0, 2, 3))[:4].numpy()))
    dz_ref = acts['conv_%d' % k].grad
    print('     conv_%d out grad (dz) ref absmax %.3e mean %.3e' % (k, dz_ref.abs().max(), dz_ref.mean()))
print('--- LeakyReLU kink check: elements whose sign differs between HIP and the fp64 oracle')
for k in range(5, 0, -1):
    a, ah = acts['relu_%d' % k], hacts['relu_%d' % k].permute(0, 3, 1, 2).detach().double().cpu()
    flips = ((a > 0) != (ah > 0))
    print('relu_%d: %d sign flips of %d; |act| at flips: %s' % (k, int(flips.sum()), a.numel(), a[flips].abs()[:5].tolist()))
    da, dah = a.grad, hacts['relu_%d' % k].grad.permute(0, 3, 1, 2).double().cpu()
    print('       grad rel L2 err %.2e' % float((dah - da).norm() / da.norm()))
