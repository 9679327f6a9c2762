"""Diagnostic (not collected by pytest): CapsuleNet with the closed-form weights of the golden fixtures -- where does the gradient of
decoder.0.weight (Linear(16, 256) -> ReLU) differ from the fp64 oracle?  Prints the decoder's pre-activations closest to the ReLU
kink and the per-row gradient errors.   python tests/diag_capsule_decoder.py"""
import copy
import os
import sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from helpers import closed_form_state, make_params, synth_images
import capsyolo_amd
from capsyolo_amd import loss_fns, models
from oracle import loss_fns as OL
from oracle import models as OM

T = torch.from_numpy
x = T(synth_images(4, 32, seed=21))
y = T(np.array([3, 42, 0, 17], dtype=np.int64))
p = make_params(model='capsule', recon=True, device='cuda')
po = make_params(model='capsule', recon=True)
o32 = OM.CapsuleNet(po)
o32.load_state_dict(closed_form_state(o32))
o64 = copy.deepcopy(o32).double()
pre = {}
for tag, net, xx in (('o64', o64, x.double()), ('o32', o32, x)):
    net.train()
    h = net.decoder[0].register_forward_hook(lambda m, i, o, tag=tag: pre.__setitem__(tag, (i[0].detach().clone(), o.detach().clone())))
    s, r = net(xx, y, True)
    OL.capsule_loss(s, y, po, xx, r).backward()
    h.remove()
hip = models.CapsuleNet(p)
hip.load_state_dict(closed_form_state(hip))
hip.cuda().train()
s, r = hip(x.cuda(), y.cuda(), True)
loss_fns.capsule_loss(s, y.cuda(), p, x.cuda(), r).backward()
t64, h64 = pre['o64']
t32, h32 = pre['o32']
print('decoder input (picked capsule) fp32-oracle vs fp64 rel err %.2e' % float((t32.double() - t64).norm() / t64.norm()))
a = h64.abs().flatten()
k = torch.argsort(a)[:8]
print('pre-activations closest to 0 (fp64): ', [(int(i) // 256, int(i) % 256, float(h64.flatten()[i]), float(h32.flatten()[i])) for i in k])
g64 = o64.decoder[0].weight.grad
g32 = o32.decoder[0].weight.grad.double()
gh = hip.decoder[0].weight.grad.double().cpu()
rows = ((gh - g64).norm(dim=1) / g64.norm()).numpy()
rows32 = ((g32 - g64).norm(dim=1) / g64.norm()).numpy()
top = np.argsort(-rows)[:6]
print('rows of decoder.0.weight.grad with the largest error (row, HIP err / |g|, oracle-fp32 err / |g|):',
      [(int(i), float(rows[i]), float(rows32[i])) for i in top])
print('total rel L2: HIP %.3e, oracle fp32 %.3e' % (float((gh - g64).norm() / g64.norm()), float((g32 - g64).norm() / g64.norm())))
gb = hip.decoder[0].bias.grad.double().cpu()
print('bias grad rel L2: HIP %.3e' % float((gb - o64.decoder[0].bias.grad).norm() / o64.decoder[0].bias.grad.norm()))
for n, q in hip.named_parameters():
    if q.grad is not None and dict(o64.named_parameters())[n].grad is not None:
        g6 = dict(o64.named_parameters())[n].grad
        print('  %-40s HIP %.2e  oracle-fp32 %.2e' % (n, float((q.grad.double().cpu() - g6).norm() / g6.norm()),
                                                       float((dict(o32.named_parameters())[n].grad.double() - g6).norm() / g6.norm())))
