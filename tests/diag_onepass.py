"""Diagnostic (not collected by pytest): the first block's three modes on the loss-curve recipes -- (a) two-pass recompute,
(b) patch-moment statistics + two-pass backward, (c) patch-moment statistics + one-pass backward -- against the fp64 oracle:
forward output, bn_1 running statistics, and the gradients of the first layers.   python tests/diag_onepass.py"""
import copy
import os
import sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from helpers import closed_form_state, load_golden, make_params, synth_gtsdb_labels, synth_images
import capsyolo_amd
from capsyolo_amd import loss_fns, models, ops
from oracle import loss_fns as OL
from oracle import models as OM

T = torch.from_numpy


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-300))


for tag, golden in (('dc64', 'curves'), ('dc256', 'curves256')):
    g = load_golden(golden)
    H, gg, B, seed = (int(v) for v in g[tag + '_cfg'])
    p = make_params(model='darkcapsule', n_grid=gg, darknet_input=H, recon=False, device='cuda')
    po = make_params(model='darkcapsule', n_grid=gg, darknet_input=H, recon=False)
    x, y = T(synth_images(B, H, seed=seed)), T(synth_gtsdb_labels(B, gg, 43, seed=seed + 1))
    o64 = OM.DarkCapsuleNet(po)
    o64.load_state_dict(closed_form_state(o64))
    o64 = o64.double().train()
    out64 = o64(x.double())
    OL.darkcapsule_loss(out64, y, po).backward()
    g64 = dict((n, q.grad) for n, q in o64.named_parameters())
    b64 = dict(o64.named_buffers())
    names = ['conv.conv_1.weight', 'conv.bn_1.weight', 'conv.bn_1.bias', 'conv.conv_2.weight', 'conv.bn_2.weight', 'conv.conv_5.weight',
             'traffic_sign_capsules.route_weights']
    for mode, sw in (('two-pass', dict(CONV1_MOMENTS_MIN_PIXELS=1 << 62)),
                     ('moments + two-pass bwd', dict(CONV1_MOMENTS_MIN_PIXELS=0, USE_CONV1_ONEPASS=False)),
                     ('moments + one-pass bwd', dict(CONV1_MOMENTS_MIN_PIXELS=0, USE_CONV1_ONEPASS=True))):
        old = dict((k, getattr(ops, k)) for k in sw)
        for k, v in sw.items():
            setattr(ops, k, v)
        try:
            net = models.DarkCapsuleNet(p)
            net.load_state_dict(closed_form_state(net))
            net.cuda().train()
            ops.timer.reset(); ops.timer.enabled = True
            out = net(x.cuda())
            loss_fns.darkcapsule_loss(out, y.cuda(), p).backward()
            torch.cuda.synchronize()
            keys = sorted(k for k in ops.timer.summary() if k.startswith('conv1'))
            ops.timer.enabled = False
        finally:
            for k, v in old.items():
                setattr(ops, k, v)
        gr = dict((n, q.grad) for n, q in net.named_parameters())
        bf = dict(net.named_buffers())
        print('%s %-24s out %.2e  bn_1.running_mean %.2e running_var %.2e | grads: %s | %s'
              % (tag, mode, rel(out, out64), rel(bf['conv.bn_1.running_mean'], b64['conv.bn_1.running_mean']),
                 rel(bf['conv.bn_1.running_var'], b64['conv.bn_1.running_var']),
                 ' '.join('%s %.1e' % (n.split('.')[-2][-6:], rel(gr[n], g64[n])) for n in names), ','.join(k.split('/')[0] for k in keys)), flush=True)
