"""Per-parameter gradient error of the HIP models against the fp64 oracle, next to the fp32 oracle's own
error against fp64 (the reference path's rounding noise).  Run on the GPU box: python tools/compare_models.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import closed_form_state, make_params, synth_gtsdb_labels, synth_images  # noqa: E402
import capsyolo_amd  # noqa: E402,F401
from capsyolo_amd import loss_fns, models  # noqa: E402
from oracle import loss_fns as OL  # noqa: E402
from oracle import models as OM  # noqa: E402

T = torch.from_numpy


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def run(name, make_hip, make_ora, fwd_hip, fwd_ora, which=None):
    o64 = make_ora()
    o64.load_state_dict(closed_form_state(o64))
    o64.double().train()
    out64, l64 = fwd_ora(o64, torch.float64)
    l64.backward()
    o32 = make_ora()
    o32.load_state_dict(closed_form_state(o32))
    o32.train()
    out32, l32 = fwd_ora(o32, torch.float32)
    l32.backward()
    h = make_hip()
    h.load_state_dict(closed_form_state(h))
    h.cuda().train()
    outh, lh = fwd_hip(h)
    lh.backward()
    print('== %s: loss hip %.8f  o32 %.8f  o64 %.8f | out err hip %.2e o32 %.2e' % (
        name, lh.item(), l32.item(), l64.item(), relerr(outh, out64), relerr(out32, out64)))
    g64 = dict((n, p.grad) for n, p in o64.named_parameters())
    g32 = dict((n, p.grad) for n, p in o32.named_parameters())
    for n, p in h.named_parameters():
        if p.grad is None or g64[n] is None:
            continue
        eh, e32 = relerr(p.grad, g64[n]), relerr(g32[n], g64[n])
        flag = '  <<<' if eh > 10 * e32 + 1e-5 else ''
        print('   %-44s hip %.2e   o32 %.2e%s' % (n, eh, e32, flag))


if __name__ == '__main__':
    sel = sys.argv[1:] or ['capsule', 'darkcapsule', 'darknet']
    if 'capsule' in sel:
        x = T(synth_images(4, 32, seed=21))
        y = T(np.array([3, 42, 0, 17], dtype=np.int64))
        p = make_params(model='capsule', recon=True, device='cuda')

        def fh(net):
            s, r = net(x.cuda(), y.cuda(), True)
            return s, loss_fns.capsule_loss(s, y.cuda(), p, x.cuda(), r)

        def fo(net, dt):
            s, r = net(x.to(dt), y, True)
            return s, OL.capsule_loss(s, y, p, x.to(dt), r)
        run('capsule(recon)', lambda: models.CapsuleNet(p), lambda: OM.CapsuleNet(p), fh, fo)
    if 'darkcapsule' in sel:
        p = make_params(model='darkcapsule', n_grid=2, darknet_input=64, recon=False, device='cuda')
        x = T(synth_images(4, 64, seed=22))
        y = T(synth_gtsdb_labels(4, 2, 43, seed=23))

        def fh(net):
            o = net(x.cuda())
            return o, loss_fns.darkcapsule_loss(o, y.cuda(), p)

        def fo(net, dt):
            o = net(x.to(dt))
            return o, OL.darkcapsule_loss(o, y, p)
        run('darkcapsule', lambda: models.DarkCapsuleNet(p), lambda: OM.DarkCapsuleNet(p), fh, fo)
    if 'darknet' in sel:
        p = make_params(model='darknet_d', n_grid=4, n_boxes=2, n_classes=0, darknet_input=128, dropout=0.0, device='cuda')
        x = T(synth_images(8, 128, seed=22))
        y = T(synth_gtsdb_labels(8, 4, 0, seed=27))

        def fh(net):
            o = net(x.cuda())
            return o, loss_fns.dark_loss(o, y.cuda(), p)

        def fo(net, dt):
            o = net(x.to(dt))
            return o, OL.dark_loss(o, y, p)[0]
        run('darknet_d(128,B=8)', lambda: models.DarkNet(p), lambda: OM.DarkNet(p), fh, fo)
