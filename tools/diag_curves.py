"""Per-step deviation of the 20-step Adam loss curves (tests/golden/curves*.npz) from the reference's fp32 and fp64 curves under
different kernel switches: which kernel family moves the curve, and from which step on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import capsyolo_amd
from capsyolo_amd import loss_fns, models, ops, optim
from helpers import closed_form_state, load_golden, make_params, synth_gtsdb_labels, synth_images

T = torch.from_numpy


def curve(tag, golden, **sw):
    g = load_golden(golden)
    H, gg, B, seed = (int(v) for v in g[tag + '_cfg'])
    p = make_params(model='darkcapsule', n_grid=gg, darknet_input=H, recon=False, device='cuda')
    x, y = T(synth_images(B, H, seed=seed)).cuda(), T(synth_gtsdb_labels(B, gg, 43, seed=seed + 1)).cuda()
    old = dict((k, getattr(ops, k)) for k in sw)
    for k, v in sw.items():
        setattr(ops, k, v)
    try:
        net = models.DarkCapsuleNet(p)
        net.load_state_dict(closed_form_state(net))
        net.cuda().train()
        opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
        out = []
        for _ in range(20):
            loss = loss_fns.darkcapsule_loss(net(x), y, p)
            opt.zero_grad(); loss.backward(); opt.step()
            out.append(loss.item())
    finally:
        for k, v in old.items():
            setattr(ops, k, v)
    return np.array(out), g[tag + '_curve'], g[tag + '_curve_ulp'], load_golden('curves64')[tag + '_curve64']


if __name__ == '__main__':
    BIG = 1 << 62
    cases = [('default', {}),
             ('two-pass first block', dict(CONV1_MOMENTS_MIN_PIXELS=BIG)),
             ('moments forced', dict(CONV1_MOMENTS_MIN_PIXELS=0)),
             ('direct kernels', dict(USE_WINOGRAD=False, CONV1_MOMENTS_MIN_PIXELS=BIG)),
             ('direct, generic first block', dict(USE_WINOGRAD=False, USE_CONV1=False, USE_CONV1_BWD=False, CONV1_MOMENTS_MIN_PIXELS=BIG)),
             ('direct, no fusions', dict(USE_WINOGRAD=False, USE_CONV1=False, USE_CONV1_BWD=False, CONV1_MOMENTS_MIN_PIXELS=BIG,
                                         FUSE_BN_BWD_REDUCE=False, FUSE_BN_BWD_APPLY=False, FUSE_INPUT_AFFINE=False)),
             ('winograd, no fusions', dict(CONV1_MOMENTS_MIN_PIXELS=BIG, FUSE_BN_BWD_REDUCE=False, FUSE_BN_BWD_APPLY=False,
                                           FUSE_INPUT_AFFINE=False))]
    for tag, golden in (('dc256', 'curves256'), ('dc96', 'curves'), ('dc64', 'curves')):
        for name, sw in cases:
            c, ref, ulp, r64 = curve(tag, golden, **sw)
            span = float(ref.max() - ref.min())
            d32, d64 = np.abs(c - ref) / span, np.abs(c - r64) / span
            print('%-6s %-28s max dev: %.3f %% (fp32 ref) %.3f %% (fp64 ref) | ref band %.3f %%, ref32-ref64 %.3f %% | per step (fp64, %%): %s'
                  % (tag, name, 100 * d32.max(), 100 * d64.max(), 100 * np.abs(ulp - ref).max() / span, 100 * np.abs(ref - r64).max() / span,
                     ' '.join('%.2f' % (100 * v) for v in d64)), flush=True)
        print('%-6s reference fp32 vs fp64 per step (%%): %s' % (tag, ' '.join('%.2f' % (100 * v) for v in np.abs(ref - r64) / span)))
