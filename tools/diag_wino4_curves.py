"""Loss-curve deviation with conv_2 forward / input gradient on Winograd F(4x4,3x3) (winograd4.hip) against F(2x2,3x3):
the closed-form-weight recipes (tests/golden/curves.npz, curves256.npz) and the default-initialisation recipes
(curves_init.npz), 20 Adam steps each, against the reference's fp32 curve, its one-ulp band and its fp64 curve."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import capsyolo_amd
from capsyolo_amd import ops
import test_gpu_models as tm
from diag_curves import curve

if __name__ == '__main__':
    for f4 in (False, True):
        ops.USE_WINOGRAD4, ops.WINOGRAD4_MIN_PIXELS = f4, 0
        name = 'F(4x4,3x3)' if f4 else 'F(2x2,3x3)'
        for tag, golden in (('dc96', 'curves'), ('dc256', 'curves256')):
            c, ref, ulp, r64 = curve(tag, golden)
            span = float(ref.max() - ref.min())
            print('%-11s closed-form %-6s dev %.3f %% of range (fp32 ref), %.3f %% (fp64 ref) | ref one-ulp band %.3f %%, ref32-ref64 %.3f %%'
                  % (name, tag, 100 * np.abs(c - ref).max() / span, 100 * np.abs(c - r64).max() / span,
                     100 * np.abs(ulp - ref).max() / span, 100 * np.abs(ref - r64).max() / span), flush=True)
        for tag in ('di96', 'di256'):
            r = tm._hip_curve_default_init(tag)
            print('%-11s default-init %-6s dev %.3f %% of range (fp32 ref), %.3f %% (fp64 ref) | ref one-ulp band %.3f %%, ref32-ref64 %.3f %%'
                  % (name, tag, 100 * r['dev'], 100 * r['dev64'], 100 * r['band'], 100 * r['ref_dev64']), flush=True)
