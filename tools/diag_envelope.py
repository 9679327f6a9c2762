"""Per-step loss-curve deviations of the kernels against the ensemble envelopes of tests/golden/curves_ens.npz (GPU): the default
kernels on the three recipes, every F(4x4,.) family forced onto the well-conditioned recipe, the direct kernels, and the fault
injection (tests/helpers.py: Winograd4Perturbation) at 1e-3 / 1e-4.  The numbers the bounds of tests/test_gpu_models.py were set from.

    python3 tools/diag_envelope.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import capsyolo_amd  # noqa: E402,F401
from capsyolo_amd import ops  # noqa: E402
from helpers import Winograd4Perturbation, curve_envelope, curve_in_envelope, hip_curve_default_init, load_golden  # noqa: E402

g = load_golden('curves_ens')
np.set_printoptions(precision=2, linewidth=260)


def report(label, tag, curve):
    env = curve_envelope(g, tag, floor_frac=None if tag == 'dw64' else 2e-4)
    dev, ok = curve_in_envelope(curve, env)
    print('%-44s %s: %2d/20 inside; dev/range %s' % (label, tag, int(ok.sum()), np.array2string(dev / env['span'], formatter={'float_kind': lambda v: '%.1e' % v})))
    print('%-44s      bound/range %s' % ('', np.array2string(env['bound'] / env['span'], formatter={'float_kind': lambda v: '%.1e' % v})))
    print('%-44s      |c - fp64 ref|/range %s' % ('', np.array2string(np.abs(curve - env['curve64']) / env['span'], formatter={'float_kind': lambda v: '%.1e' % v})))
    return {'label': label, 'tag': tag, 'dev_over_range': (dev / env['span']).tolist(), 'inside': ok.tolist()}


saved = dict((k, getattr(ops, k)) for k in ('USE_WINOGRAD', 'USE_WINOGRAD4', 'WINOGRAD4_MIN_PIXELS', 'WINOGRAD4_S2_MIN_PIXELS',
                                            'CONV1_MOMENTS_MIN_PIXELS', 'USE_WINOGRAD4_WGRAD', 'USE_WINOGRAD4_S2', 'USE_WINOGRAD4_S2_DGRAD'))
out = []
for tag in ('di96', 'di256', 'dw64'):
    out.append(report('default kernels', tag, hip_curve_default_init(g, tag)))
ops.WINOGRAD4_MIN_PIXELS = ops.WINOGRAD4_S2_MIN_PIXELS = ops.CONV1_MOMENTS_MIN_PIXELS = 0
out.append(report('all gates open (F(4x4,2x2) on conv_3..5)', 'dw64', hip_curve_default_init(g, 'dw64')))
out.append(report('all gates open', 'di96', hip_curve_default_init(g, 'di96')))
for k, v in saved.items():
    setattr(ops, k, v)
ops.USE_WINOGRAD4 = ops.USE_WINOGRAD4_WGRAD = ops.USE_WINOGRAD4_S2 = ops.USE_WINOGRAD4_S2_DGRAD = False
out.append(report('F(2x2,.) kernels only', 'dw64', hip_curve_default_init(g, 'dw64')))
ops.USE_WINOGRAD = False
out.append(report('direct kernels', 'dw64', hip_curve_default_init(g, 'dw64')))
for k, v in saved.items():
    setattr(ops, k, v)
for eps in (1e-3, 1e-4):
    with Winograd4Perturbation(ops, eps) as pt:
        out.append(report('F(4x4,3x3) U row 1 scaled by 1 + %g' % eps, 'dw64', hip_curve_default_init(g, 'dw64')))
        print('   (%d pack calls perturbed)' % pt.hits)
    with Winograd4Perturbation(ops, eps) as pt:
        out.append(report('F(4x4,3x3) U row 1 scaled by 1 + %g' % eps, 'di256', hip_curve_default_init(g, 'di256')))
print(json.dumps(out))
