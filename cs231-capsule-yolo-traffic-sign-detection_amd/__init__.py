"""MI355X-native training hot path of cs231-capsule-yolo-traffic-sign-detection.

Importable as ``capsyolo_amd`` (see the alias module at the repository root).  The package holds the
host-side mirror of the reference's plugin surface (models / loss_fns / predict_fns / utils /
config, same names and call signatures) over hand-written gfx950 kernels in ``csrc/`` reached
through the C-ABI of ``include/capsyolo_hip.h``.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ['config', 'models', 'loss_fns', 'predict_fns', 'utils', 'optim', 'dp', 'synth', 'ops']
