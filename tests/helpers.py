"""Shared test helpers: deterministic closed-form weights/inputs and synthetic labels.

Weights are regenerated from a formula (never shipped), so the golden fixtures
only have to hold inputs' recipe + the reference's outputs.
"""
import math
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')


def _phase(name):
    return (sum((i + 1) * ord(c) for i, c in enumerate(name)) % 1000) / 1000.0 * 2 * math.pi


def wave(shape, phase, amp=1.0, freq=0.37, dtype=np.float32):
    """amp*sin(freq*k + phase) over the flattened index k (computed in float64)."""
    n = int(np.prod(shape)) if len(shape) else 1
    k = np.arange(n, dtype=np.float64)
    return (amp * np.sin(freq * k + phase)).astype(dtype).reshape(shape)


def closed_form_state(model):
    """Deterministic, numerically healthy values for every entry of model.state_dict()."""
    out = {}
    for name, t in model.state_dict().items():
        shape, ph = tuple(t.shape), _phase(name)
        if name.endswith('num_batches_tracked'):
            out[name] = torch.zeros_like(t)
            continue
        if name.endswith('running_var'):
            v = 1.0 + 0.2 * wave(shape, ph)
        elif name.endswith('running_mean'):
            v = 0.1 * wave(shape, ph)
        elif name.endswith('route_weights'):
            v = 0.1 * wave(shape, ph, freq=0.731)
        elif '.bn_' in name and name.endswith('weight') or (len(shape) == 1 and name.endswith('weight')):
            v = 1.0 + 0.1 * wave(shape, ph)
        elif name.endswith('bias'):
            v = 0.05 * wave(shape, ph)
        else:  # conv / linear weight
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            v = math.sqrt(3.0 / fan_in) * wave(shape, ph, freq=0.737)
        out[name] = torch.from_numpy(np.ascontiguousarray(v)).to(t.dtype)
    return out


def load_closed_form(model):
    model.load_state_dict(closed_form_state(model))
    return model


def make_params(**kw):
    d = dict(device='cpu', n_classes=43, n_grid=2, n_boxes=2, dropout=0.0, recon=False, recon_coef=5e-4,
             l_coord=5, l_noobj=0.5, darknet_input=64, batch_size=4, model='darkcapsule')
    d.update(kw)
    return types.SimpleNamespace(**d)


def synth_images(n, hw, seed, nchw=True):
    """uint8 U{0..255} NHWC -> (x-128)/128 float32 (utils.py:122-123), optionally permuted to NCHW."""
    rng = np.random.default_rng(seed)
    x = (rng.integers(0, 256, (n, hw, hw, 3), dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)) if nchw else x


def synth_gtsdb_labels(n, g, n_classes, seed):
    """float64 [n,g,g,5+C]: 1..3 object cells per image (build_data.py:84-103 shapes)."""
    rng = np.random.default_rng(seed)
    y = np.zeros((n, g, g, 5 + n_classes), dtype=np.float64)
    for i in range(n):
        k = min(int(rng.integers(1, 4)), g * g)
        cells = rng.choice(g * g, size=k, replace=False)
        for c in cells:
            r, col = divmod(int(c), g)
            y[i, r, col, 0] = 1.0
            y[i, r, col, 1:3] = rng.uniform(0.0, 1.0, 2)
            y[i, r, col, 3:5] = rng.uniform(0.02, 0.15, 2)
            if n_classes > 0:
                y[i, r, col, 5 + int(rng.integers(0, n_classes))] = 1.0
    return y


def routing_case(ci, R, N, C, Din, Dout):
    """Inputs of routing golden case ``ci`` (regenerated, not stored): u, W, output cotangent G."""
    u = wave((R, N, Din), 0.11 * (ci + 1), amp=0.8, freq=0.913)
    W = 0.1 * wave((1, N, C, Din, Dout), 0.7 + ci, freq=0.731) + 0.05 * wave((1, N, C, Din, Dout), 0.2, freq=0.0517)
    G = wave((R, C, Dout), 1.3 + ci, freq=1.31)
    return u, W, G


def grad_digest(t):
    """Compact fingerprint of a (possibly huge) gradient tensor: sum, abs-sum, a strided sample."""
    f = t.detach().double().reshape(-1)
    step = max(1, f.numel() // 256)
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[::step][:256].numpy()])


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))


def perturb_one_ulp(x):
    """Every element of a float32 array moved by exactly one ulp, up for even flat indices and down for odd ones."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    flat = x.reshape(-1)
    up = (np.arange(flat.size) % 2) == 0
    out = np.where(up, np.nextafter(flat, np.float32(np.inf)), np.nextafter(flat, np.float32(-np.inf)))
    return out.astype(np.float32).reshape(x.shape)


def perturb_ulps(x, seed, ulps=1):
    """Every element of a float32 array moved by exactly ``ulps`` ulps, up or down by a coin flip seeded with ``seed`` (the members of
    the loss-curve ensembles: tests/golden/curves_ens.npz)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    up = np.random.default_rng(seed).integers(0, 2, x.size).astype(bool).reshape(x.shape)
    hi, lo = x.copy(), x.copy()
    for _ in range(int(ulps)):
        hi = np.nextafter(hi, np.float32(np.inf))
        lo = np.nextafter(lo, np.float32(-np.inf))
    return np.where(up, hi, lo).astype(np.float32)


_DARKNET_COUT = [32, 64, 128, 64, 128, 256, 128, 256, 512, 256, 512, 256, 512, 1024, 512, 1024, 512, 1024]
_DARKNET_K = [3, 3, 3, 1, 3, 3, 1, 3, 3, 1, 3, 1, 3, 3, 1, 3, 1, 3]


def write_tf_style_darknet_npz(path, n_layers):
    """A synthetic stand-in for ./darknet19_weights.npz (absent from the reference repo; models.py:238-269 reads it):
    keys '<idx>-<scope>/<name>:0' with HWIO kernels and the four BatchNorm vectors of the first n_layers layers,
    closed-form values."""
    arrays, cin = {}, 3
    for idx in range(n_layers):
        cout, k = _DARKNET_COUT[idx], _DARKNET_K[idx]
        ph = 0.1 * (idx + 1)
        arrays['%d-convolutional/kernel:0' % idx] = (np.sqrt(3.0 / (cin * k * k)) * wave((k, k, cin, cout), ph, freq=0.613)).astype(np.float32)
        arrays['%d-convolutional/biases:0' % idx] = 0.05 * wave((cout,), ph + 1)
        arrays['%d-convolutional/gamma:0' % idx] = 1.0 + 0.1 * wave((cout,), ph + 2)
        arrays['%d-convolutional/moving_mean:0' % idx] = 0.1 * wave((cout,), ph + 3)
        arrays['%d-convolutional/moving_variance:0' % idx] = 1.0 + 0.2 * wave((cout,), ph + 4)
        cin = cout
    np.savez(path, **arrays)
