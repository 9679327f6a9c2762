"""Parity of every HIP kernel (through the C-ABI) against the CPU oracle / torch fp64 on seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, make_params, routing_case, synth_gtsdb_labels, wave

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def dev():
    return torch.device('cuda:0')


def close(a, b, rtol, atol_rel=0.0, atol=0.0):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol + atol_rel * (np.abs(b).max() if b.size else 0.0))


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).float()


# ------------------------------------------------------------------------------------------------ convolution
CONV_CASES = [
    # B, Cin, H, Cout, k, s, p, nchw
    (2, 128, 12, 256, 3, 1, 1, False),     # conv_2 shape class (vec loader, 2 N tiles)
    (2, 256, 16, 64, 4, 2, 1, False),      # conv_3 class (k4 s2, BN=64)
    (3, 64, 10, 128, 4, 2, 1, False),      # conv_4 class
    (2, 3, 20, 128, 3, 1, 1, True),        # conv_1: NCHW input, scalar loader, K=27
    (2, 3, 17, 256, 9, 1, 0, True),        # CapsuleNet conv1 (K=243), odd size
    (2, 256, 12, 128, 8, 2, 0, False),     # fused primary capsules conv (64 taps)
    (2, 1024, 4, 10, 1, 1, 0, False),      # DarkNet conv_19 (N=10 padded)
    (3, 16, 8, 4, 3, 1, 1, False),         # decoder conv (tiny channels)
    (5, 32, 9, 64, 3, 1, 1, False),        # M not a multiple of the tile, Cin=32
    (4, 128, 16, 256, 4, 2, 1, False),     # conv_5 at H=64 (as inside the DarkCapsuleNet golden model)
    (4, 16, 32, 3, 3, 1, 1, False),        # decoder's last conv (Cout=3: dgrad runs the scalar loader with Cin=3)
    (4, 64, 32, 128, 4, 2, 1, False),      # conv_4 at H=64
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
    from capsyolo_amd import ops
    B, Cin, H, Cout, k, s, p, nchw = case
    x = rnd((B, Cin, H, H), 1)
    w = rnd((Cout, Cin, k, k), 2, (1.0 / (Cin * k * k)) ** 0.5)
    b = rnd((Cout,), 3, 0.1)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    zr = F.conv2d(xd, wd, b.double(), stride=s, padding=p)
    gz = rnd(tuple(zr.shape), 4)
    zr.backward(gz.double())
    xg = x.to(dev()) if nchw else x.permute(0, 2, 3, 1).contiguous().to(dev())
    z = ops.conv_forward(xg, w.to(dev()), b.to(dev()), k, s, p, nchw)
    close(z.permute(0, 3, 1, 2), zr, 2e-5, 2e-5)
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    dW = ops.conv_wgrad(xg, gzd, k, s, p, nchw)
    close(dW, wd.grad, 5e-5, 5e-5)
    if not nchw:
        dx = ops.conv_dgrad(gzd, w.to(dev()), (B, H, H, Cin), k, s, p)
        close(dx.permute(0, 3, 1, 2), xd.grad, 5e-5, 5e-5)


@pytest.mark.parametrize('case', [(32, 256, 24, 128, 8, 2, 0), (5, 480, 7, 224, 5, 1, 2), (2, 1024, 4, 10, 1, 1, 0)])
def test_conv_gemm_split_reduction(case, monkeypatch):
    """Under-filled grids split their reduction over several blocks per output tile (cy_conv_gemm_ws_floats > 0; round 4): the
    CapsuleNet primary-capsule convolution at its full shape (batch 32: 42 tiles x 648 K tiles -> 12 shares), a padded-N layer with
    a ragged last share, the 1x1 head.  Forward and input gradient against torch fp64, against the unsplit launch of the same
    kernel (no workspace), and bit-identical on repetition (the shares are added in a fixed order)."""
    from capsyolo_amd import ops
    from capsyolo_amd._lib import ConvGemm, query
    import ctypes as C
    B, Cin, H, Cout, k, s, p = case
    x = rnd((B, Cin, H, H), 1)
    w = rnd((Cout, Cin, k, k), 2, (1.0 / (Cin * k * k)) ** 0.5)
    b = rnd((Cout,), 3, 0.1)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    zr = F.conv2d(xd, wd, b.double(), stride=s, padding=p)
    gz = rnd(tuple(zr.shape), 4)
    zr.backward(gz.double())
    xg, wg, bg = x.permute(0, 2, 3, 1).contiguous().to(dev()), w.to(dev()), b.to(dev())
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    seen = []
    orig = ops._gemm_workspace

    def spy(a, like):
        ws = orig(a, like)
        seen.append(0 if ws is None else ws.numel())
        return ws
    monkeypatch.setattr(ops, '_gemm_workspace', spy)
    z = ops.conv_forward(xg, wg, bg, k, s, p, False)
    dx = ops.conv_dgrad(gzd, wg, (B, H, H, Cin), k, s, p)
    assert seen and seen[0] > 0, seen                          # the forward did split
    z2 = ops.conv_forward(xg, wg, bg, k, s, p, False)
    dx2 = ops.conv_dgrad(gzd, wg, (B, H, H, Cin), k, s, p)
    assert torch.equal(z, z2) and torch.equal(dx, dx2)
    monkeypatch.setattr(ops, '_gemm_workspace', lambda a, like: None)
    z1 = ops.conv_forward(xg, wg, bg, k, s, p, False)
    dx1 = ops.conv_dgrad(gzd, wg, (B, H, H, Cin), k, s, p)
    close(z.permute(0, 3, 1, 2), zr, 2e-5, 2e-5)
    close(dx.permute(0, 3, 1, 2), xd.grad, 5e-5, 5e-5)
    close(z, z1, 1e-5, 1e-5)
    close(dx, dx1, 1e-5, 1e-5)


@pytest.mark.parametrize('case', [(16, 512, 13, 1024, 2), (4, 256, 13, 768, 4), (3, 128, 16, 384, 3), (16, 512, 26, 256, 1)])
def test_conv3x3_winograd_input_gradient_splits_its_reduction(case, monkeypatch):
    """F(2x2,3x3) launches without an epilogue whose tiles fill at most half of the CUs run several blocks per tile on shares of the
    reduction channels (cy_conv3x3_winograd_ws / cy_wino_split_ws_floats; round 4): DarkNet's 13 x 13 input gradients (conv_14 /
    16 / 18: 128 tiles, 2 shares), a four-share and a three-share case, and a shape that must NOT split (26 x 26: 256 tiles).
    Against torch fp64, against the unsplit launch of the same kernel, bit-identical on repetition; the forward (bias, statistics:
    an epilogue) never splits."""
    from capsyolo_amd import ops
    from capsyolo_amd._lib import query
    monkeypatch.setattr(ops, 'USE_WINOGRAD4', False)
    B, Cin, H, Cout, shares = case
    x = rnd((B, Cin, H, H), 71)
    w = rnd((Cout, Cin, 3, 3), 72, (1.0 / (Cin * 9)) ** 0.5)
    xd = x.double().requires_grad_(True)
    zr = F.conv2d(xd, w.double(), None, padding=1)
    gz = rnd(tuple(zr.shape), 74)
    zr.backward(gz.double())
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    wg = w.to(dev())
    n = query('cy_wino_split_ws_floats', B, H, H, Cout, Cin, 1)        # (the input gradient reduces over the layer's Cout)
    assert n == (shares * B * H * H * Cin if shares > 1 else 0), (n, shares)
    assert query('cy_wino_split_ws_floats', B, H, H, Cin, Cout, 0) == 0
    dx = ops.conv_dgrad(gzd, wg, (B, H, H, Cin), 3, 1, 1)
    dx_again = ops.conv_dgrad(gzd, wg, (B, H, H, Cin), 3, 1, 1)
    assert torch.equal(dx, dx_again)
    close(dx.permute(0, 3, 1, 2), xd.grad, 1e-4, 1e-4)
    real_query = ops.query
    monkeypatch.setattr(ops, 'query', lambda name, *a: 0 if name == 'cy_wino_split_ws_floats' else real_query(name, *a))
    dx_unsplit = ops.conv_dgrad(gzd, wg, (B, H, H, Cin), 3, 1, 1)
    close(dx, dx_unsplit, 2e-5, 2e-5)


@pytest.mark.parametrize('case', [(2, 128, 16, 256), (1, 8, 10, 64), (3, 64, 37, 40), (2, 256, 18, 128), (2, 32, 15, 10),
                                  (3, 8, 200, 64), (2, 16, 104, 128)])   # > 256 tiles: persistent blocks walk several tiles
@pytest.mark.parametrize('f4', [False, True])
def test_conv3x3_winograd_matches_direct_and_fp64(case, f4, monkeypatch):
    """Fused Winograd F(2x2,3x3) / F(4x4,3x3) (forward + input gradient, bias, BN statistics, odd sizes, padded channels)
    against torch fp64, and switched off against the direct implicit GEMM."""
    from capsyolo_amd import ops
    monkeypatch.setattr(ops, 'USE_WINOGRAD4', f4)
    monkeypatch.setattr(ops, 'WINOGRAD4_MIN_PIXELS', 0)
    B, Cin, H, Cout = case
    x = rnd((B, Cin, H, H), 61)
    w = rnd((Cout, Cin, 3, 3), 62, (1.0 / (Cin * 9)) ** 0.5)
    b = rnd((Cout,), 63, 0.1)
    xd, wd = x.double().requires_grad_(True), w.double()
    zr = F.conv2d(xd, wd, b.double(), padding=1)
    gz = rnd(tuple(zr.shape), 64)
    zr.backward(gz.double())
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev())
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    assert ops.USE_WINOGRAD
    stats = torch.zeros((ops.STATS_COPIES, Cout, 2), dtype=torch.float64, device=dev())
    z = ops.conv_forward(xg, w.to(dev()), b.to(dev()), 3, 1, 1, False, stats)
    stats = stats.sum(0)
    dx = ops.conv_dgrad(gzd, w.to(dev()), (B, H, H, Cin), 3, 1, 1)
    close(z.permute(0, 3, 1, 2), zr, 5e-5, 5e-5)
    close(dx.permute(0, 3, 1, 2), xd.grad, 1e-4, 1e-4)
    close(stats[:, 0], zr.sum(dim=(0, 2, 3)), 1e-4, 1e-4)
    close(stats[:, 1], (zr ** 2).sum(dim=(0, 2, 3)), 1e-4, 1e-4)
    try:
        ops.USE_WINOGRAD = False
        z2 = ops.conv_forward(xg, w.to(dev()), b.to(dev()), 3, 1, 1)
        dx2 = ops.conv_dgrad(gzd, w.to(dev()), (B, H, H, Cin), 3, 1, 1)
    finally:
        ops.USE_WINOGRAD = True
    close(z, z2, 1e-4, 1e-4)
    close(dx, dx2, 2e-4, 2e-4)


@pytest.mark.parametrize('case', [(2, 128, 16, 256, 0.0), (1, 64, 10, 64, 3.0), (3, 64, 37, 128, -1.0), (2, 256, 18, 64, 0.5),
                                  (5, 64, 3, 64, 0.0), (1, 192, 9, 128, 0.0), (2, 128, 40, 64, 20.0)])
@pytest.mark.parametrize('premasked', [0, 1])
def test_conv3x3_winograd_wgrad_with_fused_batchnorm_backward(case, premasked):
    """cy_conv3x3_winograd_wgrad_bn (BatchNorm + LeakyReLU backward pass 2 applied inside the weight-gradient kernel, dz written
    for the input-gradient kernel) against the two-kernel path (cy_bn_bwd_apply, then cy_conv3x3_winograd_wgrad) and against
    the fp64 formula: 1 / 2 / 3 / 4 input-channel blocks (who writes dz), border chunks, one-chunk ranges, a channel mean far
    from zero (the (z - mean) term must not cancel).  premasked: the gradient arrives as d = dA * lrelu'(y) already (what the
    stride-2 input-gradient kernel stores next to its fused sums): the kernel then applies no mask."""
    from capsyolo_amd import ops
    from capsyolo_amd._lib import call, query
    B, Cin, H, Cout, zmean = case
    W_ = H + 3 if H > 8 else H
    P = B * H * W_
    x = rnd((B, H, W_, Cin), 171).to(dev())
    z = (rnd((B, H, W_, Cout), 172) * 1.5 + zmean).to(dev())
    da = rnd((B, H, W_, Cout), 173).to(dev())
    gamma = (rnd((Cout,), 174).abs() + 0.5).to(dev())
    beta = rnd((Cout,), 175, 0.3).to(dev())
    slope = 0.1
    zd = z.double().reshape(P, Cout)
    mean, var = zd.mean(0), zd.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = (gamma.double() * invstd), (beta.double() - mean * gamma.double() * invstd)
    # an activation within fp32 rounding of 0 may legitimately take either LeakyReLU branch (and moves its channel's sums with
    # it): such elements are moved away from 0 (the kernels take mean / invstd as inputs: they need not be z's own)
    y = zd * scale + shift
    near = (y.abs() < 1e-4 * (zd.abs() * scale.abs() + shift.abs())).reshape(z.shape)
    z = torch.where(near, z + 0.05, z)
    zd = z.double().reshape(P, Cout)
    y = zd * scale + shift
    assert not (y.abs() < 1e-4 * (zd.abs() * scale.abs() + shift.abs())).any()
    d = torch.where(y > 0, da.double().reshape(P, Cout), slope * da.double().reshape(P, Cout))
    xh = (zd - mean) * invstd
    dz64 = scale * (d - d.mean(0) - xh * (d * xh).mean(0))
    f = lambda t: t.float().contiguous()
    scale_f, shift_f, mean_f, invstd_f = f(scale), f(shift), f(mean), f(invstd)
    st = torch.cuda.current_stream().cuda_stream
    red = torch.zeros(Cout, 2, dtype=torch.float64, device=dev())
    call('cy_bn_bwd_reduce', z.data_ptr(), da.data_ptr(), scale_f.data_ptr(), shift_f.data_ptr(), mean_f.data_ptr(),
         invstd_f.data_ptr(), slope, red.data_ptr(), P, Cout, st)
    dz_ref = torch.empty_like(z)
    dg, db = torch.empty(Cout, device=dev()), torch.empty(Cout, device=dev())
    if premasked:                                                     # from here on the kernels see d and slope 1
        da = d.float().reshape(z.shape).contiguous()
        slope = 1.0
    call('cy_bn_bwd_apply', z.data_ptr(), da.data_ptr(), dz_ref.data_ptr(), scale_f.data_ptr(), shift_f.data_ptr(),
         mean_f.data_ptr(), invstd_f.data_ptr(), gamma.data_ptr(), slope, red.data_ptr(), dg.data_ptr(), db.data_ptr(), P, Cout, st)
    dw_ref = ops.conv_wgrad(x, dz_ref, 3, 1, 1)
    dz = torch.full_like(z, float('nan'))
    dw = torch.empty(Cout, Cin, 3, 3, device=dev())
    ws = torch.empty(query('cy_wino_wgrad_ws_floats', B, Cin, Cout), device=dev())
    call('cy_conv3x3_winograd_wgrad_bn', x.data_ptr(), z.data_ptr(), da.data_ptr(), dz.data_ptr(), scale_f.data_ptr(),
         shift_f.data_ptr(), mean_f.data_ptr(), invstd_f.data_ptr(), slope, premasked, red.data_ptr(), P, dw.data_ptr(), ws.data_ptr(),
         B, H, W_, Cin, Cout, st)
    dg2, db2 = torch.empty(Cout, device=dev()), torch.empty(Cout, device=dev())
    call('cy_bn_param_grad', red.data_ptr(), dg2.data_ptr(), db2.data_ptr(), Cout, st)
    torch.cuda.synchronize()
    assert torch.isfinite(dz).all()                                   # every element written
    zs = dz64.abs().max().item()
    close(dz.reshape(P, Cout), dz64, 1e-4, 0.0, 2e-5 * zs)
    close(dz, dz_ref, 1e-4, 0.0, 2e-6 * zs)
    ws_ = dw_ref.abs().max().item()
    close(dw, dw_ref, 4e-5, 4e-5 * ws_)
    assert torch.equal(dg, dg2) and torch.equal(db, db2)
    with pytest.raises(Exception):                                    # in place: several blocks read every element
        call('cy_conv3x3_winograd_wgrad_bn', x.data_ptr(), z.data_ptr(), da.data_ptr(), da.data_ptr(), scale_f.data_ptr(),
             shift_f.data_ptr(), mean_f.data_ptr(), invstd_f.data_ptr(), slope, premasked, red.data_ptr(), P, dw.data_ptr(), ws.data_ptr(),
             B, H, W_, Cin, Cout, st)


@pytest.mark.parametrize('case', [(2, 32, 64, 128, True), (3, 16, 32, 32, False), (1, 7, 32, 64, True), (2, 64, 96, 40, True), (5, 3, 32, 8, True),
                                  (1, 1, 32, 8, True), (2, 2, 64, 16, False), (3, 416, 416, 32, True)])
def test_conv1_statistics_from_patch_moments(case):
    """cy_conv1_3x3_stats: sum z and sum z^2 of the first layer's output from the 28 x 28 moment matrix of its 27-element input
    patches (csrc/conv1_moments.hip) against the sums of the layer's actual output in fp64 (image borders, odd heights, with /
    without bias, a Cout the MFMA forward kernel does not take)."""
    from capsyolo_amd._lib import call, query
    B, H, W_, Cout, has_bias = case
    x = rnd((B, 3, H, W_), 301)
    w = rnd((Cout, 3, 3, 3), 302, 0.3)
    b = rnd((Cout,), 303, 0.2) if has_bias else None
    zr = F.conv2d(x.double(), w.double(), b.double() if has_bias else None, padding=1)
    xg, wg = x.to(dev()), w.to(dev())
    bg = b.to(dev()) if has_bias else None
    stats = torch.zeros(Cout, 2, dtype=torch.float64, device=dev())
    ws = torch.empty(query('cy_conv1_3x3_stats_ws_floats', B, H), device=dev())
    call('cy_conv1_3x3_stats', xg.data_ptr(), wg.data_ptr(), bg.data_ptr() if has_bias else None, stats.data_ptr(), ws.data_ptr(),
         B, H, W_, Cout, torch.cuda.current_stream().cuda_stream)
    s1, s2 = zr.sum(dim=(0, 2, 3)), (zr ** 2).sum(dim=(0, 2, 3))
    close(stats[:, 0], s1, 1e-5, 0.0, 1e-5 * float(s2.max().sqrt()) * (B * H * W_) ** 0.5)
    close(stats[:, 1], s2, 2e-5)


def test_convolution_kernels_repeat_bit_identically():
    """tools/check_determinism.py: every convolution kernel (fp32 Winograd / bf16 MFMA, forward, input and weight gradient,
    the weight gradient with the fused BatchNorm backward) returns the same bits when it is run again on the same inputs --
    fixed-order partial sums, and no VALU work in a hardware hazard (a bf16 variant of the fused kernel that differed from
    run to run in lanes 48-63 was caught by exactly this check and not shipped)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_determinism.py'), '4'], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'check_determinism: ok' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize('case', [(2, 128, 16, 256), (1, 64, 10, 64), (3, 64, 37, 128), (2, 256, 18, 64), (5, 64, 3, 64)])
def test_conv3x3_winograd_wgrad_matches_direct_and_fp64(case):
    """Winograd F(3x3,2x2) weight gradient (odd sizes: partially filled tile groups; one image = one partial sum)
    against torch fp64 and against the direct MFMA weight-gradient kernel."""
    from capsyolo_amd import ops
    B, Cin, H, Cout = case
    x = rnd((B, Cin, H, H), 71)
    w = rnd((Cout, Cin, 3, 3), 72, (1.0 / (Cin * 9)) ** 0.5)
    wd = w.double().requires_grad_(True)
    zr = F.conv2d(x.double(), wd, None, padding=1)
    gz = rnd(tuple(zr.shape), 74)
    zr.backward(gz.double())
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev())
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    assert ops.USE_WINOGRAD
    ops.timer.reset()
    ops.timer.enabled = True
    try:
        dw = ops.conv_wgrad(xg, gzd, 3, 1, 1, False, 'wg')
    finally:
        ops.timer.enabled = False
    torch.cuda.synchronize()
    assert 'conv_wino_wgrad/wg' in ops.timer.summary()
    scale = wd.grad.abs().max().item()
    close(dw, wd.grad, 2e-5, 2e-5 * scale)
    try:
        ops.USE_WINOGRAD = False
        dw2 = ops.conv_wgrad(xg, gzd, 3, 1, 1)
    finally:
        ops.USE_WINOGRAD = True
    close(dw, dw2, 4e-5, 4e-5 * scale)


@pytest.mark.parametrize('case', [(2, 64, 32, 64), (1, 8, 10, 64), (3, 16, 70, 40), (2, 256, 36, 128), (1, 24, 6, 10),
                                  (2, 8, 128, 64)])
@pytest.mark.parametrize('f42', [False, True])
def test_conv4x4s2_winograd_matches_direct_and_fp64(case, f42, monkeypatch):
    """Fused Winograd F(2x2,2x2) forward of the 4x4 / stride 2 / pad 1 layers (bias, BatchNorm statistics, image
    borders, partial blocks, padded channels) against torch fp64 and, switched off, the direct implicit GEMM."""
    from capsyolo_amd import ops
    monkeypatch.setattr(ops, 'USE_WINOGRAD4_S2', f42)       # F(4x4,2x2) (winograd4_s2.hip) forced on / off
    monkeypatch.setattr(ops, 'WINOGRAD4_S2_MIN_PIXELS', 0)
    B, Cin, H, Cout = case
    x = rnd((B, Cin, H, H), 81)
    w = rnd((Cout, Cin, 4, 4), 82, (1.0 / (Cin * 16)) ** 0.5)
    b = rnd((Cout,), 83, 0.1)
    zr = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev())
    assert ops.USE_WINOGRAD and ops.USE_WINOGRAD_S2
    stats = torch.zeros((ops.STATS_COPIES, Cout, 2), dtype=torch.float64, device=dev())
    ops.timer.reset()
    ops.timer.enabled = True
    try:
        z = ops.conv_forward(xg, w.to(dev()), b.to(dev()), 4, 2, 1, False, stats, False, 'c3')
    finally:
        ops.timer.enabled = False
    torch.cuda.synchronize()
    assert ('conv_wino42_fwd/c3' if f42 else 'conv_wino2_fwd/c3') in ops.timer.summary()
    stats = stats.sum(0)
    close(z.permute(0, 3, 1, 2), zr, 2e-5, 2e-5)
    close(stats[:, 0], zr.sum(dim=(0, 2, 3)), 1e-4, 1e-4)
    close(stats[:, 1], (zr ** 2).sum(dim=(0, 2, 3)), 1e-4, 1e-4)
    try:
        ops.USE_WINOGRAD_S2 = False
        z2 = ops.conv_forward(xg, w.to(dev()), b.to(dev()), 4, 2, 1)
    finally:
        ops.USE_WINOGRAD_S2 = True
    close(z, z2, 4e-5, 4e-5)


@pytest.mark.parametrize('case', [(2, 64, 32, 64), (1, 32, 10, 64), (3, 32, 70, 128), (2, 256, 36, 64), (5, 64, 6, 64),
                                  (2, 128, 48, 256)])
def test_conv4x4s2_winograd_wgrad_matches_direct_and_fp64(case):
    """Winograd F(2x2,2x2) weight gradient of the 4x4 / stride 2 / pad 1 layers (image borders, partially filled tile
    groups, q blocks that span two (py, px) classes) against torch fp64 and the direct MFMA weight-gradient kernel."""
    from capsyolo_amd import ops
    B, Cin, H, Cout = case
    x = rnd((B, Cin, H, H), 91)
    w = rnd((Cout, Cin, 4, 4), 92, (1.0 / (Cin * 16)) ** 0.5)
    wd = w.double().requires_grad_(True)
    zr = F.conv2d(x.double(), wd, None, stride=2, padding=1)
    gz = rnd(tuple(zr.shape), 94)
    zr.backward(gz.double())
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev())
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    ops.timer.reset()
    ops.timer.enabled = True
    try:
        dw = ops.conv_wgrad(xg, gzd, 4, 2, 1, False, 'wg')
    finally:
        ops.timer.enabled = False
    torch.cuda.synchronize()
    assert 'conv_wino2_wgrad/wg' in ops.timer.summary()
    scale = wd.grad.abs().max().item()
    close(dw, wd.grad, 2e-5, 2e-5 * scale)
    try:
        ops.USE_WINOGRAD_S2 = False
        dw2 = ops.conv_wgrad(xg, gzd, 4, 2, 1)
    finally:
        ops.USE_WINOGRAD_S2 = True
    close(dw, dw2, 4e-5, 4e-5 * scale)


@pytest.mark.parametrize('case', [(2, 64, 32, 64), (3, 32, 70, 128), (1, 256, 12, 64)])
@pytest.mark.parametrize('f42', [False, True])
def test_conv4x4s2_winograd_fused_input_affine(case, f42, monkeypatch):
    """The 4x4 / stride-2 Winograd forward and weight gradient with the producer's BatchNorm + LeakyReLU applied on
    their loads (X = raw conv output of the previous layer) against torch fp64 on the explicitly activated input;
    the zero padding must stay zero (not lrelu(shift)).  f42: the forward on F(4x4,2x2)."""
    from capsyolo_amd import ops
    monkeypatch.setattr(ops, 'USE_WINOGRAD4_S2', f42)
    monkeypatch.setattr(ops, 'WINOGRAD4_S2_MIN_PIXELS', 0)
    B, Cin, H, Cout = case
    zprev = rnd((B, Cin, H, H), 101)
    sc, sh = rnd((Cin,), 102, 0.5) + 1.0, rnd((Cin,), 103, 0.5)
    w = rnd((Cout, Cin, 4, 4), 104, (1.0 / (Cin * 16)) ** 0.5)
    a = F.leaky_relu(zprev.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), 0.1)
    wd = w.double().requires_grad_(True)
    zr = F.conv2d(a, wd, None, stride=2, padding=1)
    gz = rnd(tuple(zr.shape), 105)
    zr.backward(gz.double())
    zg = zprev.permute(0, 2, 3, 1).contiguous().to(dev())
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    ina = (sc.to(dev()), sh.to(dev()), 0.1)
    z = ops.conv_forward(zg, w.to(dev()), None, 4, 2, 1, False, None, False, 'c', ina)
    close(z.permute(0, 3, 1, 2), zr.detach(), 2e-5, 2e-5)
    dw = ops.conv_wgrad(zg, gzd, 4, 2, 1, False, 'c', ina)
    close(dw, wd.grad, 2e-5, 2e-5 * wd.grad.abs().max().item())


@pytest.mark.parametrize('case', [(2, 64, 16, 64, 4, 2, 1), (3, 32, 18, 128, 4, 2, 1), (2, 64, 9, 64, 1, 1, 0),
                                  (2, 128, 70, 40, 4, 2, 1, 42), (3, 64, 32, 64, 4, 2, 1, 42), (1, 256, 6, 16, 4, 2, 1, 42)])
def test_conv_dgrad_epilogue_bn_backward_sums(case, monkeypatch):
    """The input-gradient kernel's optional BatchNorm-backward sums of the producer block (d = dx * lrelu'(y),
    sum d and sum d * xhat over the pixels it stores, all stride-parity classes together) against cy_bn_bwd_reduce
    run on the same dx.  Cases ending in 42: the F(4x4,2x2) kernel (winograd4_s2.hip), the others F(2x2,2x2) / the direct kernel."""
    from capsyolo_amd import ops
    from capsyolo_amd._lib import call
    monkeypatch.setattr(ops, 'USE_WINOGRAD4_S2_DGRAD', len(case) == 8)
    monkeypatch.setattr(ops, 'WINOGRAD4_S2_MIN_PIXELS', 0)
    B, Cin, Hi, Cout, k, stride, pad = case[:7]
    Ho = (Hi + 2 * pad - k) // stride + 1
    dz = rnd((B, Ho, Ho, Cout), 111).to(dev())
    w = rnd((Cout, Cin, k, k), 112, 0.1).to(dev())
    z = rnd((B, Hi, Hi, Cin), 113).to(dev())
    sc, sh = (rnd((Cin,), 114, 0.3) + 1.0).to(dev()), rnd((Cin,), 115, 0.5).to(dev())
    mu, isd = rnd((Cin,), 116, 0.2).to(dev()), (rnd((Cin,), 117, 0.1).abs() + 0.8).to(dev())
    red = torch.zeros((ops.STATS_COPIES, Cin, 2), dtype=torch.float64, device=dev())
    info = {}
    dx = ops.conv_dgrad(dz, w, (B, Hi, Hi, Cin), k, stride, pad, 'c', (z, sc, sh, mu, isd, 0.1, red), info)
    dx0 = ops.conv_dgrad(dz, w, (B, Hi, Hi, Cin), k, stride, pad)
    if info.get('premasked'):
        # the stride-2 Winograd kernel stores d = dx * lrelu'(y) when it sums (include/capsyolo_hip.h): the producer block's
        # backward then runs with slope 1
        assert k == 4 and Cin % 64 == 0
        y = z * sc + sh
        assert torch.equal(dx, torch.where(y > 0, dx0, dx0 * 0.1))
    else:
        assert torch.equal(dx, dx0)
    ref = torch.zeros((Cin, 2), dtype=torch.float64, device=dev())
    call('cy_bn_bwd_reduce', z.data_ptr(), dx0.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), 0.1,
         ref.data_ptr(), B * Hi * Hi, Cin, torch.cuda.current_stream().cuda_stream)
    close(red.sum(0), ref, 1e-5, 1e-5)


@pytest.mark.parametrize('case', [(2, 64, 32, 64), (1, 64, 10, 8), (3, 128, 70, 40), (2, 256, 36, 64), (5, 64, 6, 16)])
@pytest.mark.parametrize('f42', [False, True])
def test_conv4x4s2_winograd_dgrad_matches_direct_and_fp64(case, f42, monkeypatch):
    """Winograd F(2x2,2x2) / F(4x4,2x2) input gradient of the 4x4 / stride 2 / pad 1 layers (scatter through the space-to-depth
    view, image borders, grid overhang) against torch fp64 and the direct per-parity-class kernel."""
    from capsyolo_amd import ops
    monkeypatch.setattr(ops, 'USE_WINOGRAD4_S2_DGRAD', f42)
    monkeypatch.setattr(ops, 'WINOGRAD4_S2_MIN_PIXELS', 0)
    B, Cin, H, Cout = case
    x = rnd((B, Cin, H, H), 121)
    w = rnd((Cout, Cin, 4, 4), 122, (1.0 / (Cin * 16)) ** 0.5)
    xd = x.double().requires_grad_(True)
    zr = F.conv2d(xd, w.double(), None, stride=2, padding=1)
    gz = rnd(tuple(zr.shape), 124)
    zr.backward(gz.double())
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    ops.timer.reset()
    ops.timer.enabled = True
    try:
        dx = ops.conv_dgrad(gzd, w.to(dev()), (B, H, H, Cin), 4, 2, 1, 'dg')
    finally:
        ops.timer.enabled = False
    torch.cuda.synchronize()
    assert ('conv_wino42_dgrad/dg' if f42 else 'conv_wino2_dgrad/dg') in ops.timer.summary()
    close(dx.permute(0, 3, 1, 2), xd.grad, 2e-5, 2e-5)
    try:
        ops.USE_WINOGRAD_S2_DGRAD = False
        dx2 = ops.conv_dgrad(gzd, w.to(dev()), (B, H, H, Cin), 4, 2, 1)
    finally:
        ops.USE_WINOGRAD_S2_DGRAD = True
    close(dx, dx2, 4e-5, 4e-5)


def test_conv_relu_epilogue_and_stats():
    from capsyolo_amd import ops
    x = rnd((2, 64, 9, 9), 5)
    w = rnd((128, 64, 3, 3), 6, 0.05)
    b = rnd((128,), 7, 0.1)
    zr = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev())
    stats = torch.zeros((ops.STATS_COPIES, 128, 2), dtype=torch.float64, device=dev())
    z = ops.conv_forward(xg, w.to(dev()), b.to(dev()), 3, 1, 1, False, stats, False)
    stats = stats.sum(0)
    close(z.permute(0, 3, 1, 2), zr, 2e-5, 2e-5)
    close(stats[:, 0], zr.sum(dim=(0, 2, 3)), 1e-5, 1e-5)
    close(stats[:, 1], (zr ** 2).sum(dim=(0, 2, 3)), 1e-5, 1e-5)
    z2 = ops.conv_forward(xg, w.to(dev()), b.to(dev()), 3, 1, 1, False, None, True)
    close(z2.permute(0, 3, 1, 2), zr.clamp(min=0), 2e-5, 2e-5)


@pytest.mark.parametrize('train,cin,cout,hw,B', [(True, 32, 64, 10, 3), (False, 32, 64, 10, 3), (True, 64, 128, 32, 4),
                                                 (True, 32, 256, 12, 2), (True, 32, 1024, 6, 2), (True, 32, 32, 16, 5)])
def test_conv_bn_lrelu_block(train, cin, cout, hw, B):
    """Conv -> BatchNorm2d -> LeakyReLU(0.1) block vs torch modules (batch stats, running stats, all grads)."""
    from capsyolo_amd import models
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(cin, cout, 4, 2, 1).double()
    bn = torch.nn.BatchNorm2d(cout).double()
    bn.weight.data = 1 + 0.2 * torch.randn(cout).double()
    bn.bias.data = 0.1 * torch.randn(cout).double()
    bn.running_mean.data = 0.1 * torch.randn(cout).double()
    bn.running_var.data = 1 + 0.2 * torch.rand(cout).double()
    ref = torch.nn.Sequential(conv, bn, torch.nn.LeakyReLU(0.1)).train(train)
    seq = models.FusedBackbone()
    seq.add_module('conv_1', models.HipConv2d(cin, cout, 4, 2, 1))
    seq.add_module('bn_1', models.HipBatchNorm2d(cout))
    seq.add_module('relu_1', models.HipLeakyReLU(0.1))
    seq.conv_1.load_state_dict({k: v.float() for k, v in conv.state_dict().items()})
    seq.bn_1.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    seq.to(dev()).train(train)
    x = rnd((B, cin, hw, hw), 8)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev()).requires_grad_(True)
    yh = seq(xh, nchw_in=False)
    close(yh.permute(0, 3, 1, 2), yr, 1e-4, 1e-5)
    close(seq.bn_1.running_mean, bn.running_mean, 1e-4, 1e-6)
    close(seq.bn_1.running_var, bn.running_var, 1e-4, 1e-6)
    if train:
        g = rnd(tuple(yr.shape), 9)
        yr.backward(g.double())
        yh.backward(g.permute(0, 2, 3, 1).contiguous().to(dev()))
        close(xh.grad.permute(0, 3, 1, 2), xr.grad, 1e-3, 1e-4)
        close(seq.conv_1.weight.grad, conv.weight.grad, 1e-3, 1e-4)
        close(seq.bn_1.weight.grad, bn.weight.grad, 1e-3, 1e-4)
        close(seq.bn_1.bias.grad, bn.bias.grad, 1e-3, 1e-4)
        assert float(seq.conv_1.bias.grad.abs().max()) == 0.0       # analytically zero in front of BN


@pytest.mark.parametrize('onepass', [False, True])
@pytest.mark.parametrize('kind,edge', [('smooth', False), ('smooth', True)])
def test_first_block_on_smooth_bright_images_with_zero_sum_filters(kind, edge, onepass):
    """the patch-moment statistics (w^T M2 w - mean^2) and the one-pass backward subtract large sums; iid-noise
    images have a near-diagonal moment matrix and hide that.  Blurred noise around +0.7 (a bright, spatially correlated image,
    quantised to k / 128 like the data sets) with zero-sum (edge) filters makes var(z) a difference of numbers three to four
    digits larger.  The moment matrix is accumulated EXACTLY for such images (conv1_moments.hip: products are multiples of
    2^-14, fp32 chains flushed to double every 960 terms), so the batch variance must come out at fp32 resolution of the result
    (measured 5.7e-8 per channel, better than the recompute path's 2e-7 .. 8e-7), and the block's gradients at 64 x 64 as close
    to torch fp64 as the two-pass path's (measured 1.4e-6 / 1.5e-6 / 9e-8 for dW / dgamma / dbeta)."""
    import importlib.util
    import os
    from helpers import REPO
    from capsyolo_amd import ops
    spec = importlib.util.spec_from_file_location('cy_diag_conv1', os.path.join(REPO, 'tools', 'diag_conv1_smooth.py'))
    diag = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(diag)
    old = ops.CONV1_MOMENTS_MIN_PIXELS
    try:
        r = diag.run(kind, 128, 4, 64, 64, onepass, edge)
    finally:
        ops.CONV1_MOMENTS_MIN_PIXELS = old
    print(kind, edge, 'onepass' if onepass else 'twopass', r)
    assert r['out'] <= 2e-6
    assert r['var_elem'] <= (3e-7 if onepass else 3e-6)
    assert r['dW'] <= 2e-5 and r['dgamma'] <= 2e-5 and r['dbeta'] <= 2e-5


@pytest.mark.parametrize('cin,cout,k,stride,pad,hw,B,nchw,f4', [
    (128, 256, 3, 1, 1, 20, 2, False, False),     # fused Winograd F(2x2,3x3) forward, LeakyReLU epilogue
    (64, 128, 3, 1, 1, 13, 3, False, False),      # ... odd size
    (128, 256, 3, 1, 1, 20, 2, False, True),      # fused Winograd F(4x4,3x3) forward, LeakyReLU epilogue
    (64, 72, 3, 1, 1, 37, 3, False, True),        # ... odd size, padded channels, partial tile blocks
    (256, 64, 4, 2, 1, 24, 2, False, False),      # fused Winograd F(2x2,2x2) (4x4 / stride 2)
    (256, 64, 4, 2, 1, 24, 2, False, 42),         # fused Winograd F(4x4,2x2) forward, LeakyReLU epilogue
    (64, 72, 4, 2, 1, 38, 3, False, 42),          # ... odd map, padded channels
    (64, 128, 4, 2, 1, 10, 3, False, False),
    (128, 64, 1, 1, 0, 9, 2, False, False),       # implicit GEMM (DarkNet's 1x1 layers), act = 2
    (16, 40, 3, 1, 1, 11, 2, False, False),       # implicit GEMM, padded channels
    (3, 128, 3, 1, 1, 32, 2, True, False),        # the first layer: scale / shift in its own epilogue
])
def test_eval_block_batchnorm_folded_into_the_conv(cin, cout, k, stride, pad, hw, B, nchw, f4, monkeypatch):
    """Eval-mode conv -> BatchNorm -> LeakyReLU(0.1) as ONE launch (cy_bn_fold_eval + the conv kernel's LeakyReLU epilogue),
    every forward kernel class, against torch fp64 modules in eval mode."""
    from capsyolo_amd import _lib, models, ops
    monkeypatch.setattr(ops, 'USE_WINOGRAD4', f4 is True)
    monkeypatch.setattr(ops, 'WINOGRAD4_MIN_PIXELS', 0)
    monkeypatch.setattr(ops, 'USE_WINOGRAD4_S2', f4 == 42)
    monkeypatch.setattr(ops, 'WINOGRAD4_S2_MIN_PIXELS', 0)
    torch.manual_seed(11)
    conv = torch.nn.Conv2d(cin, cout, k, stride, pad).double()
    bn = torch.nn.BatchNorm2d(cout).double()
    bn.weight.data = (1 + 0.3 * torch.randn(cout)).double()
    bn.bias.data = (0.2 * torch.randn(cout)).double()
    bn.running_mean.data = (0.3 * torch.randn(cout)).double()
    bn.running_var.data = (0.5 + torch.rand(cout)).double()
    ref = torch.nn.Sequential(conv, bn, torch.nn.LeakyReLU(0.1)).eval()
    seq = models.FusedBackbone()
    seq.add_module('conv_1', models.HipConv2d(cin, cout, k, stride, pad))
    seq.add_module('bn_1', models.HipBatchNorm2d(cout))
    seq.add_module('relu_1', models.HipLeakyReLU(0.1))
    seq.conv_1.load_state_dict({n: v.float() for n, v in conv.state_dict().items()})
    seq.bn_1.load_state_dict({n: (v.float() if v.is_floating_point() else v) for n, v in bn.state_dict().items()})
    seq.to(dev()).eval()
    x = rnd((B, cin, hw, hw), 18)
    with torch.no_grad():
        yr = ref(x.double())
        xg = (x if nchw else x.permute(0, 2, 3, 1).contiguous()).to(dev())
        _lib.TRACE = []
        try:
            yh = seq(xg, nchw_in=nchw)
            torch.cuda.synchronize()
            calls = list(_lib.TRACE)
        finally:
            _lib.TRACE = None
    assert 'cy_affine_act' not in calls, calls
    assert ('cy_bn_fold_eval' in calls) != nchw
    assert ('cy_conv3x3_winograd4' in calls) == (f4 is True) and ('cy_conv4x4s2_winograd4' in calls) == (f4 == 42)
    close(yh.permute(0, 3, 1, 2), yr, 1e-4, 2e-5)


@pytest.mark.parametrize('onepass', [False, True])
@pytest.mark.parametrize('cout,H,W,B', [(128, 32, 32, 2), (32, 33, 64, 3), (64, 6, 96, 1)])
def test_first_block_fused_backward(cout, H, W, B, onepass):
    """Image -> Conv(3, cout, 3, 1, 1) -> BatchNorm2d -> LeakyReLU(0.1): the first block's fused paths (second forward pass
    with the activation, backward passes that recompute z: cy_conv1_bn_bwd_reduce / _wgrad) vs torch fp64 modules and
    vs the generic kernels.  onepass: the statistics from the patch moments (cy_conv1_3x3_stats) and the ONE-pass backward built on
    them (cy_conv1_bn_bwd_onepass), which large inputs take by default."""
    from capsyolo_amd import models, ops
    min_pixels = ops.CONV1_MOMENTS_MIN_PIXELS
    ops.CONV1_MOMENTS_MIN_PIXELS = 0 if onepass else (1 << 62)
    try:
        _first_block_fused_backward(cout, H, W, B, onepass)
    finally:
        ops.CONV1_MOMENTS_MIN_PIXELS = min_pixels


def _first_block_fused_backward(cout, H, W, B, onepass):
    from capsyolo_amd import models, ops
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(3, cout, 3, 1, 1).double()
    bn = torch.nn.BatchNorm2d(cout).double()
    bn.weight.data = 1 + 0.2 * torch.randn(cout).double()
    bn.bias.data = 0.1 * torch.randn(cout).double()
    ref = torch.nn.Sequential(conv, bn, torch.nn.LeakyReLU(0.1)).train()
    x = rnd((B, 3, H, W), 141)
    g = rnd((B, cout, H, W), 142)
    yr = ref(x.double())
    yr.backward(g.double())

    def run():
        seq = models.FusedBackbone()
        seq.add_module('conv_1', models.HipConv2d(3, cout, 3, 1, 1))
        seq.add_module('bn_1', models.HipBatchNorm2d(cout))
        seq.add_module('relu_1', models.HipLeakyReLU(0.1))
        seq.conv_1.load_state_dict({k: v.float() for k, v in conv.state_dict().items()})
        sd = torch.nn.BatchNorm2d(cout).state_dict()
        sd['weight'], sd['bias'] = bn.weight.data.float(), bn.bias.data.float()
        seq.bn_1.load_state_dict(sd)
        seq.to(dev()).train()
        ops.timer.reset(); ops.timer.enabled = True
        try:
            yh = seq(x.to(dev()), nchw_in=True)
            yh.backward(g.permute(0, 2, 3, 1).contiguous().to(dev()))
            torch.cuda.synchronize()
            keys = set(ops.timer.summary())
        finally:
            ops.timer.enabled = False
        return seq, yh, keys
    seq, yh, keys = run()
    assert any(k.startswith('conv1_bn_bwd_onepass/' if onepass else 'conv1_bn_bwd_wgrad/') for k in keys)
    assert any(k.startswith('conv1_fwd_act/') for k in keys)
    close(yh.permute(0, 3, 1, 2), yr, 1e-4, 1e-5)
    close(seq.bn_1.running_mean, bn.running_mean, 1e-4, 1e-6)
    close(seq.conv_1.weight.grad, conv.weight.grad, 1e-3, 1e-4)
    close(seq.bn_1.weight.grad, bn.weight.grad, 1e-3, 1e-4)
    close(seq.bn_1.bias.grad, bn.bias.grad, 1e-3, 1e-4)
    assert float(seq.conv_1.bias.grad.abs().max()) == 0.0
    ops.USE_CONV1_BWD = ops.USE_CONV1 = False
    try:
        seq0, yh0, keys0 = run()
    finally:
        ops.USE_CONV1_BWD = ops.USE_CONV1 = True
    assert not any(k.startswith('conv1_') for k in keys0)
    close(yh, yh0, 1e-5, 1e-5)
    close(seq.conv_1.weight.grad, seq0.conv_1.weight.grad, 1e-4, 1e-4)
    close(seq.bn_1.weight.grad, seq0.bn_1.weight.grad, 1e-4, 1e-4)
    close(seq.bn_1.bias.grad, seq0.bn_1.bias.grad, 1e-4, 1e-4)


# ------------------------------------------------------------------------------------------------ routing
@pytest.mark.parametrize('ci', [0, 1, 2, 3])
@pytest.mark.parametrize('n_iter', [1, 3, 5])
def test_routing_golden(ci, n_iter):
    """HIP routing vs the reference's own outputs/gradients (tests/golden/routing.npz)."""
    from capsyolo_amd import ops
    from helpers import grad_digest
    g = load_golden('routing')
    R, N, C, Din, Dout = (int(v) for v in g['c%d_shape' % ci])
    u, W, G = routing_case(ci, R, N, C, Din, Dout)
    ut = T(u).to(dev()).requires_grad_(True)
    Wt = T(W).to(dev()).requires_grad_(True)
    v = ops.routing(ut, Wt, n_iter)
    (v * T(G).to(dev())).sum().backward()
    key = 'c%d_r%d_' % (ci, n_iter)
    close(v, g[key + 'v'], 1e-4, 1e-5)
    close(ut.grad, g[key + 'du'], 1e-3, 1e-4)
    dig = grad_digest(Wt.grad.cpu())
    close(dig, g[key + 'dW_digest'], 1e-3, 1e-4)


@pytest.mark.parametrize('shape', [(37, 70, 43, 8, 16, 3), (9, 33, 7, 8, 21, 2), (130, 512, 1, 8, 5, 3),
                                   (5, 64, 64, 8, 16, 3), (3, 20, 3, 8, 5, 4),
                                   # many rows: the single-launch fused kernels (few rows take the phased path)
                                   (1100, 12, 5, 8, 16, 3), (600, 10, 7, 8, 21, 2), (1030, 9, 3, 8, 5, 3),
                                   # fused plans hand c^t, db^t from the row part to the du / dW kernel (up to 4 iterations t >= 1; 6 iterations: recomputed)
                                   (1100, 20, 43, 8, 21, 4), (1100, 16, 20, 8, 16, 5), (1050, 8, 6, 8, 21, 6), (1040, 10, 49, 8, 48, 2),
                                   (32, 1296, 43, 8, 16, 3),
                                   # DarkCapsuleNet2-like heads (Dout = 5 + 43 = 48): one j per lane / 16-lane rows with 1 and 2 capsules per lane
                                   (6, 40, 49, 8, 48, 3), (5, 30, 4, 8, 48, 2), (4, 24, 20, 8, 48, 3),
                                   # several row tiles x several chunks of input capsules
                                   (70, 300, 43, 8, 21, 3), (200, 64, 33, 8, 16, 3)])
def test_routing_vs_oracle(shape):
    from capsyolo_amd import ops
    from oracle.models import dynamic_routing
    R, N, C, Din, Dout, n_iter = shape
    u, W, G = rnd((R, N, Din), 11, 0.8), rnd((1, N, C, Din, Dout), 12, 0.15), rnd((R, C, Dout), 13)
    ur, Wr = u.double().requires_grad_(True), W.double().requires_grad_(True)
    vr = dynamic_routing(ur, Wr, n_iter)
    (vr * G.double()).sum().backward()
    ut, Wt = u.to(dev()).requires_grad_(True), W.to(dev()).requires_grad_(True)
    v = ops.routing(ut, Wt, n_iter)
    (v * G.to(dev())).sum().backward()
    close(v, vr, 1e-4, 1e-5)
    close(ut.grad, ur.grad, 1e-3, 1e-4)
    close(Wt.grad, Wr.grad, 1e-3, 1e-4)


@pytest.mark.parametrize('shape', [(37, 70, 43, 8, 16, 3), (9, 33, 7, 8, 21, 2), (1100, 12, 5, 8, 16, 3), (600, 10, 7, 8, 21, 2),
                                   (1100, 20, 43, 8, 21, 4), (1100, 16, 20, 8, 16, 5), (32, 1296, 43, 8, 16, 3), (70, 300, 43, 8, 21, 3),
                                   (200, 64, 33, 8, 16, 3), (1000, 40, 48, 8, 21, 3)])
def test_routing_forward_on_mfma_vs_oracle(shape, monkeypatch):
    """The opt-in forward with u_hat = u W on v_mfma_f32_16x16x4_f32 (csrc/routing_mfma.hip, CY_ROUTING_MFMA=1: 16-row tiles,
    capsules along the MFMA's N, output components split over the block's four waves, partial logits exchanged through LDS),
    fused (many rows) and phased (few rows) plans, against the fp64 oracle at the tolerance of the vector kernel; the backward
    (vector kernels) runs on the s_hist this forward leaves."""
    from capsyolo_amd import ops
    from oracle.models import dynamic_routing
    monkeypatch.setenv('CY_ROUTING_MFMA', '1')
    R, N, C, Din, Dout, n_iter = shape
    u, W, G = rnd((R, N, Din), 11, 0.8), rnd((1, N, C, Din, Dout), 12, 0.15), rnd((R, C, Dout), 13)
    ur, Wr = u.double().requires_grad_(True), W.double().requires_grad_(True)
    vr = dynamic_routing(ur, Wr, n_iter)
    (vr * G.double()).sum().backward()
    ut, Wt = u.to(dev()).requires_grad_(True), W.to(dev()).requires_grad_(True)
    v = ops.routing(ut, Wt, n_iter)
    (v * G.to(dev())).sum().backward()
    close(v, vr, 1e-4, 1e-5)
    close(ut.grad, ur.grad, 1e-3, 1e-4)
    close(Wt.grad, Wr.grad, 1e-3, 1e-4)
    monkeypatch.setenv('CY_ROUTING_MFMA', '0')
    v0 = ops.routing(u.to(dev()), W.to(dev()), n_iter)
    close(v.detach(), v0, 2e-5, 2e-6)                 # the two forward kernels agree far inside the oracle tolerance


@pytest.mark.parametrize('C,Dout,n_iter', [(1, 5, 3), (3, 21, 3)])
def test_routing_cell_gather(C, Dout, n_iter):
    """Gather fused into the routing loads == oracle cell_gather + routing (DarkCapsuleNet / DarkCapsuleNet3 heads)."""
    from capsyolo_amd import ops
    from oracle.models import cell_gather, dynamic_routing
    g, B = 3, 5
    feat = rnd((B, 256, 4 * g, 4 * g), 21, 0.7)
    W = rnd((1, 512, C, 8, Dout), 22, 0.1)
    fr, Wr = feat.double().requires_grad_(True), W.double().requires_grad_(True)
    vr = dynamic_routing(cell_gather(fr, g), Wr, n_iter)                   # [g*g*B, C, Dout]
    vr = vr.reshape(g, g, B, C, Dout).permute(2, 0, 1, 3, 4)
    G = rnd(tuple(vr.shape), 23)
    (vr * G.double()).sum().backward()
    fh = feat.permute(0, 2, 3, 1).contiguous().to(dev()).requires_grad_(True)
    Wh = W.to(dev()).requires_grad_(True)
    v = ops.routing(fh, Wh, n_iter, g, B)
    (v * G.to(dev())).sum().backward()
    close(v, vr, 1e-4, 1e-5)
    close(fh.grad.permute(0, 3, 1, 2), fr.grad, 1e-3, 1e-4)
    close(Wh.grad, Wr.grad, 1e-3, 1e-4)


def test_routing_singleton_is_iteration_independent():
    from capsyolo_amd import ops
    u, W = rnd((40, 512, 8), 31).to(dev()), rnd((1, 512, 1, 8, 5), 32, 0.1).to(dev())
    outs = [ops.routing(u, W, r) for r in (1, 3, 5)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


def test_squash_and_length():
    from capsyolo_amd import ops
    from oracle.models import squash
    g = load_golden('squash')
    x = T(g['v']).to(dev()).requires_grad_(True)
    y = ops.squash(x)
    close(y, g['out'], 2e-5, 1e-6)
    xr = T(g['v']).double().requires_grad_(True)
    G = rnd(tuple(xr.shape), 41)
    (squash(xr) * G.double()).sum().backward()
    (y * G.to(dev())).sum().backward()
    close(x.grad, xr.grad, 1e-4, 1e-5)
    x2 = T(g['v']).to(dev()).requires_grad_(True)
    x2r = T(g['v']).double().requires_grad_(True)
    G2 = rnd(tuple(xr.shape[:-1]), 42)
    (ops.length(x2) * G2.to(dev())).sum().backward()
    (((x2r ** 2).sum(-1) ** 0.5) * G2.double()).sum().backward()
    close(x2.grad, x2r.grad, 1e-4, 1e-5)
    assert torch.isnan(ops.squash(torch.zeros(1, 4, device=dev()))).all()


# ------------------------------------------------------------------------------------------------ losses
def test_losses_golden():
    from capsyolo_amd import loss_fns
    g = load_golden('losses')
    p = make_params(recon=False, device='cuda')
    ct = T(g['dc_caps']).to(dev()).requires_grad_(True)
    l = loss_fns.darkcapsule_loss(ct, T(g['dc_y']).to(dev()), p)
    (l * 1.0).backward()
    close(l, g['dc_loss'], 2e-5)
    close(ct.grad, g['dc_dcaps'], 1e-4, 1e-5)
    xim = T(wave((5, 3, 32, 32), 0.1, amp=0.9, freq=0.211)).to(dev())
    for recon in (False, True):
        pc = make_params(recon=recon, device='cuda')
        st = T(g['cap_scores']).to(dev()).requires_grad_(True)
        rt = T(wave((5, 3, 32, 32), 2.1, amp=0.9, freq=0.173)).to(dev()).requires_grad_(True)
        l = loss_fns.capsule_loss(st, T(g['cap_y']).to(dev()), pc, xim, rt)
        (l * 2.0).backward()                                         # upstream gradient != 1
        tag = 'cap_recon%d_' % int(recon)
        close(l, g[tag + 'loss'], 2e-5)
        close(st.grad, 2.0 * g[tag + 'dscores'], 1e-4, 1e-5)
    for tag in ('dk_d', 'dk_r'):
        nb, C, inp, gd = (int(v) for v in g[tag + '_cfg'])
        pd = make_params(n_boxes=nb, n_classes=C, darknet_input=inp, n_grid=gd, device='cuda')
        pt = T(g[tag + '_pred']).float().to(dev()).requires_grad_(True)
        l = loss_fns.dark_loss(pt, T(g[tag + '_y']).to(dev()), pd)
        l.backward()
        close(l, g[tag + '_loss'], 5e-5)
        close(pd.avg_iou, g[tag + '_avg_iou'], 5e-5)
        close(pt.grad, g[tag + '_dpred'], 2e-4, 1e-5)
    # darkcapsule2_loss / darkcapsule3_loss (loss_fns.py:145-184) on their own kernels, vs the reference's values
    from helpers import grad_digest, synth_gtsdb_labels
    y = T(synth_gtsdb_labels(3, 4, 43, seed=7)).to(dev())
    c2 = T(wave((3, 4, 4, 48), 1.8, amp=0.3, freq=0.477)).to(dev()).requires_grad_(True)
    l2 = loss_fns.darkcapsule2_loss(c2, y, p)
    (l2 * 1.5).backward()
    close(l2, g['dc2_loss'], 2e-5)
    close(c2.grad, 1.5 * g['dc2_dcaps'], 1e-4, 1e-5)
    c3 = T(wave((3, 4, 4, 43, 21), 0.8, amp=0.3, freq=0.377)).to(dev()).requires_grad_(True)
    l3 = loss_fns.darkcapsule3_loss(c3, y, p)
    l3.backward()
    close(l3, g['dc3_loss'], 2e-5)
    close(grad_digest(c3.grad.cpu()), g['dc3_dcaps_digest'], 1e-4, 1e-5)


def test_adam_matches_torch():
    from capsyolo_amd import optim
    torch.manual_seed(0)
    shapes = [(1000,), (33, 7), (5, 3, 3, 3), (70000,)]
    ps_ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    ps_hip = [torch.nn.Parameter(p.detach().clone().to(dev())) for p in ps_ref]
    o_ref, o_hip = torch.optim.Adam(ps_ref, lr=1e-2), optim.Adam(ps_hip, lr=1e-2)
    for step in range(5):
        for pr, ph in zip(ps_ref, ps_hip):
            gr = torch.randn(pr.shape, generator=torch.Generator().manual_seed(step * 10 + pr.numel() % 7))
            pr.grad, ph.grad = gr.clone(), gr.clone().to(dev())
        o_ref.step()
        o_hip.step()
    for pr, ph in zip(ps_ref, ps_hip):
        close(ph, pr, 1e-5, 1e-6)


def test_misc_ops():
    from capsyolo_amd import ops
    x = rnd((2, 6, 8, 8), 51)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev()).requires_grad_(True)
    xr = x.double().requires_grad_(True)
    G = rnd((2, 6, 4, 4), 52)
    (F.max_pool2d(xr, 2) * G.double()).sum().backward()
    y = ops.maxpool2(xh)
    (y * G.permute(0, 2, 3, 1).contiguous().to(dev())).sum().backward()
    close(y.permute(0, 3, 1, 2), F.max_pool2d(x, 2), 0, 0)
    close(xh.grad.permute(0, 3, 1, 2), xr.grad, 1e-6)
    # ... and with a multiple of 4 channels (four channels per thread), incl. a window of equal values (the first position wins)
    x4 = rnd((3, 32, 10, 12), 54)
    x4[0, :, :2, :2] = 0.5
    x4h = x4.permute(0, 2, 3, 1).contiguous().to(dev()).requires_grad_(True)
    x4r = x4.double().requires_grad_(True)
    G4 = rnd((3, 32, 5, 6), 55)
    (F.max_pool2d(x4r, 2) * G4.double()).sum().backward()
    y4 = ops.maxpool2(x4h)
    (y4 * G4.permute(0, 2, 3, 1).contiguous().to(dev())).sum().backward()
    close(y4.permute(0, 3, 1, 2), F.max_pool2d(x4, 2), 0, 0)
    close(x4h.grad.permute(0, 3, 1, 2), x4r.grad, 1e-6)
    # upsample
    xh2 = x.permute(0, 2, 3, 1).contiguous().to(dev()).requires_grad_(True)
    xr2 = x.double().requires_grad_(True)
    G2 = rnd((2, 6, 16, 16), 53)
    (F.interpolate(xr2, scale_factor=2, mode='nearest') * G2.double()).sum().backward()
    y2 = ops.upsample_nearest(xh2, 2)
    (y2 * G2.permute(0, 2, 3, 1).contiguous().to(dev())).sum().backward()
    close(y2.permute(0, 3, 1, 2), F.interpolate(x, scale_factor=2, mode='nearest'), 0, 0)
    close(xh2.grad.permute(0, 3, 1, 2), xr2.grad, 1e-5, 1e-6)
    # layout permutes round trip
    close(ops.nhwc_to_nchw(ops.nchw_to_nhwc(x.to(dev()))), x, 0, 0)
    close(ops.nchw_to_nhwc(x.to(dev())), x.permute(0, 2, 3, 1), 0, 0)
    # yolo head
    h = rnd((3, 4, 4, 13), 54)
    hr = h.double().requires_grad_(True)
    ref = torch.cat((torch.sigmoid(hr[..., :10]), F.softmax(hr[..., 10:], dim=-1)), dim=-1)
    G3 = rnd((3, 4, 4, 13), 55)
    (ref * G3.double()).sum().backward()
    hh = h.to(dev()).requires_grad_(True)
    yh = ops.yolo_head(hh, 10, 3)
    (yh * G3.to(dev())).sum().backward()
    close(yh, ref, 1e-5, 1e-6)
    close(hh.grad, hr.grad, 1e-4, 1e-5)


def test_center_u8_and_device_feeder_match_reference_expression():
    """cy_center_u8 and the double-buffered feeder give exactly torch.from_numpy(x).float().permute(0,3,1,2)
    (main.py:57-59) for uint8-representable data (float32 and float64 storage) and for augmented floats."""
    from capsyolo_amd import _lib
    from capsyolo_amd.input_pipeline import DeviceFeeder, quantize_if_exact
    rng = np.random.default_rng(11)
    u = rng.integers(0, 256, (5, 12, 10, 3), dtype=np.uint8)
    ud = torch.from_numpy(u).to(dev())
    for to_nchw in (0, 1):
        out = torch.empty((5, 3, 12, 10) if to_nchw else (5, 12, 10, 3), device=dev())
        _lib.call('cy_center_u8', ud.data_ptr(), out.data_ptr(), 5, 12, 10, 3, to_nchw, torch.cuda.current_stream().cuda_stream)
        ref = (torch.from_numpy(u).float() - 128.0) / 128.0
        assert torch.equal(out.cpu(), ref.permute(0, 3, 1, 2).contiguous() if to_nchw else ref)
    x64 = (u.astype(np.float64) - 128.0) / 128.0
    xaug = x64.astype(np.float32) * np.float32(1.03)
    y = rng.random((5, 2, 2, 48))
    assert np.array_equal(quantize_if_exact(x64), u) and quantize_if_exact(xaug) is None
    for x in (x64, x64.astype(np.float32), xaug, u):
        splits = list(zip(np.array_split(x, 3), np.array_split(y, 3)))
        got = [(a.clone(), b.clone()) for a, b in DeviceFeeder(splits, 'cuda')]
        assert len(got) == 3
        for (xa, ya), (xb, yb) in zip(got, splits):
            xf = (xb.astype(np.float32) - 128.0) / 128.0 if xb.dtype == np.uint8 else xb
            want = torch.from_numpy(np.ascontiguousarray(xf)).float().permute(0, 3, 1, 2).contiguous()
            assert xa.dtype == torch.float32 and torch.equal(xa.cpu(), want)
            assert torch.equal(ya.cpu(), torch.from_numpy(yb))


@pytest.mark.parametrize('tag', ['det', 'rec', 'sq', 'none'])
def test_y_to_boxes_vec_device_matches_reference(tag):
    """cy_yolo_decode_boxes against the reference's utils.y_to_boxes_vec outputs (order, pixels in double, classes)."""
    import types
    from capsyolo_amd import utils
    from helpers import load_golden
    gold = load_golden('boxes')
    seed, B, g, nb, C, use_hw, th = [int(v) for v in gold[tag + '_cfg']]
    rng = np.random.default_rng(seed)
    y = rng.random((B, g, g, 5 * nb + C)).astype(np.float32)
    y[..., 0:5 * nb:5] *= 0.7
    hw = rng.integers(200, 900, (B, 2)).astype(np.int64)
    if tag == 'none':
        y[..., 0::5] = 0.1
    p = types.SimpleNamespace(n_classes=C, darknet_input=416)
    idx, xy, cls = utils.y_to_boxes_vec(y, p, image_hw=hw if use_hw else None, conf_th=th / 1000.0)
    assert np.array_equal(idx, gold[tag + '_idx'])
    np.testing.assert_allclose(xy, gold[tag + '_xy'].reshape(-1, 4), rtol=1e-14, atol=1e-11)
    if C:
        assert np.array_equal(cls, gold[tag + '_cls'])
    else:
        assert cls is None


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_detect_acc_device_matches_reference(tag):
    """Device-side decode + IoU matching (cy_yolo_decode_boxes, cy_detect_confusion) against the reference's
    metrics.detect_acc and its TP / FP / FN."""
    import types
    from capsyolo_amd import metrics
    from helpers import load_golden
    from test_oracle_golden import _detect_case
    gold = load_golden('metrics')
    seed, B, g, nb = [int(v) for v in gold[tag + '_cfg']]
    y, y_hat = _detect_case(seed, B, g, nb)
    p = types.SimpleNamespace(n_classes=0, darknet_input=416)
    assert list(metrics.detect_confusion(y, y_hat, p)) == [int(v) for v in gold[tag + '_tpfpfn']]
    assert abs(metrics.detect_acc(y, y_hat, p) - float(gold[tag + '_f1'])) < 1e-15
    if tag + '_ap' in gold.files:
        assert abs(metrics.detect_AP(y, y_hat, p) - float(gold[tag + '_ap'])) < 1e-12
    bad = y_hat.copy()
    bad[0, 0, 0, 0], bad[0, 0, 0, 3] = 0.9, -0.2              # negative width: x1 > x2
    with pytest.raises(AssertionError):
        metrics.detect_confusion(y, bad, p)


@pytest.mark.parametrize('tag', ['ra', 'rb'])
def test_detect_and_recog_acc_device_matches_reference(tag):
    """Per-(image, class) matching on the device against the reference's metrics.detect_and_recog_acc (metrics.py:264-282)
    and the numpy oracle's TP / FP / FN."""
    import types
    from capsyolo_amd import metrics
    from helpers import load_golden
    from oracle import utils_np
    from test_oracle_golden import _detect_recog_case
    gold = load_golden('metrics')
    seed, B, g, nb, C = [int(v) for v in gold[tag + '_cfg']]
    y, y_hat = _detect_recog_case(seed, B, g, nb, C)
    p = types.SimpleNamespace(n_classes=C, darknet_input=416)
    assert list(metrics.detect_and_recog_confusion(y, y_hat, p)) == [int(v) for v in utils_np.detect_and_recog_confusion(y, y_hat, C, 416)]
    assert abs(metrics.detect_and_recog_acc(y, y_hat, p) - float(gold[tag + '_f1'])) < 1e-15


@pytest.mark.parametrize('case', [(2, 32, 32, 128), (3, 64, 96, 32), (1, 5, 32, 64), (2, 33, 64, 128)])
def test_first_layer_kernel_matches_direct_and_fp64(case):
    """cy_conv1_3x3_fwd (3 input channels, NCHW image, persistent waves) against torch fp64 and the implicit-GEMM kernel,
    incl. bias and BatchNorm statistics, image borders, one-segment rows."""
    from capsyolo_amd import ops
    B, H, W, Cout = case
    x = rnd((B, 3, H, W), 131)
    w = rnd((Cout, 3, 3, 3), 132, 0.2)
    b = rnd((Cout,), 133)
    zr = F.conv2d(x.double(), w.double(), b.double(), stride=1, padding=1)
    st = torch.zeros((ops.STATS_COPIES, Cout, 2), dtype=torch.float64, device=dev())
    ops.timer.reset(); ops.timer.enabled = True
    try:
        z = ops.conv_forward(x.to(dev()), w.to(dev()), b.to(dev()), 3, 1, 1, True, st, False, 'c1')
        torch.cuda.synchronize()
        assert any(k.startswith('conv1_fwd/') for k in ops.timer.summary())
    finally:
        ops.timer.enabled = False
    close(z.permute(0, 3, 1, 2), zr, 2e-5, 2e-5)
    s = st.sum(0).cpu()
    close(s[:, 0], zr.sum((0, 2, 3)), 1e-5, 1e-5)
    close(s[:, 1], (zr * zr).sum((0, 2, 3)), 1e-5, 1e-5)
    ops.USE_CONV1 = False
    try:
        z0 = ops.conv_forward(x.to(dev()), w.to(dev()), b.to(dev()), 3, 1, 1, True, None, False, 'c1')
    finally:
        ops.USE_CONV1 = True
    close(z, z0, 1e-5, 1e-5)
    # weight gradient (cy_conv1_3x3_wgrad: pixels are the MFMA k dimension, per-wave slabs)
    xd, wd = x.double(), w.double().requires_grad_(True)
    gz = rnd(tuple(zr.shape), 134)
    F.conv2d(xd, wd, None, stride=1, padding=1).backward(gz.double())
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    ops.timer.reset(); ops.timer.enabled = True
    try:
        dw = ops.conv_wgrad(x.to(dev()), gzd, 3, 1, 1, True, 'c1')
        torch.cuda.synchronize()
        assert any(k.startswith('conv1_wgrad/') for k in ops.timer.summary())
    finally:
        ops.timer.enabled = False
    close(dw, wd.grad, 2e-5, 2e-5)
    ops.USE_CONV1 = False
    try:
        dw0 = ops.conv_wgrad(x.to(dev()), gzd, 3, 1, 1, True, 'c1')
    finally:
        ops.USE_CONV1 = True
    close(dw, dw0, 2e-5, 2e-5)


# ------------------------------------------------------------------------ full size (BASELINE configs[2]) properties
def _dot64(a, b, chunk=1 << 26):
    """sum(a * b) accumulated in float64, in chunks (the tensors are several GB)."""
    a, b = a.reshape(-1), b.reshape(-1)
    s = torch.zeros((), dtype=torch.float64, device=a.device)
    for i in range(0, a.numel(), chunk):
        s += torch.dot(a[i:i + chunk].double(), b[i:i + chunk].double())
    return s.item()


@pytest.mark.parametrize('layer', ['conv_2', 'conv_3'])
def test_full_size_conv_sampled_outputs_and_adjoint_identities(layer):
    """BASELINE configs[2] shapes (batch 32, 416x416): the oracle cannot run them in seconds, so the forward result is
    checked on 256 sampled outputs against their exact fp64 dot products (incl. image corners and edges), and the two
    gradients through the adjoint identities <conv(x), dz> = <x, dgrad(dz)> = <w, wgrad(x, dz)> over ALL elements."""
    from capsyolo_amd import ops
    B, H = 32, 416
    Cin, Cout, k, s, p = (128, 256, 3, 1, 1) if layer == 'conv_2' else (256, 64, 4, 2, 1)
    Ho = (H + 2 * p - k) // s + 1
    g = torch.Generator(device='cuda').manual_seed(7 if layer == 'conv_2' else 8)
    x = torch.randn((B, H, H, Cin), generator=g, device=dev())
    w = torch.randn((Cout, Cin, k, k), generator=g, device=dev()) * (1.0 / (Cin * k * k)) ** 0.5
    b = torch.randn((Cout,), generator=g, device=dev())
    y = ops.conv_forward(x, w, b, k, s, p)
    assert tuple(y.shape) == (B, Ho, Ho, Cout)
    # ---- sampled outputs: corners, edges and random interior points
    rs = np.random.RandomState(3)
    pts = [(0, 0, 0, 0), (B - 1, Ho - 1, Ho - 1, Cout - 1), (1, 0, Ho - 1, 5), (2, Ho - 1, 0, 7), (3, 0, Ho // 2, 1),
           (4, Ho // 2, Ho - 1, 2)]
    pts += [(int(rs.randint(B)), int(rs.randint(Ho)), int(rs.randint(Ho)), int(rs.randint(Cout))) for _ in range(250)]
    wd = w.double()
    for (bi, oy, ox, co) in pts:
        acc = b[co].double()
        for kh in range(k):
            iy = oy * s - p + kh
            if iy < 0 or iy >= H:
                continue
            for kw in range(k):
                ix = ox * s - p + kw
                if ix < 0 or ix >= H:
                    continue
                acc = acc + torch.dot(x[bi, iy, ix].double(), wd[co, :, kh, kw])
        got, ref = y[bi, oy, ox, co].item(), acc.item()
        assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref)), (layer, bi, oy, ox, co, got, ref)
    # ---- adjoint identities (bias removed: <y - b, dz>)
    dz = torch.randn((B, Ho, Ho, Cout), generator=g, device=dev())
    y0 = ops.conv_forward(x, w, None, k, s, p)
    lhs = _dot64(y0, dz)
    del y, y0
    dx = ops.conv_dgrad(dz, w, (B, H, H, Cin), k, s, p)
    mid = _dot64(x, dx)
    del dx
    dw = ops.conv_wgrad(x, dz, k, s, p)
    rhs = _dot64(w, dw)
    scale = (_dot64(dz, dz) ** 0.5) * ((B * Ho * Ho * Cout) ** 0.5)      # ~ |y| |dz|
    assert abs(lhs - mid) <= 1e-6 * scale and abs(lhs - rhs) <= 1e-6 * scale, (lhs, mid, rhs, scale)


def test_full_size_routing_head_matches_fp64_and_gather_is_address_only():
    """BASELINE configs[2] head: R = 13*13*32 = 5408 rows, N = 512, C = 1, 8 -> 5, 3 iterations.  With one output capsule
    the coupling is identically 1, so v = squash(sum_i u_i W_i): checked in fp64 over all rows, with its gradients; the
    cell gather folded into the loads must give exactly what the oracle's explicit gather + the plain kernel give."""
    from capsyolo_amd import ops
    from oracle.models import cell_gather
    g, B, N, Din, Dout = 13, 32, 512, 8, 5
    feat = rnd((B, 256, 4 * g, 4 * g), 51, 0.7)
    W = rnd((1, N, 1, Din, Dout), 52, 0.1)
    u = cell_gather(feat, g).contiguous()                                  # [g*g*B, 512, 8] on the CPU (oracle)
    assert tuple(u.shape) == (g * g * B, N, Din)
    ud, Wd = u.to(dev()).requires_grad_(True), W.to(dev()).requires_grad_(True)
    v = ops.routing(ud, Wd, 3)
    G = rnd(tuple(v.shape), 53).to(dev())
    (v * G).sum().backward()
    u64, W64 = u.to(dev()).double().requires_grad_(True), W.to(dev()).double().requires_grad_(True)
    s = torch.einsum('rnd,ndo->ro', u64, W64[0, :, 0])
    n2 = (s * s).sum(-1, keepdim=True)
    vr = (n2 / (1 + n2)) * s / n2.sqrt()
    (vr * G.double().reshape(vr.shape)).sum().backward()
    close(v.reshape(vr.shape), vr, 2e-5, 1e-6)
    close(ud.grad, u64.grad, 1e-4, 1e-5)
    close(Wd.grad, W64.grad, 1e-4, 1e-5)
    fh = feat.permute(0, 2, 3, 1).contiguous().to(dev())
    vg = ops.routing(fh, W.to(dev()), 3, g, B)                             # [B, g, g, 1, 5]
    vp = v.detach().reshape(g, g, B, 1, Dout).permute(2, 0, 1, 3, 4)
    assert torch.equal(vg, vp.contiguous())


@pytest.mark.parametrize('case', [(2, 32, 16, 16, 64), (1, 64, 4, 32, 128), (3, 32, 20, 48, 64), (2, 128, 8, 16, 256), (1, 32, 12, 16, 64),
                                  (9, 64, 24, 48, 128)])
def test_conv3x3_winograd4_wgrad_matches_fp64_and_f22(case, monkeypatch):
    """Winograd F(3x3,4x4) weight gradient (winograd4_wgrad.hip: halo of the input patch at all four image borders, one-chunk-wide
    and one-tile-row-high maps, several input- / output-channel blocks, tile ranges that cut through images) against torch fp64
    and against the F(3x3,2x2) kernel.  Its transforms hold 4, 5, 8 and 1/24: 6e-6 relative to max against 1e-6 for F(3x3,2x2)
    (tools/probe/wino_f34_wgrad_numerics.py), hence the bound of 5e-5."""
    from capsyolo_amd import ops
    monkeypatch.setattr(ops, 'WINOGRAD4_MIN_PIXELS', 0)
    B, Cin, H, W_, Cout = case
    x = rnd((B, Cin, H, W_), 81)
    x = torch.where(x > 0, x, 0.1 * x)
    w = rnd((Cout, Cin, 3, 3), 82, (1.0 / (Cin * 9)) ** 0.5)
    wd = w.double().requires_grad_(True)
    zr = F.conv2d(x.double(), wd, None, padding=1)
    gz = rnd(tuple(zr.shape), 84)
    zr.backward(gz.double())
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev())
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev())
    ops.timer.reset()
    ops.timer.enabled = True
    try:
        dw = ops.conv_wgrad(xg, gzd, 3, 1, 1, False, 'wg')
    finally:
        ops.timer.enabled = False
    torch.cuda.synchronize()
    assert 'conv_wino4_wgrad/wg' in ops.timer.summary()
    scale = wd.grad.abs().max().item()
    close(dw, wd.grad, 5e-5, 5e-5 * scale)
    monkeypatch.setattr(ops, 'USE_WINOGRAD4_WGRAD', False)
    dw2 = ops.conv_wgrad(xg, gzd, 3, 1, 1)
    close(dw, dw2, 6e-5, 6e-5 * scale)


@pytest.mark.parametrize('case', [(2, 32, 16, 32, 64, 0.0), (2, 64, 8, 16, 128, 3.0), (3, 128, 12, 48, 64, -1.0), (5, 128, 24, 32, 128, 2.0)])
def test_conv3x3_winograd4_wgrad_with_fused_batchnorm_backward(case):
    """cy_conv3x3_winograd4_wgrad_bn (premasked gradient: dz = d scale + (z - mean) kb + kc formed on the way in and written out
    by the blocks of the first input-channel block) against cy_bn_bwd_apply + the F(3x3,2x2) weight gradient and the fp64 formula."""
    from capsyolo_amd import ops
    from capsyolo_amd._lib import call, query
    B, Cin, H, W_, Cout, zmean = case
    P = B * H * W_
    x = rnd((B, H, W_, Cin), 171).to(dev())
    z = (rnd((B, H, W_, Cout), 172) * 1.5 + zmean).to(dev())
    d = rnd((B, H, W_, Cout), 173).to(dev())
    gamma = (rnd((Cout,), 174).abs() + 0.5).to(dev())
    zd = z.double().reshape(P, Cout)
    mean, var = zd.mean(0), zd.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.double() * invstd
    dd = d.double().reshape(P, Cout)
    xh = (zd - mean) * invstd
    dz64 = scale * (dd - dd.mean(0) - xh * (dd * xh).mean(0))
    f = lambda t: t.float().contiguous()
    scale_f, mean_f, invstd_f = f(scale), f(mean), f(invstd)
    red = torch.stack([dd.sum(0), (dd * xh).sum(0)], 1).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    dz = torch.full_like(z, float('nan'))
    dw = torch.empty(Cout, Cin, 3, 3, device=dev())
    ws = torch.empty(query('cy_wino4_wgrad_ws_floats', B, H, W_, Cin, Cout), device=dev())
    call('cy_conv3x3_winograd4_wgrad_bn', x.data_ptr(), z.data_ptr(), d.data_ptr(), dz.data_ptr(), scale_f.data_ptr(), mean_f.data_ptr(),
         invstd_f.data_ptr(), red.data_ptr(), P, dw.data_ptr(), ws.data_ptr(), B, H, W_, Cin, Cout, st)
    torch.cuda.synchronize()
    assert torch.isfinite(dz).all()                                   # every element written
    zs = dz64.abs().max().item()
    close(dz.reshape(P, Cout), dz64, 1e-4, 0.0, 2e-5 * zs)
    old = ops.USE_WINOGRAD4_WGRAD
    try:
        ops.USE_WINOGRAD4_WGRAD = False
        dw_ref = ops.conv_wgrad(x, dz64.float().reshape(z.shape).contiguous(), 3, 1, 1)
    finally:
        ops.USE_WINOGRAD4_WGRAD = old
    close(dw, dw_ref, 6e-5, 6e-5 * dw_ref.abs().max().item())


def test_round3_winograd_kernels_repeat_bit_for_bit(monkeypatch):
    """The F(4x4,3x3) / F(3x3,4x4) / F(4x4,2x2) kernels add their partial sums in a fixed order (tile ranges, chunk order) and read
    every hand-waited register behind a sufficient wait: the same inputs give the same bits, run after run (the double atomics of the
    BatchNorm sums are the one documented exception and are not compared).  tools/check_determinism.py runs more shapes."""
    from capsyolo_amd import ops
    monkeypatch.setattr(ops, 'WINOGRAD4_MIN_PIXELS', 0)
    monkeypatch.setattr(ops, 'WINOGRAD4_S2_MIN_PIXELS', 0)
    B, H = 3, 48
    x = rnd((B, H, H, 128), 301).to(dev())
    w3 = rnd((128, 128, 3, 3), 302, 0.03).to(dev())
    dz3 = rnd((B, H, H, 128), 303).to(dev())
    w4 = rnd((64, 128, 4, 4), 304, 0.03).to(dev())
    dz4 = rnd((B, H // 2, H // 2, 64), 305).to(dev())
    z = rnd((B, H, H, 128), 306).to(dev())
    sc, sh = (rnd((128,), 307, 0.3) + 1.0).to(dev()), rnd((128,), 308, 0.5).to(dev())
    mu, isd = rnd((128,), 309, 0.2).to(dev()), (rnd((128,), 310, 0.1).abs() + 0.8).to(dev())

    def bn_dgrad():
        red = torch.zeros((ops.STATS_COPIES, 128, 2), dtype=torch.float64, device=dev())
        return ops.conv_dgrad(dz4, w4, (B, H, H, 128), 4, 2, 1, 'c', (z, sc, sh, mu, isd, 0.1, red), {})
    cases = {
        'F(4x4,3x3) forward': lambda: ops.conv_forward(x, w3, None, 3, 1, 1),
        'F(4x4,3x3) input gradient': lambda: ops.conv_dgrad(dz3, w3, (B, H, H, 128), 3, 1, 1),
        'F(3x3,4x4) weight gradient': lambda: ops.conv_wgrad(x, dz3, 3, 1, 1),
        'F(4x4,2x2) forward': lambda: ops.conv_forward(x, w4, None, 4, 2, 1),
        'F(4x4,2x2) forward, input affine': lambda: ops.conv_forward(x, w4, None, 4, 2, 1, False, None, False, 'c', (sc, sh, 0.1)),
        'F(4x4,2x2) input gradient': lambda: ops.conv_dgrad(dz4, w4, (B, H, H, 128), 4, 2, 1),
        'F(4x4,2x2) input gradient with the BatchNorm sums (stored gradient)': bn_dgrad,
    }
    ops.timer.reset()
    ops.timer.enabled = True
    try:
        for name, fn in cases.items():
            first = fn()
            for _ in range(3):
                assert torch.equal(fn(), first), name
    finally:
        ops.timer.enabled = False
    keys = ops.timer.summary()
    for pre in ('conv_wino4_fwd/', 'conv_wino4_dgrad/', 'conv_wino4_wgrad/', 'conv_wino42_fwd/', 'conv_wino42_dgrad/'):
        assert any(k.startswith(pre) for k in keys), pre


@pytest.mark.parametrize('case', [(3, 64, 12, 10), (2, 128, 6, 14), (1, 512, 4, 2), (2, 32, 8, 8)])
def test_fused_batchnorm_activation_maxpool_and_its_backward(case):
    """cy_affine_act_maxpool2 / cy_maxpool2_bwd_bn (round 4: DarkNet's conv -> BatchNorm -> LeakyReLU -> MaxPool blocks, models.py:135 ...
    195): forward against torch (F.max_pool2d of the activated tensor; ties go to the first position like nn.MaxPool2d: checked on a
    tensor with repeated values), backward: the premasked gradient against autograd's gradient of z, the BatchNorm-backward sums
    against cy_bn_bwd_reduce's contract (sum d, sum d xhat) in fp64."""
    from capsyolo_amd import ops
    B, C, Ho, Wo = case
    z = rnd((B, 2 * Ho, 2 * Wo, C), 301)
    z[0, :2, :2, :] = 0.25                                     # a window of equal values: the first position must win
    scale, shift = rnd((C,), 302).abs() + 0.5, rnd((C,), 303, 0.3)
    mean, invstd = rnd((C,), 304, 0.2), rnd((C,), 305).abs() + 0.5
    slope = 0.1
    dy = rnd((B, Ho, Wo, C), 306)
    zd = z.double().requires_grad_(True)
    act = F.leaky_relu(zd * scale.double() + shift.double(), slope)
    pooled, ind = F.max_pool2d(act.permute(0, 3, 1, 2), 2, return_indices=True)
    pooled.backward(dy.permute(0, 3, 1, 2).double())
    holder = {'mean': mean.to(dev()), 'invstd': invstd.to(dev()), 'red': None}
    zg = z.to(dev()).requires_grad_(True)
    y = ops.affine_act_maxpool(zg, scale.to(dev()), shift.to(dev()), slope, holder)
    close(y.permute(0, 3, 1, 2), pooled, 1e-6, 1e-6)
    assert int(torch.count_nonzero(y[0, 0, 0] - float(F.leaky_relu(torch.tensor(0.25) * scale + shift, slope)[0]))) <= C   # (shape check only)
    ops.zero_pool.reset(torch.device('cuda', 0))
    y.backward(dy.to(dev()))
    d = zg.grad
    # autograd's dz = d * scale (chain through z * scale + shift); the kernel hands over d itself (the producer applies the BatchNorm)
    close(d * scale.to(dev()), zd.grad.float(), 1e-5, 1e-6)
    assert holder['premasked'] is True
    dref = (zd.grad / scale.double()).detach()
    xhat = (z.double() - mean.double()) * invstd.double()
    red = holder['red'].cpu()
    close(red[:, 0], dref.sum(dim=(0, 1, 2)), 1e-5, 1e-6)          # (fp32 terms d, d * xhat summed in double: the terms' own rounding)
    close(red[:, 1], (dref * xhat).sum(dim=(0, 1, 2)), 1e-4, 1e-5)
    # ties: position 0 of the constant window got the gradient
    assert float(d[0, 0, 0].abs().sum()) > 0 and float(d[0, 0, 1].abs().sum()) == 0 and float(d[0, 1, 0].abs().sum()) == 0
