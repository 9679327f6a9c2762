"""The bf16 path (params.json "precision": "bf16", BASELINE configs[4]): every bf16 kernel against torch fp64 on inputs
that are exactly representable in bf16 (so that only the kernel's own arithmetic -- fp32 accumulation, one rounding of
the result to bf16 -- is measured), and single conv -> BatchNorm -> LeakyReLU blocks at the SURVEY H4 tolerance for bf16 (2e-2), and the whole DarkCapsuleNet
against a CPU restatement with the same roundings and against the REFERENCE's fp32 results (statistical bounds)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import closed_form_state, grad_digest, load_golden, make_params, synth_gtsdb_labels, synth_images

pytestmark = pytest.mark.gpu
T = torch.from_numpy
BF = torch.bfloat16


def dev():
    return torch.device('cuda:0')


def rnd_bf(shape, seed, scale=1.0):
    """Random values rounded to bf16 (returned as float32: exactly representable)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF).float()


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


# B, Cin, H, W, Cout, k, s  (pad 1): every layer class of the backbone behind the first layer, odd sizes, partial tiles
BF16_CASES = [
    (2, 128, 12, 12, 256, 3, 1),      # conv_2 class
    (2, 256, 16, 16, 64, 4, 2),       # conv_3 class (N = 64 tile, weight-gradient blocks of 64 x 64)
    (3, 64, 10, 14, 128, 4, 2),       # conv_4 class
    (2, 128, 16, 16, 256, 4, 2),      # conv_5 class
    (1, 64, 37, 45, 128, 3, 1),       # odd sizes, pixel tiles and 32-pixel row segments with tails
    (2, 128, 9, 70, 128, 4, 2),       # odd height, rows longer than one segment
]


@pytest.mark.parametrize('cout,B,H,W', [(128, 2, 12, 64), (64, 1, 37, 96), (32, 3, 5, 32)])
def test_first_block_activation_on_the_bf16_matrix_cores(cout, B, H, W):
    """cy_conv1_3x3_fwd_act_bf16 (first block of the bf16 path: lrelu((conv(x) + b) * scale + shift) -> bf16) multiplies on
    v_mfma_f32_32x32x16_bf16: with image and weights exactly representable in bf16 the only roundings are the fp32 accumulation and
    the ONE rounding of the result (half an ulp of bf16 = 2^-9 of the value, at most 2^-8 = 3.9e-3 of the maximum); with arbitrary fp32 inputs the
    operands' rounding to bf16 adds its 2^-9 per term (1e-2).  Border tiles (rows 0 / H-1, first / last 32-pixel segment) included."""
    from capsyolo_amd import ops
    for exact in (True, False):
        g = torch.Generator().manual_seed(11 + cout)
        x = torch.randn(B, 3, H, W, generator=g) * 60.0
        w = torch.randn(cout, 3, 3, 3, generator=g) * 0.2
        if exact:
            x, w = x.to(BF).float(), w.to(BF).float()
        b = torch.randn(cout, generator=g) * 0.1
        sc = torch.rand(cout, generator=g) * 0.05 + 0.01
        sh = torch.randn(cout, generator=g) * 0.3
        ref = F.leaky_relu((F.conv2d(x.double(), w.double(), b.double(), padding=1) * sc.double().view(1, -1, 1, 1)
                            + sh.double().view(1, -1, 1, 1)), 0.1).permute(0, 2, 3, 1)
        out = ops.conv1_affine_act(x.to(dev()), w.to(dev()), b.to(dev()), sc.to(dev()), sh.to(dev()), 0.1, out_bf16=True)
        assert out.dtype == BF and tuple(out.shape) == (B, H, W, cout)
        err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < (4e-3 if exact else 1e-2), (exact, err)


@pytest.mark.parametrize('B,H,W,cout', [(4, 32, 64, 128), (2, 48, 96, 64), (4, 256, 256, 128)])
def test_first_block_of_the_bf16_path(B, H, W, cout):
    """The first conv -> BatchNorm -> LeakyReLU block of the bf16 path (Cin = 3: csrc/conv1.hip on v_mfma_f32_32x32x16_bf16, bf16
    activation out, bf16 gradient in) against torch fp64 modules on a bf16-representable image and weights: activation, running
    statistics and every gradient, through the two-pass backward (small inputs) and the one-pass backward on the patch moments
    (from 2^18 pixels on).  What differs from fp64: the rounding of the activation and of d = dA * lrelu' to bf16."""
    from capsyolo_amd import models, ops
    x = rnd_bf((B, 3, H, W), 31, 40.0)
    conv = torch.nn.Conv2d(3, cout, 3, 1, padding=1).double()
    bn = torch.nn.BatchNorm2d(cout).double()
    with torch.no_grad():
        conv.weight.copy_(rnd_bf(tuple(conv.weight.shape), 32, 0.2).double())
        conv.bias.copy_(rnd_bf((cout,), 33, 0.1).double())
        bn.weight.copy_((rnd_bf((cout,), 34, 0.2) + 1.0).double())
        bn.bias.copy_(rnd_bf((cout,), 35, 0.2).double())
    a = F.leaky_relu(bn(conv(x.double())), 0.1)
    ga = rnd_bf(tuple(a.shape), 36)
    a.backward(ga.double())
    hc = models.HipConv2d(3, cout, 3, 1, 1)
    hb = models.HipBatchNorm2d(cout)
    with torch.no_grad():
        hc.weight.copy_(conv.weight.float()); hc.bias.copy_(conv.bias.float())
        hb.weight.copy_(bn.weight.float()); hb.bias.copy_(bn.bias.float())
    hc.cuda(); hb.cuda().train()
    cfg = ops.ConvBlockCfg(3, 1, 1, True, hb, 0.1, 'first')
    cfg.out_bf16 = True
    ops.timer.reset(); ops.timer.enabled = True
    try:
        out = ops.conv_block(x.cuda(), hc.weight, hc.bias, hb.weight, hb.bias, cfg)
        assert out.dtype == BF and tuple(out.shape) == (B, H, W, cout)
        out.backward(ga.permute(0, 2, 3, 1).contiguous().cuda().to(BF))
        keys = set(k.split('/')[0] for k in ops.timer.events)
    finally:
        ops.timer.enabled = False
    onepass = B * H * W >= ops.CONV1_MOMENTS_MIN_PIXELS
    assert ('conv1_bn_bwd_onepass' in keys) == onepass and ('conv1_bn_bwd_wgrad' in keys) == (not onepass), keys
    assert rel_l2(out.float().permute(0, 3, 1, 2), a) < 1e-2
    assert rel_l2(hc.weight.grad, conv.weight.grad) < 2e-2
    assert rel_l2(hb.weight.grad, bn.weight.grad) < 2e-2 and rel_l2(hb.bias.grad, bn.bias.grad) < 2e-2
    assert rel_l2(hb.running_mean, bn.running_mean) < 1e-3 and rel_l2(hb.running_var, bn.running_var) < 1e-3


@pytest.mark.parametrize('case', BF16_CASES)
def test_conv_bf16_fwd_dgrad_wgrad(case):
    from capsyolo_amd import ops
    B, Cin, H, W, Cout, k, s = case
    x = rnd_bf((B, Cin, H, W), 1)
    w = rnd_bf((Cout, Cin, k, k), 2, (1.0 / (Cin * k * k)) ** 0.5)
    b = rnd_bf((Cout,), 3, 0.1)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    zr = F.conv2d(xd, wd, b.double(), stride=s, padding=1)
    gz = rnd_bf(tuple(zr.shape), 4)
    zr.backward(gz.double())
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev()).to(BF)
    stats = torch.zeros(ops.STATS_COPIES, Cout, 2, dtype=torch.float64, device=dev())
    z = ops.conv_forward_bf16(xg, w.to(dev()), b.to(dev()), k, s, 1, stats)
    assert z.dtype == BF
    zr_nhwc = zr.detach().permute(0, 2, 3, 1)
    # fp32 accumulation of exact products, then ONE rounding to bf16 (2^-9 relative)
    assert rel_l2(z.float(), zr_nhwc) < 3e-3
    np.testing.assert_allclose(z.float().cpu().double().numpy(), zr_nhwc.numpy(), rtol=5e-3, atol=5e-3 * float(zr.abs().max()))
    # the statistics come from the fp32 accumulators, before the rounding
    st = stats.sum(0).cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), zr.detach().sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[:, 1].numpy(), (zr.detach() ** 2).sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    gzd = gz.permute(0, 2, 3, 1).contiguous().to(dev()).to(BF)
    for out_f32 in (False, True):
        dx = ops.conv_dgrad_bf16(gzd, w.to(dev()), (B, H, W, Cin), k, s, 1, out_f32)
        assert dx.dtype == (torch.float32 if out_f32 else BF)
        assert rel_l2(dx.float().permute(0, 3, 1, 2), xd.grad) < (2e-5 if out_f32 else 3e-3)
    dw = ops.conv_wgrad_bf16(xg, gzd, k, s, 1)
    assert dw.dtype == torch.float32
    assert rel_l2(dw, wd.grad) < 2e-5            # exact products, fp32 accumulation, fp32 result


@pytest.mark.parametrize('case', BF16_CASES + [(4, 64, 40, 36, 64, 4, 2), (2, 256, 44, 44, 128, 3, 1)])
def test_conv_dgrad_bf16_with_fused_batchnorm_backward_sums(case):
    """The bf16 input-gradient kernel's optional BatchNorm-backward sums of the PRODUCER block (ops.conv_dgrad_bf16 bn_fuse): the
    stored gradient is bf16(dx * lrelu'(z * scale + shift)) with dx the fp32 accumulator (= the out_f32 launch), and the sums are
    those of the STORED values -- what cy_bn_bwd_reduce_bf16 would compute from them with slope 1 -- over all parity classes."""
    from capsyolo_amd import ops
    from capsyolo_amd._lib import call
    B, Cin, H, W, Cout, k, s = case
    Ho, Wo = (H + 2 - k) // s + 1, (W + 2 - k) // s + 1
    w = rnd_bf((Cout, Cin, k, k), 12, (1.0 / (Cin * k * k)) ** 0.5).to(dev())
    gz = rnd_bf((B, Ho, Wo, Cout), 14).to(dev()).to(BF)
    z = rnd_bf((B, H, W, Cin), 15).to(dev()).to(BF)
    g = torch.Generator().manual_seed(16)
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(dev()), (torch.randn(Cin, generator=g) * 0.5).to(dev())
    mu, isd = (torch.randn(Cin, generator=g) * 0.2).to(dev()), (torch.rand(Cin, generator=g) + 0.5).to(dev())
    red = torch.zeros(ops.STATS_COPIES, Cin, 2, dtype=torch.float64, device=dev())
    d = ops.conv_dgrad_bf16(gz, w, (B, H, W, Cin), k, s, 1, False, 'c', (z, sc, sh, mu, isd, 0.1, red))
    dx32 = ops.conv_dgrad_bf16(gz, w, (B, H, W, Cin), k, s, 1, True)
    assert d.dtype == BF
    y = torch.addcmul(sh, z.float(), sc)
    ref = torch.where(y > 0, dx32, dx32 * 0.1).to(BF)
    # (the kernel forms y with one fused multiply-add: a pre-activation within an ulp of zero may take the other branch)
    differ = (d != ref)
    assert int(differ.sum()) <= max(2, d.numel() // 100000), int(differ.sum())
    assert bool((y.abs()[differ] < 1e-5).all())
    # the sums against the reduce kernel run on the stored gradient (already masked: slope 1)
    want = torch.zeros(Cin, 2, dtype=torch.float64, device=dev())
    call('cy_bn_bwd_reduce_bf16', z.data_ptr(), d.data_ptr(), 0, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), 1.0,
         want.data_ptr(), B * H * W, Cin, torch.cuda.current_stream().cuda_stream)
    got = red.sum(0)
    scale_ = want.abs().max(dim=0).values.clamp(min=1e-30)
    assert float(((got - want).abs() / scale_).max()) < 1e-5, ((got - want).abs() / scale_).max()


def test_bn_act_bf16_kernels():
    """affine_act / bn_bwd_reduce / bn_bwd_apply on bf16 tensors vs the formulas in fp64."""
    from capsyolo_amd._lib import call
    P, N = 777, 128
    z = rnd_bf((P, N), 11)
    da = rnd_bf((P, N), 12)
    sc, sh = rnd_bf((N,), 13).abs() + 0.5, rnd_bf((N,), 14, 0.3)
    mu, isd = rnd_bf((N,), 15, 0.2), rnd_bf((N,), 16).abs() + 0.5
    slope = 0.1
    st = torch.cuda.current_stream().cuda_stream
    zg, dag = z.to(dev()).to(BF), da.to(dev()).to(BF)
    scg, shg, mug, isg = (v.to(dev()) for v in (sc, sh, mu, isd))
    y64 = z.double() * sc.double() + sh.double()
    a64 = torch.where(y64 > 0, y64, y64 * slope)
    for out_f32 in (0, 1):
        out = torch.empty((P, N), dtype=torch.float32 if out_f32 else BF, device=dev())
        call('cy_affine_act_bf16', zg.data_ptr(), out.data_ptr(), scg.data_ptr(), shg.data_ptr(), slope, P, N, out_f32, st)
        assert rel_l2(out.float(), a64) < (1e-6 if out_f32 else 3e-3)
    for da_f32 in (0, 1):
        dag_x = da.to(dev()) if da_f32 else dag
        red = torch.empty((N, 2), dtype=torch.float64, device=dev())
        call('cy_bn_bwd_reduce_bf16', zg.data_ptr(), dag_x.data_ptr(), da_f32, scg.data_ptr(), shg.data_ptr(), mug.data_ptr(),
             isg.data_ptr(), slope, red.data_ptr(), P, N, st)
        d64 = torch.where(y64 > 0, da.double(), da.double() * slope)
        xh = (z.double() - mu.double()) * isd.double()
        r0, r1 = d64.sum(0), (d64 * xh).sum(0)
        np.testing.assert_allclose(red[:, 0].cpu().numpy(), r0.numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(red[:, 1].cpu().numpy(), r1.numpy(), rtol=1e-4, atol=1e-3)
        dz = torch.empty((P, N), dtype=BF, device=dev())
        dg, db = torch.empty(N, device=dev()), torch.empty(N, device=dev())
        call('cy_bn_bwd_apply_bf16', zg.data_ptr(), dag_x.data_ptr(), da_f32, dz.data_ptr(), scg.data_ptr(), shg.data_ptr(),
             mug.data_ptr(), isg.data_ptr(), slope, red.data_ptr(), dg.data_ptr(), db.data_ptr(), P, N, st)
        ref = sc.double() * (d64 - r0 / P - xh * (r1 / P))
        assert rel_l2(dz.float(), ref) < 4e-3
        np.testing.assert_allclose(dg.cpu().numpy(), r1.numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(db.cpu().numpy(), r0.numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('case', [(4, 128, 24, 256, 3, 1, False), (4, 256, 32, 64, 4, 2, False), (4, 128, 16, 256, 4, 2, True)])
def test_conv_bn_lrelu_block_bf16(case):
    """One conv -> BatchNorm -> LeakyReLU block on the bf16 kernels (forward, running statistics, all gradients) against
    torch fp64 modules on bf16-representable inputs and weights: no chain of layers amplifies anything here, so the block
    is held to the bf16 tolerance of SURVEY H4 (2e-2; measured 3e-3 .. 1e-2)."""
    from capsyolo_amd import models, ops
    B, Cin, H, Cout, k, s, out_f32 = case
    x = rnd_bf((B, Cin, H, H), 21)
    conv = torch.nn.Conv2d(Cin, Cout, k, s, padding=1).double()
    bn = torch.nn.BatchNorm2d(Cout).double()
    with torch.no_grad():
        conv.weight.copy_(rnd_bf(tuple(conv.weight.shape), 22, (1.0 / (Cin * k * k)) ** 0.5).double())
        conv.bias.copy_(rnd_bf((Cout,), 23, 0.1).double())
        bn.weight.copy_((rnd_bf((Cout,), 24, 0.2) + 1.0).double())
        bn.bias.copy_(rnd_bf((Cout,), 25, 0.2).double())
    xd = x.double().requires_grad_(True)
    a = F.leaky_relu(bn(conv(xd)), 0.1)
    ga = rnd_bf(tuple(a.shape), 26)
    a.backward(ga.double())
    hc = models.HipConv2d(Cin, Cout, k, s, 1)
    hb = models.HipBatchNorm2d(Cout)
    with torch.no_grad():
        hc.weight.copy_(conv.weight.float()); hc.bias.copy_(conv.bias.float())
        hb.weight.copy_(bn.weight.float()); hb.bias.copy_(bn.bias.float())
    hc.cuda(); hb.cuda().train()
    cfg = ops.ConvBlockCfg(k, s, 1, False, hb, 0.1, 'blk')
    cfg.out_f32, cfg.in_f32 = out_f32, False
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().to(BF).requires_grad_(True)
    out = ops.conv_block_bf16(xg, hc.weight, hc.bias, hb.weight, hb.bias, cfg)
    assert out.dtype == (torch.float32 if out_f32 else BF)
    gag = ga.permute(0, 2, 3, 1).contiguous().cuda()
    out.backward(gag if out_f32 else gag.to(BF))
    assert rel_l2(out.float().permute(0, 3, 1, 2), a) < 1e-2
    assert rel_l2(xg.grad.float().permute(0, 3, 1, 2), xd.grad) < 2e-2
    assert rel_l2(hc.weight.grad, conv.weight.grad) < 2e-2
    assert rel_l2(hb.weight.grad, bn.weight.grad) < 2e-2 and rel_l2(hb.bias.grad, bn.bias.grad) < 2e-2
    assert rel_l2(hb.running_mean, bn.running_mean) < 1e-3 and rel_l2(hb.running_var, bn.running_var) < 1e-3
    assert int(hb.num_batches_tracked) == 1


@pytest.mark.parametrize('case', BF16_CASES + [(3, 64, 33, 47, 128, 3, 1), (2, 256, 26, 26, 64, 4, 2)])
def test_weight_gradient_with_the_batchnorm_backward_on_the_way_in(case):
    """cy_conv_wgrad_bf16_bn (BatchNorm-backward pass 2 formed in the weight gradient's loader from the premasked gradient and z, dz
    written out for the input-gradient kernel) against the two launches it replaces -- cy_bn_bwd_apply_bf16 with slope 1 and
    cy_conv_wgrad_bf16: dz, dW, dgamma and dbeta BIT-IDENTICAL (the same expression per element, the same order of accumulation), on
    every layer class incl. odd sizes and partial row pairs; run twice (no race between the blocks that share a dz tile)."""
    from capsyolo_amd import ops
    from capsyolo_amd._lib import call, query
    B, Cin, H, W, Cout, k, s_ = case
    Ho, Wo = (H + 2 - k) // s_ + 1, (W + 2 - k) // s_ + 1
    nws = query('cy_conv_wgrad_bf16_bn_ws_floats', B, Ho, Wo, Cin, Cout, k, s_)
    if nws < 0:
        pytest.skip('layer class not built for the bf16 weight gradient')
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, Cin, generator=g).to(BF).to(dev())
    d = torch.randn(B, Ho, Wo, Cout, generator=g).to(BF).to(dev())
    z = torch.randn(B, Ho, Wo, Cout, generator=g).to(BF).to(dev())
    sc = (torch.rand(Cout, generator=g) + 0.5).to(dev()); sh = torch.randn(Cout, generator=g).to(dev())
    mu = (torch.randn(Cout, generator=g) * 0.2).to(dev()); isd = (torch.rand(Cout, generator=g) + 0.5).to(dev())
    P = B * Ho * Wo
    red = (torch.randn(Cout, 2, generator=g, dtype=torch.float64) * P * 0.01).to(dev())
    st = torch.cuda.current_stream().cuda_stream
    dz0 = torch.empty_like(z); dg0 = torch.empty(Cout, device=dev()); db0 = torch.empty(Cout, device=dev())
    call('cy_bn_bwd_apply_bf16', z.data_ptr(), d.data_ptr(), 0, dz0.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), 1.0,
         red.data_ptr(), dg0.data_ptr(), db0.data_ptr(), P, Cout, st)
    dW0 = ops.conv_wgrad_bf16(x, dz0, k, s_, 1)
    for rep in range(2):
        dz1 = torch.full_like(z, float('nan')); dg1 = torch.empty(Cout, device=dev()); db1 = torch.empty(Cout, device=dev())
        dW1 = torch.empty(Cout, Cin, k, k, device=dev()); ws = torch.empty(nws, device=dev())
        call('cy_conv_wgrad_bf16_bn', x.data_ptr(), d.data_ptr(), z.data_ptr(), dz1.data_ptr(), dW1.data_ptr(), ws.data_ptr(), sc.data_ptr(),
             mu.data_ptr(), isd.data_ptr(), red.data_ptr(), dg1.data_ptr(), db1.data_ptr(), B, H, W, Cin, Ho, Wo, Cout, k, s_, st)
        ne = dz1.view(torch.int16) != dz0.view(torch.int16)
        if bool(ne.any()):
            m1 = (red[:, 0] / P).float(); m2 = (red[:, 1] / P).float()
            ka, kb, kc = sc, -sc * isd * m2, sc * (mu * isd * m2 - m1)
            want = (d.float() * ka + (z.float() * kb + kc)).to(BF)
            ix = tuple(ne.nonzero()[0].tolist())
            info = (rep, int(ne.sum()), ne.nonzero()[:4].tolist(), 'fused', float(dz1[ix]), 'apply', float(dz0[ix]), 'torch', float(want[ix]),
                    'fused==torch', int((dz1.view(torch.int16) != want.view(torch.int16)).sum()), 'apply==torch', int((dz0.view(torch.int16) != want.view(torch.int16)).sum()))
            print('WGBN', info)       # (pytest truncates the tuple in the assertion message)
            assert False, info[:3]
        assert torch.equal(dW1, dW0), (rep, float((dW1 - dW0).abs().max()))
        assert torch.equal(dg1, dg0) and torch.equal(db1, db0)


def test_weight_gradient_bf16_random_shapes():
    """Seeded random shapes (odd sizes, one-row and one-segment images, partial row pairs, every layer class) through the bf16 weight
    gradient's hand-waited loader -- border flags, two output rows per chunk, phantom chunks behind a block's range -- plain against
    torch fp64 and run to run, and fused with the BatchNorm backward against the two launches it replaces (bit-identical)."""
    import random
    from capsyolo_amd import ops
    from capsyolo_amd._lib import call, query
    rng = random.Random(2024)
    st = torch.cuda.current_stream().cuda_stream
    done = 0
    for it in range(40):
        k, s_ = rng.choice([(4, 2), (3, 1)])
        B = rng.choice([1, 2, 3])
        if k == 3:
            Cin, Cout = rng.choice([64, 128]), rng.choice([128, 256])
        else:
            Cout = rng.choice([64, 128])
            Cin = rng.choice([64, 128]) if Cout % 128 else rng.choice([32, 64])
        H, W = rng.randint(2, 70), rng.randint(2, 70)
        if k == 4:
            H, W = 2 * (H // 2 + 1), 2 * (W // 2 + 1)
        Ho, Wo = (H + 2 - k) // s_ + 1, (W + 2 - k) // s_ + 1
        nws = query('cy_conv_wgrad_bf16_bn_ws_floats', B, Ho, Wo, Cin, Cout, k, s_)
        if Ho < 1 or Wo < 1 or nws < 0:
            continue
        g = torch.Generator().manual_seed(it)
        x = torch.randn(B, H, W, Cin, generator=g).to(BF).to(dev())
        d = torch.randn(B, Ho, Wo, Cout, generator=g).to(BF).to(dev())
        z = torch.randn(B, Ho, Wo, Cout, generator=g).to(BF).to(dev())
        g1, g2 = ops.conv_wgrad_bf16(x, d, k, s_, 1), ops.conv_wgrad_bf16(x, d, k, s_, 1)
        w0 = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, device=dev(), requires_grad=True)
        F.conv2d(x.double().permute(0, 3, 1, 2), w0, None, stride=s_, padding=1).backward(d.double().permute(0, 3, 1, 2))
        err = float((g1.double() - w0.grad).abs().max() / w0.grad.abs().max().clamp(min=1e-30))
        assert err < 2e-5 and torch.equal(g1, g2), (k, s_, B, H, W, Cin, Cout, err)
        sc = (torch.rand(Cout, generator=g) + 0.5).to(dev()); sh = torch.zeros(Cout, device=dev())
        mu = (torch.randn(Cout, generator=g) * 0.2).to(dev()); isd = (torch.rand(Cout, generator=g) + 0.5).to(dev())
        P = B * Ho * Wo
        red = (torch.randn(Cout, 2, generator=g, dtype=torch.float64) * P * 0.01).to(dev())
        dz0 = torch.empty_like(z); dg = torch.empty(Cout, device=dev()); db = torch.empty(Cout, device=dev())
        call('cy_bn_bwd_apply_bf16', z.data_ptr(), d.data_ptr(), 0, dz0.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), 1.0,
             red.data_ptr(), dg.data_ptr(), db.data_ptr(), P, Cout, st)
        dW0 = ops.conv_wgrad_bf16(x, dz0, k, s_, 1)
        dz1 = torch.full_like(z, float('nan')); dW1 = torch.empty(Cout, Cin, k, k, device=dev()); ws = torch.empty(nws, device=dev())
        call('cy_conv_wgrad_bf16_bn', x.data_ptr(), d.data_ptr(), z.data_ptr(), dz1.data_ptr(), dW1.data_ptr(), ws.data_ptr(), sc.data_ptr(),
             mu.data_ptr(), isd.data_ptr(), red.data_ptr(), None, None, B, H, W, Cin, Ho, Wo, Cout, k, s_, st)
        assert torch.equal(dz1.view(torch.int16), dz0.view(torch.int16)) and torch.equal(dW1, dW0), (k, s_, B, H, W, Cin, Cout)
        done += 1
    assert done >= 25


def _oracle_bf16_forward(net, x, g):
    """The oracle's DarkCapsuleNet forward with the bf16 path's roundings restated on the CPU: block 1 with bf16 operands of its
    convolution, fp32 statistics and its activation rounded to bf16; blocks 2..5 with bf16 weights, fp32 accumulation, BatchNorm statistics from the fp32
    conv output, normalisation applied to the bf16-rounded conv output, activations rounded to bf16 (the last one
    stays fp32: the routing head is an fp32 kernel).  torch's cast is differentiable (identity), so autograd gives the
    matching gradients up to the roundings of the backward tensors."""
    from oracle import models as OM
    rb = lambda t: t.to(BF).float()
    mods = dict(net.conv.named_children())
    h = x
    for i in range(1, 6):
        conv, bn, act = mods['conv_%d' % i], mods['bn_%d' % i], mods['relu_%d' % i]
        if i == 1:
            # the first block multiplies on the bf16 matrix cores (image and weights rounded to bf16, fp32 accumulation) and normalises
            # with the statistics of the EXACT fp32 convolution (they come from the patch moments / the fp32 statistics pass)
            ze = conv(h)
            mean, var = ze.mean(dim=(0, 2, 3)), ze.var(dim=(0, 2, 3), unbiased=False)
            zb = F.conv2d(rb(h), rb(conv.weight), conv.bias, stride=conv.stride, padding=conv.padding)
            yv = (zb - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + bn.eps) * bn.weight[None, :, None, None] \
                + bn.bias[None, :, None, None]
            h = rb(act(yv))
            continue
        z = F.conv2d(h, rb(conv.weight), conv.bias, stride=conv.stride, padding=conv.padding)
        mean, var = z.mean(dim=(0, 2, 3)), z.var(dim=(0, 2, 3), unbiased=False)
        yv = (rb(z) - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + bn.eps) * bn.weight[None, :, None, None] \
            + bn.bias[None, :, None, None]
        h = F.leaky_relu(yv, 0.1)
        if i < 5:
            h = rb(h)
    B = x.shape[0]
    v = net.traffic_sign_capsules(OM.cell_gather(h.contiguous(), g))[:, 0, 0, 0, :]
    return v.view(g, g, B, 5).permute(2, 0, 1, 3)


def test_darkcapsule_net_bf16():
    """DarkCapsuleNet with params.precision = 'bf16', whole net, from torch's default (seeded) initialisation at 96 x 96,
    batch 8, against the fp32 oracle and against the oracle with the bf16 roundings restated on the CPU.
    Output: 2e-2 relative L2 against fp32 (SURVEY H4's bf16 tolerance; the CPU restatement itself sits at 9e-3), loss
    1e-2.  Gradients: bf16 storage alone moves them by 6-13 % relative L2 in the four blocks below the last one (CPU
    restatement vs fp32: LeakyReLU / BatchNorm turn 0.4 % activation roundings into that), so they are bounded at 2.5e-1
    against fp32 and must be at least as close to the restatement as the restatement is to fp32 (x 1.5).
    (With the closed-form weights of the golden fixtures this net is chaotic under ANY bf16 rounding -- the CPU
    restatement's gradients differ from fp32 by 60-100 % there -- which is why this test does not use them.)"""
    from capsyolo_amd import loss_fns, models, optim
    from oracle import loss_fns as OL
    from oracle import models as OM
    H, g, B = 96, 3, 8
    p = make_params(model='darkcapsule', n_grid=g, darknet_input=H, recon=False, device='cuda', precision='bf16')
    po = make_params(model='darkcapsule', n_grid=g, darknet_input=H, recon=False)
    xc, yc = T(synth_images(B, H, seed=41)), T(synth_gtsdb_labels(B, g, 43, seed=42))
    x, y = xc.cuda(), yc.cuda()
    res = {}
    for mode in ('bf16', 'fp32'):
        torch.manual_seed(0)
        onet = OM.DarkCapsuleNet(po).train()
        oo = _oracle_bf16_forward(onet, xc, g) if mode == 'bf16' else onet(xc)
        ol = OL.darkcapsule_loss(oo, yc, po)
        ol.backward()
        res[mode] = (oo.detach(), dict((n, q.grad) for n, q in onet.named_parameters()), ol.item(), onet.state_dict())
    torch.manual_seed(0)
    net = models.DarkCapsuleNet(p)
    net.load_state_dict(res['fp32'][3])
    net.cuda().train()
    out = net(x)
    loss = loss_fns.darkcapsule_loss(out, y, p)
    loss.backward()
    e32, ebf = rel_l2(out, res['fp32'][0]), rel_l2(out, res['bf16'][0])
    assert e32 < 2e-2 and ebf < 2e-2, (e32, ebf)
    assert abs(loss.item() - res['fp32'][2]) < 1e-2 * abs(res['fp32'][2])
    worst = 0.0
    for name, q in net.named_parameters():
        g32, gbf = res['fp32'][1][name], res['bf16'][1][name]
        if q.grad is None:
            assert g32 is None, name
            continue
        if '.conv_' in name and name.endswith('bias'):
            continue
        err32, errbf, base = rel_l2(q.grad, g32), rel_l2(q.grad, gbf), rel_l2(gbf, g32)
        worst = max(worst, err32)
        assert err32 < 2.5e-1, '%s: %.3e against fp32' % (name, err32)
        assert errbf < max(2e-2, 1.5 * base), '%s: %.3e against the bf16 restatement (restatement vs fp32: %.3e)' % (name, errbf, base)
    # 20 Adam steps against the fp32 oracle's curve from the same initialisation
    torch.manual_seed(0)
    onet = OM.DarkCapsuleNet(po).train()
    onet.load_state_dict(res['fp32'][3])
    oopt = torch.optim.Adam([q for q in onet.parameters() if q.requires_grad], lr=1e-3)
    net = models.DarkCapsuleNet(p)
    net.load_state_dict(res['fp32'][3])
    net.cuda().train()
    opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
    curve, ocurve = [], []
    for _ in range(20):
        l = loss_fns.darkcapsule_loss(net(x), y, p)
        opt.zero_grad(); l.backward(); opt.step()
        curve.append(l.item())
        ol = OL.darkcapsule_loss(onet(xc), yc, po)
        oopt.zero_grad(); ol.backward(); oopt.step()
        ocurve.append(ol.item())
    span = max(ocurve) - min(ocurve)
    dev_frac = float(np.abs(np.array(curve) - np.array(ocurve)).max()) / span
    print('bf16 whole net: output %.2e vs fp32 / %.2e vs the bf16 restatement, worst gradient %.2e vs fp32, 20-step loss curve '
          'max deviation %.2f %% of its range' % (e32, ebf, worst, 100 * dev_frac))
    assert dev_frac < 1.5e-1 and np.all(np.isfinite(curve))       # measured 8.6 % (a flat 20-step curve: small range)


def test_bf16_rejects_unsupported_blocks():
    from capsyolo_amd import _lib, models
    p = make_params(model='darknet_d', n_grid=2, n_boxes=2, n_classes=0, darknet_input=64, dropout=0.0, device='cuda', precision='bf16')
    net = models.DarkNet(p).cuda()          # DarkNet does not opt in: precision is a DarkCapsuleNet key
    assert net.model.precision == 'fp32'
    p2 = make_params(model='darkcapsule', n_grid=2, darknet_input=64, recon=False, device='cuda', precision='fp16')
    with pytest.raises(ValueError):
        models.DarkCapsuleNet(p2)
    assert issubclass(_lib.HipExtensionError, RuntimeError)


def test_bf16_eval_forward_folds_batchnorm():
    """Eval mode on the bf16 path: every block behind the first is ONE bf16 GEMM launch on weights / bias with the BatchNorm
    folded in (cy_bn_fold_eval -> fp32 masters -> bf16 pack) and a LeakyReLU epilogue; against the unfolded bf16 path (same
    kernels + cy_affine_act_bf16) within two bf16 roundings, and against the fp32 eval forward at the bf16 tolerance."""
    from capsyolo_amd import _lib, models, ops
    H, g, B = 96, 3, 4
    p = make_params(model='darkcapsule', n_grid=g, darknet_input=H, recon=False, device='cuda', precision='bf16')
    p32 = make_params(model='darkcapsule', n_grid=g, darknet_input=H, recon=False, device='cuda')
    torch.manual_seed(0)
    net = models.DarkCapsuleNet(p).cuda()
    x = T(synth_images(B, H, seed=43)).cuda()
    net.train()
    with torch.no_grad():
        for _ in range(3):                  # running statistics that are not the initial (0, 1)
            net(x)
    net32 = models.DarkCapsuleNet(p32).cuda()
    net32.load_state_dict(net.state_dict())
    net.eval(); net32.eval()
    with torch.no_grad():
        _lib.TRACE = []
        try:
            out = net(x)
            torch.cuda.synchronize()
            calls = list(_lib.TRACE)
        finally:
            _lib.TRACE = None
        ops.FOLD_EVAL_BN = False
        try:
            out0 = net(x)
        finally:
            ops.FOLD_EVAL_BN = True
        out32 = net32(x)
    assert not any(c.startswith('cy_affine_act') for c in calls), calls
    assert calls.count('cy_bn_fold_eval') == 4 and calls.count('cy_conv_gemm_bf16') == 4
    assert rel_l2(out, out0) < 1e-2, rel_l2(out, out0)
    assert rel_l2(out, out32) < 2e-2, rel_l2(out, out32)
