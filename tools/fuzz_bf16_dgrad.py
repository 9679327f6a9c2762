"""Randomised shapes for the bf16 input gradient: the parity classes in one launch against one launch per class (bit-identical), and the
fused BatchNorm-backward sums against the reduce kernel on the stored gradient.   python3 tools/fuzz_bf16_dgrad.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import random
import torch
import capsyolo_amd  # noqa: F401
from capsyolo_amd import ops
from capsyolo_amd._lib import call

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
BF = torch.bfloat16
bad = 0
for it in range(n):
    k, s = rng.choice([(4, 2), (4, 2), (3, 1)])
    B = rng.choice([1, 2, 3, 5])
    Cin, Cout = rng.choice([64, 128, 256]), rng.choice([64, 128, 256] if k == 4 else [128, 256])
    H, W = rng.randint(3, 70), rng.randint(3, 70)
    if k == 4 and rng.random() < 0.7:
        H, W = 2 * (H // 2 + 1), 2 * (W // 2 + 1)
    Ho, Wo = (H + 2 - k) // s + 1, (W + 2 - k) // s + 1
    if Ho < 1 or Wo < 1:
        continue
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05)
    gz = torch.randn(B, Ho, Wo, Cout, device=dev).to(BF)
    z = torch.randn(B, H, W, Cin, device=dev).to(BF)
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
    mu, isd = torch.randn(Cin, device=dev) * 0.2, torch.rand(Cin, device=dev) + 0.5
    outs = []
    for one in (True, False):
        ops.BF16_DGRAD_ONE_LAUNCH = one
        red = torch.zeros(ops.STATS_COPIES, Cin, 2, dtype=torch.float64, device=dev)
        d = ops.conv_dgrad_bf16(gz, w, (B, H, W, Cin), k, s, 1, False, 'c', (z, sc, sh, mu, isd, 0.1, red))
        p = ops.conv_dgrad_bf16(gz, w, (B, H, W, Cin), k, s, 1)
        outs.append((d, p, red.sum(0)))
    ops.BF16_DGRAD_ONE_LAUNCH = True
    (d1, p1, r1), (d0, p0, r0) = outs
    want = torch.zeros(Cin, 2, dtype=torch.float64, device=dev)
    call('cy_bn_bwd_reduce_bf16', z.data_ptr(), d1.data_ptr(), 0, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), 1.0,
         want.data_ptr(), B * H * W, Cin, torch.cuda.current_stream().cuda_stream)
    scale_ = want.abs().max(dim=0).values.clamp(min=1e-30)
    e_sum = float(((r1 - want).abs() / scale_).max())
    ok = torch.equal(d1, d0) and torch.equal(p1, p0) and e_sum < 1e-5 and float(((r1 - r0).abs() / scale_).max()) < 1e-9
    print('k%d s%d B%d %2dx%-2d %3d->%-3d one launch == per class: %s %s, sums %.1e %s' % (k, s, B, H, W, Cin, Cout, torch.equal(d1, d0), torch.equal(p1, p0), e_sum, '' if ok else 'FAIL'), flush=True)
    bad += 0 if ok else 1
print('fuzz_bf16_dgrad:', 'ok' if bad == 0 else '%d FAILED' % bad)
sys.exit(1 if bad else 0)
