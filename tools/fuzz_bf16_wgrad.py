"""Randomised shapes for the bf16 weight gradient (hand-waited buffer loads with border flags, two output rows per chunk, phantom
chunks behind a block's range): against torch's fp64 weight gradient of the same bf16 tensors, and run to run (bit-identical).
python3 tools/fuzz_bf16_wgrad.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import random
import torch
import torch.nn.functional as F
import capsyolo_amd  # noqa: F401
from capsyolo_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
BF = torch.bfloat16
bad = 0
for it in range(n):
    k, s = rng.choice([(4, 2), (4, 2), (3, 1)])
    B = rng.choice([1, 2, 3, 5])
    if k == 3:
        Cin, Cout = rng.choice([64, 128]), rng.choice([128, 256])
    else:
        Cout = rng.choice([64, 128, 256])
        Cin = rng.choice([64, 128, 256]) if Cout % 128 else rng.choice([32, 64, 128])
    H, W = rng.randint(2, 75), rng.randint(2, 75)
    if k == 4 and rng.random() < 0.7:
        H, W = 2 * (H // 2 + 1), 2 * (W // 2 + 1)
    Ho, Wo = (H + 2 - k) // s + 1, (W + 2 - k) // s + 1
    if Ho < 1 or Wo < 1:
        continue
    x = torch.randn(B, H, W, Cin, device=dev).to(BF)
    dz = torch.randn(B, Ho, Wo, Cout, device=dev).to(BF)
    try:
        g1 = ops.conv_wgrad_bf16(x, dz, k, s, 1)
    except Exception as e:
        print('k%d s%d B%d %2dx%-2d %3d->%-3d unsupported: %s' % (k, s, B, H, W, Cin, Cout, str(e)[:60]))
        continue
    g2 = ops.conv_wgrad_bf16(x, dz, k, s, 1)
    w0 = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, device=dev, requires_grad=True)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w0, None, stride=s, padding=1)
    y.backward(dz.double().permute(0, 3, 1, 2))
    err = float((g1.double() - w0.grad).abs().max() / w0.grad.abs().max().clamp(min=1e-30))
    ok = err < 2e-5 and torch.equal(g1, g2)
    print('k%d s%d B%d %2dx%-2d %3d->%-3d max err / max %.1e, run to run %s %s' % (k, s, B, H, W, Cin, Cout, err, torch.equal(g1, g2), '' if ok else 'FAIL'), flush=True)
    bad += 0 if ok else 1
print('fuzz_bf16_wgrad:', 'ok' if bad == 0 else '%d FAILED' % bad)
sys.exit(1 if bad else 0)
