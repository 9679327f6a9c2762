"""Randomised shapes for the round-3 kernels against the direct kernels (not a test of the suite: a one-off sweep for latent
indexing bugs -- tile groups that do not divide, grids smaller than the CU count, odd tile ranges, one-tile images).
    python3 tools/fuzz_wino4.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import random
import torch
import capsyolo_amd  # noqa: F401
from capsyolo_amd import ops
from capsyolo_amd._lib import query

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
bad = 0


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def both(fn, flags):
    out = []
    for v in (True, False):
        old = [getattr(ops, f) for f in flags]
        for f in flags:
            setattr(ops, f, v)
        try:
            out.append(fn())
        finally:
            for f, o in zip(flags, old):
                setattr(ops, f, o)
    return out


ops.WINOGRAD4_MIN_PIXELS = 0
ops.WINOGRAD4_S2_MIN_PIXELS = 0
for it in range(n):
    kind = rng.choice(['s2', 's2', '3x3'])
    B = rng.choice([1, 2, 3, 5])
    if kind == 's2':
        Cin, Cout = rng.choice([64, 128, 192, 256]), rng.choice([8, 16, 40, 64, 72, 128])
        H, W = 2 * rng.randint(1, 45), 2 * rng.randint(1, 45)
        x = torch.randn(B, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.05
        dz = torch.randn(B, H // 2, W // 2, Cout, device=dev)
        z = torch.randn(B, H, W, Cin, device=dev)
        sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
        mu, isd = torch.randn(Cin, device=dev) * 0.2, torch.rand(Cin, device=dev) + 0.5
        st = [torch.zeros(ops.STATS_COPIES, Cout, 2, dtype=torch.float64, device=dev) for _ in range(2)]
        k = [0]

        def fwd():
            s = st[k[0]]; k[0] += 1
            return ops.conv_forward(x, w, None, 4, 2, 1, False, s, False, 'c', (sc, sh, 0.1) if Cout % 64 == 0 else None)
        y1, y0 = both(fwd, ['USE_WINOGRAD4_S2'])
        reds = [torch.zeros(ops.STATS_COPIES, Cin, 2, dtype=torch.float64, device=dev) for _ in range(2)]
        k2 = [0]

        def dg():
            r = reds[k2[0]]; k2[0] += 1
            return ops.conv_dgrad(dz, w, (B, H, W, Cin), 4, 2, 1, 'c', (z, sc, sh, mu, isd, 0.1, r), {})
        d1, d0 = both(dg, ['USE_WINOGRAD4_S2_DGRAD'])
        e = (rel(y1, y0), rel(st[0].sum(0), st[1].sum(0)), rel(d1, d0), rel(reds[0].sum(0), reds[1].sum(0)))
        ok = e[0] < 5e-5 and e[1] < 1e-4 and e[2] < 5e-5 and e[3] < 2e-4
        print('%-4s B%d %3dx%-3d %3d->%-3d fwd %.1e stats %.1e dgrad %.1e sums %.1e %s' % (kind, B, H, W, Cin, Cout, e[0], e[1], e[2], e[3], '' if ok else 'FAIL'), flush=True)
    else:
        Cin, Cout = rng.choice([32, 64, 96, 128]), rng.choice([64, 128, 192])
        H, W = 4 * rng.randint(1, 12), 16 * rng.randint(1, 4)
        x = torch.randn(B, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        dz = torch.randn(B, H, W, Cout, device=dev)
        assert query('cy_wino4_wgrad_ok', B, H, W, Cin, Cout)
        g1, g0 = both(lambda: ops.conv_wgrad(x, dz, 3, 1, 1), ['USE_WINOGRAD4_WGRAD'])
        y1, y0 = both(lambda: ops.conv_forward(x, w, None, 3, 1, 1), ['USE_WINOGRAD4'])
        d1, d0 = both(lambda: ops.conv_dgrad(dz, w, (B, H, W, Cin), 3, 1, 1), ['USE_WINOGRAD4'])
        e = (rel(g1, g0), rel(y1, y0), rel(d1, d0))
        ok = e[0] < 1e-4 and e[1] < 1e-4 and e[2] < 1e-4
        print('%-4s B%d %3dx%-3d %3d->%-3d wgrad %.1e fwd %.1e dgrad %.1e %s' % (kind, B, H, W, Cin, Cout, e[0], e[1], e[2], '' if ok else 'FAIL'), flush=True)
    bad += 0 if ok else 1
    torch.cuda.synchronize()
print('fuzz_wino4:', 'ok' if bad == 0 else '%d FAILED' % bad)
sys.exit(1 if bad else 0)
