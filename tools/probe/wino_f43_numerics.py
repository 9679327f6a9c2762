"""Numerics of Winograd F(4x4,3x3) against the shipped F(2x2,3x3) on a conv_2-shaped layer.

F(4x4,3x3) needs 36 multiplies per 16 outputs (2.25 per output) against F(2x2,3x3)'s 16 per 4 (4.0): 1.78x fewer MFMA FLOPs for
conv_2 (77 % of the model).  Its transform matrices hold 4, 5, 8 and 1/24, so fp32 products of transformed operands cancel harder.
This probe evaluates both algorithms in float32 (numpy: exact fp32 products, fp32 accumulation over the 128 input channels in
channel order like the MFMA chain) on activations shaped like conv_2's input (LeakyReLU(0.1) of unit-variance values) and
He-scaled weights, against the direct convolution in float64.  CPU only.

    python3 tools/probe/wino_f43_numerics.py        ->  relative L2 / max errors of both, three point sets for F(4,3)"""
import numpy as np


def cook_toom(points, m, r):
    """A^T (m x n), G (n x r), B^T (n x n) of F(m, r) for the given n - 1 = m + r - 2 finite points plus infinity."""
    n = m + r - 1
    pts = list(points)
    assert len(pts) == n - 1
    # polynomial multiplication by evaluation / interpolation (Toom-Cook), transposed into the FIR form
    V = lambda cols: np.array([[p ** k for k in range(cols)] for p in pts] + [[0.0] * (cols - 1) + [1.0]], dtype=np.float64)
    Vm, Vr, Vn = V(m), V(r), V(n)
    Vn_inv = np.linalg.inv(Vn)
    # y = A^T [(G g) * (B^T d)]:  A^T = Vm^T, G = Vr, B^T = Vn^{-T}
    return Vm.T.copy(), Vr.copy(), Vn_inv.T.copy()


def scale_rows(AT, G, BT):
    """Move the interpolation's row scalings from B^T into G (the usual form: B^T keeps small integers)."""
    n = G.shape[0]
    for i in range(n):
        s = np.abs(BT[i]).max()
        s = 1.0 if s == 0 else s
        # keep B^T's rows integer-friendly: divide the row by its smallest non-zero magnitude
        nz = np.abs(BT[i][np.abs(BT[i]) > 1e-12]).min()
        BT[i] /= nz
        G[i] *= nz
    return AT, G, BT


def winograd2d(x, w, AT, G, BT, m, dtype):
    """x [H, W, C] (H, W multiples of m after the pad), w [K, C, 3, 3] -> y [H, W, K]; everything in `dtype`."""
    H, W, C = x.shape
    K = w.shape[0]
    n = m + 2
    AT, G, BT = AT.astype(dtype), G.astype(dtype), BT.astype(dtype)
    xp = np.zeros((H + 2, W + 2, C), dtype)
    xp[1:-1, 1:-1] = x
    U = np.einsum('ia,kcab,jb->ijkc', G, w.astype(dtype), G).astype(dtype)          # [n, n, K, C]
    y = np.zeros((H, W, K), dtype)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = xp[ty:ty + n, tx:tx + n]                                              # [n, n, C]
            V = np.einsum('ia,abc,jb->ijc', BT, d, BT).astype(dtype)
            M = np.zeros((n, n, K), dtype)
            for c in range(C):                                                        # channel-ordered fp32 accumulation (the MFMA chain)
                M += (U[:, :, :, c] * V[:, :, None, c]).astype(dtype)
            y[ty:ty + m, tx:tx + m] = np.einsum('ia,abk,jb->ijk', AT, M, AT).astype(dtype)
    return y


def direct64(x, w):
    H, W, C = x.shape
    xp = np.zeros((H + 2, W + 2, C))
    xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, w.shape[0]))
    for a in range(3):
        for b in range(3):
            y += np.einsum('hwc,kc->hwk', xp[a:a + H, b:b + W], w[:, :, a, b].astype(np.float64))
    return y


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    H = W = 24
    C, K = 128, 32
    z = rng.standard_normal((H, W, C))
    x = np.where(z > 0, z, 0.1 * z).astype(np.float32)                               # conv_2's input: LeakyReLU(0.1) of BatchNorm output
    w = (rng.standard_normal((K, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    ref = direct64(x.astype(np.float64), w)
    rel = lambda y: (np.linalg.norm(y - ref) / np.linalg.norm(ref), np.abs(y - ref).max() / np.abs(ref).max())
    # the direct convolution itself in fp32, channel-and-tap ordered chain of 1152 (what conv_gemm does)
    yd = np.zeros((H, W, K), np.float32)
    xp = np.zeros((H + 2, W + 2, C), np.float32); xp[1:-1, 1:-1] = x
    for a in range(3):
        for b in range(3):
            for c in range(C):
                yd += xp[a:a + H, b:b + W, c][:, :, None] * w[None, None, :, c, a, b]
    print('direct fp32 (chain of 1152)              rel L2 %.2e  max %.2e' % rel(yd))
    AT, G, BT = scale_rows(*cook_toom([0, 1, -1], 2, 3))
    print('F(2x2,3x3), points 0, +-1 (shipped)      rel L2 %.2e  max %.2e' % rel(winograd2d(x, w, AT, G, BT, 2, np.float32)))
    for name, pts in (('0, +-1, +-2 (Lavin)', [0, 1, -1, 2, -2]), ('0, +-1, +-1/2', [0, 1, -1, 0.5, -0.5]),
                      ('0, +-1/2, +-2 ', [0, 0.5, -0.5, 2, -2])):
        AT, G, BT = scale_rows(*cook_toom(pts, 4, 3))
        y32 = winograd2d(x, w, AT, G, BT, 4, np.float32)
        y64 = winograd2d(x.astype(np.float64), w, AT, G, BT, 4, np.float64)
        print('F(4x4,3x3), points %-22s rel L2 %.2e  max %.2e   (the same algorithm in fp64: %.1e)' % ((name,) + rel(y32) + (rel(y64)[0],)))
