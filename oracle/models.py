"""Oracle (TEST INFRASTRUCTURE ONLY): PyTorch-CPU restatement of the reference's models.

Every class cites the reference lines it restates (paths relative to the
reference root).  The modules keep the reference's ``state_dict`` key names so
that the same weights can be loaded into the reference, the oracle and the HIP
product models.  Nothing here is used by the product path.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- capsule primitives
def squash(s):
    """models.py:64-67 -- |s|^2/(1+|s|^2) * s/|s| over the last dim, no epsilon (0 -> NaN)."""
    n2 = (s * s).sum(dim=-1, keepdim=True)
    return (n2 / (1.0 + n2)) * s / n2.sqrt()


def prediction_vectors(u, W):
    """models.py:71 -- u [R,N,Din], W [1,N,C,Din,Dout] -> u_hat [R,N,C,Dout]."""
    return torch.einsum('rnd,ncdo->rnco', u, W[0])


def dynamic_routing(u, W, n_iter=3, return_all=False):
    """models.py:70-79 (routing branch of CapsuleLayer.forward).

    u [R,N,Din], W [1,N,C,Din,Dout] -> v [R,C,Dout].
    Softmax over the output-capsule axis C (models.py:75); gradients flow
    through every iteration (no detach, models.py:74-79).  The reference keeps
    its logits at the redundant 5-D shape [R,N,C,1,Dout]; the values along Dout
    are identical so a [R,N,C] logit tensor is the same computation.
    """
    u_hat = prediction_vectors(u, W)                      # [R,N,C,Dout]
    b = torch.zeros(u_hat.shape[:3], dtype=u.dtype)       # [R,N,C]
    vs = []
    v = None
    for it in range(n_iter):
        c = F.softmax(b, dim=2)                           # over C
        s = (c.unsqueeze(-1) * u_hat).sum(dim=1)          # [R,C,Dout]
        v = squash(s)
        vs.append(v)
        if it != n_iter - 1:
            b = b + (u_hat * v.unsqueeze(1)).sum(dim=-1)
    return (v, vs) if return_all else v


def cell_gather(feat, g):
    """models.py:393-398 -- raw-memory view/chunk 'grid cell' gather.

    feat [B,256,4g,4g] (NCHW, contiguous) -> u [g*g*B, 512, 8], row = k*B + b.
    Capsule i = pos*32 + chg (pos = 4r+c), component d reads channel chg*8+d at
    flat spatial index r*4g^2 + 4k + c of the row-major (4g x 4g) plane.
    """
    B = feat.shape[0]
    flat = feat.reshape(B, 256, 4, g * g, 4)              # [B, ch, r, k, c]
    u = flat.permute(3, 0, 2, 4, 1)                       # [k, B, r, c, ch]
    return u.reshape(g * g * B, 16 * 32, 8)


class RoutingCapsules(nn.Module):
    """models.py:46-79 with n_nodes != -1: holds ``route_weights`` [1,N,C,Din,Dout]."""

    def __init__(self, n_caps, n_nodes, in_C, out_C, n_iter=3):
        super().__init__()
        self.n_iter = n_iter
        self.route_weights = nn.Parameter(0.1 * torch.randn(1, n_nodes, n_caps, in_C, out_C))

    def forward(self, u):
        v = dynamic_routing(u, self.route_weights, self.n_iter)   # [R,C,Dout]
        return v[:, None, :, None, :]                              # reference shape [R,1,C,1,Dout]


class PrimaryCapsules(nn.Module):
    """models.py:58-62,80-82 with n_nodes == -1: n_caps parallel convs, cat on a new last dim, squash."""

    def __init__(self, n_caps, in_C, out_C, kernel, stride):
        super().__init__()
        self.capsules = nn.ModuleList([nn.Conv2d(in_C, out_C, kernel, stride=stride) for _ in range(n_caps)])

    def forward(self, x):
        B = x.shape[0]
        comps = torch.stack([cap(x).reshape(B, -1) for cap in self.capsules], dim=-1)   # [B, out_C*h*w, n_caps]
        return squash(comps)


def _decoder():
    """models.py:96-111 (same block repeated at 372-387, 435-450)."""
    class _UnFlatten(nn.Module):
        def forward(self, x):
            return x.view(-1, 16, 4, 4)
    return nn.Sequential(
        nn.Linear(16, 256), nn.ReLU(), _UnFlatten(), nn.Upsample((8, 8)),
        nn.Conv2d(16, 4, 3, padding=1), nn.ReLU(), nn.Upsample((16, 16)),
        nn.Conv2d(4, 8, 3, padding=1), nn.ReLU(), nn.Upsample((32, 32)),
        nn.Conv2d(8, 16, 3, padding=1), nn.ReLU(),
        nn.Conv2d(16, 3, 3, padding=1), nn.Tanh())


# --------------------------------------------------------------------------- models
class ConvNet(nn.Module):
    """models.py:22-43."""

    def __init__(self, params):
        super().__init__()
        p = params.dropout
        self.cnn = nn.Sequential(
            nn.Conv2d(3, 64, 3, padding=1), nn.BatchNorm2d(64), nn.LeakyReLU(inplace=True), nn.Dropout(p),
            nn.Conv2d(64, 128, 3, padding=1), nn.BatchNorm2d(128), nn.LeakyReLU(inplace=True), nn.Dropout(p),
            nn.MaxPool2d(2), nn.Flatten(), nn.Linear(128 * 16 * 16, 128), nn.ReLU(),
            nn.Linear(128, params.n_classes))

    def forward(self, x):
        return self.cnn(x)


class CapsuleNet(nn.Module):
    """models.py:86-124."""

    def __init__(self, params, n_iter=3):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 256, 9)
        self.primary_capsules = PrimaryCapsules(8, 256, 16, 8, 2)
        self.traffic_sign_capsules = RoutingCapsules(params.n_classes, 16 * 9 * 9, 8, 16, n_iter)
        self.decoder = _decoder()

    def forward(self, x, y=None, recon=False):
        h = F.relu(self.conv1(x))
        u = self.primary_capsules(h)                                # [B,1296,8]
        caps = self.traffic_sign_capsules(u)[:, 0, :, 0, :]         # [B,C,16]  (reference: .squeeze())
        scores = (caps ** 2).sum(dim=-1) ** 0.5
        if not recon:
            return scores
        picked = caps[torch.arange(caps.shape[0]), y]               # models.py:122 gather of the true capsule
        return scores, self.decoder(picked)


def _cbl(seq, idx, cin, cout, k, stride=1, pad=0, bias=True, momentum=0.1):
    seq['conv_%d' % idx] = nn.Conv2d(cin, cout, k, stride, padding=pad, bias=bias)
    seq['bn_%d' % idx] = nn.BatchNorm2d(cout, momentum=momentum)
    seq['relu_%d' % idx] = nn.LeakyReLU(0.1)


# (cout, kernel, what follows: 'M' maxpool / 'D' dropout)  -- models.py:131-224
_DARKNET_PLAN = [
    (32, 3, 'M'), (64, 3, 'M'), (128, 3, 'D'), (64, 1, 'D'), (128, 3, 'M'),
    (256, 3, 'D'), (128, 1, 'D'), (256, 3, 'M'),
    (512, 3, 'D'), (256, 1, 'D'), (512, 3, 'D'), (256, 1, 'D'), (512, 3, 'M'),
    (1024, 3, 'D'), (512, 1, 'D'), (1024, 3, 'D'), (512, 1, 'D'), (1024, 3, 'D'),
]


class DarkNet(nn.Module):
    """models.py:126-236."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        seq = OrderedDict()
        cin, n_pool = 3, 0
        for idx, (cout, k, after) in enumerate(_DARKNET_PLAN, start=1):
            _cbl(seq, idx, cin, cout, k, 1, k // 2, bias=False, momentum=0.01)
            if after == 'M':
                n_pool += 1
                seq['maxpool_%d' % n_pool] = nn.MaxPool2d(2)
            else:
                seq['drop_%d' % idx] = nn.Dropout(params.dropout)
            cin = cout
        seq['conv_19'] = nn.Conv2d(1024, 5 * params.n_boxes + params.n_classes, 1, bias=False)
        self.model = nn.Sequential(seq)

    def forward(self, x):
        out = self.model(x).permute(0, 2, 3, 1)
        split = 5 * self.params.n_boxes
        y_box = torch.sigmoid(out[..., :split])
        if self.params.n_classes == 0:
            return y_box
        return torch.cat((y_box, F.softmax(out[..., split:], dim=-1)), dim=-1)


def _darkcaps_backbone():
    """models.py:346-366 (identical at 409-429): 5 x (Conv(bias) -> BN -> LeakyReLU 0.1)."""
    seq = OrderedDict()
    _cbl(seq, 1, 3, 128, 3, 1, 1)
    _cbl(seq, 2, 128, 256, 3, 1, 1)
    _cbl(seq, 3, 256, 64, 4, 2, 1)
    _cbl(seq, 4, 64, 128, 4, 2, 1)
    _cbl(seq, 5, 128, 256, 4, 2, 1)
    return nn.Sequential(seq)


class DarkCapsuleNet(nn.Module):
    """models.py:340-400.  n_caps=1 => coupling == 1 exactly."""

    def __init__(self, params, n_iter=3):
        super().__init__()
        self.params = params
        self.conv = _darkcaps_backbone()
        self.traffic_sign_capsules = RoutingCapsules(1, 16 * 32, 8, 5, n_iter)
        self.decoder = _decoder()          # never used in forward

    def forward(self, x):
        B, g = x.shape[0], self.params.n_grid
        u = cell_gather(self.conv(x).contiguous(), g)                 # [g*g*B,512,8]
        v = self.traffic_sign_capsules(u)[:, 0, 0, 0, :]              # [g*g*B,5]
        return v.view(g, g, B, 5).permute(2, 0, 1, 3)


class DarkCapsuleNet3(nn.Module):
    """models.py:403-463: same backbone, C = n_classes output capsules of 5+16 dims."""

    def __init__(self, params, n_iter=3):
        super().__init__()
        self.params = params
        self.conv = _darkcaps_backbone()
        self.traffic_sign_capsules = RoutingCapsules(params.n_classes, 16 * 32, 8, 21, n_iter)
        self.decoder = _decoder()

    def forward(self, x):
        B, g, C = x.shape[0], self.params.n_grid, self.params.n_classes
        u = cell_gather(self.conv(x).contiguous(), g)
        v = self.traffic_sign_capsules(u)[:, 0, :, 0, :]              # [g*g*B,C,21]
        return v.reshape(g, g, B, C, 21).permute(2, 0, 1, 3, 4)


class DarkCapsuleNet2(nn.Module):
    """models.py:271-337: only ``conv2`` + primary caps (1x1) + routing head are used in forward."""

    def __init__(self, params, n_iter=3):
        super().__init__()
        self.params = params

        def tower(n):
            seq, cin = OrderedDict(), 3
            for idx, cout in enumerate([32, 64, 128, 256, 512][:n], start=1):
                _cbl(seq, idx, cin, cout, 4, 2, 1)
                seq['drop_%d' % idx] = nn.Dropout(params.dropout)
                cin = cout
            return nn.Sequential(seq)
        self.conv = tower(4)      # declared, unused (models.py:276-296)
        self.conv2 = tower(5)
        self.primary_capsules = PrimaryCapsules(8, 512, 16, 1, 1)
        self.traffic_sign_capsules = RoutingCapsules(params.n_grid ** 2, 16 * 7 * 7, 8, 5 + params.n_classes, n_iter)

    def forward(self, x):
        B, g = x.shape[0], self.params.n_grid
        u = self.primary_capsules(self.conv2(x))
        v = self.traffic_sign_capsules(u)[:, 0, :, 0, :]              # [B,g*g,5+C]
        return v.reshape(B, g, g, -1)
