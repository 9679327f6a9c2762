"""Model zoo on the HIP kernels: same classes, constructor signature (``ModelCls(params)``), forward
contract and ``state_dict`` keys as the reference's models.py, so reference checkpoints load and the
``main.py`` registry is unchanged.  Internally activations are NHWC and every layer is a call into
libcapsyolo_hip.so (see ops.py); inputs arrive NCHW fp32 as in main.py:57.

New optional ``params`` keys: ``n_iter`` (routing iterations, default 3); ``precision`` ('fp32' default |
'bf16': the DarkCapsuleNet / DarkCapsuleNet3 backbone on bf16 MFMA kernels, BASELINE configs[4]).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# ------------------------------------------------------------------------------------------------ building blocks
class HipConv2d(nn.Module):
    """Parameter holder with nn.Conv2d's names/shapes/init (models.py: every nn.Conv2d)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__()
        ref = nn.Conv2d(cin, cout, k, stride, padding=padding, bias=bias)     # torch's default init
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone()) if bias else None
        self.k, self.stride, self.padding = k, stride, padding
        self.tag = 'conv'            # name of the layer in the kernel timer (ops.timer); the owner may set it

    def forward(self, x, nchw_in=False, slope=None):
        cfg = ops.ConvBlockCfg(self.k, self.stride, self.padding, nchw_in, None, slope, name=self.tag)
        return ops.conv_block(x, self.weight, self.bias, None, None, cfg)


class HipBatchNorm2d(nn.Module):
    """Parameter/buffer holder with nn.BatchNorm2d's names (weight, bias, running_*, num_batches_tracked)."""

    def __init__(self, n, momentum=0.1, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self.momentum, self.eps = momentum, eps

    # The eval-mode fold cache (ops.fold_eval_bn) is keyed by the tensors' version counters, which a write through `param.data`
    # (manual initialisation, EMA, clipping, c10d's parameter broadcast) does not bump: every switch of mode and every state-dict
    # load therefore starts a new cache epoch -- an eval forward never sees W' / b' folded from older values.
    def train(self, mode=True):
        ops._bump_param_epoch()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        ops._bump_param_epoch()
        return super()._load_from_state_dict(*args, **kwargs)


class HipLeakyReLU(nn.Module):
    def __init__(self, slope):
        super().__init__()
        self.slope = slope


class HipMaxPool2(nn.Module):
    def forward(self, x):
        return ops.maxpool2(x)


class FusedBackbone(nn.Sequential):
    """nn.Sequential of conv_k / bn_k / relu_k / maxpool_k / drop_k children (reference names);
    forward walks the children and fuses each conv -> bn -> relu triple into one conv block."""

    def _triple(self, mods, i):
        """(bn, act, index after the block) of the conv at position i."""
        bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], HipBatchNorm2d) else None
        j = i + (2 if bn is not None else 1)
        act = mods[j] if j < len(mods) and isinstance(mods[j], HipLeakyReLU) else None
        return bn, act, j + (1 if act is not None else 0)

    precision = 'fp32'      # 'bf16': every block behind the first runs on the bf16 MFMA kernels (params.json "precision")

    def _forward_bf16(self, x, nchw_in):
        """params.precision == 'bf16' (BASELINE configs[4]): the first block (3 input channels, store-bound) stays on
        its fp32 kernels, writes its activation as bf16 and reads the bf16 gradient; every further conv -> BatchNorm -> LeakyReLU block
        runs on v_mfma_f32_32x32x16_bf16 with bf16 activations, fp32 accumulation and fp32 / double statistics; the
        last block hands its activation over in fp32 (the routing head is an fp32 kernel)."""
        names, mods = zip(*self.named_children())
        for nm, m in zip(names, mods):
            # this walk only knows conv -> BatchNorm -> LeakyReLU triples: any other child (max-pool, dropout with p > 0)
            # would be skipped silently and a DIFFERENT network computed
            if not (isinstance(m, (HipConv2d, HipBatchNorm2d, HipLeakyReLU)) or (isinstance(m, nn.Dropout) and m.p == 0)):
                raise ops._lib.HipExtensionError('precision bf16: child %s (%s) has no bf16 path; the bf16 backbone is built '
                                                 'for conv -> BatchNorm -> LeakyReLU blocks only' % (nm, type(m).__name__))
        convs = [i for i, m in enumerate(mods) if isinstance(m, HipConv2d)]
        first = True
        holder = None                            # producer block -> consumer block: what the fused BatchNorm-backward sums need
        for n, i in enumerate(convs):
            m = mods[i]
            bn, act, _ = self._triple(mods, i)
            slope = act.slope if act is not None else None
            if first:
                cfg = ops.ConvBlockCfg(m.k, m.stride, m.padding, nchw_in, bn, slope, names[i])
                cfg.out_bf16 = bn is not None      # its activation leaves as bf16 and it takes a bf16 gradient (csrc/conv1.hip)
                x = ops.conv_block(x, m.weight, m.bias, bn.weight if bn is not None else None,
                                   bn.bias if bn is not None else None, cfg)
                first = False
                continue
            cin, cout = m.weight.shape[1], m.weight.shape[0]
            if bn is None or slope is None or not ops.bf16_layer_ok(m.k, m.stride, m.padding, cin, cout):
                raise ops._lib.HipExtensionError('precision bf16: no bf16 kernel for block %s (k=%d s=%d p=%d %d -> %d channels); '
                                                 'built: 3x3/s1/p1 and 4x4/s2/p1 with channels in multiples of 64'
                                                 % (names[i], m.k, m.stride, m.padding, cin, cout))
            cfg = ops.ConvBlockCfg(m.k, m.stride, m.padding, False, bn, slope, names[i])
            cfg.in_f32 = (n == 1) and x.dtype == torch.float32   # (a first block without BatchNorm hands over fp32)
            cfg.out_f32 = (n == len(convs) - 1)
            cfg.in_holder, cfg.out_holder = holder, {}
            holder = cfg.out_holder
            x = ops.conv_block_bf16(x, m.weight, m.bias, bn.weight, bn.bias, cfg)
        return x

    def forward(self, x, nchw_in=True):
        if self.training:
            ops.zero_pool.reset(x.device)     # one zero-fill per step for all statistics / backward-sum scratch
        if self.precision == 'bf16':
            return self._forward_bf16(x, nchw_in)
        names, mods = zip(*self.named_children())
        i = 0
        lazy = None                               # (scale, shift, slope) of a producer that deferred its activation
        while i < len(mods):
            m = mods[i]
            if isinstance(m, HipConv2d):
                bn, act, nxt = self._triple(mods, i)
                slope = act.slope if act is not None else None
                # hand the raw output + scale/shift to the next block when that block is a 4x4 / stride-2 layer whose
                # Winograd forward and weight-gradient kernels apply BatchNorm + LeakyReLU on their loads: the
                # activation tensor (5.7 GB after conv_2 at the headline shape) is never written
                defer = False
                if (ops.FUSE_INPUT_AFFINE and self.training and bn is not None and slope is not None and 0.0 < slope <= 1.0
                        and nxt < len(mods) and isinstance(mods[nxt], HipConv2d)):
                    n = mods[nxt]
                    hin = x.shape[2] if nchw_in else x.shape[1]
                    win = x.shape[3] if nchw_in else x.shape[2]
                    ho = (hin + 2 * m.padding - m.k) // m.stride + 1
                    wo = (win + 2 * m.padding - m.k) // m.stride + 1
                    defer = ops.s2_fusable(n.k, n.stride, n.padding, m.weight.shape[0], n.weight.shape[0], ho, wo)
                # ... and when a max-pool follows (DarkNet): activation + pooling in ONE pass over z, the pooling's backward with this
                # block's BatchNorm-backward sums in one pass (not for the first layer: its recompute kernels keep their own path)
                pool = False
                if (self.training and bn is not None and bn.training and nxt < len(mods) and isinstance(mods[nxt], HipMaxPool2)
                        and not (nchw_in and ops.conv1_ok(x, m.weight, m.k, m.stride, m.padding, nchw_in)) and not nchw_in):
                    ho = (x.shape[1] + 2 * m.padding - m.k) // m.stride + 1
                    wo = (x.shape[2] + 2 * m.padding - m.k) // m.stride + 1
                    pool = defer = ops.pool_fusable(m.weight.shape[0], ho, wo, slope)
                cfg = ops.ConvBlockCfg(m.k, m.stride, m.padding, nchw_in, bn, slope, names[i], defer_act=defer,
                                       in_slope=lazy[2] if lazy is not None else None)
                cfg.in_holder = lazy[3] if lazy is not None else None
                out = ops.conv_block(x, m.weight, m.bias, bn.weight if bn is not None else None,
                                     bn.bias if bn is not None else None, cfg,
                                     lazy[0] if lazy is not None else None, lazy[1] if lazy is not None else None)
                if pool:
                    x, lazy = ops.affine_act_maxpool(out[0], out[1], out[2], slope, getattr(cfg, 'out_holder', None)), None
                    nxt += 1                                  # the max-pool module is done
                elif defer:
                    x, lazy = out[0], (out[1], out[2], slope, getattr(cfg, 'out_holder', None))
                else:
                    x, lazy = out, None
                nchw_in = False
                i = nxt
            elif isinstance(m, nn.Dropout):
                if m.p > 0 and self.training:
                    x = F.dropout(x, m.p, True)      # torch RNG kept (RNG parity)
                i += 1
            else:
                x = m(x)
                i += 1
        return x


def _cbl(seq, idx, cin, cout, k, stride=1, pad=0, bias=True, momentum=0.1):
    seq['conv_%d' % idx] = HipConv2d(cin, cout, k, stride, pad, bias)
    seq['bn_%d' % idx] = HipBatchNorm2d(cout, momentum)
    seq['relu_%d' % idx] = HipLeakyReLU(0.1)


class CapsuleLayer(nn.Module):
    """models.py:46-83.  n_nodes != -1: routing layer holding ``route_weights``; n_nodes == -1:
    primary capsules (n_caps parallel convs, run as ONE fused conv with n_caps*out_C channels)."""

    def __init__(self, params, n_caps, n_nodes, in_C, out_C, kernel=None, stride=None, n_iter=3):
        super().__init__()
        self.params, self.n_iter, self.n_nodes, self.n_caps = params, n_iter, n_nodes, n_caps
        if n_nodes != -1:
            self.route_weights = nn.Parameter(0.1 * torch.randn(1, n_nodes, n_caps, in_C, out_C))
        else:
            self.capsules = nn.ModuleList([HipConv2d(in_C, out_C, kernel, stride) for _ in range(n_caps)])
            self.kernel, self.stride = kernel, stride

    def forward(self, x, gather_g=0, gather_B=0):
        if self.n_nodes != -1:
            return ops.routing(x, self.route_weights, self.n_iter, gather_g, gather_B)
        # fused weight: output channel o*n_caps + cap  <-  capsules[cap].weight[o]
        w = torch.stack([c.weight for c in self.capsules], dim=1)
        w = w.reshape(-1, w.shape[2], self.kernel, self.kernel)
        b = torch.stack([c.bias for c in self.capsules], dim=1).reshape(-1)
        cfg = ops.ConvBlockCfg(self.kernel, self.stride, 0, False, None, None, name='primary_caps')
        z = ops.conv_block(x, w, b, None, None, cfg)                      # [B,h,w,out_C*n_caps]
        return ops.squash(ops.primary_caps_rows(z, self.n_caps))         # [B, out_C*h*w, n_caps]


class Decoder(nn.Sequential):
    """models.py:96-111: Linear(16,256) ReLU UnFlatten Upsample conv ReLU Upsample conv ReLU Upsample conv ReLU conv Tanh.
    Children sit at the reference's indices (0,4,7,10,12 hold parameters)."""

    def __init__(self):
        lin = nn.Linear(16, 16 * 4 * 4)
        super().__init__(lin, nn.Identity(), nn.Identity(), nn.Identity(),
                         HipConv2d(16, 4, 3, 1, 1), nn.Identity(), nn.Identity(),
                         HipConv2d(4, 8, 3, 1, 1), nn.Identity(), nn.Identity(),
                         HipConv2d(8, 16, 3, 1, 1), nn.Identity(),
                         HipConv2d(16, 3, 3, 1, 1), nn.Identity())
        for i in (4, 7, 10, 12):
            self[i].tag = 'dec_%d' % i

    def forward(self, t):
        B = t.shape[0]
        lin = self[0]
        cfg = ops.ConvBlockCfg(1, 1, 0, False, None, 0.0, name='dec_fc')
        h = ops.conv_block(t.view(B, 1, 1, 16), lin.weight.view(256, 16, 1, 1), lin.bias, None, None, cfg)  # ReLU fused
        h = ops.nchw_to_nhwc(h.view(B, 16, 4, 4))                      # UnFlatten(16,4,4) is an NCHW view
        h = self[4](ops.upsample_nearest(h, 2), slope=0.0)
        h = self[7](ops.upsample_nearest(h, 2), slope=0.0)
        h = self[10](ops.upsample_nearest(h, 2), slope=0.0)
        h = ops.tanh(self[12](h))
        return ops.nhwc_to_nchw(h)                                     # [B,3,32,32] like the reference


class CapsuleNet(nn.Module):
    """models.py:86-124."""

    def __init__(self, params):
        super().__init__()
        n_iter = getattr(params, 'n_iter', 3)
        self.conv1 = HipConv2d(3, 256, 9)
        self.conv1.tag = 'conv1'
        self.primary_capsules = CapsuleLayer(params, n_caps=8, n_nodes=-1, in_C=256, out_C=16, kernel=8, stride=2)
        self.traffic_sign_capsules = CapsuleLayer(params, n_caps=params.n_classes, n_nodes=16 * 9 * 9, in_C=8,
                                                  out_C=16, n_iter=n_iter)
        self.decoder = Decoder()

    def forward(self, x, y=None, recon=False):
        h = self.conv1(x, nchw_in=True, slope=0.0)                     # relu(conv1(x)), NHWC
        u = self.primary_capsules(h)                                   # [B,1296,8]
        caps = self.traffic_sign_capsules(u)                           # [B,C,16]
        scores = ops.length(caps)
        if not recon:
            return scores
        return scores, self.decoder(ops.pick_capsule(caps, y))


_DARKNET_PLAN = [
    (32, 3, 'M'), (64, 3, 'M'), (128, 3, 'D'), (64, 1, 'D'), (128, 3, 'M'),
    (256, 3, 'D'), (128, 1, 'D'), (256, 3, 'M'),
    (512, 3, 'D'), (256, 1, 'D'), (512, 3, 'D'), (256, 1, 'D'), (512, 3, 'M'),
    (1024, 3, 'D'), (512, 1, 'D'), (1024, 3, 'D'), (512, 1, 'D'), (1024, 3, 'D'),
]


class DarkNet(nn.Module):
    """models.py:126-269."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        seq = OrderedDict()
        cin, n_pool = 3, 0
        for idx, (cout, k, after) in enumerate(_DARKNET_PLAN, start=1):
            _cbl(seq, idx, cin, cout, k, 1, k // 2, bias=False, momentum=0.01)
            if after == 'M':
                n_pool += 1
                seq['maxpool_%d' % n_pool] = HipMaxPool2()
            else:
                seq['drop_%d' % idx] = nn.Dropout(params.dropout)
            cin = cout
        seq['conv_19'] = HipConv2d(1024, 5 * params.n_boxes + params.n_classes, 1, bias=False)
        self.model = FusedBackbone(seq)

    def forward(self, x):
        out = self.model(x)                                            # NHWC == the reference's permute(0,2,3,1)
        return ops.yolo_head(out, 5 * self.params.n_boxes, self.params.n_classes)

    def load_weights(self, weights_dir, n_load_layer):
        """models.py:238-269: TF-style npz ('<idx>-<scope>/<name>:0', HWIO kernels) into the first layers."""
        import numpy as np
        names = {'kernel:0': ('conv', 'weight'), 'biases:0': ('bn', 'bias'), 'gamma:0': ('bn', 'weight'),
                 'moving_mean:0': ('bn', 'running_mean'), 'moving_variance:0': ('bn', 'running_var')}
        state = self.state_dict()
        for key, v in np.load(weights_dir).items():
            index, layer = key.split('-')
            index = int(index) + 1
            if index > n_load_layer:
                continue
            kind, pname = names[layer.split('/')[1]]
            t = torch.from_numpy(v)
            if kind == 'conv':
                t = t.permute(3, 2, 0, 1)
            state['model.%s_%d.%s' % (kind, index, pname)] = t
        self.load_state_dict(state)


def _precision(params):
    """Optional params.json key ``precision``: 'fp32' (default, the reference's arithmetic) or 'bf16'."""
    pr = getattr(params, 'precision', 'fp32')
    if pr not in ('fp32', 'bf16'):
        raise ValueError("params.precision must be 'fp32' or 'bf16', got %r" % (pr,))
    return pr


def _darkcaps_backbone():
    seq = OrderedDict()
    _cbl(seq, 1, 3, 128, 3, 1, 1)
    _cbl(seq, 2, 128, 256, 3, 1, 1)
    _cbl(seq, 3, 256, 64, 4, 2, 1)
    _cbl(seq, 4, 64, 128, 4, 2, 1)
    _cbl(seq, 5, 128, 256, 4, 2, 1)
    return FusedBackbone(seq)


class DarkCapsuleNet(nn.Module):
    """models.py:340-400.  The cell gather (393-398) is folded into the routing kernel's loads."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.conv = _darkcaps_backbone()
        self.conv.precision = _precision(params)
        self.traffic_sign_capsules = CapsuleLayer(params, n_caps=1, n_nodes=16 * 32, in_C=8, out_C=5,
                                                  n_iter=getattr(params, 'n_iter', 3))
        self.decoder = Decoder()            # unused in forward, like the reference

    def forward(self, x):
        B, g = x.shape[0], self.params.n_grid
        if x.shape[2] != 32 * g or x.shape[3] != 32 * g:
            raise ValueError('DarkCapsuleNet needs H = W = 32*n_grid (models.py:393); got %s, n_grid=%d'
                             % (tuple(x.shape), g))
        feat = self.conv(x)                                            # [B,4g,4g,256] NHWC
        v = self.traffic_sign_capsules(feat, gather_g=g, gather_B=B)   # [B,g,g,1,5]
        return v.view(B, g, g, 5)


class DarkCapsuleNet3(nn.Module):
    """models.py:403-463."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.conv = _darkcaps_backbone()
        self.conv.precision = _precision(params)
        self.traffic_sign_capsules = CapsuleLayer(params, n_caps=params.n_classes, n_nodes=16 * 32, in_C=8,
                                                  out_C=5 + 16, n_iter=getattr(params, 'n_iter', 3))
        self.decoder = Decoder()

    def forward(self, x):
        B, g = x.shape[0], self.params.n_grid
        feat = self.conv(x)
        return self.traffic_sign_capsules(feat, gather_g=g, gather_B=B)   # [B,g,g,C,21]


def _s2_tower(n, dropout):
    """models.py:276-322: n x (Conv 4x4 / stride 2 -> BN -> LeakyReLU 0.1 -> Dropout), 3 -> 32 -> 64 -> 128 -> 256 -> 512."""
    seq, cin = OrderedDict(), 3
    for idx, cout in enumerate([32, 64, 128, 256, 512][:n], start=1):
        _cbl(seq, idx, cin, cout, 4, 2, 1)
        seq['drop_%d' % idx] = nn.Dropout(dropout)
        cin = cout
    return FusedBackbone(seq)


class DarkCapsuleNet2(nn.Module):
    """models.py:271-337 (unwired in the reference's registry; registered here as ``darkcapsule2``): five stride-2
    convs -> 8 primary 1x1 convs (one fused 512 -> 128 conv) -> routing to n_grid^2 capsules of 5 + n_classes dims.
    ``conv`` (four layers) is declared and never used, like the reference; the input must be 224 x 224 (7 x 7 x 16
    primary capsules)."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.conv = _s2_tower(4, params.dropout)
        self.conv2 = _s2_tower(5, params.dropout)
        self.primary_capsules = CapsuleLayer(params, n_caps=8, n_nodes=-1, in_C=512, out_C=16, kernel=1, stride=1)
        self.traffic_sign_capsules = CapsuleLayer(params, n_caps=params.n_grid ** 2, n_nodes=16 * 7 * 7, in_C=8,
                                                  out_C=5 + params.n_classes, n_iter=getattr(params, 'n_iter', 3))

    def forward(self, x):
        B, g = x.shape[0], self.params.n_grid
        if x.shape[2] != 224 or x.shape[3] != 224:
            raise ValueError('DarkCapsuleNet2 needs 224 x 224 inputs (models.py:325-329); got %s' % (tuple(x.shape),))
        u = self.primary_capsules(self.conv2(x))                       # [B,784,8]
        return self.traffic_sign_capsules(u).view(B, g, g, -1)         # [B,g*g,5+C] -> [B,g,g,5+C]


class ConvNet(nn.Module):
    """models.py:22-43: the plain-torch CNN baseline (not a kernel target)."""

    def __init__(self, params):
        super().__init__()

        class Flatten(nn.Module):
            def forward(self, x):
                return x.view(x.size(0), -1)
        self.cnn = nn.Sequential(
            nn.Conv2d(3, 64, 3, padding=1), nn.BatchNorm2d(64), nn.LeakyReLU(inplace=True), nn.Dropout(params.dropout),
            nn.Conv2d(64, 128, 3, padding=1), nn.BatchNorm2d(128), nn.LeakyReLU(inplace=True),
            nn.Dropout(params.dropout), nn.MaxPool2d(2), Flatten(), nn.Linear(128 * 16 * 16, 128), nn.ReLU(),
            nn.Linear(128, params.n_classes))

    def forward(self, x):
        return self.cnn(x)
