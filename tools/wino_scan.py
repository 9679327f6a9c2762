"""Winograd kernel time vs number of input-channel chunks (fixed per-block cost vs per-chunk cost)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd
from capsyolo_amd import ops
dev = torch.device('cuda:0')
B, H, Cout = 8, 416, 256
for Cin in (8, 32, 64, 128, 256):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.03
    f = lambda: ops.conv_forward(x, w, None, 3, 1, 1)
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    blocks = B * 26 * 26 * 4
    per_block_us = ms * 1e3 * 256 / blocks
    print('Cin %4d chunks %3d: %8.3f ms  per-block %7.2f us  (%.0f cycles @2.3GHz)' % (Cin, Cin // 8, ms, per_block_us, per_block_us * 2300), flush=True)
