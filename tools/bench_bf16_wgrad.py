"""Times the bf16 weight gradient (wgrad_bf16_kernel + its slab sum) on the backbone's layer shapes.
usage: python3 tools/bench_bf16_wgrad.py [H] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd  # noqa: F401
from capsyolo_amd import ops

H = int(sys.argv[1]) if len(sys.argv) > 1 else 608
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device('cuda:0')
torch.manual_seed(0)
# name, Cin, Cout, k, stride, input size divisor
for name, cin, cout, k, s, div in (('conv_2', 128, 256, 3, 1, 1), ('conv_3', 256, 64, 4, 2, 1), ('conv_4', 64, 128, 4, 2, 2), ('conv_5', 128, 256, 4, 2, 4)):
    hi = H // div
    ho = hi if s == 1 else hi // 2
    x = torch.randn(B, hi, hi, cin, device=dev).to(torch.bfloat16)
    dz = torch.randn(B, ho, ho, cout, device=dev).to(torch.bfloat16)
    fn = lambda: ops.conv_wgrad_bf16(x, dz, k, s, 1)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    print('%s %d->%d k%d s%d at %d: %.3f ms  %.0f TFLOP/s (%.3f of 2500)' % (name, cin, cout, k, s, hi, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500), flush=True)
    del x, dz
