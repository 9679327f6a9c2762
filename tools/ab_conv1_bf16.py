"""First block of the bf16 path at the BASELINE configs[4] shape: launch times of its three kernels (HIP events, median).
usage: python3 tools/ab_conv1_bf16.py [B] [H] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd  # noqa: F401
from capsyolo_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 608
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
dev = torch.device('cuda:0')
torch.manual_seed(0)
x = torch.randn(B, 3, H, H, device=dev) * 60
w = torch.randn(128, 3, 3, 3, device=dev) * 0.2
b = torch.zeros(128, device=dev)
sc = torch.rand(128, device=dev) * 0.05 + 0.01
sh = torch.randn(128, device=dev) * 0.3


def med(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


t = med(lambda: ops.conv1_affine_act(x, w, b, sc, sh, 0.1, out_bf16=True))
gb = B * H * H * 128 * 2 / 1e9
print('conv1_affine_act -> bf16: %.3f ms (%.2f GB written: %.2f TB/s)' % (t, gb, gb / t))
t = med(lambda: ops.conv1_affine_act(x, w, b, sc, sh, 0.1, out_bf16=False))
print('conv1_affine_act -> fp32 (fp32 matrix cores): %.3f ms' % t)
