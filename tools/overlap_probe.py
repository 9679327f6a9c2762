"""Can an MFMA-bound conv kernel and HBM-bound elementwise kernels share the GPU on two streams?
Times conv_3's weight gradient + the BatchNorm backward passes of layer 2, back to back and concurrently."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd
from capsyolo_amd import ops
from capsyolo_amd._lib import call

dev = torch.device('cuda:0')
B, H = 32, 416
x = torch.randn(B, H, H, 256, device=dev)
dz3 = torch.randn(B, H // 2, H // 2, 64, device=dev)
z = torch.randn(B * H * H, 256, device=dev)
da = torch.randn(B * H * H, 256, device=dev)
out = torch.empty_like(z)
N = 256
sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
mu, isd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
red = torch.zeros(N, 2, dtype=torch.float64, device=dev)
side = torch.cuda.Stream()


def bn_bwd(stream):
    st = stream.cuda_stream
    P = z.shape[0]
    call('cy_bn_bwd_reduce', z.data_ptr(), da.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), 0.1, red.data_ptr(), P, N, st)
    call('cy_bn_bwd_apply', z.data_ptr(), da.data_ptr(), out.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), None, 0.1, red.data_ptr(), None, None, P, N, st)


def wgrad():
    return ops.conv_wgrad(x, dz3, 4, 2, 1)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


main = torch.cuda.current_stream()
t_w = timed(wgrad)
t_b = timed(lambda: bn_bwd(main))


def both_serial():
    wgrad(); bn_bwd(main)


def both_concurrent():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        wgrad()
    bn_bwd(main)
    main.wait_stream(side)


print('conv_3 wgrad alone %.3f ms, BN2 backward alone %.3f ms, serial %.3f ms, two streams %.3f ms' %
      (t_w, t_b, timed(both_serial), timed(both_concurrent)))
