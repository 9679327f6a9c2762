"""Train-step time of DarkNet (darknet_d: 416x416, n_grid 13, 2 boxes, batch 16; BASELINE configs[1]) with the kernel timer."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd
from capsyolo_amd import loss_fns, models, ops, optim, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2:      # A/B of the F(4x4,.) pixel threshold: python tools/bench_darknet.py 16 <log2 pixels>
    ops.WINOGRAD4_MIN_PIXELS = 1 << int(sys.argv[2])
p = types.SimpleNamespace(n_classes=0, n_grid=13, n_boxes=2, dropout=0.0, darknet_input=416, device='cuda', model='darknet_d',
                          l_coord=5.0, l_noobj=0.5)
torch.manual_seed(0)
net = models.DarkNet(p).cuda().train()
opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
x = torch.from_numpy(synth.images(B, 416)).permute(0, 3, 1, 2).contiguous().cuda()
y = torch.from_numpy(synth.gtsdb_labels(B, 13, 0)).cuda()


def step():
    out = net(x)
    loss = loss_fns.dark_loss(out, y, p)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
n = 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    loss = step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('darknet_d batch %d: %.2f ms/step, %.1f images/s, loss %.4f' % (B, 1e3 * dt / n, B * n / dt, loss.item()))
# second pass with the per-launch HIP-event timer for the breakdown (252 launches per step: the brackets cost ~20 % here)
ops.timer.reset(); ops.timer.enabled = True
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
ops.timer.enabled = False
tot = {}
for k, (cnt, ms) in ops.timer.summary().items():
    tot[k] = ms * cnt / n
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print('  %-36s %8.3f ms/step' % (k, v))
print('  timed kernels total %.2f ms/step' % sum(tot.values()))
