"""Train-step time of CapsuleNet (capsule: 32x32 GTSRB-shaped, batch 32, recon on; BASELINE configs[0] shape) on the GPU."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd
from capsyolo_amd import loss_fns, models, ops, optim, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
p = types.SimpleNamespace(n_classes=43, dropout=0.0, recon=True, recon_coef=5e-4, device='cuda', model='capsule')
torch.manual_seed(0)
net = models.CapsuleNet(p).cuda().train()
opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
x = torch.from_numpy(synth.images(B, 32)).permute(0, 3, 1, 2).contiguous().cuda()
y = torch.from_numpy(synth.gtsrb_labels(B, 43)).cuda()


def step():
    out, recon = net(x, y, True)
    loss = loss_fns.capsule_loss(out, y, p, x, recon)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
ops.timer.reset(); ops.timer.enabled = True
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n):
    loss = step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ops.timer.enabled = False
print('capsule batch %d: %.3f ms/step, %.1f images/s, loss %.4f' % (B, 1e3 * dt / n, B * n / dt, loss.item()))
tot = dict((k, ms * cnt / n) for k, (cnt, ms) in ops.timer.summary().items())
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print('  %-36s %8.3f ms/step' % (k, v))
print('  timed kernels total %.2f ms/step' % sum(tot.values()))
