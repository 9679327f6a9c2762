import sys; sys.path.insert(0,'/root/repo')
import torch, torch.nn.functional as F
import capsyolo_amd
from capsyolo_amd import ops
torch.manual_seed(0)
for (B,Cin,H,Cout) in [(4,256,64,64),(4,64,32,128),(4,128,16,256)]:
    x = torch.randn(B,Cin,H,H); w = torch.randn(Cout,Cin,4,4)*(1.0/(Cin*16))**0.5; b = torch.randn(Cout)*0.1
    zr = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    xg = x.permute(0,2,3,1).contiguous().cuda()
    res = {}
    for flag in (True, False):
        ops.USE_WINOGRAD_S2 = flag
        stats = torch.zeros((ops.STATS_COPIES, Cout, 2), dtype=torch.float64, device='cuda')
        z = ops.conv_forward(xg, w.cuda(), b.cuda(), 4, 2, 1, False, stats)
        err = (z.permute(0,3,1,2).cpu().double() - zr)
        s = stats.sum(0).cpu()
        e1 = (s[:,0] - zr.sum(dim=(0,2,3))).abs().max().item() / zr.sum(dim=(0,2,3)).abs().max().item()
        e2 = (s[:,1] - (zr**2).sum(dim=(0,2,3))).abs().max().item() / (zr**2).sum(dim=(0,2,3)).abs().max().item()
        res[flag] = (err.abs().max().item()/zr.abs().max().item(), (err.norm()/zr.norm()).item(), e1, e2)
    ops.USE_WINOGRAD_S2 = True
    print((B,Cin,H,Cout), 'winograd max/rel-L2/stats1/stats2 %.2e %.2e %.2e %.2e | direct %.2e %.2e %.2e %.2e' % (res[True]+res[False]))
