"""Forward of the C > 1 routing heads on the MFMA kernel (routing_mfma.hip) against the vector kernel (routing_rows.hip,
CY_ROUTING_MFMA=0): agreement of v and s_hist, and launch times (HIP events, median of `reps`)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd
from capsyolo_amd import ops


def run(u, W, r, g, B, mfma, reps):
    os.environ['CY_ROUTING_MFMA'] = '1' if mfma else '0'
    v = ops.routing(u, W, r, g, B if g else 0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.routing(u, W, r, g, B if g else 0); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return v, ts[len(ts) // 2]


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    g = 13
    out = {}
    for name, R, N, C, Dout, gg in (('dcn3_head', g * g * B, 512, 43, 21, g), ('capsule_head', B, 1296, 43, 16, 0),
                                    ('rows2000_c20_d16', 2000, 256, 20, 16, 0), ('rows37_c7_d21', 37, 70, 7, 21, 0)):
        u = torch.randn(B, 4 * gg, 4 * gg, 256, device=dev) if gg else torch.randn(R, N, 8, device=dev)
        W = 0.1 * torch.randn(1, N, C, 8, Dout, device=dev)
        v0, t0 = run(u, W, 3, gg, B, False, reps)
        v1, t1 = run(u, W, 3, gg, B, True, reps)
        err = float((v1 - v0).abs().max() / v0.abs().max())
        out[name] = {'R': R, 'N': N, 'C': C, 'Dout': Dout, 'vector_ms': round(t0, 4), 'mfma_ms': round(t1, 4),
                     'max_rel_diff_v': err, 'nan': bool(torch.isnan(v1).any())}
        print(name, out[name], flush=True)
    os.environ.pop('CY_ROUTING_MFMA', None)
    print(json.dumps(out))
