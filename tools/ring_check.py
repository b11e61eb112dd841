"""The producer / consumer conv kernel (csrc/conv_ring.hip) against the classic MFMA kernel on the network's layer shapes:
results (same operands, same accumulation order per element where the classic plan does not split K) and per-launch time.
usage: python tools/ring_check.py [B] [T] [layer ...] [bf16]   (bf16: bf16 activation storage, the _h entry points)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from dcsnet import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
only = [x for x in sys.argv[3:] if x != 'bf16']
BF16 = 'bf16' in sys.argv[3:]
dev = torch.device('cuda:0')
if BF16:
    ops.set_conv_precision('bf16')
adt = torch.bfloat16 if BF16 else torch.float32
t8 = T // 8
L = [('enc2', 64, T // 4, 16, 0, 32, 5, (2, 2), (1, 1)), ('enc3', 32, t8, 32, 0, 64, 5, (2, 1), (1, 1)),
     ('enc4', 16, t8, 64, 0, 128, 3, (2, 1), (1, 1)), ('enc5', 8, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('enc6', 4, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('dec0', 2, t8, 128, 128, 128, 3, (1, 1), (2, 1)), ('dec1', 4, t8, 128, 128, 128, 3, (1, 1), (2, 1)),
     ('dec2', 8, t8, 128, 128, 64, 3, (1, 1), (2, 1)), ('dec3', 16, t8, 64, 64, 32, 3, (1, 1), (2, 1)),
     ('dec4', 32, t8, 32, 32, 16, 3, (1, 1), (2, 2))]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def both(fn):
    out = {}
    for mode in ('0', '1'):
        os.environ['DCS_CONV_RING'] = mode
        r = fn()
        torch.cuda.synchronize()
        out[mode] = (r, timeit(fn))
    return out


def diff(a, b):
    if isinstance(a, (tuple, list)):
        return max(diff(x, y) for x, y in zip(a, b) if x is not None)
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


tot = [0.0, 0.0, 0.0, 0.0]
print(f'B={B} T={T}: us classic -> ring (max |diff| / max |value|)')
for name, H, W, C1, C2, Cout, k, st, up in L:
    if only and name not in only:
        continue
    tr = name.startswith('dec')
    Cin = C1 + C2
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    torch.manual_seed(1)
    w_r, w_i = torch.randn(wshape, device=dev) * 0.05, torch.randn(wshape, device=dev) * 0.05
    b_r, b_i = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
    x1 = torch.randn(B, H, W, C1, 2, device=dev).to(adt)
    x2 = torch.randn(B, H, W, C2, 2, device=dev).to(adt) if C2 else None
    pad = (k // 2, k // 2)
    wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, tr, up)
    f = both(lambda: ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up))
    y = f['0'][0]
    gy = torch.randn_like(y)
    wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, pad, up)
    d = both(lambda: ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, up, C1))
    tot[0] += f['0'][1]; tot[1] += f['1'][1]; tot[2] += d['0'][1]; tot[3] += d['1'][1]
    print(f'{name}: fwd {f["0"][1]:6.1f} -> {f["1"][1]:6.1f} ({diff(f["1"][0], f["0"][0]):.1e}) | '
          f'dgrad {d["0"][1]:6.1f} -> {d["1"][1]:6.1f} ({diff(d["1"][0], d["0"][0]):.1e})', flush=True)
print(f'total fwd {tot[0]:.1f} -> {tot[1]:.1f} us, dgrad {tot[2]:.1f} -> {tot[3]:.1f} us')
