"""Per-layer timing of the complex conv kernels (forward, data gradient, weight gradient) at the network's
shapes.  usage: python tools/conv_layers_bench.py [B] [T] [f32|bf16]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from dcsnet import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
if len(sys.argv) > 3:  # f32 | bf16 | bf16x6
    ops.set_conv_precision(sys.argv[3])
# name, Hin, Win, C1, C2, Cout, k, stride, up, transposed
t8 = T // 8
L = [('enc0', 256, T, 1, 0, 8, 7, (2, 2), (1, 1)), ('enc1', 128, T // 2, 8, 0, 16, 7, (2, 2), (1, 1)),
     ('enc2', 64, T // 4, 16, 0, 32, 5, (2, 2), (1, 1)), ('enc3', 32, t8, 32, 0, 64, 5, (2, 1), (1, 1)),
     ('enc4', 16, t8, 64, 0, 128, 3, (2, 1), (1, 1)), ('enc5', 8, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('enc6', 4, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('dec0', 2, t8, 128, 128, 128, 3, (1, 1), (2, 1)), ('dec1', 4, t8, 128, 128, 128, 3, (1, 1), (2, 1)),
     ('dec2', 8, t8, 128, 128, 64, 3, (1, 1), (2, 1)), ('dec3', 16, t8, 64, 64, 32, 3, (1, 1), (2, 1)),
     ('dec4', 32, t8, 32, 32, 16, 3, (1, 1), (2, 2)), ('dec5', 64, T // 4, 16, 16, 8, 3, (1, 1), (2, 2)),
     ('dec6', 128, T // 2, 8, 8, 1, 3, (1, 1), (2, 2))]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
gf_tot = 0.0
print(f'B={B} T={T}   (us, algorithmic TFLOP/s)')
for name, H, W, C1, C2, Cout, k, st, up in L:
    tr = name.startswith('dec')
    Cin = C1 + C2
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    w_r, w_i = torch.randn(wshape, device=dev) * 0.05, torch.randn(wshape, device=dev) * 0.05
    b_r, b_i = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
    x1 = torch.randn(B, H, W, C1, 2, device=dev)
    x2 = torch.randn(B, H, W, C2, 2, device=dev) if C2 else None
    pad = (k // 2, k // 2)
    wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, tr, up)
    y = ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up)
    gy = torch.randn_like(y)
    wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, pad, up)
    gflop = 8.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * Cin * k * k / 1e9
    f = timeit(lambda: ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up))
    d = timeit(lambda: ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, up, C1))
    w = timeit(lambda: ops.cconv2d_bwd_weight(x1, x2, gy, wshape, True, (k, k), st, pad, up, tr))
    tot['fwd'] += f; tot['dgrad'] += d; tot['wgrad'] += w; gf_tot += gflop
    tf = lambda us: gflop / us * 1e3
    print(f'{name}: {gflop:7.2f} GF | fwd {f:7.1f} us {tf(f):6.1f} TF | dgrad {d:7.1f} us {tf(d):6.1f} TF | wgrad {w:7.1f} us {tf(w):6.1f} TF')
print(f'total {gf_tot:.1f} GF: fwd {tot["fwd"]/1e3:.2f} ms ({gf_tot/tot["fwd"]*1e3:.1f} TF), dgrad {tot["dgrad"]/1e3:.2f} ms '
      f'({gf_tot/tot["dgrad"]*1e3:.1f} TF), wgrad {tot["wgrad"]/1e3:.2f} ms ({gf_tot/tot["wgrad"]*1e3:.1f} TF)')
