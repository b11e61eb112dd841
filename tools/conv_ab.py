"""A/B of the two fp32 MFMA conv schedules (classic one-patch-per-workgroup vs the persistent LDS-DMA pipeline) in ONE
process, interleaved rounds, per layer of the network: forward and data gradient; checks bit-identity on the way.
usage: python tools/conv_ab.py [B] [T] [rounds]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from dcsnet import ops
ops.set_conv_precision('f32')      # the two schedules exist for the native fp32 MFMA path only

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
only = sys.argv[4].split(',') if len(sys.argv) > 4 else None
dev = torch.device('cuda:0')
t8 = T // 8
L = [('enc1', 128, T // 2, 8, 0, 16, 7, (2, 2), (1, 1)),
     ('enc2', 64, T // 4, 16, 0, 32, 5, (2, 2), (1, 1)), ('enc3', 32, t8, 32, 0, 64, 5, (2, 1), (1, 1)),
     ('enc4', 16, t8, 64, 0, 128, 3, (2, 1), (1, 1)), ('enc5', 8, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('enc6', 4, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('dec0', 2, t8, 128, 128, 128, 3, (1, 1), (2, 1)), ('dec1', 4, t8, 128, 128, 128, 3, (1, 1), (2, 1)),
     ('dec2', 8, t8, 128, 128, 64, 3, (1, 1), (2, 1)), ('dec3', 16, t8, 64, 64, 32, 3, (1, 1), (2, 1)),
     ('dec4', 32, t8, 32, 32, 16, 3, (1, 1), (2, 2)), ('dec5', 64, T // 4, 16, 16, 8, 3, (1, 1), (2, 2))]


def timeit(fn, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


tot = {}
print(f'B={B} T={T}   us (median of {R} interleaved rounds) [algorithmic TFLOP/s]')
for name, H, W, C1, C2, Cout, k, st, up in L:
    if only and name not in only:
        continue
    tr = name.startswith('dec')
    Cin = C1 + C2
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    g = torch.Generator(device='cpu').manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    w_r, w_i = rnd(*wshape) * 0.05, rnd(*wshape) * 0.05
    b_r, b_i = rnd(Cout), rnd(Cout)
    x1 = rnd(B, H, W, C1, 2)
    x2 = rnd(B, H, W, C2, 2) if C2 else None
    pad = (k // 2, k // 2)
    wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, tr, up)
    ops.set_conv_schedule('classic')
    y0 = ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up)
    gy = torch.randn(y0.shape, generator=g).to(dev)
    wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, pad, up)
    d0 = ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, up, C1)
    ops.set_conv_schedule('pipe')
    y1 = ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up)
    d1 = ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, up, C1)
    torch.cuda.synchronize()
    same = torch.equal(y0, y1) and all((a is None and b is None) or torch.equal(a, b) for a, b in zip(d0, d1))
    gflop = 8.0 * y0.shape[0] * y0.shape[1] * y0.shape[2] * Cout * Cin * k * k / 1e9
    fns = {'fwd': lambda: ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up),
           'dgrad': lambda: ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, up, C1)}
    res = {}
    for what, fn in fns.items():
        ts = {'classic': [], 'pipe': []}
        for r in range(R):
            for mode in ('classic', 'pipe'):
                ops.set_conv_schedule(mode)
                if r == 0:
                    fn(); fn()
                ts[mode].append(timeit(fn))
        res[what] = {m_: sorted(v)[len(v) // 2] for m_, v in ts.items()}
        for m_ in res[what]:
            tot[(what, m_)] = tot.get((what, m_), 0.0) + res[what][m_]
            tot[(what, 'gf')] = tot.get((what, 'gf'), 0.0) + gflop / 2
    f, d = res['fwd'], res['dgrad']
    tf = lambda us: gflop / us * 1e3
    print(f"{name}: {gflop:6.2f} GF  identical={same} | fwd classic {f['classic']:7.1f} [{tf(f['classic']):5.1f}] pipe {f['pipe']:7.1f} "
          f"[{tf(f['pipe']):5.1f}] | dgrad classic {d['classic']:7.1f} [{tf(d['classic']):5.1f}] pipe {d['pipe']:7.1f} [{tf(d['pipe']):5.1f}]",
          flush=True)
for what in ('fwd', 'dgrad'):
    if (what, 'classic') in tot:
        print(f"total {what}: classic {tot[(what, 'classic')] / 1e3:.3f} ms, pipe {tot[(what, 'pipe')] / 1e3:.3f} ms")
