import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/dcs-net_amd')
from dcsnet import ops
dev = torch.device('cuda:0')
cases = [('enc1', 64, 64, 8, 0, 16, 7, (2, 2), (1, 1), False), ('enc2', 32, 32, 16, 0, 32, 5, (2, 2), (1, 1), False),
         ('enc5', 8, 32, 128, 0, 128, 3, (2, 1), (1, 1), False), ('dec1', 4, 32, 128, 128, 128, 3, (1, 1), (2, 1), True),
         ('dec4', 32, 32, 32, 32, 16, 3, (1, 1), (2, 2), True), ('fc1x1', 1, 64, 128, 0, 128, 1, (1, 1), (1, 1), False),
         ('plain64', 16, 16, 64, 0, 64, 3, (1, 1), (1, 1), False)]
for name, H, W, C1, C2, Cout, k, st, up, tr in cases:
    torch.manual_seed(1)
    Cin = C1 + C2
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    w_r, w_i = torch.randn(wshape, device=dev) * 0.05, torch.randn(wshape, device=dev) * 0.05
    b_r, b_i = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
    x1 = torch.randn(4, H, W, C1, 2, device=dev)
    x2 = torch.randn(4, H, W, C2, 2, device=dev) if C2 else None
    pad = (k // 2, k // 2)
    outs = {}
    for mode in ('f32', 'bf16'):
        ops.set_conv_precision(mode)
        wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, tr, up)
        y = ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up)
        gy = torch.ones_like(y) * 0.5 + y * 0
        wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, pad, up)
        gx1, gx2 = ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, up, C1)
        outs[mode] = (y.clone(), gx1.clone())
    ops.set_conv_precision('f32')
    ey = float((outs['bf16'][0] - outs['f32'][0]).abs().max()) / float(outs['f32'][0].abs().max())
    eg = float((outs['bf16'][1] - outs['f32'][1]).abs().max()) / float(outs['f32'][1].abs().max())
    print(f'{name:8s} fwd rel {ey:.3e}  dgrad rel {eg:.3e}')

# detail for the failing data gradient
name, H, W, C1, C2, Cout, k, st, up, tr = cases[4]
torch.manual_seed(1)
Cin = C1 + C2
w_r, w_i = torch.randn((Cin, Cout, k, k), device=dev) * 0.05, torch.randn((Cin, Cout, k, k), device=dev) * 0.05
res = {}
for mode in ('f32', 'bf16'):
    ops.set_conv_precision(mode)
    wp, bias = ops.pack_conv_weight(w_r, w_i, None, None, True, up)
    gy = torch.randn(4, H * up[0], W * up[1], Cout, 2, device=dev)
    torch.manual_seed(3); gy = torch.randn_like(gy)
    wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, (1, 1), up)
    gx1, gx2 = ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, (1, 1), up, C1)
    res[mode] = (gx1.clone(), gx2.clone())
ops.set_conv_precision('f32')
for i in (0, 1):
    a, b = res['bf16'][i], res['f32'][i]
    print('gx%d' % (i + 1), 'bf16 absmax', float(a.abs().max()), 'f32 absmax', float(b.abs().max()), 'nan', bool(torch.isnan(a).any()),
          'relerr', float((a - b).abs().max()) / float(b.abs().max()))
