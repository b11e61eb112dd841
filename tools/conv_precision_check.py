"""Error of the MFMA conv forward / data gradient against an fp64 reference in each precision mode
('f32' exact fp32 MFMA, 'bf16' rounded operands, 'bf16x6' fp32 emulated on the bf16 MFMA), and the layer times.
usage: python tools/conv_precision_check.py"""
import os, sys, torch
import torch.nn.functional as tF
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from dcsnet import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)


def ref_conv(x, w_r, w_i, b_r, b_i, st, pad, gy=None):
    """fp64 complex conv with the two-real-layer bias convention (re: b_r - b_i, im: b_r + b_i); with gy also the
    fp64 gradient with respect to x"""
    x64 = x.double().cpu().requires_grad_(gy is not None)
    xr, xi = x64[..., 0].permute(0, 3, 1, 2), x64[..., 1].permute(0, 3, 1, 2)
    wr, wi = w_r.double().cpu(), w_i.double().cpu()
    c = lambda a, w: tF.conv2d(a, w, None, st, pad)
    yr = c(xr, wr) - c(xi, wi) + (b_r - b_i).double().cpu()[None, :, None, None]
    yi = c(xi, wr) + c(xr, wi) + (b_r + b_i).double().cpu()[None, :, None, None]
    y = torch.stack((yr, yi), -1).permute(0, 2, 3, 1, 4)
    if gy is None:
        return y.detach()
    wr.requires_grad_(True); wi.requires_grad_(True)
    yr = c(xr, wr) - c(xi, wi)
    yi = c(xi, wr) + c(xr, wi)
    y2 = torch.stack((yr, yi), -1).permute(0, 2, 3, 1, 4)
    (y2 * gy.double().cpu()).sum().backward()
    return y.detach(), x64.grad, wr.grad, wi.grad


cases = [('3x3 s(2,1) 64->128', 4, 16, 32, 64, 128, 3, (2, 1)), ('5x5 s(2,2) 16->32', 4, 64, 64, 16, 32, 5, (2, 2)),
         ('3x3 s1 128->128', 2, 8, 32, 128, 128, 3, (1, 1))]
for name, B, H, W, Cin, Cout, k, st in cases:
    x = torch.randn(B, H, W, Cin, 2, device=dev)
    w_r, w_i = torch.randn(Cout, Cin, k, k, device=dev) * 0.05, torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    b_r, b_i = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
    pad = (k // 2, k // 2)
    Ho, Wo = (H + 2 * pad[0] - k) // st[0] + 1, (W + 2 * pad[1] - k) // st[1] + 1
    gy = torch.randn(B, Ho, Wo, Cout, 2, device=dev)
    ref, gx_ref, gwr_ref, gwi_ref = ref_conv(x, w_r, w_i, b_r, b_i, st, pad, gy)
    print(name)
    for mode in ('f32', 'bf16', 'bf16x6'):
        ops.set_conv_precision(mode)
        wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, False, (1, 1))
        y = ops.cconv2d(x, None, wp, bias, (k, k), st, pad, (1, 1))
        wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, pad, (1, 1))
        gx = ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, (1, 1), Cin)
        gx = gx[0] if isinstance(gx, tuple) else gx
        e, ge = y.double().cpu() - ref, gx.double().cpu() - gx_ref
        gw = ops.cconv2d_bwd_weight(x, None, gy, tuple(w_r.shape), True, (k, k), st, pad, (1, 1), False)
        we = torch.cat(((gw[0].double().cpu() - gwr_ref).flatten(), (gw[1].double().cpu() - gwi_ref).flatten()))
        wn = torch.cat((gwr_ref.flatten(), gwi_ref.flatten()))
        print(f'  {mode:7s} forward: max|err|/max|y| {e.abs().max().item() / ref.abs().max().item():.3e}  rel-L2 '
              f'{(e.norm() / ref.norm()).item():.3e}   data gradient: max {ge.abs().max().item() / gx_ref.abs().max().item():.3e}'
              f'  rel-L2 {(ge.norm() / gx_ref.norm()).item():.3e}   weight gradient: rel-L2 {(we.norm() / wn.norm()).item():.3e}')
ops.set_conv_precision('f32')

# every MFMA layer of the network at the bench's shapes: bf16x6 against the fp32 MFMA (forward and data gradient)
for B, T in ((32, 256), (16, 2000)):
    t8 = T // 8
    L = [('enc1', 128, T // 2, 8, 0, 16, 7, (2, 2), (1, 1)),
         ('enc2', 64, T // 4, 16, 0, 32, 5, (2, 2), (1, 1)), ('enc3', 32, t8, 32, 0, 64, 5, (2, 1), (1, 1)),
         ('enc4', 16, t8, 64, 0, 128, 3, (2, 1), (1, 1)), ('enc5', 8, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
         ('enc6', 4, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
         ('dec0', 2, t8, 128, 128, 128, 3, (1, 1), (2, 1)), ('dec1', 4, t8, 128, 128, 128, 3, (1, 1), (2, 1)),
         ('dec2', 8, t8, 128, 128, 64, 3, (1, 1), (2, 1)), ('dec3', 16, t8, 64, 64, 32, 3, (1, 1), (2, 1)),
         ('dec4', 32, t8, 32, 32, 16, 3, (1, 1), (2, 2)), ('dec5', 64, T // 4, 16, 16, 8, 3, (1, 1), (2, 2))]
    print(f'B={B} T={T}: max |bf16x6 - f32| / max |f32|')
    for name, H, W, C1, C2, Cout, k, st, up in L:
        tr = name.startswith('dec')
        Cin = C1 + C2
        wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
        w_r, w_i = torch.randn(wshape, device=dev) * 0.05, torch.randn(wshape, device=dev) * 0.05
        b_r, b_i = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
        x1 = torch.randn(B, H, W, C1, 2, device=dev)
        x2 = torch.randn(B, H, W, C2, 2, device=dev) if C2 else None
        pad = (k // 2, k // 2)
        out = {}
        gy = None
        for mode in ('f32', 'bf16x6'):
            ops.set_conv_precision(mode)
            wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, tr, up)
            y = ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up)
            gy = torch.randn_like(y) if gy is None else gy
            wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, pad, up)
            gx = ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, up, C1)
            gx = gx if isinstance(gx, tuple) else (gx,)
            out[mode] = (y,) + tuple(g for g in gx if g is not None)
        d = [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(out['bf16x6'], out['f32'])]
        print(f'  {name}: fwd {d[0]:.2e}  dgrad ' + ' '.join(f'{v:.2e}' for v in d[1:]))
        del x1, x2, out, gy
ops.set_conv_precision('f32')
