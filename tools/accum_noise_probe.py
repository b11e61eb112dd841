"""Rounding noise of the gradient kernels against fp64, next to torch-CPU fp32 on the same operands:
err / sum|terms| per output element (rms and max).  usage: python tools/accum_noise_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from dcsnet import ops
import torch.nn.functional as TF
dev = torch.device('cuda:0')
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
torch.manual_seed(0)


def wgrad_case(name, B, H, W, Cin, Cout, k, st):
    pad = k // 2
    xr, xi = torch.randn(B, Cin, H, W), torch.randn(B, Cin, H, W)
    Ho, Wo = (H + 2 * pad - k) // st[0] + 1, (W + 2 * pad - k) // st[1] + 1
    gr, gi = torch.randn(B, Cout, Ho, Wo), torch.randn(B, Cout, Ho, Wo)
    # complex conv y = conv(x, w): gw_r = corr(x_r, g_r) + corr(x_i, g_i)  (conv_r sees x_r -> y_r and x_i -> y_i)
    def ref(dt):
        f = lambda a, g: torch.nn.grad.conv2d_weight(a.to(dt), (Cout, Cin, k, k), g.to(dt), stride=st, padding=pad)
        return f(xr, gr) + f(xi, gi)
    w64, w32 = ref(torch.float64), ref(torch.float32).double()
    mag = (torch.nn.grad.conv2d_weight(xr.abs().double(), (Cout, Cin, k, k), gr.abs().double(), stride=st, padding=pad) +
           torch.nn.grad.conv2d_weight(xi.abs().double(), (Cout, Cin, k, k), gi.abs().double(), stride=st, padding=pad))
    x = torch.stack([xr, xi], -1).permute(0, 2, 3, 1, 4).contiguous().to(dev)
    g = torch.stack([gr, gi], -1).permute(0, 2, 3, 1, 4).contiguous().to(dev)
    gw_r, gw_i, _, _ = ops.cconv2d_bwd_weight(x, None, g, (Cout, Cin, k, k), True, (k, k), st, (pad, pad))
    h = gw_r.cpu().double()
    for tag, v in (('hip', h), ('cpu32', w32)):
        e = (v - w64).abs() / mag
        print(f'{name:8s} wgrad {tag:6s} err/sum|terms|: rms {float(e.pow(2).mean().sqrt()):.2e} max {float(e.max()):.2e}')


wgrad_case('enc4', 32, 16, 32, 64, 128, 3, (2, 1))
wgrad_case('enc1', 32, 128, 128, 8, 16, 7, (2, 2))
wgrad_case('enc6', 32, 4, 32, 128, 128, 3, (2, 1))
wgrad_case('dec1like', 32, 8, 32, 256, 128, 3, (1, 1))
