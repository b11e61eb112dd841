"""The last decoder stage's backward at the train shape: the factored form (tap-sum tensor, 1x1 tap conv gradients, scatter) against the
direct kernels of conv_up1.hip, us per call (events around 50 calls in a replayed graph would hide nothing here: every kernel is > 10 us).
    python tools/dec6_bwd_bench.py [B] [Hs] [Ws] [bf16]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'dcs-net_amd'))
import torch
from dcsnet import ops

args = [a for a in sys.argv[1:] if a != 'bf16']
bf = 'bf16' in sys.argv[1:]
B, Hs, Ws = (int(args[0]) if args else 32), (int(args[1]) if len(args) > 1 else 128), (int(args[2]) if len(args) > 2 else 128)
dev = torch.device('cuda:0')
if bf:
    ops.set_conv_precision('bf16')
dt = torch.bfloat16 if bf else torch.float32
g = torch.Generator().manual_seed(0)
rn = lambda *sh: torch.randn(*sh, generator=g).to(dev)
x1, x2 = rn(B, Hs, Ws, 8, 2).to(dt), rn(B, Hs, Ws, 8, 2).to(dt)
w_r, w_i = rn(16, 1, 3, 3) * 0.2, rn(16, 1, 3, 3) * 0.2
gy = rn(B, 2 * Hs, 2 * Ws, 1, 2)
wt, _ = ops.pack_tap_rows(w_r, w_i, 16)
wtb = ops.pack_conv_weight_bwd(wt, (1, 1))
gb = (torch.empty(1, device=dev), torch.empty(1, device=dev))
sink = (torch.zeros(16, 1, 3, 3, device=dev), torch.zeros(16, 1, 3, 3, device=dev))


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def old_all():
    gz = ops.tapsum((B, Hs, Ws, 16, 2), (3, 3), (2, 2), (1, 1), backward=True, grad=gy, bias_grad=gb, out_dtype=dt)
    gt_r, gt_i, _, _ = ops.cconv2d_bwd_weight(x1, x2, gz, (16, 16, 1, 1), False, (1, 1), (1, 1), (0, 0))
    ops.tap_rows_scatter(gt_r, gt_i, (16, 1, 3, 3), sink)
    ops.cconv2d_bwd_data(gz, wtb, (Hs, Ws, 16), (1, 1), (1, 1), (0, 0), (1, 1), 8)


def old_main():
    gz = ops.tapsum((B, Hs, Ws, 16, 2), (3, 3), (2, 2), (1, 1), backward=True, grad=gy, bias_grad=gb, out_dtype=dt)
    ops.cconv2d_bwd_data(gz, wtb, (Hs, Ws, 16), (1, 1), (1, 1), (0, 0), (1, 1), 8)


print(f'B={B} Hs={Hs} Ws={Ws} {"bf16" if bf else "fp32"} storage; us per call')
print(f'factored: tap sum + data gradient (main chain) {timed(old_main):7.1f}   all six launches {timed(old_all):7.1f}')
print(f'direct:   data gradient                        {timed(lambda: ops.cconv_up2_single_bwd_data(gy, wt, 8, 8, dt)):7.1f}   '
      f'weight + bias gradient {timed(lambda: ops.cconv_up2_single_bwd_weight(gy, x1, x2, (16, 1, 3, 3), sink, gb)):7.1f}')
mb = (gy.numel() * 4 + x1.numel() * x1.element_size() * 2) / 1e6
print(f'(g_y + gradient = {mb:.0f} MB per kernel)')
