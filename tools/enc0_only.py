"""Run the first encoder conv forward a few times (kernel timing / counter collection).
usage: python tools/enc0_only.py [B] [T]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from dcsnet import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device('cuda:0')
x = torch.randn(B, 256, T, 1, 2, device=dev)
w_r, w_i = torch.randn(8, 1, 7, 7, device=dev) * 0.05, torch.randn(8, 1, 7, 7, device=dev) * 0.05
b_r, b_i = torch.randn(8, device=dev), torch.randn(8, device=dev)
wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, False, (1, 1))
for _ in range(3):
    y = ops.cconv2d(x, None, wp, bias, (7, 7), (2, 2), (3, 3), (1, 1))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = ops.cconv2d(x, None, wp, bias, (7, 7), (2, 2), (3, 3), (1, 1))
e1.record(); torch.cuda.synchronize()
print(f'enc0 fwd B={B} T={T}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us')
