import os, sys
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'dcs-net_amd'))
import torch
from dcsnet import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
rn = lambda *sh: torch.randn(*sh, generator=g).to(dev)
B, Hs, Ws = 32, 128, 128
x1, x2 = rn(B, Hs, Ws, 8, 2), rn(B, Hs, Ws, 8, 2)
w_r, w_i = rn(16, 1, 3, 3) * 0.2, rn(16, 1, 3, 3) * 0.2
b_r, b_i = rn(1), rn(1)
wt, _ = ops.pack_tap_rows(w_r, w_i, 16)
for _ in range(20):
    y = ops.cconv_up2_single(x1, x2, wt, b_r, b_i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    y = ops.cconv_up2_single(x1, x2, wt, b_r, b_i)
e1.record(); torch.cuda.synchronize()
print('cconv_up2_single fwd us', e0.elapsed_time(e1) / 50 * 1e3)
