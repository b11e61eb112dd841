"""Phase times inside the enc0 forward kernel.  Needs a -DDCS_ENC0_DIAG build of the library:
  python tools/enc0_diag.py --build   (here, no GPU: writes dcs-net_amd/lib/diag/libdcsnet_enc0diag.so), then on the GPU box
  DCS_LIB_PATH=dcs-net_amd/lib/diag/libdcsnet_enc0diag.so python tools/enc0_diag.py [B] [T]
usage: python tools/enc0_diag.py [B] [T].  Per persistent workgroup: fill = tile-loop top -> after the barrier (waits for the
prefetched patch, LDS writes), compute = MFMAs + stores; core clocks printed as us at 2.4 GHz."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
if '--build' in sys.argv:
    import build
    out = os.path.join(ROOT, 'dcs-net_amd', 'lib', 'diag')
    os.makedirs(out, exist_ok=True)
    build.build(flags=build.FLAGS + ['-DDCS_ENC0_DIAG'], verbose=False, lib=os.path.join(out, 'libdcsnet_enc0diag.so'), objdir=os.path.join(out, 'obj'))
    sys.exit(0)
from dcsnet import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.dcs_debug_set_enc0_buffer.argtypes = [ctypes.c_void_p]
w_r, w_i = torch.randn(8, 1, 7, 7, device=dev) * 0.05, torch.randn(8, 1, 7, 7, device=dev) * 0.05
b_r, b_i = torch.randn(8, device=dev), torch.randn(8, device=dev)
x = torch.randn(B, 256, T, 1, 2, device=dev)
wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, False, (1, 1))
run = lambda: ops.cconv2d(x, None, wp, bias, (7, 7), (2, 2), (3, 3), (1, 1))
for _ in range(3):
    run()
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
lib.dcs_debug_set_enc0_buffer(dbg.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.dcs_debug_set_enc0_buffer(None)
d = dbg.view(-1, 8).cpu().double()
d = d[d[:, 2] > 0]
us = lambda v: float(v) / 2400.0
print(f'enc0 fwd B={B} T={T}: call {e0.elapsed_time(e1) * 1e3:.1f} us | {d.shape[0]} workgroups, tiles/WG {float(d[:, 3].mean()):.1f} | per WG (us): '
      f'fill {us(d[:, 0].mean()):.1f}  compute {us(d[:, 1].mean()):.1f}  life {us(d[:, 2].mean()):.1f} (max {us(d[:, 2].max()):.1f}) | '
      f'per tile: fill {us(d[:, 0].sum() / d[:, 3].sum()):.2f}  compute {us(d[:, 1].sum() / d[:, 3].sum()):.2f} | '
      f'prologue (B fragments) {us(d[:, 7].mean()):.2f}')
hw = d[:, 5].long()
cu = (d[:, 6].long() & 15) * 4096 + ((hw >> 8) & 0xff)
ids, inv, cnt = torch.unique(cu, return_inverse=True, return_counts=True)
per = cnt[inv]
print('   ' + ', '.join(f'{int(k)} WGs/CU: {int((cnt == k).sum())} CUs, life {us(d[per == k, 2].mean()):.1f}' for k in torch.unique(cnt)))
lf = torch.sort(d[:, 2]).values
n = lf.shape[0]
print('   life p5/p25/p50/p75/p95: ' + ' '.join(f'{us(lf[int(f * (n - 1))]):.1f}' for f in (.05, .25, .5, .75, .95)),
      '| start spread', us(d[:, 4].max() - d[:, 4].min()))
# s_memtime counts per XCD: spans are only comparable inside one
xcd = d[:, 6].long() & 15
for x in torch.unique(xcd)[:8]:
    m = xcd == x
    st_, en_ = d[m, 4], d[m, 4] + d[m, 2]
    o = torch.sort(st_ - st_.min()).values
    print(f'   XCD {int(x)}: {int(m.sum())} WGs, first start -> last start {us(o[-1]):.1f} us (p50 {us(o[len(o) // 2]):.1f}), first start -> last end {us(en_.max() - st_.min()):.1f} us')
