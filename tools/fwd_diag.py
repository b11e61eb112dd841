"""Phase times inside the MFMA forward / data-gradient conv kernel (needs a diagnostic build:
DCS_EXTRA_HIPCC_FLAGS=-DDCS_FWD_DIAG python dcs-net_amd/build.py).  usage: python tools/fwd_diag.py [B] [T] [layers] [fwd|dgrad]
Per workgroup (wave 0): setup = start -> chunk loop (table build, first B loads), gather = loop-top barrier -> second
barrier per chunk, mfma = tap loop, epilogue = stores; s_memtime = core clocks, printed as us at 2.4 GHz."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from dcsnet import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
only = sys.argv[3].split(',') if len(sys.argv) > 3 and sys.argv[3] != 'all' else None
what = sys.argv[4] if len(sys.argv) > 4 else 'fwd'
dev = torch.device('cuda:0')
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.dcs_debug_set_fwd_buffer.argtypes = [ctypes.c_void_p]
t8 = T // 8
L = [('enc1', 128, T // 2, 8, 0, 16, 7, (2, 2), (1, 1)),
     ('enc2', 64, T // 4, 16, 0, 32, 5, (2, 2), (1, 1)), ('enc3', 32, t8, 32, 0, 64, 5, (2, 1), (1, 1)),
     ('enc4', 16, t8, 64, 0, 128, 3, (2, 1), (1, 1)), ('enc5', 8, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('enc6', 4, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('dec0', 2, t8, 128, 128, 128, 3, (1, 1), (2, 1)), ('dec1', 4, t8, 128, 128, 128, 3, (1, 1), (2, 1)),
     ('dec2', 8, t8, 128, 128, 64, 3, (1, 1), (2, 1)), ('dec3', 16, t8, 64, 64, 32, 3, (1, 1), (2, 1)),
     ('dec4', 32, t8, 32, 32, 16, 3, (1, 1), (2, 2))]
us = lambda v: float(v) / 2400.0
for name, H, W, C1, C2, Cout, k, st, up in L:
    if only and name not in only:
        continue
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
    if what == 'fwd':
        run = lambda: ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up)
    else:
        run = lambda: ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, up, C1)
    for _ in range(3):
        run()
    dbg = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
    lib.dcs_debug_set_fwd_buffer(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    lib.dcs_debug_set_fwd_buffer(None)
    d = dbg.view(-1, 8).cpu().double()
    d = d[d[:, 3] > 0]
    n = d.shape[0]
    if n == 0:
        print(f'{name}: no stamps (not on cconv_mfma_kernel)')
        continue
    t0 = d[:, 5].min()
    span = (d[:, 5] + d[:, 3]).max() - t0
    hw = d[:, 6].long()
    cu = (d[:, 7].long() & 15) * 4096 + ((hw >> 8) & 0xff)
    ids, inv, cnt = torch.unique(cu, return_inverse=True, return_counts=True)
    lf = torch.sort(d[:, 3]).values
    q = lambda f: us(lf[int(f * (n - 1))])
    st_ = torch.sort(d[:, 5] - t0).values
    print(f'{name} {what}: call {e0.elapsed_time(e1) * 1e3:6.1f} us, MFMA-bound {gflop / 157.3 * 1e3:5.1f} us | {n} WGs on {len(ids)} CUs '
          f'({float(cnt.float().mean()):.1f}/CU, max {int(cnt.max())}) | per WG (us): setup {us(d[:, 4].mean()):4.1f} gather {us(d[:, 0].mean()):5.1f} '
          f'mfma {us(d[:, 1].mean()):5.1f} epilogue {us(d[:, 2].mean()):4.1f} life {us(d[:, 3].mean()):5.1f} | first start -> last end {us(span):6.1f}')
    # co-resident pairs: workgroup ids sharing a CU
    wgid = torch.nonzero(dbg.view(-1, 8)[:, 3].cpu() > 0).flatten()
    pairs = {}
    for w_, c_ in zip(wgid.tolist(), cu.tolist()):
        pairs.setdefault(c_, []).append(w_)
    ex = list(pairs.values())[:3]
    print(f'      WG ids on the first CUs: {ex}')
    if os.environ.get('DCS_FDIAG_DUMP'):
        import numpy as np
        os.makedirs(os.environ['DCS_FDIAG_DUMP'], exist_ok=True)
        np.save(os.path.join(os.environ['DCS_FDIAG_DUMP'], f'{name}_{what}.npy'), torch.cat([wgid.double()[:, None], d], 1).numpy())
    print(f'      life p5/p50/p95 {q(.05):.1f} {q(.5):.1f} {q(.95):.1f} | start p50/p95/max {us(st_[n // 2]):.1f} {us(st_[int(.95 * (n - 1))]):.1f} {us(st_[-1]):.1f}')
