"""Phase times inside the MFMA weight-gradient kernel.  Needs a -DDCS_WGRAD_DIAG build of the library:
  python tools/wgrad_diag.py --build   (here, no GPU: writes dcs-net_amd/lib/diag/libdcsnet_wgraddiag.so), then on the GPU box
  DCS_LIB_PATH=dcs-net_amd/lib/diag/libdcsnet_wgraddiag.so python tools/wgrad_diag.py [B] [T] [layers]
Per workgroup (wave 0): gather = tile-loop top -> after the gather's second barrier (incl. waiting for the other waves),
mfma = the k-step loop, epilogue = slab stores; s_memtime = core clocks, printed as us at 2.4 GHz."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
if '--build' in sys.argv:
    import build
    out = os.path.join(ROOT, 'dcs-net_amd', 'lib', 'diag')
    os.makedirs(out, exist_ok=True)
    build.build(flags=build.FLAGS + ['-DDCS_WGRAD_DIAG'], verbose=False, lib=os.path.join(out, 'libdcsnet_wgraddiag.so'),
                objdir=os.path.join(out, 'obj_wgrad'))
    sys.exit(0)
from dcsnet import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
dev = torch.device('cuda:0')
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.dcs_debug_set_wgrad_buffer.argtypes = [ctypes.c_void_p]
t8 = T // 8
L = [('enc1', 128, T // 2, 8, 0, 16, 7, (2, 2), (1, 1)),
     ('enc2', 64, T // 4, 16, 0, 32, 5, (2, 2), (1, 1)), ('enc3', 32, t8, 32, 0, 64, 5, (2, 1), (1, 1)),
     ('enc4', 16, t8, 64, 0, 128, 3, (2, 1), (1, 1)), ('enc5', 8, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('enc6', 4, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('dec0', 2, t8, 128, 128, 128, 3, (1, 1), (2, 1)), ('dec1', 4, t8, 128, 128, 128, 3, (1, 1), (2, 1)),
     ('dec2', 8, t8, 128, 128, 64, 3, (1, 1), (2, 1)), ('dec3', 16, t8, 64, 64, 32, 3, (1, 1), (2, 1)),
     ('dec4', 32, t8, 32, 32, 16, 3, (1, 1), (2, 2)), ('dec5', 64, T // 4, 16, 16, 8, 3, (1, 1), (2, 2))]
for name, H, W, C1, C2, Cout, k, st, up in L:
    if only and name not in only:
        continue
    tr = name.startswith('dec')
    Cin = C1 + C2
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    x1 = torch.randn(B, H, W, C1, 2, device=dev)
    x2 = torch.randn(B, H, W, C2, 2, device=dev) if C2 else None
    Ho = H * up[0] if tr else (H + 2 * (k // 2) - k) // st[0] + 1
    Wo = W * up[1] if tr else (W + 2 * (k // 2) - k) // st[1] + 1
    gy = torch.randn(B, Ho, Wo, Cout, 2, device=dev)
    pad = (k // 2, k // 2)
    run = lambda: ops.cconv2d_bwd_weight(x1, x2, gy, wshape, True, (k, k), st, pad, up, tr)
    for _ in range(3):
        run()
    dbg = torch.zeros(8192 * 4 * 8, dtype=torch.int64, device=dev)
    lib.dcs_debug_set_wgrad_buffer(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    lib.dcs_debug_set_wgrad_buffer(None)
    d = dbg.view(-1, 4, 8)[:, 0].cpu().double()      # wave 0 of every workgroup
    d = d[d[:, 3] > 0]
    n = d.shape[0]
    us = lambda v: float(v) / 2400.0           # s_memtime counts core clocks (~2.4 GHz under MFMA load)
    t0 = d[:, 5].min()
    last_end = (d[:, 5] + d[:, 3]).max() - t0
    print(f'{name}: call {e0.elapsed_time(e1) * 1e3:7.1f} us | {n} WGs, tiles/WG {float(d[:, 4].mean()):.1f} | per WG (us): '
          f'gather {us(d[:, 0].mean()):6.1f}  mfma {us(d[:, 1].mean()):6.1f}  epilogue {us(d[:, 2].mean()):5.1f}  '
          f'life {us(d[:, 3].mean()):6.1f} (max {us(d[:, 3].max()):6.1f}) | first start -> last end {us(last_end):6.1f}, '
          f'start spread {us(d[:, 5].max() - t0):5.1f}')
    hw = d[:, 6].long()
    cu = (d[:, 7].long() & 15) * 4096 + ((hw >> 8) & 0xff)            # (xcc, se, sh, cu)
    ids, inv, cnt = torch.unique(cu, return_inverse=True, return_counts=True)
    per = cnt[inv]
    desc = ', '.join(f'{int(k)} WGs/CU: {int((cnt == k).sum())} CUs, life {us(d[per == k, 3].mean()):.1f}' for k in torch.unique(cnt))
    print(f'      {len(ids)} CUs used; {desc}')
    lf = torch.sort(d[:, 3]).values
    q = lambda f: us(lf[int(f * (n - 1))])
    mf = torch.sort(d[:, 1]).values
    print(f'      life percentiles 5/25/50/75/95: {q(.05):.1f} {q(.25):.1f} {q(.5):.1f} {q(.75):.1f} {q(.95):.1f} | mfma p5/p50/p95: '
          f'{us(mf[int(.05*(n-1))]):.1f} {us(mf[int(.5*(n-1))]):.1f} {us(mf[int(.95*(n-1))]):.1f}')
