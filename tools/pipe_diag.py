"""Cycle shares inside the pipelined conv kernel (needs a diagnostic build: DCS_EXTRA_HIPCC_FLAGS=-DDCS_PIPE_DIAG
python dcs-net_amd/build.py).  usage: python tools/pipe_diag.py [B] [T] [layers]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from dcsnet import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
dev = torch.device('cuda:0')
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.dcs_debug_set_buffer.argtypes = [ctypes.c_void_p]
t8 = T // 8
L = [('enc1', 128, T // 2, 8, 0, 16, 7, (2, 2), (1, 1)),
     ('enc2', 64, T // 4, 16, 0, 32, 5, (2, 2), (1, 1)), ('enc3', 32, t8, 32, 0, 64, 5, (2, 1), (1, 1)),
     ('enc4', 16, t8, 64, 0, 128, 3, (2, 1), (1, 1)), ('enc5', 8, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('enc6', 4, t8, 128, 0, 128, 3, (2, 1), (1, 1)),
     ('dec1', 4, t8, 128, 128, 128, 3, (1, 1), (2, 1)), ('dec4', 32, t8, 32, 32, 16, 3, (1, 1), (2, 2))]
ops.set_conv_schedule('pipe')
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
    for _ in range(3):
        ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up)
    dbg = torch.zeros(4096 * 5 * 8, dtype=torch.int64, device=dev)
    lib.dcs_debug_set_buffer(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up)
    e1.record()
    torch.cuda.synchronize()
    lib.dcs_debug_set_buffer(None)
    d = dbg.view(-1, 5, 8).cpu().double()
    used = d[:, 0, 4] > 0
    d = d[used]
    n = d.shape[0]
    mf = d[:, :4].mean(1)            # mean over the 4 MFMA waves
    ld = d[:, 4]
    f = lambda v: f'{float(v.mean()):9.0f}'
    print(f'{name}: {e0.elapsed_time(e1) * 1e3:7.1f} us, {n} workgroups, items/WG {float(ld[:, 5].mean()):.1f}')
    print(f'   MFMA waves (cycles/WG, s_memtime ticks = 100 MHz*?):  barrier wait {f(mf[:, 0])}  compute {f(mf[:, 1])}  epilogue {f(mf[:, 2])}  life {f(mf[:, 4])}  (max life {float(mf[:, 4].max()):.0f})')
    print(f'   loader:  table {f(ld[:, 0])}  issue {f(ld[:, 1])}  vmcnt wait {f(ld[:, 2])}  barrier wait {f(ld[:, 3])}  life {f(ld[:, 4])}')
