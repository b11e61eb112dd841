"""dcs_gemm_f32 against torch.mm / bmm (rocBLAS) at the LSTM projection shapes of the train step ([32,256,256]: rows 4096, in 128,
8H 512) and of the inference pass.  GPU box: python tools/gemm_bench.py [rows]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'dcs-net_amd'))
from dcsnet import ops  # noqa: E402


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    eager = len(sys.argv) > 2 and sys.argv[2] == 'eager'          # a few plain launches of the in-tree kernels (counter passes)
    dev = torch.device('cuda:0')
    K, G8 = 128, 512
    x = torch.randn(M, K, device=dev)
    x3 = torch.randn(2, M, K, device=dev)
    w = torch.randn(2, G8, K, device=dev)
    g = torch.randn(2, M, G8, device=dev)
    c1 = torch.empty(M, 2 * G8, device=dev)
    c2 = torch.empty(2, M, G8, device=dev)
    d1 = torch.empty(M, K, device=dev)
    d2 = torch.empty(2, M, K, device=dev)
    cases = [
        ('projection, shared input   [M,128]x[128,1024]', lambda: ops.gemm_f32(x, w, c1, M, 2 * G8, K, K, K, 2 * G8, True),
         lambda: torch.mm(x, w.reshape(2 * G8, K).t(), out=c1)),
        ('projection, per-set input 2x[M,128]x[128,512]', lambda: ops.gemm_f32(x3, w, c2, M, G8, K, K, K, G8, True, nbatch=2, a_batch=M * K,
                                                                                b_batch=G8 * K, c_batch=M * G8),
         lambda: torch.bmm(x3, w.transpose(1, 2), out=c2)),
        ('data gradient, sets summed [M,1024]x[1024,128]', lambda: ops.gemm_f32(g, w, d1, M, K, G8, G8, K, K, False, nseg=2, a_seg=M * G8,
                                                                                 b_seg=G8 * K),
         lambda: torch.addmm(torch.mm(g[0], w[0]), g[1], w[1])),
        ('data gradient, per set    2x[M,512]x[512,128]', lambda: ops.gemm_f32(g, w, d2, M, K, G8, G8, K, K, False, nbatch=2, a_batch=M * G8,
                                                                                b_batch=G8 * K, c_batch=M * K),
         lambda: torch.bmm(g, w, out=d2)),
    ]
    flop = 2.0 * M * K * 2 * G8
    if eager:
        for name, mine, lib in cases:
            for _ in range(3):
                mine()
        torch.cuda.synchronize()
        return
    for name, mine, lib in cases:
        tm, tl = timeit(mine), timeit(lib)
        print(f'{name}: in-tree {tm:7.2f} us ({flop / tm * 1e-6:6.1f} TFLOP/s)   rocBLAS {tl:7.2f} us', flush=True)


if __name__ == '__main__':
    main()
