"""LSTM recurrence kernels alone at the train shape (2 sets x 64 sequences x 64 steps, H = 64): python tools/lstm_bench.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from dcsnet import ops
dev = torch.device('cuda:0')
for (sets, seqs, S, H) in ((2, 64, 64, 64), (2, 32, 500, 64), (1, 32, 64, 128)):
    gx = torch.randn(sets, seqs, S, 2, 4 * H, device=dev) * 0.5
    whh = torch.randn(sets, 2, 4 * H, H, device=dev) * 0.1
    st = (seqs * S * 8 * H, S * 8 * H, 8 * H)
    out, gates, c = ops.lstm_layer(gx, whh, sets, seqs, S, st, True)
    go = torch.randn_like(out)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    f = t(lambda: ops.lstm_layer(gx, whh, sets, seqs, S, st, True))
    b = t(lambda: ops.lstm_layer_bwd(go, gates, c, whh, sets, seqs, S))
    ba, bb = torch.randn(sets, 8 * H, device=dev) * 0.1, torch.randn(sets, 8 * H, device=dev) * 0.1
    fb = t(lambda: ops.lstm_layer(gx, whh, sets, seqs, S, st, True, bias=(ba, bb)))
    fi = t(lambda: ops.lstm_layer(gx, whh, sets, seqs, S, st, False))
    fib = t(lambda: ops.lstm_layer(gx, whh, sets, seqs, S, st, False, bias=(ba, None)))
    print(f'sets {sets} seqs {seqs} S {S} H {H}: fwd {f:.1f} us ({f / S:.2f} us/step; with biases {fb:.1f}; inference form {fi:.1f}, '
          f'with bias {fib:.1f}), bwd {b:.1f} us ({b / S:.2f} us/step)')
