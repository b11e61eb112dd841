"""Why does a trivial kernel of the replayed train step last ~4.8 us (profiles/r03_f_step_sequence_train.txt) when a raw
hipGraph chain of trivial kernels costs 1.6-2.0 us per kernel (tools/micro/launch_floor.hip)?  Chains captured with
torch.cuda.graph on one stream, wall time per kernel over a replay:
  same     200 x the same tiny in-place ATen kernel
  mixed    200 launches cycling through 16 different tiny ATen kernels (instruction-cache / code-object effect)
  big      the same tiny kernels, each touching a DIFFERENT 8 MB tensor first written by the previous one (cache write-back)
  dcs      200 x a trivial kernel of the library through the C ABI (dcs_dropout on 64 elements)
"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
dev = torch.device('cuda:0')
N = 200


def bench(name, body):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 2)
    print(f'{name:8s}: {best * 1e3 / N:6.2f} us per kernel  ({N} kernels per replay)')


x = torch.zeros(64, device=dev)
fns = [lambda t: t.add_(1.0), lambda t: t.mul_(0.5), lambda t: t.sin_(), lambda t: t.cos_(), lambda t: t.exp_(), lambda t: t.neg_(),
       lambda t: t.abs_(), lambda t: t.tanh_(), lambda t: t.sigmoid_(), lambda t: t.sqrt_(), lambda t: t.clamp_(0, 1), lambda t: t.floor_(),
       lambda t: t.ceil_(), lambda t: t.relu_(), lambda t: t.sub_(1.0), lambda t: t.div_(2.0)]
bench('same', lambda: [x.add_(1.0) for _ in range(N)])
bench('mixed', lambda: [fns[i % 16](x) for i in range(N)])
bufs = [torch.zeros(2 * 1024 * 1024, device=dev) for _ in range(8)]
bench('big8MB', lambda: [bufs[i % 8].add_(1.0) for i in range(N)])
bench('bigmix', lambda: [fns[i % 16](bufs[i % 8]) for i in range(N)])
from dcsnet import ops
y = torch.zeros(64, 2, device=dev)
try:
    bench('dcs', lambda: [ops.dropout(y, 0.1, 1, out=y) for _ in range(N)])
except Exception as e:      # signature differs: report and go on
    print('dcs probe skipped:', repr(e)[:200])
