import torch, inspect
print(torch.__version__)
print(inspect.signature(torch.cuda.Event.__new__) if hasattr(torch.cuda.Event,'__new__') else '')
x = torch.randn(4096, 4096, device='cuda')
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
    e2 = torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print('no external kw:', ex); raise SystemExit
y = x @ x
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    e0.record()
    y = x @ x
    e1.record()
    z = y + 1
    e2.record()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print('matmul ms', e0.elapsed_time(e1), 'add ms', e1.elapsed_time(e2))
