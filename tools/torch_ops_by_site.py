"""Which lines of dcsnet/ launch the torch-side (ATen) kernels of one eager train step: aten op -> call site counts.
usage (GPU box): python tools/torch_ops_by_site.py"""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
sys.argv = [sys.argv[0]]
from dcsnet.config import config, hparams
from dcsnet.c_network import C_NETWORK
from dcsnet.dp import TrainStep
import bench
dev = torch.device('cuda:0')
net = C_NETWORK(config, hparams, 0).to(dev).train()
noise, noisy, clean = bench.synthetic_stft_batch(32, 256, dev, seed=0)
ts = TrainStep(net, use_graph=False)
batch = (noise, noisy, clean, list(range(32)))
for _ in range(3):
    ts(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts(batch)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if not ev.kernels:
        continue
    site = next((s for s in ev.stack if 'dcsnet' in s or 'bench.py' in s), ev.stack[0] if ev.stack else '?')
    sites[(ev.name, site.split('dcs-net_amd/')[-1][:90])] += len(ev.kernels)
for (name, site), n in sites.most_common(70):
    print(f'{n:4d}  {name:28s} {site}')
