"""Which lines of dcsnet/ run torch-side (ATen) ops during one eager train step: (aten op, call site) -> count.
A TorchDispatchMode sees every ATen call (forward thread and, through the propagated thread-local state, autograd's
backward thread) and the Python stack gives the dcsnet/ line that issued it.
usage (GPU box): python tools/torch_ops_by_site.py"""
import os, sys, collections, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
sys.argv = [sys.argv[0]]
from torch.utils._python_dispatch import TorchDispatchMode
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
SKIP = ('aten.view', 'aten.empty', 'aten._unsafe_view', 'aten.detach', 'aten.as_strided', 'aten.select', 'aten.slice',
        'aten.unsqueeze', 'aten.squeeze', 'aten.permute', 'aten.transpose', 'aten.t.', 'aten.expand', 'aten.alias',
        'aten.view_as_real', 'aten.view_as_complex', 'aten.reshape', 'aten.unbind', 'aten.split', 'aten.narrow',
        'aten.lift_fresh', 'aten._reshape_alias', 'aten.is_', 'aten.sym_', 'aten.stride', 'aten.size', 'aten.numel',
        'aten.real', 'aten.imag', 'aten.empty_like', 'aten.new_empty', 'aten.unfold', 'aten.result_type')
sites = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            st = traceback.extract_stack()
            site = next((f'{os.path.basename(f.filename)}:{f.lineno} {f.name}' for f in reversed(st)
                         if ('dcsnet' in f.filename or 'bench.py' in f.filename)), 'autograd engine (accumulate)')
            if site.startswith('autograd engine'):          # which tensors autograd sums: the shape tells the node
                shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
                site = f'{site} {shp}'
            sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    ts(batch)
    torch.cuda.synchronize()
for (name, site), n in sorted(sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f'{n:4d}  {name:34s} {site}')
print(sum(sites.values()), 'ATen calls')
