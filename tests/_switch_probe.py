"""Helper of tests/test_switches.py (run as a subprocess, so that switches read once per process — C statics, module-level
os.environ reads — take effect): three updates of the train step at [4,256,64] with the reference's dropout on (the masks are
counter-based on the device: the same in every process) and one inference pass; prints one JSON line.
usage: python tests/_switch_probe.py [attr=value ...]   (module attributes set after import, e.g. ops.STATS_EPILOGUE=0)"""
import json
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from oracle.seeded_state import fill_state, seeded_input   # noqa: E402
from dcsnet import ops, functional   # noqa: E402
from dcsnet.config import config, hparams   # noqa: E402
from dcsnet.c_network import C_NETWORK   # noqa: E402
from dcsnet.dp import TrainStep   # noqa: E402

for a in sys.argv[1:]:
    name, val = a.split('=')
    mod, attr = name.split('.')
    setattr({'ops': ops, 'functional': functional, 'C_NETWORK': C_NETWORK}[mod], attr, type(getattr({'ops': ops, 'functional': functional, 'C_NETWORK': C_NETWORK}[mod], attr))(int(val)))
dev = torch.device('cuda:0')
graph = os.environ.get('PROBE_GRAPH', '0') == '1'
clean, noise = seeded_input(4, 256, 64, 1, 0.1), seeded_input(4, 256, 64, 2, 0.05)
batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1, 2, 3])
torch.manual_seed(0)
net = fill_state(C_NETWORK(config, dict(hparams), 0), 2).to(dev).train()
ts = TrainStep(net, use_graph=graph, graph_warmup=1)
losses = [float(ts(batch)) for _ in range(3)]
torch.cuda.synchronize()
p = ts.bucket.flat.double()
net.eval()
with torch.no_grad():
    m = net((clean + noise).to(dev))
torch.cuda.synchronize()
print(json.dumps(dict(losses=losses, pnorm=float(p.norm()), psum=float(p.sum()), mask=float(torch.view_as_real(m).double().norm()),
                      finite=bool(torch.isfinite(p).all()))))
