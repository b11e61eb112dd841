"""Loss trajectories of a small DR-Net / DCS-Net train step, eager and captured, repeated in one process — and, with `nanfill`, with
every torch.empty filled with NaN (torch.utils.deterministic.fill_uninitialized_memory): a kernel that reads memory nobody wrote
shows up as a NaN or as a first-run / later-run difference.    python tools/rnet_traj.py [nanfill] [c]"""
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd')); sys.path.insert(0, ROOT)
import torch
args = sys.argv[1:]
if 'nanfill' in args:
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
from dcsnet.config import config, hparams
from dcsnet.dp import TrainStep
from oracle.seeded_state import fill_state_stream, fill_state, seeded_input
dev = torch.device('cuda:0')
hp = dict(hparams); hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
cnet = 'c' in args
sys.argv = ['train.py', 'dcs' if cnet else 'drs', '0']
for use_graph in (False, True, False, True):
    if cnet:
        from dcsnet.c_network import C_NETWORK
        net = fill_state(C_NETWORK(config, hp, 0), 2).to(dev).train()
    else:
        from dcsnet.r_network import R_NETWORK
        net = fill_state_stream(R_NETWORK(config, hp, 0), 5).to(dev).train()
    ts = TrainStep(net, use_graph=use_graph, graph_warmup=2)
    print('graph' if use_graph else 'eager', [round(float(ts(batch)), 5) for _ in range(6)], flush=True)
