"""Uninitialised-read probe: the C_NETWORK paths with every torch.empty filled with NaN (torch.utils.deterministic.fill_uninitialized_memory)
against the same paths without — bf16-storage train steps, fp32 inference, B = 1 — must give the same numbers.
    python tools/nanfill_probe.py [nanfill]"""
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd')); sys.path.insert(0, ROOT)
import torch
if 'nanfill' in sys.argv[1:]:
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
from dcsnet.config import config, hparams
from dcsnet.dp import TrainStep
from dcsnet.c_network import C_NETWORK
from dcsnet import functional as F
from oracle.seeded_state import fill_state, seeded_input
dev = torch.device('cuda:0')
sys.argv = ['train.py', 'dcs', '0']
hp = dict(hparams)
for B, T, bf in ((2, 32, True), (3, 40, False), (1, 16, False)):
    clean, noise = seeded_input(B, 256, T, 1, 0.1), seeded_input(B, 256, T, 2, 0.05)
    batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), list(range(B)))
    net = fill_state(C_NETWORK(config, hp, 0), 2).to(dev).train()
    if bf:
        net.set_activation_dtype('bf16')
    ts = TrainStep(net, use_graph=True, graph_warmup=2)
    print(f'train B={B} T={T} bf16={bf}:', [round(float(ts(batch)), 5) for _ in range(5)], flush=True)
    net.eval()
    with torch.no_grad():
        d = net(batch[1], bound=False)
        M, N, S = F.bound2_mask_apply_complex(batch[1], d if d.dim() == 3 else d.unsqueeze(0), hp['atan2_eps'])
    print(f'  eval checksum: {float(torch.view_as_real(M).double().sum()):.9f} {float(torch.view_as_real(S).double().abs().sum()):.9f}', flush=True)
    if bf:
        from dcsnet import ops
        ops.set_conv_precision('bf16x6')
