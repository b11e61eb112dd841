"""DR-Net: why does the second step's loss of the FIRST TrainStep of a process differ from later ones?  (probe)
    python tools/rnet_first_run.py"""
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd')); sys.path.insert(0, ROOT)
import torch
from dcsnet.config import config, hparams
from dcsnet.dp import TrainStep
from dcsnet import functional as F
from dcsnet.r_network import R_NETWORK
from oracle.seeded_state import fill_state_stream, seeded_input
dev = torch.device('cuda:0')
hp = dict(hparams); hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
sys.argv = ['train.py', 'drs', '0']
for r, bump in enumerate((False, True, False, False)):
    net = fill_state_stream(R_NETWORK(config, hp, 0), 5).to(dev).train()
    ts = TrainStep(net, use_graph=False)
    l0 = float(ts(batch))
    if bump:
        F.bump_param_generation()
    with torch.no_grad():
        net.eval(); le = float(ts._loss_no_sync(batch, 0)); net.train()
    l1 = float(ts(batch))
    print(f'run {r} bump={bump}: loss0 {l0:.5f}  eval-mode loss after update {le:.5f}  loss1 {l1:.5f}  generation {F.state_generation()}')
