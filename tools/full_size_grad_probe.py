"""Calibration of the full-size train-step parity test: per-parameter gradient error of the HIP step against the
oracle in fp64, next to the fp32 oracle's own error against fp64.  usage: python tools/full_size_grad_probe.py [B] [T]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from oracle.cnet_oracle import C_NETWORK_Oracle
from oracle.nf_oracle import dcs_train_losses
from oracle.seeded_state import fill_state, seeded_input

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
seed = 3
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
clean, noise = seeded_input(B, 256, T, 1, 0.1), seeded_input(B, 256, T, 2, 0.05)
noisy = clean + noise


def oracle(dtype):
    from oracle import cpt_oracle, nf_oracle
    cpt_oracle.CDTYPE = nf_oracle.CDTYPE = torch.complex128 if dtype == torch.float64 else torch.complex64
    ref = fill_state(C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), seed).train()
    c = lambda z: z.to(torch.complex128) if dtype == torch.float64 else z
    if dtype == torch.float64:
        ref = ref.double()
    loss = dcs_train_losses(ref, c(noise), c(noisy), c(clean))[2]
    loss.backward()
    return float(loss), {n: (None if p.grad is None else p.grad.detach().double()) for n, p in ref.named_parameters()}


l32, g32 = oracle(torch.float32)
l64, g64 = oracle(torch.float64)
from dcsnet.config import config, hparams
from dcsnet.c_network import C_NETWORK
from dcsnet.dp import TrainStep
dev = torch.device('cuda:0')
hp = dict(hparams); hp['dropout_conv'] = hp['dropout_fc'] = 0.0
net = fill_state(C_NETWORK(config, hp, seed), seed).to(dev).train()
net.hparams['lr'] = 0.0; net.hparams['optim_weight_decay'] = 0.0
ts = TrainStep(net)
lh = float(ts((noise.to(dev), noisy.to(dev), clean.to(dev), list(range(B)))))
print(f'loss hip {lh:.7f} o32 {l32:.7f} o64 {l64:.7f}')
pd = dict(net.named_parameters())
rows = []
for n, w in g64.items():
    if w is None:
        continue
    h = pd[n].grad.detach().cpu().double()
    o = g32[n]
    s = float(w.abs().max()) + 1e-30
    rows.append((float((h - w).abs().max()) / s, float((o - w).abs().max()) / s,
                 abs(float(h.norm()) - float(w.norm())) / (float(w.norm()) + 1e-30),
                 abs(float(o.norm()) - float(w.norm())) / (float(w.norm()) + 1e-30), n,
                 float((h - w).norm() / (w.norm() + 1e-30)), float((o - w).norm() / (w.norm() + 1e-30))))
print('maxerr_hip/max  maxerr_o32/max  norm_hip  norm_o32  name')
for r in rows:
    if r[0] < 1e3:
        print(f'{r[0]:.2e}  {r[1]:.2e}  {r[2]:.2e}  {r[3]:.2e}  {r[0] / (r[1] + 1e-12):6.1f}x  L2 {r[5]:.2e} {r[6]:.2e} {r[5] / (r[6] + 1e-12):5.1f}x  {r[4]}')
