import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/dcs-net_amd')
from dcsnet import ops
from dcsnet.config import config, hparams
from dcsnet.c_network import C_NETWORK
from oracle import cnet_oracle as cno
from oracle.seeded_state import fill_state, seeded_input
dev = torch.device('cuda:0')
hp = dict(hparams); hp['dropout_conv'] = hp['dropout_fc'] = 0.0
x = seeded_input(2, 256, 32, seed=4)
ref = fill_state(cno.C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), 6).eval()
out_r = ref(x)
w = torch.rand(out_r.shape, generator=torch.Generator().manual_seed(2))
(w * (out_r.real ** 2 + 0.5 * out_r.imag ** 2)).sum().backward()
pr = dict(ref.named_parameters())
for mode in ('f32', 'bf16'):
    ops.set_conv_precision(mode)
    net = fill_state(C_NETWORK(config, hp, 0), 6).to(dev).eval()
    out = net(x.to(dev))
    (w.to(dev) * (out.real ** 2 + 0.5 * out.imag ** 2)).sum().backward()
    ops.set_conv_precision('f32')
    errs = []
    for n, p in net.named_parameters():
        if p.grad is None: continue
        g, r = p.grad.cpu(), pr[n].grad
        errs.append((float((g - r).norm()), float(r.norm()), n))
    num = sum(e[0] ** 2 for e in errs) ** 0.5; den = sum(e[1] ** 2 for e in errs) ** 0.5
    print(mode, 'mask', float((out.detach().cpu() - out_r.detach()).norm() / out_r.detach().norm()), 'grad rel-L2', num / den)
    for e in sorted(errs, reverse=True)[:5]: print('   ', e)
