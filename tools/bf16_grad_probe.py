"""How far the bf16-operand mode moves the network's gradients from the fp32 HIP path (same kernels, same inputs)."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/dcs-net_amd')
from dcsnet import ops
from dcsnet.config import config, hparams
from dcsnet.c_network import C_NETWORK
from oracle.seeded_state import fill_state, seeded_input
dev = torch.device('cuda:0')
hp = dict(hparams); hp['dropout_conv'] = hp['dropout_fc'] = 0.0
def run(mode, B, T, train):
    ops.set_conv_precision(mode)
    net = fill_state(C_NETWORK(config, hp, 0), 6).to(dev)
    net.train(train)
    x = seeded_input(B, 256, T, seed=4).to(dev)
    out = net(x)
    w = torch.rand(out.shape, generator=torch.Generator().manual_seed(2)).to(dev)
    (w * (out.real ** 2 + 0.5 * out.imag ** 2)).sum().backward()
    g = torch.cat([p.grad.reshape(-1) for n, p in net.named_parameters() if p.grad is not None])
    ops.set_conv_precision('f32')
    return out.detach(), g
for B, T, train in ((2, 32, True), (8, 64, True), (2, 32, False), (8, 64, False)):
    o0, g0 = run('f32', B, T, train)
    o1, g1 = run('bf16', B, T, train)
    print(f'B={B} T={T} train={train}: mask rel-L2 {float((o1-o0).norm()/o0.norm()):.3e}  grad rel-L2 {float((g1-g0).norm()/g0.norm()):.3e}')
