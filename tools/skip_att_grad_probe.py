"""Where does the distance of a skip attention's FC gradient to the fp64 oracle come from at the bench size — the block's own
backward kernels, or the fp32 noise of what reaches them?  The batched skip-attention backward of one HIP train step is
intercepted; for block `blk` its inputs (x, the cotangent g_out) are re-differentiated on the CPU in fp64 through the oracle's
modules with the same weights:
   (a) HIP kernels on HIP inputs        = the parameter gradient the step produced
   (b) fp64 autograd on the SAME inputs = what an exact backward of the block gives for these (fp32-noisy) inputs
   (c) the fp64 oracle's whole step     = the reference of tests/test_full_size.py
|a - b| is the block's kernel error, |b - c| the inherited noise.   usage (GPU box): python tools/skip_att_grad_probe.py [blk]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sys.argv = sys.argv[:1]
from oracle.cnet_oracle import C_NETWORK_Oracle, ComplexChannelAttention, ComplexSpatialAttention
from oracle.nf_oracle import dcs_train_losses
from oracle import cpt_oracle, nf_oracle
from oracle.seeded_state import fill_state, seeded_input
from dcsnet import functional as F
from dcsnet.config import config, hparams
from dcsnet.c_network import C_NETWORK
from dcsnet.dp import TrainStep
B, T, seed = 32, 256, 3
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
clean, noise = seeded_input(B, 256, T, 1, 0.1), seeded_input(B, 256, T, 2, 0.05)
noisy = clean + noise



def oracle_step(cd):
    cpt_oracle.CDTYPE = nf_oracle.CDTYPE = cd
    ref = fill_state(C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), seed).train()
    if cd == torch.complex128:
        ref = ref.double()
    io = {}

    def pre(m, inp):
        io['x'] = inp[0].detach().to(torch.complex128)
        if inp[0].requires_grad:
            inp[0].register_hook(lambda gr: io.__setitem__('gx', gr.detach().to(torch.complex128)))
    ref.skip_attention[2 * blk].register_forward_pre_hook(pre)
    dcs_train_losses(ref, noise.to(cd), noisy.to(cd), clean.to(cd))[2].backward()
    cpt_oracle.CDTYPE = nf_oracle.CDTYPE = torch.complex64
    return {n: p.grad.detach().double() for n, p in ref.named_parameters() if p.grad is not None}, io


g64, io64 = oracle_step(torch.complex128)
g32, io32 = oracle_step(torch.complex64)
print('oracle steps (fp64, fp32) done', flush=True)

dev = torch.device('cuda:0')
hp = dict(hparams); hp['dropout_conv'] = hp['dropout_fc'] = 0.0
net = fill_state(C_NETWORK(config, hp, seed), seed).to(dev).train()
net.hparams['lr'] = 0.0; net.hparams['optim_weight_decay'] = 0.0
cap = {}
orig = F._AttentionBlocksFn.backward


def spy(ctx, *g_outs):
    t = ctx.saved_tensors
    cap['x'] = t[9 * blk].detach().cpu().double()
    cap['g'] = g_outs[blk].detach().cpu().double()
    out = orig(ctx, *g_outs)
    cap['gx'] = out[2 + blk].detach().cpu().double()
    return out


F._AttentionBlocksFn.backward = staticmethod(spy)
ts = TrainStep(net, use_graph=False)
ts((noise.to(dev), noisy.to(dev), clean.to(dev), list(range(B))))
torch.cuda.synchronize()
pd = dict(net.named_parameters())
x = torch.view_as_complex(cap['x'].contiguous()).permute(0, 3, 1, 2).contiguous().requires_grad_(True)      # [B,C,H,W] complex128
g = torch.view_as_complex(cap['g'].contiguous()).permute(0, 3, 1, 2).contiguous()
C = x.shape[1]
cpt_oracle.CDTYPE = nf_oracle.CDTYPE = torch.complex128
ca = ComplexChannelAttention(C, hp['channel_attention_reduction_ratio']).double()
sa = ComplexSpatialAttention(hp['spatial_attention_kernel_size']).double()
pre = f'skip_attention.{2 * blk}.'
ca.load_state_dict({k[len(pre):]: v.detach().cpu().double() for k, v in pd.items() if k.startswith(pre)})
pre2 = f'skip_attention.{2 * blk + 1}.'
sa.load_state_dict({k[len(pre2):]: v.detach().cpu().double() for k, v in pd.items() if k.startswith(pre2)})
z = ca(x) * x
y = sa(z) * z
# real-valued cotangent pairing used by torch for complex outputs: sum(Re(conj(g) y))
torch.view_as_real(y).mul(torch.view_as_real(g)).sum().backward()
rel = lambda a, b: float((a - b).norm() / b.norm())
print(f'block {blk}: C = {C}, map {tuple(x.shape[2:])}')
cx = lambda t: torch.view_as_complex(t.contiguous()).permute(0, 3, 1, 2)
print(f"input x of the block      : |hip - oracle64| {rel(x.detach(), io64['x']):.2e}   |cpu fp32 - oracle64| {rel(io32['x'], io64['x']):.2e}")
print(f"gradient w.r.t. x (total) : |hip - oracle64| see note   |cpu fp32 - oracle64| {rel(io32['gx'], io64['gx']):.2e}   (the enc1 output has a second consumer: the hip figure below is the block's share only)")
print(f"block's own g_x           : |hip - exact(block)| {rel(cx(cap['gx']), x.grad):.2e}")
for name, p in [(n, q) for n, q in ca.named_parameters()] + [(n, q) for n, q in sa.named_parameters()]:
    full = (pre2 if name.startswith('conv1.') else pre) + name
    a = pd[full].grad.detach().cpu().double()
    b = p.grad.detach()
    c = g64[full]
    print(f'{full:44s} |hip - exact(block)| {rel(a, b):.2e}   |exact(block) - oracle64| {rel(b, c):.2e}   |hip - oracle64| {rel(a, c):.2e}   |cpu fp32 - oracle64| {rel(g32[full], c):.2e}')
