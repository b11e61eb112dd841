"""GPU parity of the hand-written backward kernels: every differentiable op (through the C ABI)
against torch.autograd on the CPU oracle, and the whole network against the gradient vectors
generated from the reference's own c_network.py (tests/golden/cnet_vectors.npz).

Tolerance: 1e-4 relative to each gradient tensor's max-abs (fp32; accumulation orders differ).
"""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpt_oracle as cpt          # noqa: E402
from oracle import nf_oracle as nf            # noqa: E402
from oracle import cnet_oracle as cno         # noqa: E402
from oracle.seeded_state import fill_state    # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from dcsnet import _lib
    _lib.load()
    return torch.device('cuda:0')


def rand_c(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.complex(torch.randn(shape, generator=g) * scale, torch.randn(shape, generator=g) * scale)


def weights_like(t, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(t.shape, generator=g) - 0.3


def functional_loss(out, seed):
    """A fixed real functional of a complex tensor."""
    w = weights_like(out.real, seed).to(out.device)
    return (w * (out.real ** 2 + 0.5 * out.imag ** 2 + 0.25 * out.real * out.imag + 0.3 * out.imag)).sum()


def nhwc_leaf(z, dev):
    from dcsnet import ops
    return ops.to_nhwc(z.to(dev)).detach().requires_grad_(True)


def cgrad(x_nhwc):
    """float NHWC leaf's grad -> complex NCHW on the CPU."""
    return torch.view_as_complex(x_nhwc.grad.cpu()).permute(0, 3, 1, 2)


def close(got, want, rel=1e-4, abs_=0.0, what=''):
    got, want = got.detach().cpu(), want.detach().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err <= rel * scale + abs_, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


def dev_params(mod, dev):
    return {n: p.detach().to(dev).requires_grad_(True) for n, p in mod.named_parameters()}


# --------------------------------------------------------------------------------- convolution

@pytest.mark.parametrize('cin,cout,k,stride,hw', [(1, 8, 7, (2, 2), (40, 36)), (1, 8, 7, (2, 2), (75, 134)), (8, 16, 7, (2, 2), (20, 24)),
                                                   (16, 32, 5, (2, 1), (12, 9)), (64, 128, 3, (2, 1), (8, 16)),
                                                   (4, 2, 3, (1, 2), (19, 9)),
                                                   # >= 8 input-channel chunks: the weight gradient takes g_Y pre-split (gy_planes_kernel,
                                                   # conv_wgrad_mfma.hip) — ragged tiles, 3x3 and 5x5, 32 .. 128 output channels
                                                   (128, 64, 3, (1, 1), (9, 37)), (64, 32, 5, (2, 2), (21, 45)), (72, 40, 3, (2, 1), (11, 19))])
def test_conv2d_backward(dev, cin, cout, k, stride, hw):
    from dcsnet import functional as F
    torch.manual_seed(cin + k)
    m = cpt.ComplexConv2d(cin, cout, k, stride, k // 2)
    x = rand_c((2, cin, *hw), 5).requires_grad_(True)
    functional_loss(m(x), 1).backward()
    p = dev_params(m, dev)
    xn = nhwc_leaf(x.detach(), dev)
    y = F.cconv2d(xn, None, p['conv_r.weight'], p['conv_i.weight'], p['conv_r.bias'], p['conv_i.bias'], False,
                  (k, k), stride, (k // 2, k // 2))
    functional_loss(F.from_nhwc(y), 1).backward()
    close(cgrad(xn), x.grad, what='g_x')
    for n, q in m.named_parameters():
        close(p[n].grad, q.grad, what=n)


def test_first_encoder_conv_weight_gradient_train_size(dev):
    """BASELINE configs[2] size [32,256,256]: the weight / bias gradient of enc0 over all 2048 tiles (persistent MFMA
    kernel of conv_enc0.hip, one slab per workgroup, slab-parallel reduce) against autograd on the oracle conv."""
    from dcsnet import functional as F
    torch.manual_seed(8)
    m = cpt.ComplexConv2d(1, 8, 7, (2, 2), 3)
    x = rand_c((32, 1, 256, 256), 6, 0.5)
    functional_loss(m(x), 2).backward()
    p = dev_params(m, dev)
    xn = nhwc_leaf(x, dev)
    y = F.cconv2d(xn, None, p['conv_r.weight'], p['conv_i.weight'], p['conv_r.bias'], p['conv_i.bias'], False,
                  (7, 7), (2, 2), (3, 3))
    functional_loss(F.from_nhwc(y), 2).backward()
    for n, q in m.named_parameters():
        close(p[n].grad, q.grad, rel=3e-4, what=n)     # sums of 524 k fp32 terms in two different orders


@pytest.mark.parametrize('c1,c2,cout,up', [(128, 128, 64, (2, 1)), (16, 16, 8, (2, 2)), (8, 8, 1, (2, 2)), (16, 0, 8, (1, 1)),
                                           (96, 32, 32, (2, 2)), (64, 64, 128, (2, 1))])
def test_convtranspose_cat_upsample_backward(dev, c1, c2, cout, up):
    from dcsnet import functional as F
    torch.manual_seed(c1 + cout)
    m = cpt.ComplexConvTranspose2d(c1 + c2, cout, 3, 1, 1)
    d = rand_c((2, c1, 5, 9), 3).requires_grad_(True)
    s = rand_c((2, c2, 5, 9), 4).requires_grad_(True) if c2 else None
    cat = torch.cat((d, s), dim=1) if c2 else d
    functional_loss(m(cpt.complex_upsample(cat, scale_factor=up, mode='nearest')), 2).backward()
    p = dev_params(m, dev)
    dn = nhwc_leaf(d.detach(), dev)
    sn = nhwc_leaf(s.detach(), dev) if c2 else None
    y = F.cconv2d(dn, sn, p['conv_tran_r.weight'], p['conv_tran_i.weight'], p['conv_tran_r.bias'],
                  p['conv_tran_i.bias'], True, (3, 3), (1, 1), (1, 1), up)
    functional_loss(F.from_nhwc(y), 2).backward()
    close(cgrad(dn), d.grad, what='g_d')
    if c2:
        close(cgrad(sn), s.grad, what='g_skip')
    for n, q in m.named_parameters():
        close(p[n].grad, q.grad, what=n)


@pytest.mark.parametrize('c1,c2,up,hw', [(8, 8, (2, 2), (13, 37)), (16, 0, (2, 2), (8, 32)), (8, 8, (2, 1), (6, 11)),
                                         (8, 0, (2, 2), (5, 9))])
def test_single_output_stage(dev, c1, c2, up, hw):
    """The last decoder stage (ConvTranspose2d -> 1 channel behind cat + upsample, c_network.py:135-141) through
    F.cconv_single_output: the one-kernel forward (16 input channels, 2x2 upsample; ragged tile edges) and the factored
    tap-channel path (other geometries), both with the factored backward, against the oracle layer."""
    from dcsnet import functional as F
    torch.manual_seed(c1 + c2 + up[1])
    m = cpt.ComplexConvTranspose2d(c1 + c2, 1, 3, 1, 1)
    d = rand_c((2, c1, *hw), 3).requires_grad_(True)
    s = rand_c((2, c2, *hw), 4).requires_grad_(True) if c2 else None
    cat = torch.cat((d, s), dim=1) if c2 else d
    want = m(cpt.complex_upsample(cat, scale_factor=up, mode='nearest'))
    functional_loss(want, 2).backward()
    p = dev_params(m, dev)
    dn = nhwc_leaf(d.detach(), dev)
    sn = nhwc_leaf(s.detach(), dev) if c2 else None
    y = F.cconv_single_output(dn, sn, p['conv_tran_r.weight'], p['conv_tran_i.weight'], p['conv_tran_r.bias'],
                              p['conv_tran_i.bias'], (3, 3), (1, 1), up)
    close(F.from_nhwc(y), want, rel=2e-5, what='forward')
    functional_loss(F.from_nhwc(y), 2).backward()
    close(cgrad(dn), d.grad, what='g_d')
    if c2:
        close(cgrad(sn), s.grad, what='g_skip')
    for n, q in m.named_parameters():
        close(p[n].grad, q.grad, what=n)


def test_single_output_stage_direct_backward_equals_the_factored_one(dev):
    """conv_up1.hip's backward kernels (tap sums of the cotangent in registers: dcs_cconv_up2_single_bwd_data / _bwd_weight) against the
    factored form they replace (dcs_tapsum_bwd -> 1x1 tap conv data / weight gradient -> scatter) on a ragged multi-tile shape: same
    sums in another order (2e-5 of the tensor's max-abs, the conv tolerance); accumulation into sinks; bitwise repeatable."""
    from dcsnet import ops
    g = torch.Generator().manual_seed(11)
    B, Hs, Ws, C1, C2 = 3, 21, 70, 8, 8
    rn = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    x1, x2 = rn(B, Hs, Ws, C1, 2), rn(B, Hs, Ws, C2, 2)
    w_r, w_i = rn(16, 1, 3, 3) * 0.2, rn(16, 1, 3, 3) * 0.2
    gy = rn(B, 2 * Hs, 2 * Ws, 1, 2)
    wt, _ = ops.pack_tap_rows(w_r, w_i, 16)
    # the factored form
    gb0 = (torch.empty(1, device=dev), torch.empty(1, device=dev))
    gz = ops.tapsum((B, Hs, Ws, 16, 2), (3, 3), (2, 2), (1, 1), backward=True, grad=gy, bias_grad=gb0)
    gt_r, gt_i, _, _ = ops.cconv2d_bwd_weight(x1, x2, gz, (16, 16, 1, 1), False, (1, 1), (1, 1), (0, 0))
    gw0 = ops.tap_rows_scatter(gt_r, gt_i, (16, 1, 3, 3))
    gx0 = ops.cconv2d_bwd_data(gz, ops.pack_conv_weight_bwd(wt, (1, 1)), (Hs, Ws, 16), (1, 1), (1, 1), (0, 0), (1, 1), C1)
    # the direct kernels
    gx = ops.cconv_up2_single_bwd_data(gy, wt, C1, C2)
    gb = (torch.empty(1, device=dev), torch.empty(1, device=dev))
    gw = ops.cconv_up2_single_bwd_weight(gy, x1, x2, (16, 1, 3, 3), None, gb)
    torch.cuda.synchronize()
    for a, b_, what in ((gx[0], gx0[0], 'g_x1'), (gx[1], gx0[1], 'g_x2'), (gw[0], gw0[0], 'g_w_r'), (gw[1], gw0[1], 'g_w_i'),
                        (gb[0], gb0[0], 'g_b_r'), (gb[1], gb0[1], 'g_b_i')):
        err, scale = float((a - b_).abs().max()), float(b_.abs().max())
        assert err <= 2e-5 * scale, (what, err, scale)
    # sinks are added to; a second launch gives the same bits
    sink = (torch.ones(16, 1, 3, 3, device=dev), torch.full((16, 1, 3, 3), 2.0, device=dev))
    ops.cconv_up2_single_bwd_weight(gy, x1, x2, (16, 1, 3, 3), sink, None)
    gw2 = ops.cconv_up2_single_bwd_weight(gy, x1, x2, (16, 1, 3, 3), None, None)
    gx2_ = ops.cconv_up2_single_bwd_data(gy, wt, C1, C2)
    torch.cuda.synchronize()
    assert torch.equal(gw2[0], gw[0]) and torch.equal(gw2[1], gw[1]) and torch.equal(gx2_[0], gx[0]) and torch.equal(gx2_[1], gx[1])
    assert torch.allclose(sink[0], gw[0] + 1.0, rtol=0, atol=1e-6 * float(gw[0].abs().max()) + 1e-6)
    assert torch.allclose(sink[1], gw[1] + 2.0, rtol=0, atol=1e-6 * float(gw[1].abs().max()) + 1e-6)


def test_complex_linear_backward(dev):
    from dcsnet import functional as F
    torch.manual_seed(8)
    m = cpt.ComplexLinear(128, 128)
    z = rand_c((2, 24, 128), 2).requires_grad_(True)
    functional_loss(m(z), 3).backward()
    p = dev_params(m, dev)
    zd = z.detach().to(dev).requires_grad_(True)
    y = F.complex_linear(zd, p['fc_r.weight'], p['fc_i.weight'], p['fc_r.bias'], p['fc_i.bias'])
    functional_loss(y, 3).backward()
    close(zd.grad, z.grad, what='g_z')
    for n, q in m.named_parameters():
        close(p[n].grad, q.grad, what=n)


# --------------------------------------------------------------------------------- batch norm

@pytest.mark.parametrize('C,shape', [(1, (2, 16, 24)), (1, (1, 3, 5)), (8, (2, 12, 10)), (64, (3, 6, 8)), (128, (2, 4, 8))])
@pytest.mark.parametrize('act', ['none', 'relu', 'lrelu'])
@pytest.mark.parametrize('training', [True, False])
def test_cbn_backward(dev, C, shape, act, training):
    from dcsnet import functional as F
    B, H, W = shape
    bn = fill_state(cpt.ComplexBatchNorm2d(C), seed=C + 1)
    bn.train(training)
    post = {'none': lambda z: z, 'relu': cpt.complex_relu, 'lrelu': nf.complex_lrelu}[act]
    code = {'none': F.ACT_NONE, 'relu': F.ACT_RELU, 'lrelu': F.ACT_LRELU}[act]
    x = (rand_c((B, C, H, W), C + 2, 1.2) + (0.3 - 0.5j)).requires_grad_(True)
    rm = torch.view_as_real(bn.running_mean.clone()).to(dev).contiguous()
    rc = bn.running_covar.clone().to(dev).contiguous()
    functional_loss(post(bn(x)), 4).backward()
    p = dev_params(bn, dev)
    xn = nhwc_leaf(x.detach(), dev)
    y = F.cbn(xn, p['weight'], p['bias'], rm, rc, bn.eps, 0.1 if training else -1.0, training, code)
    functional_loss(F.from_nhwc(y), 4).backward()
    close(cgrad(xn), x.grad, rel=2e-4, what='g_x')
    close(p['weight'].grad, bn.weight.grad, rel=2e-4, what='g_weight')
    close(p['bias'].grad, bn.bias.grad, rel=2e-4, what='g_bias')


def test_cbn_backward_of_a_network_input_gives_parameter_gradients_only(dev):
    """The initial CBN (c_network.py:101,191): its input is the data, nothing wants g_x — the backward launches the
    reduction and the finalize only (dcs_cbn_bwd_add with g_x = NULL) and the parameter gradients are unchanged."""
    from dcsnet import functional as F
    bn = fill_state(cpt.ComplexBatchNorm2d(1), seed=3)
    bn.train(True)
    x = rand_c((2, 1, 16, 24), 5, 1.2) + (0.3 - 0.5j)
    rm = torch.view_as_real(bn.running_mean.clone()).to(dev).contiguous()
    rc = bn.running_covar.clone().to(dev).contiguous()
    functional_loss(bn(x), 4).backward()
    p = dev_params(bn, dev)
    xn = nhwc_leaf(x, dev).detach()                          # no gradient wanted for the input
    assert not xn.requires_grad
    y = F.cbn(xn, p['weight'], p['bias'], rm, rc, bn.eps, 0.1, True, F.ACT_NONE)
    functional_loss(F.from_nhwc(y), 4).backward()
    close(p['weight'].grad, bn.weight.grad, rel=2e-4, what='g_weight')
    close(p['bias'].grad, bn.bias.grad, rel=2e-4, what='g_bias')


def test_dropout_backward_uses_the_same_mask(dev):
    from dcsnet import functional as F
    x = torch.randn(1 << 16, device=dev, requires_grad=True)
    y = F.dropout(x, 0.2, 99)
    y.sum().backward()
    assert torch.equal(x.grad, (y.detach() != 0).float() / 0.8)
    # fused in CBN: gradient is zero exactly where the output was dropped
    bn = fill_state(cpt.ComplexBatchNorm2d(8), 1)
    xn = torch.randn(2, 6, 6, 8, 2, device=dev, requires_grad=True)
    p = dev_params(bn, dev)
    rm, rc = torch.zeros(8, 2, device=dev), torch.ones(8, 3, device=dev)
    y = F.cbn(xn, p['weight'], p['bias'], rm, rc, 1e-5, 0.1, False, F.ACT_NONE, 0.5, 7)
    y0 = F.cbn(xn, p['weight'], p['bias'], rm, rc, 1e-5, 0.1, False, F.ACT_NONE, 0.0, 7)
    keep = (y.detach() != 0)
    assert 0.35 < float(keep.float().mean()) < 0.65
    assert torch.allclose(y.detach()[keep], 2 * y0.detach()[keep])
    g = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, xn, g)
    (gx0,) = torch.autograd.grad(y0, xn, g * keep * 2)
    assert torch.allclose(gx, gx0, atol=1e-6)


# --------------------------------------------------------------------------------- attention

@pytest.mark.parametrize('C,hw,relu_in', [(8, (20, 12), True), (64, (6, 10), False), (128, (2, 8), True)])
def test_attention_block_backward(dev, C, hw, relu_in):
    from dcsnet import functional as F
    torch.manual_seed(C)
    ca_m, sa_m = cno.ComplexChannelAttention(C, 16), cno.ComplexSpatialAttention(7)
    x0 = rand_c((3, C, *hw), C + 2)
    if relu_in:                         # exact zeros (ReLU outputs) create ties in the channel arg-max
        x0 = cpt.complex_relu(x0)
    x = x0.clone().requires_grad_(True)
    z = ca_m(x) * x
    out = sa_m(z) * z
    functional_loss(out, 5).backward()
    pc, ps = dev_params(ca_m, dev), dev_params(sa_m, dev)
    xn = nhwc_leaf(x0, dev)
    y = F.attention_block(xn, pc['fc.0.conv_r.weight'], pc['fc.0.conv_i.weight'], pc['fc.2.conv_r.weight'],
                          pc['fc.2.conv_i.weight'], ps['conv1.conv_r.weight'], ps['conv1.conv_i.weight'], 7)
    close(F.from_nhwc(y), out, rel=2e-5, what='forward')
    functional_loss(F.from_nhwc(y), 5).backward()
    close(cgrad(xn), x.grad, rel=2e-4, what='g_x')
    for n, q in ca_m.named_parameters():
        close(pc[n].grad, q.grad, rel=2e-4, what=n)
    for n, q in sa_m.named_parameters():
        close(ps[n].grad, q.grad, rel=2e-4, what=n)


@pytest.mark.parametrize('C,hw,drop', [(8, (20, 12), 0.0), (64, (6, 10), 0.25), (128, (2, 8), 0.0)])
def test_cbn_attention_one_node_matches_the_two_node_chain(dev, C, hw, drop):
    """F.cbn_attention (decoder stage tail as one autograd node; the average pool's broadcast gradient is added inside the
    CBN backward kernels) against F.cbn -> F.attention_block on the same inputs: same kernels otherwise, so forward is
    bit-identical and the gradients agree to fp32 re-association."""
    from dcsnet import functional as F
    torch.manual_seed(C)
    g = torch.Generator().manual_seed(C)
    x0 = torch.randn((3, *hw, C, 2), generator=g).to(dev)
    mk = lambda *shape: ((torch.rand(shape, generator=g) - 0.4) * 0.8).to(dev)
    Ch = max(C // 16, 1)
    base = [mk(C, 3) + 1.0, mk(C, 2), mk(Ch, C, 1, 1), mk(Ch, C, 1, 1), mk(C, Ch, 1, 1), mk(C, Ch, 1, 1), mk(1, 2, 7, 7),
            mk(1, 2, 7, 7)]
    wgt = torch.rand(x0.shape, generator=g).to(dev)

    def run(fused):
        x = x0.clone().requires_grad_(True)
        p = [t.clone().requires_grad_(True) for t in base]
        rm, rc = torch.zeros(C, 2, device=dev), torch.ones(C, 3, device=dev)
        if fused:
            y = F.cbn_attention(x, p[0], p[1], rm, rc, 1e-5, 0.1, True, F.ACT_LRELU, *p[2:], 7, drop, 11)
        else:
            a = F.cbn(x, p[0], p[1], rm, rc, 1e-5, 0.1, True, F.ACT_LRELU)
            y = F.attention_block(a, *p[2:], 7, drop, 11)
        (wgt * y * y).sum().backward()
        return y.detach(), [x.grad] + [t.grad for t in p], (rm, rc)

    y1, g1, st1 = run(True)
    y0, g0, st0 = run(False)
    assert torch.equal(y1, y0) and torch.equal(st1[0], st0[0]) and torch.equal(st1[1], st0[1])
    for i, (a, b) in enumerate(zip(g1, g0)):
        close(a, b, rel=2e-5, what=f'grad {i}')


def test_attention_blocks_batched(dev):
    """The batched entry (all skip attentions in one set of launches) against the CPU oracle, block by block:
    different channel counts and plane sizes share the launches, including a ReLU input with arg-max ties."""
    from dcsnet import functional as F
    geo = [(128, (2, 8), True), (64, (6, 10), False), (8, (20, 12), True), (16, (9, 5), True), (32, (4, 4), False)]
    xs, params, want = [], [], []
    for C, hw, relu_in in geo:
        torch.manual_seed(C + 7)
        ca_m, sa_m = cno.ComplexChannelAttention(C, 16 if C >= 16 else 4), cno.ComplexSpatialAttention(7)
        x0 = rand_c((3, C, *hw), C + 3)
        if relu_in:
            x0 = cpt.complex_relu(x0)
        x = x0.clone().requires_grad_(True)
        z = ca_m(x) * x
        out = sa_m(z) * z
        functional_loss(out, C).backward()
        pc, ps = dev_params(ca_m, dev), dev_params(sa_m, dev)
        xs.append(nhwc_leaf(x0, dev))
        params.append((pc['fc.0.conv_r.weight'], pc['fc.0.conv_i.weight'], pc['fc.2.conv_r.weight'], pc['fc.2.conv_i.weight'],
                       ps['conv1.conv_r.weight'], ps['conv1.conv_i.weight']))
        want.append((out, x, ca_m, sa_m, pc, ps))
    ys = F.attention_blocks(xs, params, 7)
    total = sum(functional_loss(F.from_nhwc(y), g[0]) for y, g in zip(ys, geo))
    total.backward()
    for i, (out, x, ca_m, sa_m, pc, ps) in enumerate(want):
        close(F.from_nhwc(ys[i]), out, rel=2e-5, what=f'forward {i}')
        close(cgrad(xs[i]), x.grad, rel=2e-4, what=f'g_x {i}')
        for n, q in ca_m.named_parameters():
            close(pc[n].grad, q.grad, rel=2e-4, what=f'{i} {n}')
        for n, q in sa_m.named_parameters():
            close(ps[n].grad, q.grad, rel=2e-4, what=f'{i} {n}')


# --------------------------------------------------------------------------------- LSTM, mask

@pytest.mark.parametrize('B,S', [(2, 8), (3, 40), (1, 1)])
def test_complex_lstm_backward(dev, B, S):
    from dcsnet.c_network import ComplexLSTM
    torch.manual_seed(B * 10 + S)
    ref = cno.ComplexLSTM(128, 64, 2, True)
    mod = ComplexLSTM(128, 64, 2, True, True)
    mod.load_state_dict(ref.state_dict())
    mod.to(dev)
    z = rand_c((B, S, 128), 4, 0.8).requires_grad_(True)
    functional_loss(ref(z), 6).backward()
    zd = z.detach().to(dev).requires_grad_(True)
    out = mod(zd)
    functional_loss(out, 6).backward()
    close(zd.grad, z.grad, rel=2e-4, what='g_z')
    got = dict(mod.named_parameters())
    for n, q in ref.named_parameters():
        close(got[n].grad, q.grad, rel=2e-4, abs_=1e-7, what=n)


def test_bound_and_mask_apply_backward(dev):
    from dcsnet import functional as F
    M = rand_c((2, 64, 32), 1, 1.5)
    M.view(-1)[:4] = torch.tensor([0 + 0j, -1 + 0j, 50 - 70j, 1e-3 + 1e-3j])
    Y = rand_c((2, 64, 32), 2, 0.7)
    Mo = M.clone().requires_grad_(True)
    m, nhat, shat = nf.mask_apply_subtract(Y, Mo)
    (functional_loss(m, 1) + functional_loss(nhat, 2) + 2 * functional_loss(shat, 3)).backward()
    Md = M.to(dev).requires_grad_(True)
    m2, n2, s2 = F.bound_mask_apply_complex(Y.to(dev), Md)
    (functional_loss(m2, 1) + functional_loss(n2, 2) + 2 * functional_loss(s2, 3)).backward()
    ok = torch.ones(M.shape, dtype=torch.bool)
    ok.view(-1)[:2] = False              # the origin and the branch cut are singular for atan2
    close(Md.grad.cpu()[ok], Mo.grad[ok], rel=2e-4, what='g_M_in')
    assert torch.isfinite(torch.view_as_real(Md.grad)).all()
    # only one output used; plain bound_cRM
    Mo.grad = None
    functional_loss(nf.bound_cRM(Mo), 4).backward()
    Md.grad = None
    functional_loss(F.bound_crm_complex(Md), 4).backward()
    close(Md.grad.cpu()[ok], Mo.grad[ok], rel=2e-4, what='g_M (bound only)')


# --------------------------------------------------------------------------------- whole network

def _hip_net(dev, seed):
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    return fill_state(C_NETWORK(config, hp, seed), seed).to(dev)


def _bias_before_bn(n):
    return n.endswith('.bias') and (('.0.conv_r' in n or '.0.conv_i' in n) and n.startswith('encoder')
                                    or ('.0.conv_tran_' in n and n.startswith('decoder')))


@pytest.mark.parametrize('conv_mode', ['bf16x6', 'f32'])
def test_network_gradients_against_reference_vectors(dev, golden_dir, conv_mode):
    """Both fp32 arithmetic modes of the conv kernels (ops.set_conv_precision): 'bf16x6' = the default emulation on the bf16
    MFMA (forward, data gradient, weight gradient), 'f32' = the native fp32 MFMA kernels."""
    from dcsnet import ops
    default = ops.conv_precision()
    ops.set_conv_precision(conv_mode)
    try:
        _network_gradients_against_reference_vectors(dev, golden_dir)
    finally:
        ops.set_conv_precision(default)


def _network_gradients_against_reference_vectors(dev, golden_dir):
    cnv = np.load(os.path.join(golden_dir, 'cnet_vectors.npz'))
    tag = 'b2t32'
    net = _hip_net(dev, 0).train()
    out = net(torch.from_numpy(cnv[f'{tag}_x']).to(dev))
    w = torch.from_numpy(cnv[f'{tag}_loss_w']).to(dev)
    loss = (w * (out.real ** 2 + 0.5 * out.imag ** 2 + 0.25 * out.real * out.imag)).sum()
    loss.backward()
    assert abs(float(loss) - float(cnv[f'{tag}_loss'])) <= 1e-4 * abs(float(cnv[f'{tag}_loss']))
    pd = dict(net.named_parameters())
    names = [str(n) for n in cnv[f'{tag}_grad_names']]
    assert sorted(names) == sorted(pd.keys())
    for n, want in zip(names, cnv[f'{tag}_grad_norms']):
        g = pd[n].grad
        if want < 0:                                   # decoder_attention.12 / .13 are never run
            assert g is None, n
        elif _bias_before_bn(n):                       # analytically zero: both sides are rounding noise
            assert float(g.norm()) < 2e-3, n
        else:
            assert abs(float(g.norm()) - want) <= 2e-3 * want + 1e-6, (n, float(g.norm()), want)
    for k in cnv.files:
        if k.startswith(f'{tag}_grad_') and k not in (f'{tag}_grad_names', f'{tag}_grad_norms'):
            n = k[len(f'{tag}_grad_'):]
            if not _bias_before_bn(n):
                close(pd[n].grad, torch.from_numpy(cnv[k]), rel=2e-3, abs_=1e-6, what=n)


def test_polar_frames_forward_and_backward(dev):
    """The fused polar round trip in front of the iSTFT vs the reference's op chain (network_functions.py:140-145,
    :213-221): |z| cos/sin(atan2(z_i, z_r + eps)), one zero bin appended; HIP output is frame-major [B,T,F+1]."""
    from dcsnet import functional as F
    eps = 10e-7
    z0 = rand_c((2, 70, 45), 3, 0.8)                  # not multiples of the 32x32 transpose tile
    z0.view(-1)[:3] = torch.tensor([0 + 0j, -0.5 + 0j, 1e-4 - 2e-4j])

    def ref(z):
        mag, ph = torch.abs(z), torch.atan2(z.imag, z.real + eps)
        return torch.nn.functional.pad(torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)), (0, 0, 0, 1))

    z = z0.clone().requires_grad_(True)
    want = ref(z).transpose(1, 2)
    functional_loss(want, 7).backward()
    zd = z0.to(dev).requires_grad_(True)
    got = F.polar_frames_complex(zd, 1, eps)
    assert got.shape == (2, 45, 71) and got.is_contiguous()
    close(got, want, rel=0, abs_=2e-6, what='forward')
    functional_loss(got, 7).backward()
    ok = torch.ones(z0.shape, dtype=torch.bool)
    ok.view(-1)[:2] = False                       # origin / branch cut: atan2 is singular there
    close(zd.grad.cpu()[ok], z.grad[ok], rel=2e-4, what='g_z')
    assert torch.isfinite(torch.view_as_real(zd.grad)).all()


def test_hip_istft_equals_torch_istft_and_its_gradient(dev):
    """network_functions.istft (contiguous irfft + dcs_istft_ola_fwd; no host-synchronising NOLA check) reproduces
    torch.istft, inverts the reference's STFT (n_fft 512, hop 32, hann, normalized: data.py:112-118), and its
    backward (dcs_istft_ola_bwd) matches autograd through torch.istft."""
    from dcsnet.network_functions import istft
    torch.manual_seed(0)
    w = torch.hann_window(512)
    for T in (16, 256):
        x = torch.randn(2, 32 * T - 32)
        X = torch.stft(x, 512, 32, 512, w, return_complex=True, normalized=True)
        assert X.shape[-1] == T
        Xc = X.clone().requires_grad_(True)
        want = torch.istft(Xc, 512, 32, 512, w, normalized=True)
        gw = torch.randn(want.shape, generator=torch.Generator().manual_seed(T))
        (want * gw).sum().backward()
        Xd = X.to(dev).requires_grad_(True)
        got = istft(Xd, 512, 32, w.to(dev), True)
        assert got.shape == want.shape
        assert float((got.cpu() - want).abs().max()) < 5e-6
        assert float((got.cpu() - x).abs().max()) < 5e-6
        (got * gw.to(dev)).sum().backward()
        close(Xd.grad.cpu(), Xc.grad, rel=1e-4, what=f'g_X T={T}')


def test_fused_sisnr_value_and_gradient(dev):
    """dcs_sisnr_fwd/_bwd vs the reference's SiSNR formula (network_functions.py:30-42) and autograd through it."""
    from dcsnet import functional as F
    g = torch.Generator().manual_seed(5)
    for B, L in ((3, 1000), (32, 8160)):
        clean = torch.randn(B, L, generator=g) * 0.3
        est = (clean + 0.2 * torch.randn(B, L, generator=g)).requires_grad_(True)
        want = nf.si_snr(clean, est)
        (-0.7 * want).backward()
        ed = est.detach().to(dev).requires_grad_(True)
        got = F.sisnr_mean(clean.to(dev), ed, 1e-8)
        assert abs(float(got) - float(want)) <= 2e-5 * abs(float(want)) + 1e-5
        (-0.7 * got).backward()
        close(ed.grad, est.grad, rel=2e-4, what=f'g_est B={B}')


def test_fused_sisnr_loss_assembly(dev):
    """dcs_sisnr_losses_fwd + the two dcs_sisnr_bwd launches vs the reference's loss assembly
    (network_functions.py:168-208: speech = alpha (-SiSNR), noise = 1 - alpha (-SiSNR), total = sum)."""
    from dcsnet import functional as F
    g = torch.Generator().manual_seed(9)
    B, L, alpha = 8, 4000, 0.7
    clean, noise = torch.randn(B, L, generator=g) * 0.3, torch.randn(B, L, generator=g) * 0.1
    ec = (clean + 0.1 * torch.randn(B, L, generator=g)).requires_grad_(True)
    en = (noise + 0.05 * torch.randn(B, L, generator=g)).requires_grad_(True)
    speech = alpha * (-nf.si_snr(clean, ec))
    nl = 1 - alpha * (-nf.si_snr(noise, en))
    (nl + speech).backward()
    ecd, end_ = ec.detach().to(dev).requires_grad_(True), en.detach().to(dev).requires_grad_(True)
    got = F.sisnr_losses(clean.to(dev), ecd, noise.to(dev), end_, alpha)
    for a, b in zip(got, (nl, speech, nl + speech)):
        assert abs(float(a) - float(b)) <= 2e-5 * abs(float(b)) + 2e-5
    got[2].backward()
    close(ecd.grad, ec.grad, rel=2e-4, what='g_est_clean')
    close(end_.grad, en.grad, rel=2e-4, what='g_est_noise')


def test_fused_sisnr_pair_losses_and_guard(dev):
    """The stacked form (rows [0,B) noise, [B,2B) speech): one dcs_sisnr_fwd over 2B rows, dcs_sisnr_losses_guard_fwd (which
    also writes the train step's NaN flag) and ONE dcs_sisnr_pair_bwd, against the reference's loss assembly and autograd;
    gradients reaching only one of the three outputs; a NaN total raises the flag."""
    from dcsnet import functional as F
    g = torch.Generator().manual_seed(19)
    B, L, alpha = 6, 3000, 0.7
    clean, noise = torch.randn(B, L, generator=g) * 0.3, torch.randn(B, L, generator=g) * 0.1
    ec = (clean + 0.1 * torch.randn(B, L, generator=g)).requires_grad_(True)
    en = (noise + 0.05 * torch.randn(B, L, generator=g)).requires_grad_(True)
    speech = alpha * (-nf.si_snr(clean, ec))
    nl = 1 - alpha * (-nf.si_snr(noise, en))
    tgt = torch.cat((noise, clean)).to(dev)
    for pick in (2, 0, 1):                                   # total, noise_loss only, speech_loss only
        for t in (ec, en):
            t.grad = None
        (nl, speech, nl + speech)[pick].backward(retain_graph=True)
        est = torch.cat((en.detach(), ec.detach())).to(dev).requires_grad_(True)
        skip = torch.full((1,), 7.0, device=dev)
        got = F.sisnr_losses_pair(tgt, est, alpha, skip=skip)
        for a, b in zip(got, (nl, speech, nl + speech)):
            assert abs(float(a) - float(b)) <= 2e-5 * abs(float(b)) + 2e-5
        assert float(skip) == 0.0
        got[pick].backward()
        want = torch.cat((en.grad if en.grad is not None else torch.zeros_like(en),
                          ec.grad if ec.grad is not None else torch.zeros_like(ec)))
        close(est.grad, want, rel=2e-4, what=f'g_est (output {pick})')
    bad = est.detach().clone()
    bad[0, 0] = float('nan')
    skip = torch.zeros(1, device=dev)
    F.sisnr_losses_pair(tgt, bad, alpha, skip=skip)
    assert float(skip) == 1.0


def test_lstm_projection_gemm(dev):
    """dcs_gemm_f32 (the LSTM input projections x W_ih^T and their data gradients: inside nn.LSTM in the reference,
    c_network.py:24-31,43-46) against fp64 matmul: K-contiguous and N-contiguous B, a ragged M, two K segments (the two parameter
    sets of a shared input), batches, and the {re rows | im rows} stacking read / written in place."""
    from dcsnet import ops
    g = torch.Generator().manual_seed(41)
    f64 = torch.float64

    def run(A, Bm, M, N, K, bt, **kw):
        C = torch.full(kw.pop('c_shape'), float('nan'), device=dev)
        ops.gemm_f32(A.to(dev), Bm.to(dev), C, M, N, K, kw.pop('lda', K), kw.pop('ldb', K if bt else N), kw.pop('ldc', N), bt, **kw)
        return C.cpu()

    # (the shapes cover the three workgroup forms of csrc/gemm.hip and its group sizes 4 / 2 / 1)
    for M, N, K in ((2048, 1024, 128), (40, 64, 32), (97, 128, 512), (6400, 64, 96), (6401, 64, 64), (4096, 128, 1024)):
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g)
        want = (A.to(f64) @ W.to(f64).t()).float()
        close(run(A, W, M, N, K, True, c_shape=(M, N)), want, rel=2e-6, what=f'A W^T {M}x{N}x{K}')
        close(run(A, W.t().contiguous(), M, N, K, False, c_shape=(M, N)), want, rel=2e-6, what=f'A B {M}x{N}x{K}')
    # batches: [2][M][K] x [2][N][K]^T; segments: sum over two (A_s, B_s) pairs with B N-contiguous
    M, N, K = 72, 192, 64
    A = torch.randn(2, M, K, generator=g)
    W = torch.randn(2, N, K, generator=g)
    got = run(A, W, M, N, K, True, c_shape=(2, M, N), nbatch=2, a_batch=M * K, b_batch=N * K, c_batch=M * N)
    close(got, torch.bmm(A.to(f64), W.to(f64).transpose(1, 2)).float(), rel=2e-6, what='batched')
    A2 = torch.randn(2, 2048, 512, generator=g)                                   # the layer-1 data gradient's shape
    W2 = torch.randn(2, 512, 128, generator=g)
    got = run(A2, W2, 2048, 128, 512, False, c_shape=(2, 2048, 128), nbatch=2, a_batch=2048 * 512, b_batch=512 * 128, c_batch=2048 * 128)
    close(got, torch.bmm(A2.to(f64), W2.to(f64)).float(), rel=2e-6, what='batched, N-contiguous B')
    Bn = torch.randn(2, K, N, generator=g)
    got = run(A, Bn, M, N, K, False, c_shape=(M, N), nseg=2, a_seg=M * K, b_seg=K * N)
    close(got, torch.bmm(A.to(f64), Bn.to(f64)).sum(0).float(), rel=2e-6, what='two segments')
    # the stacking of ComplexLSTM's input: rows {re | im} of a complex [R0][K] read in place, and written in place
    R0, K, N = 36, 128, 64
    z = torch.randn(R0, K, 2, generator=g)
    W = torch.randn(N, K, generator=g)
    x2 = z.permute(2, 0, 1).reshape(2 * R0, K)
    got = run(z, W, 2 * R0, N, K, True, c_shape=(2 * R0, N), a_planes=R0)
    close(got, (x2.to(f64) @ W.to(f64).t()).float(), rel=2e-6, what='A read as planes')
    Gm = torch.randn(2 * R0, K, generator=g)
    Bn = torch.randn(K, N, generator=g)
    got = run(Gm, Bn, 2 * R0, N, K, False, c_shape=(R0, N, 2), c_planes=R0)
    close(got, (Gm.to(f64) @ Bn.to(f64)).float().view(2, R0, N).permute(1, 2, 0), rel=2e-6, what='C written as planes')
    # two runs give the same bits (fixed summation order); bad shapes are refused
    A = torch.randn(256, 128, generator=g).to(dev)
    W = torch.randn(512, 128, generator=g).to(dev)
    c1 = ops.gemm_f32(A, W, torch.empty(256, 512, device=dev), 256, 512, 128, 128, 128, 512, True)
    c2 = ops.gemm_f32(A, W, torch.empty(256, 512, device=dev), 256, 512, 128, 128, 128, 512, True)
    assert torch.equal(c1, c2)
    with pytest.raises(Exception):
        ops.gemm_f32(A, W, torch.empty(256, 512, device=dev), 256, 500, 128, 128, 128, 512, True)
    with pytest.raises(Exception):
        ops.gemm_f32(A, W, torch.empty(256, 512, device=dev), 256, 512, 100, 128, 128, 512, True)


def test_lstm_glue_kernels(dev):
    """dcs_lstm_combine_fwd / _bwd (ComplexLSTM's recombination, c_network.py:43-46) and dcs_lstm_param_grads (chunk sums of
    the recurrent-weight products, per-sequence bias sums) against their torch spellings."""
    from dcsnet import ops
    g = torch.Generator().manual_seed(23)
    B, S, W = 3, 5, 128
    o = torch.randn(2, 2 * B, S, W, generator=g)
    want = torch.complex(o[0, :B] - o[1, B:], o[0, B:] + o[1, :B])
    got = ops.lstm_combine(o.to(dev), B)
    close(torch.view_as_real(got), torch.view_as_real(want), rel=1e-6, what='lstm_combine')
    gc = torch.randn(B, S, W, 2, generator=g)
    g_o = ops.lstm_combine_bwd(gc.to(dev))
    want_o = torch.stack((torch.cat((gc[..., 0], gc[..., 1])), torch.cat((gc[..., 1], -gc[..., 0]))))
    close(g_o, want_o, rel=1e-6, what='lstm_combine_bwd')
    H, CK, seqs = 64, 16, 10
    part = torch.randn(2, 2 * CK, 4 * H, H, generator=g)
    b_part = torch.randn(2, seqs, 8 * H, generator=g)
    g_whh, g_bih, g_bhh = (torch.randn(2, 2, 4 * H, H, generator=g), torch.randn(2, 8 * H, generator=g),
                           torch.randn(2, 8 * H, generator=g))
    w_whh = g_whh + part.view(2, 2, CK, 4 * H, H).sum(2).transpose(0, 1)
    w_b = b_part.sum(1)
    d = [t.clone().to(dev) for t in (g_whh, g_bih, g_bhh)]
    ops.lstm_param_grads(part.to(dev), b_part.to(dev), d[0], d[1], d[2], CK, seqs, H)
    close(d[0], w_whh, rel=1e-5, what='g_whh')
    close(d[1], g_bih + w_b, rel=1e-5, what='g_bih')
    close(d[2], g_bhh + w_b, rel=1e-5, what='g_bhh')
    # dcs_lstm_whh_grad: the chunked g_pre^T h_prev products on the MFMA pipe vs the strided batched GEMM it replaces
    NT = 2 * 16 * 8
    g_pre = torch.randn(2, NT, 8 * H, generator=g)
    hprev = torch.randn(2, NT, 2 * H, generator=g)
    R = NT // CK
    want_p = torch.empty(2, 2 * CK, 4 * H, H)
    for dd in range(2):
        a = g_pre.view(2, NT, 2, 4 * H)[:, :, dd].reshape(2 * CK, R, 4 * H)
        h = hprev.view(2, NT, 2, H)[:, :, dd].reshape(2 * CK, R, H)
        want_p[dd] = torch.bmm(a.transpose(1, 2), h)
    got_p = ops.lstm_whh_grad(g_pre.to(dev), hprev.to(dev), NT, CK, H)
    close(got_p, want_p, rel=2e-5, what='lstm_whh_grad')
    # the general chunked A^T B + chunk sum (the W_ih gradients): per-set inputs and one shared input
    for n_in, shared in ((128, False), (256, True)):
        inp = torch.randn((NT, n_in) if shared else (2, NT, n_in), generator=g)
        out0 = torch.randn(2, 8 * H, n_in, generator=g)
        want_w = out0 + torch.stack([g_pre[s_].t() @ (inp if shared else inp[s_]) for s_ in range(2)])
        out = out0.clone().to(dev)
        ops.atb_chunks_acc(g_pre.to(dev), inp.to(dev), out, 2, 8 * H, n_in, 8 * H, n_in, NT, CK, b_shared=shared)
        close(out, want_w, rel=2e-5, what=f'atb_chunks_acc in={n_in}')
    # the same chunk sum riding dcs_lstm_param_grads' launch (dcs_lstm_param_grads_ih)
    inp = torch.randn(2, NT, 128, generator=g)
    out0 = torch.randn(2, 8 * H, 128, generator=g)
    want_w = out0 + torch.stack([g_pre[s_].t() @ inp[s_] for s_ in range(2)])
    out = out0.clone().to(dev)
    part_ih, ck_ih = ops.atb_chunks_acc(g_pre.to(dev), inp.to(dev), out, 2, 8 * H, 128, 8 * H, 128, NT, CK, reduce=False)
    d = [t.clone().to(dev) for t in (g_whh, g_bih, g_bhh)]
    ops.lstm_param_grads(part.to(dev), b_part.to(dev), d[0], d[1], d[2], CK, seqs, H, ih=(part_ih, ck_ih, out))
    close(out, want_w, rel=2e-5, what='g_wih through lstm_param_grads')
    close(d[0], w_whh, rel=1e-5, what='g_whh (with ih)')
    close(d[1], g_bih + w_b, rel=1e-5, what='g_bih (with ih)')
    close(d[2], g_bhh + w_b, rel=1e-5, what='g_bhh (with ih)')
    # the shared input as the {re rows | im rows} stacking of a complex-interleaved [R0, in, 2], read in place
    R0, n_in = NT // 2, 128
    zr = torch.randn(R0, n_in, 2, generator=g)
    x2 = zr.permute(2, 0, 1).reshape(NT, n_in)
    out0 = torch.randn(2, 8 * H, n_in, generator=g)
    want_w = out0 + torch.stack([g_pre[s_].t() @ x2 for s_ in range(2)])
    out = out0.clone().to(dev)
    ops.atb_chunks_acc_planes(g_pre.to(dev), zr.to(dev), out, 2, 8 * H, 8 * H, n_in, R0, 8)
    close(out, want_w, rel=2e-5, what='atb_chunks_acc_planes')
