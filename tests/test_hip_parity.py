"""GPU parity: every HIP entry point (called through the C ABI via dcsnet.ops) against the CPU
oracle on the same seeded inputs, and against the golden vectors generated from the reference.

Tolerances (fp32 path, stated per the north star): element-wise mask math 2e-6 absolute on a
mask of modulus < 1; convolutions / batch-norm 2e-5 relative to the tensor's max-abs (fp32
accumulation order differs from the CPU's); whole network 2e-4 absolute on the bounded mask.
"""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpt_oracle as cpt          # noqa: E402
from oracle import nf_oracle as nf            # noqa: E402
from oracle import cnet_oracle as cno         # noqa: E402
from oracle.seeded_state import fill_state, seeded_input   # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from dcsnet import _lib
    _lib.load()                                   # fail loudly if the HIP library is missing
    return torch.device('cuda:0')


def rand_c(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.complex(torch.randn(shape, generator=g) * scale, torch.randn(shape, generator=g) * scale)


def nhwc(z, dev):
    from dcsnet import ops
    return ops.to_nhwc(z.to(dev))


def back(x):
    from dcsnet import ops
    return ops.from_nhwc(x).cpu()


def close(got, want, rel=2e-5, abs_=0.0):
    got, want = got.detach().cpu(), want.detach().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = float(want.abs().max()) if want.numel() else 1.0
    err = float((got - want).abs().max()) if want.numel() else 0.0
    assert err <= rel * scale + abs_, f'max err {err:.3e} vs scale {scale:.3e}'


# --------------------------------------------------------------------------------- convolution

ENC = [(1, 8, 7, (2, 2)), (8, 16, 7, (2, 2)), (16, 32, 5, (2, 2)), (32, 64, 5, (2, 1)),
       (64, 128, 3, (2, 1)), (128, 128, 3, (2, 1))]


@pytest.mark.parametrize('cin,cout,k,stride', ENC)
def test_complex_conv2d_encoder_geometries(dev, cin, cout, k, stride):
    from dcsnet import functional as F
    torch.manual_seed(cin * 7 + k)
    m = cpt.ComplexConv2d(cin, cout, k, stride, k // 2)
    H, W = (36, 24) if cin <= 16 else (12, 16)
    x = rand_c((2, cin, H, W), 11)
    want = m(x)
    p = lambda t: t.detach().to(dev)
    y = F.cconv2d(nhwc(x, dev), None, p(m.conv_r.weight), p(m.conv_i.weight), p(m.conv_r.bias), p(m.conv_i.bias),
                  False, (k, k), stride, (k // 2, k // 2))
    close(back(y), want)


@pytest.mark.parametrize('H,W', [(257, 96), (35, 131), (15, 7)])
def test_first_encoder_conv_on_the_mfma_units(dev, H, W):
    """conv_enc0.hip (taps as the MFMA K axis): whole and ragged 8 x 32 tiles, plain and with the folded eval-mode
    CBN + ReLU epilogue of the inference path (c_network.py:107-112)."""
    from dcsnet import functional as F
    torch.manual_seed(H + W)
    m = cpt.ComplexConv2d(1, 8, 7, (2, 2), 3)
    x = rand_c((3, 1, H, W), 21)
    want = m(x)
    p = lambda t: t.detach().to(dev)
    args = (nhwc(x, dev), None, p(m.conv_r.weight), p(m.conv_i.weight), p(m.conv_r.bias), p(m.conv_i.bias), False,
            (7, 7), (2, 2), (3, 3))
    close(back(F.cconv2d(*args)), want)
    coef = torch.randn(8, 6) * 0.5
    a = coef.view(1, 8, 6, 1, 1)
    re, im = want.real, want.imag
    aff = torch.complex(torch.relu(a[:, :, 0] * re + a[:, :, 1] * im + a[:, :, 4]),
                        torch.relu(a[:, :, 2] * re + a[:, :, 3] * im + a[:, :, 5]))
    y = F.cconv2d_cbn_eval(*args, (1, 1), coef.to(dev), F.ACT_RELU)
    close(back(y), aff, rel=4e-5)


# the seven encoder layers at BASELINE configs[1] size (input [16, 256, 2000]): (cin, cout, k, stride, Hin, Win)
ENC_FULL = [(1, 8, 7, (2, 2), 256, 2000), (8, 16, 7, (2, 2), 128, 1000), (16, 32, 5, (2, 2), 64, 500),
            (32, 64, 5, (2, 1), 32, 250), (64, 128, 3, (2, 1), 16, 250), (128, 128, 3, (2, 1), 8, 250),
            (128, 128, 3, (2, 1), 4, 250)]


@pytest.mark.parametrize('cin,cout,k,stride,H,W', ENC_FULL)
def test_encoder_convs_full_size_windows(dev, cin, cout, k, stride, H, W):
    """Every encoder conv at its BASELINE configs[1] size (the tile plans, split-K choices, persistent workgroups and
    kernel variants that only full-size launches select): windows of the full-size output — corners, edges, interior,
    first and last utterance — against the oracle conv of the matching input windows (a conv is local: output rows
    r..r+h-1 need input rows s r - pad .. s (r+h-1) - pad + k - 1)."""
    from dcsnet import functional as F
    torch.manual_seed(cin + k)
    pad = k // 2
    m = cpt.ComplexConv2d(cin, cout, k, stride, pad)
    B = 16
    x = rand_c((B, cin, H, W), 31, 0.5)
    p = lambda t: t.detach().to(dev)
    y = back(F.cconv2d(nhwc(x, dev), None, p(m.conv_r.weight), p(m.conv_i.weight), p(m.conv_r.bias), p(m.conv_i.bias),
                       False, (k, k), stride, (pad, pad)))
    Ho, Wo = (H + 2 * pad - k) // stride[0] + 1, (W + 2 * pad - k) // stride[1] + 1
    assert y.shape == (B, cout, Ho, Wo)
    ref = cpt.ComplexConv2d(cin, cout, k, stride, 0)                              # same weights, windows padded by hand
    ref.load_state_dict(m.state_dict())
    h, w = min(12, Ho), min(40, Wo)
    for b, r0, c0 in [(0, 0, 0), (0, Ho - h, Wo - w), (7, (Ho - h) // 2, Wo // 2), (15, 0, Wo - w), (15, Ho - h, 0),
                      (9, min(3, Ho - h), 29), (15, Ho - h, Wo - w)]:
        rows, cols = stride[0] * (h - 1) + k, stride[1] * (w - 1) + k
        win = torch.zeros((1, cin, rows, cols), dtype=x.dtype)
        ys, xs = stride[0] * r0 - pad, stride[1] * c0 - pad
        y0, y1, x0, x1 = max(ys, 0), min(ys + rows, H), max(xs, 0), min(xs + cols, W)
        win[0, :, y0 - ys:y1 - ys, x0 - xs:x1 - xs] = x[b, :, y0:y1, x0:x1]
        close(y[b:b + 1, :, r0:r0 + h, c0:c0 + w], ref(win))


# the seven decoder stages at BASELINE configs[1] size: (c1 (previous stage), c2 (skip), cout, upsample, Hs, Ws)
DEC_FULL = [(128, 128, 128, (2, 1), 2, 250), (128, 128, 128, (2, 1), 4, 250), (128, 128, 64, (2, 1), 8, 250),
            (64, 64, 32, (2, 1), 16, 250), (32, 32, 16, (2, 2), 32, 250), (16, 16, 8, (2, 2), 64, 500),
            (8, 8, 1, (2, 2), 128, 1000)]


@pytest.mark.parametrize('c1,c2,cout,up,Hs,Ws', DEC_FULL)
def test_decoder_stages_full_size_windows(dev, c1, c2, cout, up, Hs, Ws):
    """Every decoder conv (ComplexConvTranspose2d 3x3 / stride 1 behind cat + nearest upsample: c_network.py:135-141,
    :214-217) at its BASELINE configs[1] size — the upsample-folded class launches and the one-kernel single-output
    stage as the network calls them — on output windows against the oracle applied to the matching upsampled input
    windows (output rows r..r+h-1 read upsampled rows r-1..r+h)."""
    from dcsnet import functional as F
    torch.manual_seed(c1 + cout)
    m = cpt.ComplexConvTranspose2d(c1 + c2, cout, 3, 1, 1)
    B = 16
    d, sk = rand_c((B, c1, Hs, Ws), 3, 0.5), rand_c((B, c2, Hs, Ws), 4, 0.5)
    p = lambda t: t.detach().to(dev)
    wts = (p(m.conv_tran_r.weight), p(m.conv_tran_i.weight), p(m.conv_tran_r.bias), p(m.conv_tran_i.bias))
    with torch.no_grad():
        if cout == 1:
            y = F.cconv_single_output(nhwc(d, dev), nhwc(sk, dev), *wts, (3, 3), (1, 1), up)
        else:
            y = F.cconv2d(nhwc(d, dev), nhwc(sk, dev), *wts, True, (3, 3), (1, 1), (1, 1), up)
    y = back(y)
    Hu, Wu = Hs * up[0], Ws * up[1]
    assert y.shape == (B, cout, Hu, Wu)
    h, w = min(10, Hu), min(36, Wu)
    for b, r0, c0 in [(0, 0, 0), (0, Hu - h, Wu - w), (7, (Hu - h) // 2, Wu // 2), (15, 0, Wu - w), (15, Hu - h, 0),
                      (15, Hu - h, Wu - w)]:
        ru, cu = torch.arange(r0 - 1, r0 + h + 1), torch.arange(c0 - 1, c0 + w + 1)       # upsampled coordinates
        rv, cv = (ru >= 0) & (ru < Hu), (cu >= 0) & (cu < Wu)
        rs, cs = ru.clamp(0, Hu - 1) // up[0], cu.clamp(0, Wu - 1) // up[1]
        cat = torch.cat((d[b], sk[b]), dim=0)                                            # [C, Hs, Ws]
        win = cat[:, rs][:, :, cs] * (rv[:, None] & cv[None, :])
        with torch.no_grad():
            want = m(win[None])[..., 1:-1, 1:-1]
        close(y[b:b + 1, :, r0:r0 + h, c0:c0 + w], want)


@pytest.mark.parametrize('c1,c2,cout,up', [(128, 128, 128, (2, 1)), (32, 32, 16, (2, 2)), (8, 8, 1, (2, 2)),
                                            (16, 0, 8, (1, 1))])
def test_complex_convtranspose_with_fused_cat_upsample(dev, c1, c2, cout, up):
    from dcsnet import functional as F
    torch.manual_seed(c1 + cout)
    m = cpt.ComplexConvTranspose2d(c1 + c2, cout, 3, 1, 1)
    d = rand_c((2, c1, 6, 10), 3)
    s = rand_c((2, c2, 6, 10), 4) if c2 else None
    cat = torch.cat((d, s), dim=1) if c2 else d
    want = m(cpt.complex_upsample(cat, scale_factor=up, mode='nearest'))
    p = lambda t: t.detach().to(dev)
    y = F.cconv2d(nhwc(d, dev), nhwc(s, dev) if c2 else None, p(m.conv_tran_r.weight), p(m.conv_tran_i.weight),
                  p(m.conv_tran_r.bias), p(m.conv_tran_i.bias), True, (3, 3), (1, 1), (1, 1), up)
    close(back(y), want)


def test_conv_edge_shapes_ragged_tiles_and_1x1(dev):
    from dcsnet import functional as F
    torch.manual_seed(5)
    p = lambda t: t.detach().to(dev)
    # output 19x5: not a multiple of the 16x16 tile; batch 1
    m = cpt.ComplexConv2d(4, 2, 3, (1, 2), 1)
    x = rand_c((1, 4, 19, 9), 8)
    y = F.cconv2d(nhwc(x, dev), None, p(m.conv_r.weight), p(m.conv_i.weight), p(m.conv_r.bias), p(m.conv_i.bias),
                  False, (3, 3), (1, 2), (1, 1))
    close(back(y), m(x))
    # bias-free 1x1 (channel-attention FC shape) on a 1x1 map
    m = cpt.ComplexConv2d(16, 1, 1, bias=False)
    x = rand_c((3, 16, 1, 1), 9)
    y = F.cconv2d(nhwc(x, dev), None, p(m.conv_r.weight), p(m.conv_i.weight), None, None, False, (1, 1), (1, 1), (0, 0))
    close(back(y), m(x))


def test_conv_source_of_two_gib_and_more_runs_as_sub_batches(dev):
    """the MFMA conv's source-offset tables hold 32-bit byte offsets; a source tensor of >= 2 GiB (here 16 channels x
    66 x 512 x 512 pixels in fp32 = 2.07 GiB, forward and data gradient) must still run — as sub-batch launches — and give what
    the same samples give in a small batch (fp32 accumulation order: the plans may differ), statistics rows included."""
    from dcsnet import ops
    B, H, W, C, Cout, k = 66, 512, 512, 16, 16, 3
    g = torch.Generator().manual_seed(3)
    w_r, w_i = (torch.randn(Cout, C, k, k, generator=g) * 0.1).to(dev), (torch.randn(Cout, C, k, k, generator=g) * 0.1).to(dev)
    b_r, b_i = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, False, (1, 1))
    x = torch.empty(B, H, W, C, 2, device=dev)
    assert x.numel() * 4 >= 2 ** 31
    for b in range(B):
        x[b].normal_(generator=None)
    y, stat = ops.cconv2d_stats(x, None, wp, bias, (k, k), (1, 1), (1, 1), (1, 1))
    assert stat is not None
    for sl in (slice(0, 2), slice(B - 2, B), slice(31, 33)):              # first / last sub-batch and (with 64 per launch) across the seam
        ys = ops.cconv2d(x[sl].contiguous(), None, wp, bias, (k, k), (1, 1), (1, 1), (1, 1))
        close(y[sl], ys)
    # the statistics rows cover every sample exactly once: their sums are the moments of the whole output
    raw = y - torch.stack((b_r - b_i, b_r + b_i), dim=-1)               # (the moments are taken about the effective bias)
    mom = stat[0][:, :, :stat[1]].double().sum(dim=2)                    # [C, 5]
    want = torch.stack((raw[..., 0].double().sum(dim=(0, 1, 2)), raw[..., 1].double().sum(dim=(0, 1, 2)),
                        (raw[..., 0].double() ** 2).sum(dim=(0, 1, 2)), (raw[..., 1].double() ** 2).sum(dim=(0, 1, 2)),
                        (raw[..., 0].double() * raw[..., 1].double()).sum(dim=(0, 1, 2))), dim=1)
    close(mom.float(), want.float(), rel=1e-4)
    del raw, want
    wpb = ops.pack_conv_weight_bwd(wp, (k, k), (1, 1), (1, 1), (1, 1))
    gx = ops.cconv2d_bwd_data(y, wpb, (H, W, C), (k, k), (1, 1), (1, 1), (1, 1), C)[0]
    for sl in (slice(0, 2), slice(B - 2, B)):
        gs = ops.cconv2d_bwd_data(y[sl].contiguous(), wpb, (H, W, C), (k, k), (1, 1), (1, 1), (1, 1), C)[0]
        close(gx[sl], gs)


def test_complex_linear(dev):
    from dcsnet import functional as F
    torch.manual_seed(6)
    m = cpt.ComplexLinear(128, 128)
    z = rand_c((2, 24, 128), 2)
    p = lambda t: t.detach().to(dev)
    y = F.complex_linear(z.to(dev), p(m.fc_r.weight), p(m.fc_i.weight), p(m.fc_r.bias), p(m.fc_i.bias))
    close(y, m(z))


@pytest.mark.parametrize('B,S', [(2, 8), (3, 64), (1, 1)])
def test_complex_lstm_persistent_kernel(dev, B, S):
    from dcsnet.c_network import ComplexLSTM
    torch.manual_seed(B * 100 + S)
    ref = cno.ComplexLSTM(128, 64, 2, True)
    mod = ComplexLSTM(128, 64, 2, True, True)
    mod.load_state_dict(ref.state_dict())
    z = rand_c((B, S, 128), 4, 0.8)
    with torch.no_grad():
        want = ref(z)
        got = mod.to(dev)(z.to(dev))
    close(got, want, rel=2e-5, abs_=1e-6)


# --------------------------------------------------------------------------------- batch norm

@pytest.mark.parametrize('C,shape', [(1, (2, 16, 24)), (1, (1, 3, 5)), (8, (2, 12, 10)), (64, (3, 6, 8)), (128, (2, 4, 8))])
@pytest.mark.parametrize('act', ['none', 'relu', 'lrelu'])
def test_cbn_train_and_eval(dev, C, shape, act):
    from dcsnet import ops
    B, H, W = shape
    bn = cpt.ComplexBatchNorm2d(C)
    fill_state(bn, seed=C)
    x = rand_c((B, C, H, W), C + 1, 1.3) + (0.4 - 0.2j)
    rm0, rc0 = bn.running_mean.clone(), bn.running_covar.clone()
    post = {'none': lambda z: z, 'relu': cpt.complex_relu, 'lrelu': nf.complex_lrelu}[act]
    code = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'lrelu': ops.ACT_LRELU}[act]
    bn.train()
    want_tr = post(bn(x))
    bn.eval()
    want_ev = post(bn(x))          # with the UPDATED running stats
    p = lambda t: t.detach().to(dev).contiguous()
    rm = torch.view_as_real(rm0).to(dev).contiguous()
    rc = rc0.to(dev).contiguous()
    y, stats, coef = ops.cbn(nhwc(x, dev), p(bn.weight), p(bn.bias), rm, rc, bn.eps, bn.momentum, True, code)
    close(back(y), want_tr, rel=3e-5)
    close(torch.view_as_complex(rm.cpu()), bn.running_mean, rel=1e-5, abs_=1e-7)
    close(rc, bn.running_covar, rel=1e-5)
    y2, _, _ = ops.cbn(nhwc(x, dev), p(bn.weight), p(bn.bias), rm, rc, bn.eps, bn.momentum, False, code)
    close(back(y2), want_ev, rel=3e-5)


# (B, Hin, Win, C1, C2, Cout, k, stride, pad, up, transposed): every conv -> CBN pair of the network at reduced size (the
# 7x7 one-channel layer, the emulated row kernel, split-K layers, upsample-folded decoder classes incl. a class smaller than
# the tile grid, the 16-column kernel) + ragged maps
_STAT_GEOMS = [
    (2, 32, 40, 1, 0, 8, 7, (2, 2), 3, (1, 1), False),          # enc0: conv_enc0.hip, persistent workgroups
    (3, 18, 22, 1, 0, 8, 7, (2, 2), 3, (1, 1), False),          # ... ragged
    (2, 32, 32, 8, 0, 16, 7, (2, 2), 3, (1, 1), False),         # enc1
    (2, 16, 24, 16, 0, 32, 5, (2, 2), 2, (1, 1), False),        # enc2
    (2, 16, 16, 32, 0, 64, 5, (2, 1), 2, (1, 1), False),        # enc3
    (2, 8, 32, 64, 0, 128, 3, (2, 1), 1, (1, 1), False),        # enc4
    (2, 8, 32, 128, 0, 128, 3, (2, 1), 1, (1, 1), False),       # enc5 (split-K at this size)
    (32, 4, 32, 128, 0, 128, 3, (2, 1), 1, (1, 1), False),      # enc6 at the train batch (split-K + reduce with statistics)
    (2, 2, 32, 128, 128, 128, 3, (1, 1), 1, (2, 1), True),      # dec0: classes, split-K
    (2, 8, 32, 128, 128, 64, 3, (1, 1), 1, (2, 1), True),       # dec2
    (2, 16, 16, 32, 32, 16, 3, (1, 1), 1, (2, 2), True),        # dec4: four classes, 32-column tiles
    (2, 24, 20, 16, 16, 8, 3, (1, 1), 1, (2, 2), True),         # dec5: the 16-column kernel, ragged
    (1, 5, 7, 16, 16, 8, 3, (1, 1), 1, (2, 2), True),           # ... tiny: tiles outside the map
]


@pytest.mark.parametrize('geom', _STAT_GEOMS, ids=[f'g{i}' for i in range(len(_STAT_GEOMS))])
@pytest.mark.parametrize('mode', ['bf16x6', 'f32'])
def test_cbn_statistics_from_the_conv_epilogue(dev, geom, mode):
    """dcs_cconv2d_fwd_stats + dcs_cbn_fwd_slabs against dcs_cconv2d_fwd + dcs_cbn_fwd (itself pinned to the oracle by
    test_cbn_train_and_eval): the conv output bit-identical, the moments / coefficients / running statistics / output equal
    to summation order, and every geometry of the network really takes the statistics epilogue."""
    from dcsnet import ops, functional as F
    B, H, W, C1, C2, Cout, k, st, pad, up, tr = geom
    g = torch.Generator().manual_seed(11)
    x1 = (torch.randn(B, H, W, C1, 2, generator=g) + 0.3).to(dev)
    x2 = (torch.randn(B, H, W, C2, 2, generator=g) - 0.2).to(dev) if C2 else None
    Cin = C1 + C2
    shape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    w_r, w_i = (torch.randn(shape, generator=g) * 0.1).to(dev), (torch.randn(shape, generator=g) * 0.1).to(dev)
    b_r, b_i = (torch.randn(Cout, generator=g) * 2).to(dev), (torch.randn(Cout, generator=g) * 2).to(dev)
    bnw = (torch.randn(Cout, 3, generator=g) * 0.3 + torch.tensor([1.2, 1.1, 0.1])).to(dev)
    bnb = torch.randn(Cout, 2, generator=g).to(dev)
    default = ops.conv_precision()
    ops.set_conv_precision(mode)
    try:
        wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, tr, up)
        ks, pd = (k, k), ((k - 1 - pad, k - 1 - pad) if tr else (pad, pad))
        y0 = ops.cconv2d(x1, x2, wp, bias, ks, st, pd, up, ops.ACT_NONE)
        rm0, rc0 = torch.zeros(Cout, 2, device=dev), torch.ones(Cout, 3, device=dev)
        a0, stats0, coef0 = ops.cbn(y0, bnw, bnb, rm0, rc0, 1e-5, 0.1, True, ops.ACT_LRELU)
        y1, stat = ops.cconv2d_stats(x1, x2, wp, bias, ks, st, pd, up)
        assert stat is not None, 'this geometry must take the statistics epilogue'
        assert torch.equal(y0, y1)
        rm1, rc1 = torch.zeros(Cout, 2, device=dev), torch.ones(Cout, 3, device=dev)
        a1, stats1, coef1 = ops.cbn(y1, bnw, bnb, rm1, rc1, 1e-5, 0.1, True, ops.ACT_LRELU, stat=stat)
        y2, stat2 = ops.cconv2d_stats(x1, x2, wp, bias, ks, st, pd, up)
        assert torch.equal(stat2[0][:, :, :stat2[1]], stat[0][:, :, :stat[1]])     # fixed summation order: bitwise repeatable
    finally:
        ops.set_conv_precision(default)
    close(stats1, stats0, rel=2e-5, abs_=1e-6)
    close(coef1, coef0, rel=2e-5, abs_=1e-6)
    close(rm1, rm0, rel=2e-5, abs_=1e-7)
    close(rc1, rc0, rel=2e-5, abs_=1e-7)
    close(a1, a0, rel=2e-5, abs_=1e-6)


def test_cbn_large_offset_is_stable(dev):
    """mean >> std: the pivoted one-pass statistics must not cancel catastrophically."""
    from dcsnet import ops
    C = 8
    bn = cpt.ComplexBatchNorm2d(C)
    x = rand_c((4, C, 32, 32), 3, 0.05) + (30.0 - 20.0j)
    bn.train()
    want = bn(x)
    p = lambda t: t.detach().to(dev).contiguous()
    rm = torch.zeros(C, 2, device=dev)
    rc = torch.ones(C, 3, device=dev)
    y, _, _ = ops.cbn(nhwc(x, dev), p(bn.weight), p(bn.bias), rm, rc, bn.eps, 0.1, True, ops.ACT_NONE)
    close(back(y), want, rel=2e-3)       # the CPU two-pass result itself carries ~1e-3 here


def test_dropout_keep_rate_and_scale(dev):
    from dcsnet import ops
    x = torch.ones(1 << 20, device=dev)
    y = ops.dropout(x, 0.1, 1234)
    kept = float((y != 0).float().mean())
    assert abs(kept - 0.9) < 3e-3
    vals = torch.unique(y)
    assert torch.allclose(vals, torch.tensor([0.0, 1 / 0.9], device=dev))
    assert torch.equal(y, ops.dropout(x, 0.1, 1234))          # regenerated, not random
    assert not torch.equal(y, ops.dropout(x, 0.1, 1235))
    assert torch.equal(ops.dropout(x, 0.0, 1), x)


# --------------------------------------------------------------------------------- attention

@pytest.mark.parametrize('C,hw', [(8, (20, 12)), (64, (6, 10)), (128, (2, 8))])
def test_channel_and_spatial_attention(dev, C, hw):
    from dcsnet import functional as F
    torch.manual_seed(C)
    ca_m = cno.ComplexChannelAttention(C, 16)
    sa_m = cno.ComplexSpatialAttention(7)
    x = rand_c((3, C, *hw), C + 2)
    want_ca = ca_m(x)
    z = want_ca * x
    want_sa = sa_m(z)
    want_out = want_sa * z
    p = lambda t: t.detach().to(dev)
    xn = nhwc(x, dev)
    ca = F.channel_attention(xn, p(ca_m.fc[0].conv_r.weight), p(ca_m.fc[0].conv_i.weight),
                             p(ca_m.fc[2].conv_r.weight), p(ca_m.fc[2].conv_i.weight))
    close(torch.view_as_complex(ca.cpu()), want_ca.view(3, C), rel=2e-5)
    sa = F.spatial_attention(xn, ca, p(sa_m.conv1.conv_r.weight), p(sa_m.conv1.conv_i.weight), 7)
    close(back(sa), want_sa, rel=2e-5)
    out = F.attention_apply(xn, ca, sa)
    close(back(out), want_out, rel=2e-5)
    # identity attention operands
    close(back(F.attention_apply(xn, None, None)), x, rel=0, abs_=0)


# --------------------------------------------------------------------------------- mask math

@pytest.fixture(scope='module')
def nfv(golden_dir):
    return np.load(os.path.join(golden_dir, 'nf_vectors.npz'))


@pytest.mark.parametrize('tag', ['small', 'mid'])
def test_mask_math_against_reference_vectors(dev, nfv, tag):
    from dcsnet import functional as F
    t = lambda k: torch.from_numpy(nfv[f'{tag}_{k}'])
    M, Y, S = t('M'), t('Y'), t('S')
    b1 = F.bound_crm_complex(M.to(dev))
    close(b1, t('bound1'), rel=0, abs_=2e-6)
    m2, nhat, shat = F.bound_mask_apply_complex(Y.to(dev), b1)
    close(m2, t('bound2'), rel=0, abs_=2e-6)
    close(nhat, t('nhat'), rel=0, abs_=2e-6 * float(Y.abs().max()))
    close(shat, t('shat'), rel=0, abs_=2e-6 * float(Y.abs().max()))
    # cRM divides by |Y|^2 + 1e-8: compare relative to each element's own magnitude
    got, want = F.crm_complex(S.to(dev), Y.to(dev)).cpu(), t('cRM')
    assert float(((got - want).abs() / (want.abs() + 1e-3)).max()) < 1e-5
    close(F.complex_lrelu(M.to(dev)), t('lrelu'), rel=0, abs_=0)
    close(F.complex_sigmoid(M.to(dev)), t('sigmoid'), rel=0, abs_=3e-7)


def test_mask_apply_properties_full_size(dev):
    """BASELINE config-2 size [16,256,2000]: size-independent properties of the subtractive step."""
    from dcsnet import functional as F
    g = torch.Generator(device='cpu').manual_seed(0)
    Y = torch.complex(torch.randn(16, 256, 2000, generator=g), torch.randn(16, 256, 2000, generator=g)).to(dev)
    M = torch.complex(torch.randn(16, 256, 2000, generator=g), torch.randn(16, 256, 2000, generator=g)).to(dev) * 2
    m, nhat, shat = F.bound_mask_apply_complex(Y, M)
    assert float(m.abs().max()) <= 1.0 + 2e-7                           # bounded (fp32 tanh saturates at 1.0)
    assert torch.allclose(m.abs(), torch.tanh(M.abs()), atol=2e-6)      # modulus is tanh|M|
    assert torch.equal(nhat + shat, Y) or float((nhat + shat - Y).abs().max()) < 1e-6   # S = Y - N
    assert torch.allclose(nhat, Y * m, atol=1e-5)
    # idempotence of the direction: bounding a bounded mask keeps its phase
    m2 = F.bound_crm_complex(m)
    ang = torch.angle(m2 * torch.conj(m))
    big = m.abs() > 0.1                       # eps = 1e-6 on the real part rotates tiny masks by ~eps/|m|
    assert float(ang[big].abs().max()) < 1e-4


# --------------------------------------------------------------------------------- whole network

@pytest.fixture(scope='module')
def cnv(golden_dir):
    return np.load(os.path.join(golden_dir, 'cnet_vectors.npz'))


def _hip_net(dev, seed, dropout=False):
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    hp = dict(hparams)
    if not dropout:
        hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    net = C_NETWORK(config, hp, seed)
    fill_state(net, seed)
    return net.to(dev)


@pytest.mark.parametrize('conv_mode', ['bf16x6', 'f32'])
@pytest.mark.parametrize('tag,seed', [('b2t32', 0), ('b1t16', 1), ('b3t8', 2)])
def test_network_forward_against_reference_vectors(dev, cnv, tag, seed, conv_mode):
    from dcsnet import ops
    default = ops.conv_precision()
    ops.set_conv_precision(conv_mode)          # both fp32 arithmetic modes: the bf16-MFMA emulation (default) and the native MFMA
    try:
        _network_forward_against_reference_vectors(dev, cnv, tag, seed)
    finally:
        ops.set_conv_precision(default)


def _network_forward_against_reference_vectors(dev, cnv, tag, seed):
    net = _hip_net(dev, seed)
    x = torch.from_numpy(cnv[f'{tag}_x']).to(dev)
    net.eval()
    with torch.no_grad():
        ev = net(x)
        ev2 = net(x)          # second pass: cached inference constants, CBNs folded into the conv epilogues
    close(ev, torch.from_numpy(cnv[f'{tag}_eval']), rel=0, abs_=2e-4)
    assert torch.equal(ev, ev2)
    net.train()
    with torch.no_grad():
        tr = net(x)
    close(tr, torch.from_numpy(cnv[f'{tag}_train']), rel=0, abs_=2e-4)
    sd = net.state_dict()
    for k in ('initial_batchnorm.running_mean', 'initial_batchnorm.running_covar', 'encoder.3.1.running_mean',
              'encoder.3.1.running_covar', 'decoder.2.1.running_mean', 'decoder.2.1.running_covar'):
        close(sd[k], torch.from_numpy(cnv[f'{tag}_after_{k}']), rel=1e-4, abs_=1e-6)
    assert int(sd['encoder.0.1.num_batches_tracked']) == 1


def test_network_forward_against_oracle_t256(dev):
    """Native patch size [2,256,256] (config.py:72-75), oracle run here on the host cores."""
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    seed = 5
    oracle = fill_state(cno.C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), seed).eval()
    net = _hip_net(dev, seed).eval()
    x = seeded_input(2, 256, 256, seed=seed)
    with torch.no_grad():
        want, inter = oracle(x, return_intermediates=True)
        got = net(x.to(dev))
    close(got, want, rel=0, abs_=2e-4)


def test_layerwise_dropin_modules(dev):
    """The complexPyTorch-surface modules, called one by one on NCHW complex tensors."""
    from dcsnet import complexLayers as L
    from dcsnet import complexFunctions as CF
    torch.manual_seed(3)
    x = rand_c((2, 8, 12, 10), 1)
    ref = cpt.ComplexConv2d(8, 16, 5, (2, 1), 2)
    mod = L.ComplexConv2d(8, 16, 5, (2, 1), 2)
    mod.load_state_dict(ref.state_dict())
    y = mod.to(dev)(x.to(dev))
    assert y.shape == (2, 16, 6, 10)
    close(y, ref(x))
    rbn, mbn = cpt.ComplexBatchNorm2d(16), L.ComplexBatchNorm2d(16)
    fill_state(rbn, 4)
    mbn.load_state_dict(rbn.state_dict())
    mbn.to(dev).train()
    rbn.train()
    close(mbn(y), rbn(ref(x)), rel=5e-5)
    close(mbn.running_covar, rbn.running_covar, rel=1e-5)
    close(CF.complex_relu(y), cpt.complex_relu(ref(x)))
    close(CF.complex_upsample(y, scale_factor=(2, 1)), cpt.complex_upsample(ref(x), scale_factor=(2, 1)))
    rt, mt = cpt.ComplexConvTranspose2d(16, 4, 3, 1, 1), L.ComplexConvTranspose2d(16, 4, 3, 1, 1)
    mt.load_state_dict(rt.state_dict())
    close(mt.to(dev)(y), rt(ref(x)))


def test_ops_fail_loudly_on_cpu_tensors():
    from dcsnet import ops, DcsHipError
    with pytest.raises(DcsHipError):
        ops.bound_crm(torch.zeros(4, 2))


def test_device_stft_front_end_matches_the_loader_side_stft(dev):
    """dcsnet.frontend.stft_batch (HIP framing + rocFFT + HIP bin slice / transpose) vs the three torch.stft calls of
    the reference's Dataset (data.py:104-134, restated in oracle.nf_oracle.stft_frontend)."""
    from dcsnet.frontend import stft_batch
    from dcsnet.config import config
    g = torch.Generator().manual_seed(3)
    for B, T in ((3, 16), (2, 256)):
        L = 32 * (T - 1)
        clean = 0.1 * torch.randn(B, L, generator=g)
        noisy = clean + 0.05 * torch.randn(B, L, generator=g)
        want = nf.stft_frontend(clean, noisy)
        got = stft_batch(clean.to(dev), noisy.to(dev), config)
        for name, a, b in zip(('noise', 'noisy', 'clean'), got, want):
            assert a.shape == b.shape == (B, 256, T) and a.is_contiguous(), name
            err = float((a.cpu() - b).abs().max())
            assert err <= 1e-5 * float(b.abs().max()) + 1e-7, (name, T, err)


def _bias_before_bn(name):
    """conv biases directly in front of a batch-statistics CBN have analytically zero gradients (noise only)"""
    import re
    return bool(re.match(r'(encoder\.\d+\.0|decoder\.[0-5]\.0)\..*bias$', name))


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def test_bf16_operand_mode_of_the_mfma_conv(dev):
    """dcs_set_conv_precision(1): bf16 operands, fp32 accumulate (BASELINE configs[4]).  Against the fp32 oracle run on
    bf16-ROUNDED inputs and weights the kernel must agree to fp32 accumulation order (the products of two bf16 numbers
    are exact in fp32); against the unrounded oracle it is a bf16 computation: 2e-2 of the output's max-abs."""
    from dcsnet import functional as F, ops
    torch.manual_seed(11)
    m = cpt.ComplexConv2d(32, 64, 3, (2, 1), 1)
    x = rand_c((2, 32, 12, 20), 5)
    mr = cpt.ComplexConv2d(32, 64, 3, (2, 1), 1)
    with torch.no_grad():
        for a, b in zip(mr.parameters(), m.parameters()):
            a.copy_(_bf16_round(b) if b.dim() == 4 else b)                 # biases stay fp32 (added in the epilogue)
        want_exact = mr(torch.complex(_bf16_round(x.real), _bf16_round(x.imag)))
        want_f32 = m(x)
    p = {n: q.detach().to(dev) for n, q in m.named_parameters()}
    xn = ops.to_nhwc(x.to(dev))
    default = ops.conv_precision()
    assert default in ('f32', 'bf16x6')                # the library default, or DCS_CONV_PRECISION=0 for a native-MFMA run
    try:
        ops.set_conv_precision('bf16')
        assert ops.conv_precision() == 'bf16'
        y = F.from_nhwc(F.cconv2d(xn, None, p['conv_r.weight'], p['conv_i.weight'], p['conv_r.bias'], p['conv_i.bias'],
                                  False, (3, 3), (2, 1), (1, 1))).cpu()
    finally:
        ops.set_conv_precision(default)
    scale = float(want_f32.abs().max())
    assert float((y - want_exact).abs().max()) <= 2e-5 * scale
    err = float((y - want_f32).abs().max())
    assert 1e-6 * scale < err <= 2e-2 * scale          # genuinely a bf16 computation, within bf16 tolerance


def test_bf16_mode_whole_network_and_gradients(dev):
    """The whole network in bf16-operand mode (folded / strided class kernels, split-K, data gradients) against the fp32
    oracle, as relative L2 error: train-mode mask 2e-2; gradients of all parameters together 3e-2, taken with running
    statistics (eval) — through BATCH statistics of a 2-utterance, 32-frame batch the same 2^-8 operand rounding is
    amplified to ~1e-1 (6e-2 at B=8, T=64: tools/bf16_grad_probe.py), which measures the batch, not the kernels."""
    from dcsnet import ops
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    x = seeded_input(2, 256, 32, seed=4)
    w = None

    def rel_l2(a, b):
        return float((a - b).abs().pow(2).sum().sqrt()) / (float(b.abs().pow(2).sum().sqrt()) + 1e-20)

    def loss_of(out, w_):
        return (w_ * (out.real ** 2 + 0.5 * out.imag ** 2)).sum()

    with torch.no_grad():                                  # (its own instance: a train-mode forward moves the running stats)
        mask_train = fill_state(cno.C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), 6).train()(x)
    ref = fill_state(cno.C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), 6).eval()
    out_r = ref(x)
    w = torch.rand(out_r.shape, generator=torch.Generator().manual_seed(2))
    loss_of(out_r, w).backward()
    default = ops.conv_precision()
    try:
        ops.set_conv_precision('bf16')
        net = fill_state(C_NETWORK(config, hp, 0), 6).to(dev)
        net.train()
        with torch.no_grad():
            got_train = net(x.to(dev)).cpu()
        net = fill_state(C_NETWORK(config, hp, 0), 6).to(dev).eval()
        out = net(x.to(dev))
        loss_of(out, w.to(dev)).backward()
        grads = {n: p.grad.detach().cpu() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        ops.set_conv_precision(default)
    e_mask = rel_l2(got_train, mask_train)
    pr = dict(ref.named_parameters())
    num = sum(float((g - pr[n].grad).pow(2).sum()) for n, g in grads.items()) ** 0.5
    e_all = num / sum(float(pr[n].grad.pow(2).sum()) for n in grads) ** 0.5
    print(f'bf16 mode: train-mode mask rel-L2 {e_mask:.3e}; eval-mode gradients rel-L2 {e_all:.3e}')
    assert 1e-4 < e_mask <= 2e-2, e_mask
    assert e_all <= 3e-2, e_all


# ---- BASELINE configs[0]: DR-Net (r_network.py) on the HIP path -------------------------------------------------

def test_real_conv_on_the_mfma_kernel(dev):
    """dcs_rconv2d_fwd: a real conv (cat of two sources, nearest upsample, stride) = the complex kernel's GEMM with an
    unstructured B panel, against torch.nn.functional.conv2d."""
    from dcsnet import r_network as rn
    g = torch.Generator().manual_seed(3)
    cases = [(2, 12, 20, 32, 0, 48, 3, (2, 1), (1, 1)), (3, 6, 8, 16, 16, 16, 3, (1, 1), (2, 2)),
             (2, 9, 7, 32, 32, 64, 5, (2, 2), (1, 1)), (1, 4, 32, 256, 256, 128, 3, (1, 1), (2, 1))]
    for B, H, W, c1, c2, co, k, stride, up in cases:
        x1 = torch.randn((B, H, W, c1), generator=g)
        x2 = torch.randn((B, H, W, c2), generator=g) if c2 else None
        w = torch.randn((co, c1 + c2, k, k), generator=g) * 0.1
        b = torch.randn(co, generator=g)
        xin = x1 if x2 is None else torch.cat([x1, x2], -1)
        xin = xin.permute(0, 3, 1, 2)
        if up != (1, 1):
            xin = torch.nn.functional.interpolate(xin, scale_factor=up, mode='nearest')
        want = torch.nn.functional.conv2d(xin, w, b, stride, k // 2).permute(0, 2, 3, 1)
        got = rn.rconv2d(x1.to(dev), None if x2 is None else x2.to(dev), rn.pack_real_panel(w.to(dev)), b.to(dev), co,
                         (k, k), stride, (k // 2, k // 2), up)
        close(got, want, rel=2e-5)


@pytest.mark.parametrize('tag,B,T', [('b1t256', 1, 256), ('b2t32', 2, 32)])
def test_rnetwork_forward_against_reference_vectors(dev, golden_dir, tag, B, T):
    """R_NETWORK (DR-Net) forward on the HIP path against the outputs of the reference's own r_network.py
    (tests/golden/rnet_vectors.npz: seeded state -> eval and train-mode output) — configs[0]."""
    from dcsnet.config import config, hparams
    from dcsnet.r_network import R_NETWORK
    from oracle.seeded_state import fill_state_stream
    v = np.load(os.path.join(golden_dir, 'rnet_vectors.npz'))
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    net = fill_state_stream(R_NETWORK(config, hp, 0), 5).to(dev)
    assert sum(p.numel() for p in net.parameters()) == 5808753
    x = torch.from_numpy(v[f'{tag}_x']).to(dev)
    net.eval()
    with torch.no_grad():
        ev = net(x)
        ev2 = net(x)                                                     # cached panels / coefficients
    assert torch.equal(ev, ev2)
    want = torch.from_numpy(v[f'{tag}_eval'])
    assert ev.shape == want.shape                                         # [256, T] when B == 1 (squeeze quirk)
    close(ev, want, rel=0, abs_=2e-5)
    net.train()
    with torch.no_grad():
        tr = net(x)
    close(tr, torch.from_numpy(v[f'{tag}_train']), rel=0, abs_=5e-5)


@pytest.mark.parametrize('C,hw', [(16, (37, 5)), (64, (9, 11)), (256, (2, 32))])
def test_real_attention_pair(dev, C, hw):
    """dcs_rattention_fwd against the oracle's RealChannelAttention / RealSpatialAttention (r_network.py:8-42)."""
    from oracle.rnet_oracle import RealChannelAttention as OCA, RealSpatialAttention as OSA
    from dcsnet.r_network import R_NETWORK, RealChannelAttention, RealSpatialAttention
    torch.manual_seed(C)
    oca, osa = OCA(C, 16), OSA(7)
    x = torch.randn(3, C, *hw)
    with torch.no_grad():
        z = oca(x) * x
        want = (osa(z) * z).permute(0, 2, 3, 1)
    ca, sa = RealChannelAttention(C, 16), RealSpatialAttention(7)
    ca.load_state_dict(oca.state_dict()); sa.load_state_dict(osa.state_dict())
    ca, sa = ca.to(dev), sa.to(dev)
    with torch.no_grad():
        got = R_NETWORK._attend(ca, sa, x.permute(0, 2, 3, 1).contiguous().to(dev))
    close(got, want, rel=2e-5)


@pytest.mark.parametrize('C,hw,relu_in', [(16, (37, 5), True), (64, (9, 11), False), (256, (2, 32), True), (32, (16, 16), True)])
def test_real_attention_pair_backward(dev, C, hw, relu_in):
    """The training node of the DR-Net attention pair (_RAttendFn: dcs_rattention_pool/apply_fwd/_bwd + the complex conv
    entries) against autograd through the oracle's modules on the CPU (r_network.py:8-42: AdaptiveMaxPool2d and
    torch.max(dim=1), whose gradients go to the FIRST maximum).  relu_in: a post-ReLU input (exact zeros: ties in the
    per-pixel channel maximum, two all-zero pixels included) as the skip attentions see it (r_network.py:155-158)."""
    from oracle.rnet_oracle import RealChannelAttention as OCA, RealSpatialAttention as OSA
    from dcsnet.r_network import R_NETWORK, RealChannelAttention, RealSpatialAttention
    torch.manual_seed(C + 1)
    oca, osa = OCA(C, 16), OSA(7)
    x = torch.randn(3, C, *hw)
    if relu_in:
        x = torch.relu(x)
        x[0, :, 0, 0] = 0.0
        x[2, :, -1, -1] = 0.0
    gy = torch.randn(3, C, *hw)
    xo = x.clone().requires_grad_(True)
    z = oca(xo) * xo
    (osa(z) * z).backward(gy)
    ca, sa = RealChannelAttention(C, 16), RealSpatialAttention(7)
    ca.load_state_dict(oca.state_dict()); sa.load_state_dict(osa.state_dict())
    ca, sa = ca.to(dev), sa.to(dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    y = R_NETWORK._attend(ca, sa, xd)
    assert type(y.grad_fn).__name__ == '_RAttendFnBackward'
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    close(y.detach(), (osa(z) * z).detach().permute(0, 2, 3, 1), rel=2e-5)
    close(xd.grad, xo.grad.permute(0, 2, 3, 1), rel=5e-5)
    close(ca.fc[0].weight.grad, oca.fc[0].weight.grad, rel=5e-5)
    close(ca.fc[2].weight.grad, oca.fc[2].weight.grad, rel=5e-5)
    close(sa.conv1.weight.grad, osa.conv1.weight.grad, rel=5e-5)


@pytest.mark.parametrize('frames', [(3, 7), (2, 64), (1, 1)])
def test_fft512_pair(dev, frames):
    """dcs_irfft512_frames / dcs_rfft512_frames (fft512.hip) against torch.fft on the CPU: unnormalised inverse of
    one-sided spectra with COMPLEX bins 0 / 256 (their imaginary parts are ignored, as by torch's c2r), forward without
    scaling.  Tolerance: 2e-6 of the output scale (fp32 radix-4, 512 points)."""
    from dcsnet import ops
    g = torch.Generator().manual_seed(sum(frames))
    X = torch.complex(torch.randn((*frames, 257), generator=g), torch.randn((*frames, 257), generator=g))
    want = torch.fft.irfft(X, n=512, dim=-1, norm='forward')
    got = ops.irfft512(torch.view_as_real(X).contiguous().to(dev))
    close(got, want, rel=2e-6)
    y = torch.randn((*frames, 512), generator=g)
    close(ops.rfft512(y.to(dev)), torch.view_as_real(torch.fft.rfft(y, dim=-1)), rel=2e-6)


@pytest.mark.parametrize('B,T,hop', [(3, 40, 128), (1, 2, 128), (2, 33, 64), (2, 19, 32), (1, 256, 32)])      # (32: the configured hop)
def test_forward_fft_of_the_overlap_add_adjoint_without_the_frames(dev, B, T, hop):
    """dcs_rfft512_ola_frames (the synthesis backward, network_functions.py:140-150: rfft of the windowed cotangent frames,
    read on the fly from the cotangent of the waveform) against dcs_istft_ola_bwd + dcs_rfft512_frames: same products in the
    same order, bit-identical."""
    from dcsnet import ops
    g = torch.Generator().manual_seed(B * T + hop)
    window = torch.hann_window(512).to(dev)
    inv_env = ops.istft_envelope(window, T, hop)
    gy = torch.randn(B, hop * (T - 1), generator=g).to(dev)
    want = ops.rfft512(ops.istft_ola((B, T, 512), window, inv_env, hop, 0.37, grad=gy))
    got = ops.rfft512_ola(gy, window, inv_env, T, hop, 0.37)
    assert tuple(got.shape) == (B, T, 257, 2) and torch.equal(got, want)


@pytest.mark.parametrize('B,T,hop', [(3, 40, 128), (1, 2, 128), (2, 33, 64), (2, 257, 128), (1, 14, 256), (2, 17, 128)])
def test_inverse_fft_and_overlap_add_in_one_kernel(dev, B, T, hop):
    """dcs_irfft512_ola_frames (torch.istft's synthesis behind the inverse FFT, network_functions.py:140-150, the frames staying in
    LDS) against dcs_irfft512_frames + dcs_istft_ola_fwd: the same sums in the same order, bit-identical — frame counts below,
    at and above a workgroup's 16 frames, every supported hop."""
    from dcsnet import ops
    g = torch.Generator().manual_seed(B * T + hop)
    window = torch.hann_window(512).to(dev) + 0.01
    inv_env = ops.istft_envelope(window, T, hop)
    X = torch.randn(B, T, 257, 2, generator=g).to(dev)
    want = ops.istft_ola(ops.irfft512(X), window, inv_env, hop, 0.37)
    got = ops.irfft512_ola(X, window, inv_env, hop, 0.37)
    assert tuple(got.shape) == (B, hop * (T - 1)) and torch.equal(got, want)


def test_rnetwork_gradients_against_reference_vectors(dev, golden_dir):
    """DR-Net training path on the HIP kernels: train-mode forward + backward of R_NETWORK against the gradients of the
    reference's own r_network.py (tests/golden/rnet_grad_vectors.npz — stock torch layers, no stand-in anywhere): the
    real MFMA conv and its data gradient, weight gradients through the complex kernels (x and conj x), BatchNorm on the
    CBN kernels, the H = 128 LSTM recurrence + BPTT, enc0 / dec6 / attention convs through the complex conv nodes."""
    from dcsnet.config import config, hparams
    from dcsnet.r_network import R_NETWORK
    from oracle.seeded_state import fill_state_stream
    v = np.load(os.path.join(golden_dir, 'rnet_grad_vectors.npz'))
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    net = fill_state_stream(R_NETWORK(config, hp, 0), 5).to(dev).train()
    out = net(torch.from_numpy(v['x']).to(dev))
    close(out, torch.from_numpy(v['out']), rel=0, abs_=5e-5)
    loss = (torch.from_numpy(v['loss_w']).to(dev) * (out ** 2 + 0.3 * out)).sum()
    loss.backward()
    assert abs(float(loss) - float(v['loss'])) <= 1e-4 * abs(float(v['loss']))
    pd = dict(net.named_parameters())
    names = [str(n) for n in v['grad_names']]
    assert sorted(names) == sorted(pd)
    for n, want in zip(names, v['grad_norms']):
        g = pd[n].grad
        if want < 0:                                   # decoder_attention.12 / .13: built, never run
            assert g is None, n
        elif n.endswith('.0.bias') and not n.startswith('decoder.6'):
            assert float(g.norm()) <= 2e-3 * max(1.0, want) + 1e-3, n          # conv bias in front of a batch-statistics BN
        else:
            assert abs(float(g.norm()) - want) <= 3e-3 * want + 2e-5, (n, float(g.norm()), want)
    for k in v.files:
        if k.startswith('grad_') and k not in ('grad_names', 'grad_norms'):
            n = k[5:]
            if n.endswith('.0.bias') and not n.startswith('decoder.6'):
                continue
            g, want = pd[n].grad.cpu(), torch.from_numpy(v[k])
            got = g if g.numel() <= 20000 else g.flatten()[::37]
            close(got, want, rel=3e-3, abs_=1e-6)
    sd = net.state_dict()
    for k in ('initial_batchnorm.running_mean', 'initial_batchnorm.running_var', 'encoder.2.1.running_mean',
              'encoder.2.1.running_var', 'decoder.1.1.running_var'):
        close(sd[k], torch.from_numpy(v[f'after_{k}']), rel=1e-4, abs_=1e-6)


def test_rnetwork_train_step_against_oracle(dev):
    """network_functions.py:224-232 (dtype "real") + the optimizer: three DRS-Net training steps through
    train_batch_2_loss / TrainStep on the HIP path follow the CPU oracle's loss trajectory."""
    import sys
    from dcsnet.config import config, hparams
    from dcsnet.r_network import R_NETWORK
    from dcsnet.dp import TrainStep
    from oracle.rnet_oracle import R_NETWORK_Oracle
    from oracle.nf_oracle import drs_train_losses
    from oracle.seeded_state import fill_state_stream
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    net = fill_state_stream(R_NETWORK(config, hp, 0), 5).to(dev).train()
    ref = fill_state_stream(R_NETWORK_Oracle(dropout_conv=0.0, dropout_fc=0.0), 5).train()
    clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
    noisy = clean + noise
    argv = sys.argv
    sys.argv = ['train.py', 'drs', '0']
    try:
        ts = TrainStep(net)
        opt = torch.optim.Adam(ref.parameters(), lr=hp['lr'], eps=hp['optim_eps'], weight_decay=hp['optim_weight_decay'],
                               amsgrad=True)
        batch = (noise.to(dev), noisy.to(dev), clean.to(dev), [0, 1])
        got, want = [], []
        for _ in range(3):
            got.append(float(ts(batch)))
            opt.zero_grad()
            loss = drs_train_losses(ref, noise, noisy, clean)[2]
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 100.0)
            opt.step()
            want.append(float(loss.detach()))
    finally:
        sys.argv = argv
    for a, b in zip(got, want):
        assert abs(a - b) <= 2e-3 * abs(b) + 1e-3, (got, want)


def _conv_fp64(x, w_r, w_i, b_r, b_i, stride, pad, gy):
    """fp64 complex conv of channels-last x [B,H,W,C,2] with the two-real-layer bias convention of complexPyTorch
    (re: b_r - b_i, im: b_r + b_i) and its gradient with respect to x for the cotangent gy."""
    import torch.nn.functional as tF
    x64 = x.double().cpu().requires_grad_(True)
    xr, xi = x64[..., 0].permute(0, 3, 1, 2), x64[..., 1].permute(0, 3, 1, 2)
    wr, wi = w_r.double().cpu(), w_i.double().cpu()
    c = lambda a, w: tF.conv2d(a, w, None, stride, pad)
    yr = c(xr, wr) - c(xi, wi) + (b_r - b_i).double().cpu()[None, :, None, None]
    yi = c(xi, wr) + c(xr, wi) + (b_r + b_i).double().cpu()[None, :, None, None]
    y = torch.stack((yr, yi), -1).permute(0, 2, 3, 1, 4)
    wr.requires_grad_(True); wi.requires_grad_(True)
    y2 = torch.stack((c(xr, wr) - c(xi, wi), c(xi, wr) + c(xr, wi)), -1).permute(0, 2, 3, 1, 4)
    (y2 * gy.double().cpu()).sum().backward()
    return y.detach(), x64.grad, torch.cat((wr.grad.flatten(), wi.grad.flatten()))


@pytest.mark.parametrize('geom', [(4, 16, 32, 64, 128, 3, (2, 1)), (2, 48, 40, 16, 32, 5, (2, 2)), (2, 8, 32, 128, 128, 3, (1, 1)),
                                  (2, 64, 96, 1, 8, 7, (2, 2))])       # the last: enc0 (conv_enc0.hip, cconv_enc0b_kernel: 16x16x32 bf16 MFMAs)
def test_f32_emulation_on_the_bf16_mfma_is_at_least_as_accurate_as_the_native_mfma(dev, geom):
    """Precision mode 'bf16x6' (the default): every fp32 operand split exactly into three bf16 terms, six bf16 MFMAs per
    product group, fp32 accumulation (conv_mfma.hip, PR = 2; conv_wgrad_mfma.hip, cconv_wgrad_x6_kernel).  Criterion: distance
    to an fp64 evaluation, forward, data gradient and weight gradient — no larger than the native fp32 MFMA kernels' own
    distance (+25 % slack for the sample), and both at fp32 rounding level (1e-6)."""
    from dcsnet import ops
    B, H, W, Cin, Cout, k, st = geom
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, Cin, 2, generator=g).to(dev)
    w_r, w_i = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(dev), (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(dev)
    b_r, b_i = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    pad = (k // 2, k // 2)
    Ho, Wo = (H + 2 * pad[0] - k) // st[0] + 1, (W + 2 * pad[1] - k) // st[1] + 1
    gy = torch.randn(B, Ho, Wo, Cout, 2, generator=g).to(dev)
    y64, gx64, gw64 = _conv_fp64(x, w_r, w_i, b_r, b_i, st, pad, gy)
    default = ops.conv_precision()
    err = {}
    try:
        for mode in ('f32', 'bf16x6'):
            ops.set_conv_precision(mode)
            wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, False, (1, 1))
            y = ops.cconv2d(x, None, wp, bias, (k, k), st, pad, (1, 1))
            wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, pad, (1, 1))
            gx = ops.cconv2d_bwd_data(gy, wpb, (H, W, Cin), (k, k), st, pad, (1, 1), Cin)[0]
            gw = ops.cconv2d_bwd_weight(x, None, gy, (Cout, Cin, k, k), True, (k, k), st, pad, (1, 1), False)
            gw = torch.cat((gw[0].double().cpu().flatten(), gw[1].double().cpu().flatten()))
            err[mode] = (float((y.double().cpu() - y64).norm() / y64.norm()), float((gx.double().cpu() - gx64).norm() / gx64.norm()),
                         float((y.double().cpu() - y64).abs().max() / y64.abs().max()), float((gw - gw64).norm() / gw64.norm()))
    finally:
        ops.set_conv_precision(default)
    assert err['bf16x6'][0] <= 1.25 * err['f32'][0] and err['bf16x6'][1] <= 1.25 * err['f32'][1], err
    assert err['bf16x6'][3] <= 1.25 * err['f32'][3] and err['f32'][3] <= 1e-6, err            # weight gradient (cconv_wgrad_x6_kernel)
    assert err['bf16x6'][2] <= 1e-6 and err['f32'][2] <= 1e-6, err


def test_folded_cbn_epilogue_under_co_resident_bf16_mfma_workgroups(dev):
    """The enc5 forward at the inference bench shape [16,8,250,128] -> 128 with folded eval-mode CBN coefficients, in
    both fp32 modes: the epilogue's affine map used to come out without its q1 * im term in ~300 of 4 M outputs when
    the compiler paired its FMAs into v_pk_fma_f32 and other workgroups' bf16 MFMAs were in flight (conv_mfma.hip
    epilogue comment) — the two modes must agree to rounding everywhere."""
    from dcsnet import ops, functional as F
    g = torch.Generator().manual_seed(9)
    B, H, W, C, k, st = 16, 8, 250, 128, 3, (2, 1)
    x = torch.randn(B, H, W, C, 2, generator=g).to(dev)
    w_r, w_i = (torch.randn(C, C, k, k, generator=g) * 0.05).to(dev), (torch.randn(C, C, k, k, generator=g) * 0.05).to(dev)
    b_r, b_i = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    coef = torch.randn(C, 6, generator=g).to(dev)
    default = ops.conv_precision()
    out = {}
    try:
        for mode in ('f32', 'bf16x6'):
            ops.set_conv_precision(mode)
            wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, False, (1, 1))
            out[mode] = [ops.cconv2d(x, None, wp, bias, (k, k), st, (1, 1), (1, 1), F.ACT_RELU, coef=coef) for _ in range(3)]
    finally:
        ops.set_conv_precision(default)
    scale = float(out['f32'][0].abs().max())
    for a in out['bf16x6']:
        assert torch.equal(a, out['bf16x6'][0])                       # run-to-run identical
        assert float((a - out['f32'][0]).abs().max()) <= 2e-5 * scale


def test_double_bound_mask_application_in_one_kernel(dev):
    """dcs_bound2_mask_apply_fwd / _bwd (the network's own bound_cRM, c_network.py:225, fused with the step function's second
    bound + multiply + subtract, network_functions.py:240-243) against the two-kernel chain it replaces — itself pinned to
    the reference's vectors (test_mask_math_against_reference_vectors): same arithmetic per element, so bit-identical forward,
    and identical cotangents of the raw output incl. the singular points of atan2."""
    from dcsnet import functional as F
    d0 = rand_c((3, 256, 40), 5, 1.5)
    d0.view(-1)[:4] = torch.tensor([0 + 0j, -1e-6 + 0j, 1e-7 - 1e-7j, -2.0 + 0j])
    Y = rand_c((3, 256, 40), 6, 0.8).to(dev)
    gM = torch.view_as_real(rand_c((3, 256, 40), 7)).to(dev)
    gNS = torch.view_as_real(rand_c((2, 3, 256, 40), 8)).to(dev)
    outs = []
    for fused in (False, True):
        d = d0.clone().to(dev).requires_grad_(True)
        if fused:
            M, NS = F.bound2_mask_apply_pair_complex(Y, d, 10e-7)
        else:
            M, NS = F.bound_mask_apply_pair_complex(Y, F.bound_crm_complex(d, 10e-7), 10e-7)
        ((torch.view_as_real(M) * gM).sum() + (torch.view_as_real(NS) * gNS).sum()).backward()
        outs.append((M.detach(), NS.detach(), d.grad.detach()))
    for a, b, what in zip(outs[0], outs[1], ('M', 'NS', 'g_D')):
        assert torch.equal(torch.view_as_real(a), torch.view_as_real(b)), what
    # the last stage's dropout handed to the same kernels: identical to running dcs_dropout_fwd first (same mask convention)
    res = []
    for inside in (False, True):
        d = d0.clone().to(dev).requires_grad_(True)
        if inside:
            M, NS = F.bound2_mask_apply_pair_complex(Y, d, 10e-7, (0.3, 4242))
        else:
            dd = torch.view_as_complex(F.dropout(torch.view_as_real(d), 0.3, 4242))
            M, NS = F.bound2_mask_apply_pair_complex(Y, dd, 10e-7)
        ((torch.view_as_real(M) * gM).sum() + (torch.view_as_real(NS) * gNS).sum()).backward()
        res.append((M.detach(), NS.detach(), d.grad.detach()))
    for a, b, what in zip(res[0], res[1], ('M', 'NS', 'g_D')):
        assert torch.equal(torch.view_as_real(a), torch.view_as_real(b)), ('dropout inside', what)
    assert not torch.equal(torch.view_as_real(res[0][0]), torch.view_as_real(outs[0][0]))      # (the mask did something)
    M, N, S = F.bound2_mask_apply_complex(Y, d0.to(dev), 10e-7)
    assert torch.equal(torch.view_as_real(M), torch.view_as_real(outs[0][0]))
    assert torch.equal(torch.view_as_real(N), torch.view_as_real(outs[0][1][0]))
    assert torch.equal(torch.view_as_real(S), torch.view_as_real(outs[0][1][1]))


@pytest.mark.parametrize('B,T,drop', [(3, 40, (0.0, 0)), (2, 72, (0.3, 4242)), (1, 8, (0.0, 0))])
def test_mask_application_fused_with_the_polar_round_trip(dev, B, T, drop):
    """dcs_bound2_apply_polar_frames_fwd / _bwd (network_functions.py:240-247 as one kernel each way inside
    F.bound2_apply_polar_wave_pair) against the chain it replaces, F.bound2_mask_apply_pair_complex -> F.polar_wave — itself
    pinned to the reference's vectors.  Same arithmetic per element in the same order: bit-identical waveforms, mask and
    cotangent of the raw output, incl. the singular points of atan2, ragged tiles (T not a multiple of 32) and the dropout mask."""
    from dcsnet import functional as F, ops
    n_fft, hop = 512, 128
    d0 = rand_c((B, 256, T), 5, 1.5)
    d0.view(-1)[:4] = torch.tensor([0 + 0j, -1e-6 + 0j, 1e-7 - 1e-7j, -2.0 + 0j])
    Y = rand_c((B, 256, T), 6, 0.8).to(dev)
    window = torch.hann_window(n_fft).to(dev)
    inv_env = ops.istft_envelope(window, T, hop)
    L = hop * (T - 1)
    gw = torch.randn(2 * B, L, generator=torch.Generator().manual_seed(9)).to(dev)
    gM = torch.view_as_real(rand_c((B, 256, T), 7)).to(dev)
    for with_mask in (False, True):
        outs = []
        for fused in (False, True):
            d = d0.clone().to(dev).requires_grad_(True)
            if fused:
                M, wave = F.bound2_apply_polar_wave_pair(Y, d, window, inv_env, n_fft, hop, 1.0, 10e-7, drop, want_mask=with_mask)
            else:
                M, NS = F.bound2_mask_apply_pair_complex(Y, d, 10e-7, drop)
                wave = F.polar_wave(NS.reshape(2 * B, 256, T), window, inv_env, n_fft, hop, 1.0, 10e-7)
            loss = (wave * gw).sum()
            if with_mask:
                loss = loss + (torch.view_as_real(M) * gM).sum()
            loss.backward()
            outs.append((wave.detach(), d.grad.detach(), M.detach() if with_mask else None))
        assert tuple(outs[1][0].shape) == (2 * B, L)
        assert torch.equal(outs[0][0], outs[1][0]), 'waveforms'
        assert torch.equal(torch.view_as_real(outs[0][1]), torch.view_as_real(outs[1][1])), 'g_D'
        if with_mask:
            assert torch.equal(torch.view_as_real(outs[0][2]), torch.view_as_real(outs[1][2])), 'mask'
        else:
            assert M is None


@pytest.mark.parametrize('C,H,W', [(8, 24, 20), (64, 8, 16), (128, 4, 32)])
def test_cbn_apply_that_pools_for_the_channel_attention(dev, C, H, W):
    """dcs_cbn_fwd_slabs_pool + dcs_channel_attention_fc_fwd (a decoder stage's CBN + CLReLU + channel attention in three
    launches) against dcs_cbn_fwd_slabs + dcs_channel_attention_fwd: same activation bit for bit, same pooled means / hidden /
    ca (the sums are taken in the same chunk geometry and order)."""
    from dcsnet import ops
    B, Ch = 3, max(C // 16, 1)
    g = torch.Generator().manual_seed(C)
    x1 = torch.randn(B, H // 2, W, 2 * C if False else C, 2, generator=g).to(dev)
    w_r, w_i = (torch.randn(C, C, 3, 3, generator=g) * 0.1).to(dev), (torch.randn(C, C, 3, 3, generator=g) * 0.1).to(dev)
    b_r, b_i = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, True, (2, 1))
    y, stat = ops.cconv2d_stats(x1, None, wp, bias, (3, 3), (1, 1), (1, 1), (2, 1))
    assert stat is not None
    bnw = (torch.randn(C, 3, generator=g) * 0.3 + torch.tensor([1.2, 1.1, 0.1])).to(dev)
    bnb = torch.randn(C, 2, generator=g).to(dev)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.2).to(dev)
    w1, _ = ops.pack_conv_weight(rnd(Ch, C, 1, 1), rnd(Ch, C, 1, 1))
    w2, _ = ops.pack_conv_weight(rnd(C, Ch, 1, 1), rnd(C, Ch, 1, 1))
    rm0, rc0 = torch.zeros(C, 2, device=dev), torch.ones(C, 3, device=dev)
    a0, stats0, coef0 = ops.cbn(y, bnw, bnb, rm0, rc0, 1e-5, 0.1, True, ops.ACT_LRELU, stat=stat)
    ca0, pooled0, hidden0 = ops.channel_attention(a0, w1, w2)
    rm1, rc1 = torch.zeros(C, 2, device=dev), torch.ones(C, 3, device=dev)
    a1, stats1, coef1, ca1, pooled1, hidden1 = ops.cbn_channel_attention(y, bnw, bnb, rm1, rc1, 1e-5, 0.1, ops.ACT_LRELU, stat,
                                                                         w1, w2)
    for p_, q_, n in ((a1, a0, 'a'), (stats1, stats0, 'stats'), (coef1, coef0, 'coef'), (rm1, rm0, 'running mean'),
                      (rc1, rc0, 'running covar'), (pooled1, pooled0, 'pooled'), (hidden1, hidden0, 'hidden'), (ca1, ca0, 'ca')):
        assert torch.equal(p_, q_), n
