"""GPU parity of the opt-in producer / consumer conv kernel (csrc/conv_ring.hip, DCS_CONV_RING=1) — the complex convolutions of
c_network.py:107-112 (encoder) and :135-147 (decoder, behind cat + nearest upsample) and their data gradients.

The kernel runs the same bf16 MFMAs on the same operands as cconv_mfma_kernel and sums them per output element in (chunk, tap,
k-group) order; the classic plans differ in chunk depth and in how they split K over waves / workgroups, so the two outputs
agree to fp32 accumulation order (2e-5 of the tensor's max-abs, the conv tolerance of test_hip_parity.py) — and are equal bit
for bit where the plans coincide (tools/ring_check.py shows 0.0 on the inference shapes).
One geometry is also checked against the CPU oracle directly, and the CBN statistics epilogue against the classic epilogue's
moments.  Both switches are read per call, so one process runs both paths."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpt_oracle as cpt          # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from dcsnet import _lib
    _lib.load()
    return torch.device('cuda:0')


class _ring:
    """DCS_CONV_RING / DCS_RING_MIN_WG for the calls inside the block (MIN_WG=1: also launches that would not fill the chip)."""
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ('DCS_CONV_RING', 'DCS_RING_MIN_WG')}
        os.environ['DCS_CONV_RING'] = '1' if self.on else '0'
        os.environ['DCS_RING_MIN_WG'] = '1'

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _close(a, b, rel, what):
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert err <= rel * scale, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


# (B, H, W, C1, C2, Cout, k, stride, up, transposed): geometries whose forward and / or data gradient fit the 128 x 64 tile
GEOMS = [
    (8, 4, 32, 128, 128, 128, 3, (1, 1), (2, 1), True),     # dec1: two folded classes of 2x3 taps; dgrad: 4x3 taps, stride (2, 1), cat split
    (4, 8, 32, 128, 128, 64, 3, (1, 1), (2, 1), True),      # dec2
    (3, 16, 40, 64, 64, 32, 3, (1, 1), (2, 1), True),       # dec3, ragged: W = 40 is not a multiple of the 32-pixel tile row
    (2, 6, 250, 64, 64, 32, 3, (1, 1), (2, 1), True),       # ... the inference width, H = 6: tiles outside the class extent
    (4, 16, 32, 64, 0, 128, 3, (2, 1), (1, 1), False),      # enc4: forward falls back (9 taps, 8-channel chunks), dgrad = two strided classes
    (4, 8, 32, 128, 0, 128, 3, (2, 1), (1, 1), False),      # enc5
    (2, 32, 32, 32, 0, 64, 5, (2, 1), (1, 1), False),       # enc3: dgrad classes of 3x5 and 2x5 taps, 64 columns
    (1, 4, 32, 128, 128, 128, 3, (1, 1), (2, 1), True),     # B = 1
]


def _layer(geom, dev, seed=3):
    from dcsnet import ops
    B, H, W, C1, C2, Cout, k, st, up, tr = geom
    g = torch.Generator().manual_seed(seed)
    Cin = C1 + C2
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    w_r, w_i = (torch.randn(wshape, generator=g) * 0.05).to(dev), (torch.randn(wshape, generator=g) * 0.05).to(dev)
    b_r, b_i = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    x1 = (torch.randn(B, H, W, C1, 2, generator=g) + 0.1).to(dev)
    x2 = torch.randn(B, H, W, C2, 2, generator=g).to(dev) if C2 else None
    wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, tr, up)
    return x1, x2, wp, bias, (w_r, w_i, b_r, b_i)


@pytest.mark.parametrize('geom', GEOMS, ids=[f'g{i}' for i in range(len(GEOMS))])
def test_ring_kernel_equals_the_classic_kernel(dev, geom):
    from dcsnet import ops
    B, H, W, C1, C2, Cout, k, st, up, tr = geom
    x1, x2, wp, bias, _ = _layer(geom, dev)
    pad = (k // 2, k // 2)
    out = {}
    for on in (False, True):
        with _ring(on):
            y = ops.cconv2d(x1, x2, wp, bias, (k, k), st, pad, up)
            ys, stat = ops.cconv2d_stats(x1, x2, wp, bias, (k, k), st, pad, up)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(dev)
            wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, pad, up)
            gx = ops.cconv2d_bwd_data(gy, wpb, (H, W, C1 + C2), (k, k), st, pad, up, C1)
            torch.cuda.synchronize()
            assert torch.equal(y, ys), 'the statistics epilogue must not change the output'
            # the moments summed over the launch's rows: {S_r, S_i, S_rr, S_ii, S_ri} per channel
            mom = stat[0][:, :, :stat[1]].double().sum(dim=2) if stat is not None else None
            out[on] = (y, gx, mom)
    y0, gx0, m0 = out[False]
    y1, gx1, m1 = out[True]
    _close(y1, y0, 2e-5, 'forward')
    for a, b_ in zip(gx1, gx0):
        if a is not None:
            _close(a, b_, 2e-5, 'data gradient')
    if m0 is not None and m1 is not None:
        _close(m1, m0, 2e-5, 'CBN moments of the raw output')


def test_ring_kernel_is_bitwise_repeatable(dev):
    """No atomics, a fixed accumulation order per element ((chunk, tap, k-group), the six emulation terms smallest first), a
    fixed order of the statistics' partial sums: two launches give the same bits — forward, statistics rows, data gradient."""
    from dcsnet import ops
    geom = (32, 16, 32, 64, 64, 32, 3, (1, 1), (2, 1), True)           # dec3 at the train batch
    B, H, W, C1, C2, Cout, k, st, up, tr = geom
    x1, x2, wp, bias, _ = _layer(geom, dev, seed=9)
    wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, (1, 1), up)
    runs = []
    with _ring(True):
        for _ in range(2):
            y, stat = ops.cconv2d_stats(x1, x2, wp, bias, (k, k), st, (1, 1), up)
            gx = ops.cconv2d_bwd_data(y, wpb, (H, W, C1 + C2), (k, k), st, (1, 1), up, C1)
            torch.cuda.synchronize()
            runs.append((y.clone(), stat[0][:, :, :stat[1]].clone(), [g_.clone() for g_ in gx if g_ is not None]))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    for a, b_ in zip(runs[0][2], runs[1][2]):
        assert torch.equal(a, b_)


def test_ring_kernel_against_the_oracle(dev):
    """dec1's ComplexConvTranspose2d behind cat + nearest upsample (c_network.py:135-141, :214-217) against the CPU oracle."""
    from dcsnet import functional as F, ops
    torch.manual_seed(21)
    c1 = c2 = 128
    m = cpt.ComplexConvTranspose2d(c1 + c2, 128, 3, 1, 1)
    g = torch.Generator().manual_seed(2)
    d = torch.complex(torch.randn(2, c1, 4, 32, generator=g), torch.randn(2, c1, 4, 32, generator=g)) * 0.5
    s = torch.complex(torch.randn(2, c2, 4, 32, generator=g), torch.randn(2, c2, 4, 32, generator=g)) * 0.5
    with torch.no_grad():
        want = m(cpt.complex_upsample(torch.cat((d, s), dim=1), scale_factor=(2, 1), mode='nearest'))
    p = lambda t: t.detach().to(dev)
    with _ring(True):
        y = F.cconv2d(ops.to_nhwc(d.to(dev)), ops.to_nhwc(s.to(dev)), p(m.conv_tran_r.weight), p(m.conv_tran_i.weight),
                      p(m.conv_tran_r.bias), p(m.conv_tran_i.bias), True, (3, 3), (1, 1), (1, 1), (2, 1))
        torch.cuda.synchronize()
    _close(ops.from_nhwc(y).cpu().abs(), want.abs(), 1.0, 'sanity')      # same shape / scale
    got = ops.from_nhwc(y).cpu()
    err = float((got - want).abs().max())
    assert err <= 2e-5 * float(want.abs().max()), err


def test_ring_kernel_bf16_storage(dev):
    """bf16 activation storage (the _h entry points): the ring kernel's output equals the classic kernel's to one bf16 rounding
    of a differently-ordered fp32 sum."""
    from dcsnet import ops
    default = ops.conv_precision()
    ops.set_conv_precision('bf16')
    try:
        geom = (8, 4, 32, 128, 128, 128, 3, (1, 1), (2, 1), True)
        B, H, W, C1, C2, Cout, k, st, up, tr = geom
        x1, x2, wp, bias, _ = _layer(geom, dev, seed=4)
        x1, x2 = x1.to(torch.bfloat16), x2.to(torch.bfloat16)
        res = {}
        for on in (False, True):
            with _ring(on):
                y = ops.cconv2d(x1, x2, wp, bias, (k, k), st, (1, 1), up)
                gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(dev).to(torch.bfloat16)
                wpb = ops.pack_conv_weight_bwd(wp, (k, k), st, (1, 1), up)
                gx = ops.cconv2d_bwd_data(gy, wpb, (H, W, C1 + C2), (k, k), st, (1, 1), up, C1)
                torch.cuda.synchronize()
                res[on] = (y.float(), [g_.float() for g_ in gx if g_ is not None])
        _close(res[True][0], res[False][0], 1e-2, 'forward (bf16 storage)')
        for a, b_ in zip(res[True][1], res[False][1]):
            _close(a, b_, 1e-2, 'data gradient (bf16 storage)')
    finally:
        ops.set_conv_precision(default)
