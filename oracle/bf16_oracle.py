"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path.

The CPU oracle of `cnet_oracle.py` with the rounding points of the build's bf16 activation storage
(BASELINE.json configs[4]; `C_NETWORK.set_activation_dtype('bf16')`, DESIGN.md §3 "bf16 activation storage").  The
reference trains at precision 32 (config.py:70, train.py:144) — bf16 storage is the build's extension — so this file
restates THE BUILD'S numerical contract on top of the reference's arithmetic, not a reference behaviour:

  * stored in bf16 (round to nearest even, once, on store): the initial CBN's output, every encoder / decoder conv
    output, every CBN + activation output, every attention block's output, the latent after LSTM + fc + dropout — and
    the cotangent of each of those (the straight-through rounding node `_store`: value rounded forward, cotangent
    rounded backward);
  * bf16 MFMA operands: the weights of the encoder convs, the decoder conv-transposes (not the last one, a VALU kernel
    with fp32 weights) and the latent fc are rounded to bf16 where they enter the product (fp32 master weights: the
    gradient passes straight through to them), the fc's fp32 input is rounded as an operand;
  * fp32 everywhere else: network input, mask, CBN statistics / coefficients, attention maps and their 1x1 / 7x7 convs,
    the LSTM, all accumulation, parameter gradients.
The decoder's weights are rounded AFTER the nearest upsample has been folded into them (per output-parity class: two taps
that read the same source pixel are one weight, their fp32 sum), as the build's pack does (`_cconvT_folded`).
Not modelled (second-order, inside the test's bounds): the build takes CBN statistics from the conv's fp32 accumulators
(here: from the stored bf16 values).
Checked against the HIP path in tests/test_hip_bf16.py::test_bf16_storage_network_against_the_oracle."""
import torch
import torch.nn.functional as TF

from . import cpt_oracle as cpt
from . import nf_oracle as nf
from .cnet_oracle import C_NETWORK_Oracle, UPSAMPLE


def _bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def _round_complex(z):
    return torch.complex(_bf16(z.real), _bf16(z.imag)) if z.is_complex() else _bf16(z)


# What-if switch of the contract (tests/test_oracle.py::test_bf16_contract_gradient_cost_with_fp32_cotangents):
# cotangents of maps with at most this many pixels (H * W of the stored tensor) stay fp32 on the way back; 0 = the build's
# contract (every stored tensor's cotangent is rounded to bf16), a huge value = no cotangent is rounded.
COTANGENT_FP32_MAX_PIXELS = 0


class _Store(torch.autograd.Function):
    """A tensor stored in bf16 in HBM: its value is rounded on the way forward, its cotangent on the way back."""

    @staticmethod
    def forward(ctx, z):
        return _round_complex(z)

    @staticmethod
    def backward(ctx, g):
        if g.dim() >= 2 and g.shape[-2] * g.shape[-1] <= COTANGENT_FP32_MAX_PIXELS:
            return g
        return _round_complex(g)


class _Operand(torch.autograd.Function):
    """An fp32 value rounded to bf16 where it enters an MFMA product: gradient passes through unchanged."""

    @staticmethod
    def forward(ctx, w):
        return _round_complex(w)

    @staticmethod
    def backward(ctx, g):
        return g


_store, _operand = _Store.apply, _Operand.apply


def _apply_complex(fr, fi, z):
    return torch.complex(fr(z.real) - fi(z.imag), fr(z.imag) + fi(z.real))


def _cconv(m, z):                                   # cpt.ComplexConv2d with bf16-operand weights
    f = lambda c: (lambda t: TF.conv2d(t, _operand(c.weight), c.bias, c.stride, c.padding))
    return _apply_complex(f(m.conv_r), f(m.conv_i), z)


def _cconvT(m, z):                                  # cpt.ComplexConvTranspose2d with bf16-operand weights
    f = lambda c: (lambda t: TF.conv_transpose2d(t, _operand(c.weight), c.bias, c.stride, c.padding))
    return _apply_complex(f(m.conv_tran_r), f(m.conv_tran_i), z)


def _fold_axis(w, axis, parity):
    """Fold of a 3-tap correlation kernel over an axis that was nearest-upsampled by 2: on output positions of the given
    parity two of the three taps read the SAME source pixel, so only their sum matters — the build packs that sum as one
    weight (DESIGN.md §2 "folded sub-kernels") and rounds the SUM to bf16.  Returned as a 3-tap kernel on the upsampled
    grid with the sum on one of the two taps and zero on the other (same result, same rounding)."""
    w0, w1, w2 = w.unbind(axis)
    z = torch.zeros_like(w0)
    taps = (w0, w1 + w2, z) if parity == 0 else (z, w0 + w1, w2)
    return torch.stack(taps, dim=axis)


def _cconvT_folded(m, z, up):
    """cpt.ComplexConvTranspose2d (stride 1, k 3, p 1 = correlation with the flipped, in/out-swapped kernel) over an input that
    was nearest-upsampled by `up`, with the weights folded per output-parity class BEFORE they are rounded to bf16 operands."""
    def real_layer(c):
        wc = c.weight.flip(2, 3).transpose(0, 1)                       # correlation kernel [Cout, Cin, 3, 3]
        def f(t):
            H, W = t.shape[-2:]
            out = None
            for py in ((0, 1) if up[0] == 2 else (None,)):
                for px in ((0, 1) if up[1] == 2 else (None,)):
                    k = wc
                    if py is not None:
                        k = _fold_axis(k, 2, py)
                    if px is not None:
                        k = _fold_axis(k, 3, px)
                    y = TF.conv2d(t, _operand(k), c.bias, 1, 1)
                    mask = torch.ones(H, W, dtype=t.dtype)
                    if py is not None:
                        mask = mask * ((torch.arange(H) % 2) == py).to(t.dtype)[:, None]
                    if px is not None:
                        mask = mask * ((torch.arange(W) % 2) == px).to(t.dtype)[None, :]
                    out = y * mask if out is None else out + y * mask
            return out
        return f
    return _apply_complex(real_layer(m.conv_tran_r), real_layer(m.conv_tran_i), z)


def _clinear(m, z):
    f = lambda c: (lambda t: TF.linear(t, _operand(c.weight), c.bias))
    return _apply_complex(f(m.fc_r), f(m.fc_i), _operand(z))


class C_NETWORK_Bf16Oracle(C_NETWORK_Oracle):
    """Same parameters, state_dict and wiring as C_NETWORK_Oracle (c_network.py:88-226); forward with the rounding points
    listed in the module docstring."""

    def encode(self, x):
        feats = [_store(self.initial_batchnorm(x.view(x.shape[0], -1, x.shape[1], x.shape[2])))]
        for blk in self.encoder:
            c = _store(_cconv(blk[0], feats[-1]))
            feats.append(_store(self._drop(self.dropout_conv, blk[2](blk[1](c)))))
        return feats

    def latent(self, e):
        seq = torch.flatten(e, 2, 3).permute(0, 2, 1)
        z = _store(self._drop(self.dropout_fc, _clinear(self.fc, self.lstm(seq))))
        return z.permute(0, 2, 1).reshape(e.shape)

    def decode(self, d, feats):
        L = self.hp['no_of_layers']
        for i in range(L):
            skip = feats[L - i]
            skip = self.skip_attention[2 * i](skip) * skip
            skip = _store(self.skip_attention[2 * i + 1](skip) * skip)
            d = cpt.complex_upsample(torch.cat((d, skip), dim=1), scale_factor=UPSAMPLE[i], mode='nearest')
            if i == L - 1:                      # last stage: VALU kernel, fp32 weights, fp32 result (the raw mask)
                d = self.decoder[i](d)
            else:
                stage = self.decoder[i]
                y = _store(_cconvT_folded(stage[0], d, UPSAMPLE[i]))
                a = _store(stage[2](stage[1](y)))
                a = a * self.decoder_attention[2 * i](a)
                d = _store(a * self.decoder_attention[2 * i + 1](a))
            d = self._drop(self.dropout_conv, d)
        return d
