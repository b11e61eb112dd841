"""Functional layer between the nn.Module surface and the C-ABI wrappers (ops.py).

Everything here takes / returns channels-last float32 views ``[B,H,W,C,2]`` unless it says
"complex" in its name.  Packed weights are cached per parameter version, so inference packs
once and training re-packs after each optimizer step.
"""
import weakref

import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID, to_nhwc, from_nhwc  # noqa: F401
from ._lib import DcsHipError

_pack_cache = {}


def _ver(t):
    return (0, 0) if t is None else (id(t), t._version)


def packed_weight(w_r, w_i, b_r, b_i, transposed):
    """Packed (wp, bias) for a weight pair; cached per tensor OBJECT and version (the weakrefs
    guard against a recycled id()).  1x1 / Linear weights may be passed 2-D."""
    key = (_ver(w_r), _ver(w_i), _ver(b_r), _ver(b_i), bool(transposed))
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is w_r and hit[1]() is w_i:
        return hit[2]
    if len(_pack_cache) > 512:
        _pack_cache.clear()
    d = lambda t: None if t is None else t.detach()
    wr, wi = d(w_r), d(w_i)
    if wr.dim() == 2:
        wr, wi = wr.view(*wr.shape, 1, 1), wi.view(*wi.shape, 1, 1)
    packed = ops.pack_conv_weight(wr, wi, d(b_r), d(b_i), transposed)
    _pack_cache[key] = (weakref.ref(w_r), weakref.ref(w_i), packed)
    return packed


def cconv2d(x1, x2, w_r, w_i, b_r, b_i, transposed, ksize, stride, pad, up=(1, 1), act=ACT_NONE):
    wp, bias = packed_weight(w_r, w_i, b_r, b_i, transposed)
    return ops.cconv2d(x1, x2, wp, bias, ksize, stride, pad, up, act)


def cbn(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act=ACT_NONE,
        drop_p=0.0, seed=0):
    d = lambda t: None if t is None else t.detach()
    y, _, _ = ops.cbn(x, d(weight), d(bias), running_mean, running_covar, eps, momentum, use_batch_stats,
                      act, drop_p, seed)
    return y


def channel_attention(x, fc0_r, fc0_i, fc2_r, fc2_i):
    """ComplexChannelAttention (c_network.py:53-69).  fc*_r/_i: the 1x1 conv weights."""
    w1, _ = packed_weight(fc0_r, fc0_i, None, None, False)     # [1, C, Ch, 2]
    w2, _ = packed_weight(fc2_r, fc2_i, None, None, False)     # [1, Ch, C, 2]
    ca, _, _ = ops.channel_attention(x, w1, w2)
    return ca


def spatial_attention(x, ca, conv_r, conv_i, ksize):
    """ComplexSpatialAttention (c_network.py:71-84) of z = ca * x, without materialising z."""
    pooled = ops.spatial_pool(x, ca)
    wp, bias = packed_weight(conv_r, conv_i, None, None, False)
    k = (ksize, ksize)
    return ops.cconv2d(pooled, None, wp, bias, k, (1, 1), (ksize // 2, ksize // 2), (1, 1), ACT_SIGMOID)


def attention_apply(x, ca, sa, drop_p=0.0, seed=0):
    return ops.attention_apply(x, ca, sa, drop_p, seed)


def dropout(x, drop_p, seed):
    return ops.dropout(x, drop_p, seed)


# ---- complex-tensor conveniences for the drop-in layer surface ---------------------------------

def _as_real(z):
    if z.dtype != torch.complex64:
        raise DcsHipError(f'expected complex64, got {z.dtype}')
    return torch.view_as_real(z.contiguous() if not _dense(z) else z)


def _dense(z):
    # dense in SOME permutation (e.g. channels_last): element-wise kernels may run in place order
    return z.is_contiguous() or (z.dim() == 4 and z.is_contiguous(memory_format=torch.channels_last))


def _elementwise(z, act):
    if z.dim() == 4 and z.is_contiguous(memory_format=torch.channels_last) and not z.is_contiguous():
        x = to_nhwc(z)
        return from_nhwc(ops.complex_act(x, act))
    x = torch.view_as_real(z.contiguous())
    return torch.view_as_complex(ops.complex_act(x, act))


def complex_relu(z):
    return _elementwise(z, ACT_RELU)


def complex_lrelu(z):
    return _elementwise(z, ACT_LRELU)


def complex_sigmoid(z):
    return _elementwise(z, ACT_SIGMOID)


def complex_upsample(z, scale_factor):
    sf = (scale_factor, scale_factor) if isinstance(scale_factor, (int, float)) else tuple(scale_factor)
    up = (int(sf[0]), int(sf[1]))
    if up != tuple(sf) or min(up) < 1:
        raise DcsHipError(f'complex_upsample: integer scale factors only, got {scale_factor}')
    return from_nhwc(ops.complex_upsample(to_nhwc(z), up))


def complex_linear(z, w_r, w_i, b_r, b_i):
    """ComplexLinear (complexPyTorch apply_complex over two nn.Linear) as a 1x1 complex conv."""
    in_f = z.shape[-1]
    lead = z.shape[:-1]
    x = torch.view_as_real(z.reshape(-1, in_f).contiguous())          # [N, in, 2]
    N = x.shape[0]
    W = 16 if N % 16 == 0 else 1
    x = x.view(1, N // W, W, in_f, 2)
    y = cconv2d(x, None, w_r, w_i, b_r, b_i, False, (1, 1), (1, 1), (0, 0))
    return torch.view_as_complex(y.view(N, -1, 2)).view(*lead, -1)


def complex_lstm(z, real_lstm, imag_lstm, save=False):
    """ComplexLSTM.forward (c_network.py:33-47) for two bidirectional batch_first nn.LSTM
    parameter containers.  Per layer: one input-projection GEMM for all time steps (rocBLAS via
    torch) + one persistent HIP launch for the recurrence of all 4 passes x 2 directions."""
    if not (real_lstm.bidirectional and real_lstm.batch_first and real_lstm.hidden_size == 64):
        raise DcsHipError('complex_lstm: the HIP path implements the reference geometry '
                          '(bidirectional, batch_first, hidden 64: c_network.py:118-123)')
    B, S, I = z.shape
    sets = (real_lstm, imag_lstm)
    # rows 0..B-1: real parts, rows B..2B-1: imaginary parts
    x = torch.view_as_real(z).permute(3, 0, 1, 2).reshape(2 * B, S, I)
    inp = None
    for layer in range(real_lstm.num_layers):
        names = [f'_l{layer}', f'_l{layer}_reverse']
        w_ih = [torch.cat([getattr(m, 'weight_ih' + n) for n in names]).detach() for m in sets]          # [8H, in]
        bias = [torch.cat([getattr(m, 'bias_ih' + n) + getattr(m, 'bias_hh' + n) for n in names]).detach()
                for m in sets]
        w_hh = torch.stack([torch.stack([getattr(m, 'weight_hh' + n) for n in names]) for m in sets]).detach()
        w_hh = w_hh.contiguous()
        G = w_ih[0].shape[0]                                    # 2 dirs * 4H
        if layer == 0:
            gx = torch.addmm(torch.cat(bias), x.reshape(2 * B * S, I), torch.cat(w_ih).t())    # (n, t, set, dir, 4H)
            strides = (G, S * 2 * G, 2 * G)
        else:
            gx = torch.baddbmm(torch.stack(bias).unsqueeze(1), inp.reshape(2, 2 * B * S, -1),
                               torch.stack(w_ih).transpose(1, 2))                               # (set, n, t, dir, 4H)
            strides = (2 * B * S * G, S * G, G)
        out, _, _ = ops.lstm_layer(gx.contiguous(), w_hh, 2, 2 * B, S, strides, save)
        inp = out.view(2, 2 * B, S, -1)
    rr, ir = inp[0, :B], inp[0, B:]        # real_lstm(re), real_lstm(im)
    ri, ii = inp[1, :B], inp[1, B:]        # imag_lstm(re), imag_lstm(im)
    return torch.complex(rr - ii, ir + ri)


def bound_crm_complex(M, eps=10e-7):
    x = torch.view_as_real(M.contiguous())
    return torch.view_as_complex(ops.bound_crm(x, eps))


def bound_mask_apply_complex(Y, M_in, eps=10e-7):
    y = torch.view_as_real(Y.contiguous())
    m = torch.view_as_real(M_in.contiguous())
    M, N, S = ops.bound_mask_apply(y, m, eps)
    return torch.view_as_complex(M), torch.view_as_complex(N), torch.view_as_complex(S)


def crm_complex(S, Y, eps=1e-8):
    return torch.view_as_complex(ops.crm(torch.view_as_real(S.contiguous()), torch.view_as_real(Y.contiguous()), eps))
