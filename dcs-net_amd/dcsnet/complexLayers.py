"""HIP-backed layers with the surface of ``complexPyTorch.complexLayers`` (0.3) — the operator
API the reference's hot path is written against (c_network.py:5, config.py:5).

Same class names, constructor arguments, sub-module / parameter / buffer names (``conv_r``,
``conv_i``, ``conv_tran_r``, ``conv_tran_i``, ``fc_r``, ``fc_i``, ``weight[C,3]``,
``bias[C,2]``, ``running_mean`` (complex64), ``running_covar[C,3]``, ``num_batches_tracked``)
so the reference's checkpoints load and its ``weights_init`` (c_network.py:174-184) finds the
inner ``nn.Conv2d`` / ``nn.ConvTranspose2d`` / ``nn.Linear`` modules.  The arithmetic runs in
libdcsnet_hip.so; tensors cross the module boundary as complex64 ``[B,C,H,W]`` in
channels_last memory (any input layout is accepted and converted once).
"""
import weakref

import torch
from torch.nn import Module, Parameter, Conv2d, ConvTranspose2d, Linear

from . import functional as F
from ._lib import DcsHipError

SQRT2 = 1.4142135623730951


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class ComplexReLU(Module):
    def forward(self, input):
        return F.complex_relu(input)


class ComplexConv2d(Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True):
        super().__init__()
        if _pair(dilation) != (1, 1) or groups != 1:
            raise DcsHipError('ComplexConv2d: the HIP path implements dilation=1, groups=1 (all the reference uses)')
        self.conv_r = Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.conv_i = Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)

    def forward(self, input):
        return F.from_nhwc(F.cconv2d(F.to_nhwc(input), None, self.conv_r.weight, self.conv_i.weight,
                                     self.conv_r.bias, self.conv_i.bias, False,
                                     self.kernel_size, self.stride, self.padding))


class ComplexConvTranspose2d(Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 output_padding=0, groups=1, bias=True, dilation=1, padding_mode='zeros'):
        super().__init__()
        if _pair(stride) != (1, 1) or _pair(output_padding) != (0, 0) or groups != 1 or _pair(dilation) != (1, 1) \
                or padding_mode != 'zeros':
            raise DcsHipError('ComplexConvTranspose2d: the HIP path implements stride 1, no output padding, '
                              'dilation 1, groups 1 (config.py:84,92-100)')
        self.conv_tran_r = ConvTranspose2d(in_channels, out_channels, kernel_size, stride, padding,
                                           output_padding, groups, bias, dilation, padding_mode)
        self.conv_tran_i = ConvTranspose2d(in_channels, out_channels, kernel_size, stride, padding,
                                           output_padding, groups, bias, dilation, padding_mode)
        self.kernel_size, self.padding = _pair(kernel_size), _pair(padding)
        # stride-1 transposed conv == correlation with the flipped kernel and padding k-1-p
        self.corr_padding = (self.kernel_size[0] - 1 - self.padding[0], self.kernel_size[1] - 1 - self.padding[1])
        if min(self.corr_padding) < 0:
            raise DcsHipError('ComplexConvTranspose2d: padding > kernel_size - 1 is not supported')

    def forward(self, input):
        return F.from_nhwc(F.cconv2d(F.to_nhwc(input), None, self.conv_tran_r.weight, self.conv_tran_i.weight,
                                     self.conv_tran_r.bias, self.conv_tran_i.bias, True,
                                     self.kernel_size, (1, 1), self.corr_padding))


class ComplexLinear(Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.fc_r = Linear(in_features, out_features)
        self.fc_i = Linear(in_features, out_features)

    def forward(self, input):
        return F.complex_linear(input, self.fc_r.weight, self.fc_i.weight, self.fc_r.bias, self.fc_i.bias)


_EVAL_COEF = weakref.WeakKeyDictionary()      # CBN module -> (tensor refs, versions, (stats, coef)) of eval-mode calls


class _ComplexBatchNorm(Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.momentum = momentum
        self.affine = affine
        self.track_running_stats = track_running_stats
        if self.affine:
            self.weight = Parameter(torch.Tensor(num_features, 3))
            self.bias = Parameter(torch.Tensor(num_features, 2))
        else:
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)
        if self.track_running_stats:
            self.register_buffer('running_mean', torch.zeros(num_features, dtype=torch.complex64))
            self.register_buffer('running_covar', torch.zeros(num_features, 3))
            self.running_covar[:, 0] = SQRT2
            self.running_covar[:, 1] = SQRT2
            self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        else:
            self.register_parameter('running_mean', None)
            self.register_parameter('running_covar', None)
            self.register_parameter('num_batches_tracked', None)
        self.reset_parameters()

    def reset_running_stats(self):
        if self.track_running_stats:
            self.running_mean.zero_()
            self.running_covar.zero_()
            self.running_covar[:, 0] = SQRT2
            self.running_covar[:, 1] = SQRT2
            self.num_batches_tracked.zero_()

    def reset_parameters(self):
        self.reset_running_stats()
        if self.affine:
            with torch.no_grad():
                self.weight[:, :2] = SQRT2
                self.weight[:, 2] = 0
                self.bias.zero_()

    def _hip_forward(self, x_nhwc, act=F.ACT_NONE, drop_p=0.0, seed=0, count=True, attention=None, two=False, stat=None):
        """x: float [B,H,W,C,2].  Shared with the fused C_NETWORK.forward (which advances all the
        num_batches_tracked counters of the network with one launch and passes count=False).
        attention = (fc0_r, fc0_i, fc2_r, fc2_i, conv1_r, conv1_i, ksize, drop_p, seed): the attention block that
        follows this CBN in a decoder stage, run as one autograd node with it (F.cbn_attention)."""
        use_batch = self.training or not self.track_running_stats
        momentum = -1.0
        if self.training and self.track_running_stats:
            if count or self.momentum is None:
                self.num_batches_tracked += 1
            momentum = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
            F.note_state_update()              # the kernel rewrites the running statistics in place
        rm = torch.view_as_real(self.running_mean) if self.track_running_stats else None
        if attention is not None and drop_p:
            raise F.DcsHipError('CBN + attention: dropout belongs to the attention block')
        if not use_batch and not torch.is_grad_enabled():
            y = self._eval_forward(x_nhwc, rm, act, drop_p, seed)
            if two:
                return y, y
            return y if attention is None else F.attention_block(y, *attention)
        if not use_batch:
            stat = None                        # (stat: the batch statistics the producing conv's epilogue left, F.cconv2d_with_stats)
        if attention is not None:
            return F.cbn_attention(x_nhwc, self.weight, self.bias, rm, self.running_covar, self.eps, momentum, use_batch,
                                   act, *attention, stat)
        if two and torch.is_grad_enabled():    # the output has two consumers: one tensor each (F._CbnTwoFn)
            return F.cbn_two(x_nhwc, self.weight, self.bias, rm, self.running_covar, self.eps, momentum, use_batch,
                             act, drop_p, seed, stat)
        y = F.cbn(x_nhwc, self.weight, self.bias, rm, self.running_covar, self.eps, momentum, use_batch,
                  act, drop_p, seed, stat)
        return (y, y) if two else y

    def eval_coef(self):
        """The cached inference-time coefficients [C, 6] of this CBN (see _eval_forward), or None when there is no valid
        entry (first eval pass, training mode, autograd on, parameters or running statistics changed since)."""
        if self.training or torch.is_grad_enabled() or not self.track_running_stats:
            return None
        tensors = (self.weight, self.bias, self.running_mean, self.running_covar)
        vers = (F.state_generation(),) + tuple(None if t is None else (t._version, t.data_ptr()) for t in tensors)
        ent = _EVAL_COEF.get(self)
        if ent is not None and ent[1] == vers and all((r is None and t is None) or (r is not None and r() is t)
                                                      for r, t in zip(ent[0], tensors)):
            return ent[2][1]
        return None

    def _eval_forward(self, x, rm, act, drop_p, seed):
        """Inference: the whitening + affine coefficients are constants of (weight, bias, running statistics): computed
        by the first call, kept per module in a weak dictionary (guarded by the identity and version of the four tensors) and re-used —
        one launch per CBN instead of two."""
        tensors = (self.weight, self.bias, self.running_mean, self.running_covar)
        vers = (F.state_generation(),) + tuple(None if t is None else (t._version, t.data_ptr()) for t in tensors)
        ent = _EVAL_COEF.get(self)
        cached = None
        if ent is not None and ent[1] == vers and all((r is None and t is None) or (r is not None and r() is t)
                                                      for r, t in zip(ent[0], tensors)):
            cached = ent[2]
        y, stats, coef = F.ops.cbn(x, self.weight, self.bias, rm, self.running_covar, self.eps, -1.0, False, act,
                                   drop_p, seed, coef_cached=cached)
        if cached is None and not (x.is_cuda and torch.cuda.is_current_stream_capturing()):
            refs = tuple(None if t is None else weakref.ref(t) for t in tensors)
            _EVAL_COEF[self] = (refs, vers, (stats, coef))
        return y


class ComplexBatchNorm2d(_ComplexBatchNorm):
    def forward(self, input):
        return F.from_nhwc(self._hip_forward(F.to_nhwc(input)))
