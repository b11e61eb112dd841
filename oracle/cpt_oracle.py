"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path.

CPU restatement (pure PyTorch, fp32) of the third-party operator library the
reference hot path is built on: ``complexPyTorch==0.3`` (pinned in the
reference at requirements.txt:38; its source is NOT vendored in
/root/reference and is not installed in this image).

Call sites in the reference that this file answers for:
  c_network.py:5-7      star-import of complexLayers, complex_upsample, complex_relu
  c_network.py:58-60,74 ComplexConv2d (1x1 and 7x7, bias=False) inside the attentions
  c_network.py:101,113,148  ComplexBatchNorm2d
  c_network.py:107      ComplexConv2d (encoder)
  c_network.py:124      ComplexLinear
  c_network.py:135,142  ComplexConvTranspose2d (decoder)
  c_network.py:215      complex_upsample
  config.py:5,103       ComplexReLU

PARITY STATUS: **parity unpinned at the complexPyTorch boundary.**  The
reference holds no tests or golden vectors for these layers (SURVEY.md §4,
§8c) and the package cannot be imported here, so what follows restates the
package's published 0.3 algorithm (complexLayers.py / complexFunctions.py):
  * apply_complex(fr, fi, x) = (fr(x.re) - fi(x.im)) + j (fr(x.im) + fi(x.re))
    with TWO independent real layers, each carrying its own bias;
  * ComplexBatchNorm2d = 2x2 whitening with biased batch covariance, eps added
    to Crr and Cii only, closed-form inverse matrix square root, 3-parameter
    symmetric affine weight initialised (sqrt2, sqrt2, 0), running_covar
    initialised (sqrt2, sqrt2, 0), running update scaled by n/(n-1);
  * complex_relu / complex_upsample act on .real and .imag independently.
Self-consistency of this restatement is pinned in tests/test_oracle.py by an
independent formulation (native torch complex conv; R C R = I whitening).
Everything the reference itself defines (network_functions.py, c_network.py
wiring) IS pinned by fixtures generated from the reference: see make_golden.py.

Attribute names (conv_r, conv_i, conv_tran_r, conv_tran_i, fc_r, fc_i,
running_covar, ...) are complexPyTorch's, because they are the reference's
checkpoint contract.
"""
import torch
from torch.nn import Module, Parameter, Conv2d, ConvTranspose2d, Linear
from torch.nn.functional import relu, interpolate

# complexPyTorch casts with a literal torch.complex64; a calibration run (tools/full_size_grad_probe.py) sets this to
# complex128 to get an fp64 ground truth of the same arithmetic.  Every test leaves it at complex64.
CDTYPE = torch.complex64

SQRT2 = 1.4142135623730951


def apply_complex(fr, fi, input, dtype=None):
    dtype = CDTYPE if dtype is None else dtype
    return (fr(input.real) - fi(input.imag)).type(dtype) \
        + 1j * (fr(input.imag) + fi(input.real)).type(dtype)


def complex_relu(input):
    return relu(input.real).type(CDTYPE) + 1j * relu(input.imag).type(CDTYPE)


def complex_upsample(input, size=None, scale_factor=None, mode='nearest',
                     align_corners=None, recompute_scale_factor=None):
    outp_real = interpolate(input.real, size=size, scale_factor=scale_factor, mode=mode,
                            align_corners=align_corners, recompute_scale_factor=recompute_scale_factor)
    outp_imag = interpolate(input.imag, size=size, scale_factor=scale_factor, mode=mode,
                            align_corners=align_corners, recompute_scale_factor=recompute_scale_factor)
    return outp_real.type(CDTYPE) + 1j * outp_imag.type(CDTYPE)


class ComplexReLU(Module):
    def forward(self, input):
        return complex_relu(input)


class ComplexConv2d(Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True):
        super().__init__()
        self.conv_r = Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.conv_i = Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)

    def forward(self, input):
        return apply_complex(self.conv_r, self.conv_i, input)


class ComplexConvTranspose2d(Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 output_padding=0, groups=1, bias=True, dilation=1, padding_mode='zeros'):
        super().__init__()
        self.conv_tran_r = ConvTranspose2d(in_channels, out_channels, kernel_size, stride, padding,
                                           output_padding, groups, bias, dilation, padding_mode)
        self.conv_tran_i = ConvTranspose2d(in_channels, out_channels, kernel_size, stride, padding,
                                           output_padding, groups, bias, dilation, padding_mode)

    def forward(self, input):
        return apply_complex(self.conv_tran_r, self.conv_tran_i, input)


class ComplexLinear(Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.fc_r = Linear(in_features, out_features)
        self.fc_i = Linear(in_features, out_features)

    def forward(self, input):
        return apply_complex(self.fc_r, self.fc_i, input)


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
            torch.nn.init.constant_(self.weight[:, :2], SQRT2)
            torch.nn.init.zeros_(self.weight[:, 2])
            torch.nn.init.zeros_(self.bias)


class ComplexBatchNorm2d(_ComplexBatchNorm):
    def forward(self, input):
        exponential_average_factor = 0.0
        if self.training and self.track_running_stats:
            if self.num_batches_tracked is not None:
                self.num_batches_tracked += 1
                if self.momentum is None:
                    exponential_average_factor = 1.0 / float(self.num_batches_tracked)
                else:
                    exponential_average_factor = self.momentum

        if self.training or (not self.training and not self.track_running_stats):
            mean_r = input.real.mean([0, 2, 3]).type(CDTYPE)
            mean_i = input.imag.mean([0, 2, 3]).type(CDTYPE)
            mean = mean_r + 1j * mean_i
        else:
            mean = self.running_mean

        if self.training and self.track_running_stats:
            with torch.no_grad():
                self.running_mean = exponential_average_factor * mean \
                    + (1 - exponential_average_factor) * self.running_mean

        input = input - mean[None, :, None, None]

        if self.training or (not self.training and not self.track_running_stats):
            n = input.numel() / input.size(1)
            Crr = 1. / n * input.real.pow(2).sum(dim=[0, 2, 3]) + self.eps
            Cii = 1. / n * input.imag.pow(2).sum(dim=[0, 2, 3]) + self.eps
            Cri = (input.real.mul(input.imag)).mean(dim=[0, 2, 3])
        else:
            Crr = self.running_covar[:, 0] + self.eps
            Cii = self.running_covar[:, 1] + self.eps
            Cri = self.running_covar[:, 2]

        if self.training and self.track_running_stats:
            with torch.no_grad():
                self.running_covar[:, 0] = exponential_average_factor * Crr * n / (n - 1) \
                    + (1 - exponential_average_factor) * self.running_covar[:, 0]
                self.running_covar[:, 1] = exponential_average_factor * Cii * n / (n - 1) \
                    + (1 - exponential_average_factor) * self.running_covar[:, 1]
                self.running_covar[:, 2] = exponential_average_factor * Cri * n / (n - 1) \
                    + (1 - exponential_average_factor) * self.running_covar[:, 2]

        det = Crr * Cii - Cri.pow(2)
        s = torch.sqrt(det)
        t = torch.sqrt(Cii + Crr + 2 * s)
        inverse_st = 1.0 / (s * t)
        Rrr = (Cii + s) * inverse_st
        Rii = (Crr + s) * inverse_st
        Rri = -Cri * inverse_st

        input = (Rrr[None, :, None, None] * input.real + Rri[None, :, None, None] * input.imag).type(CDTYPE) \
            + 1j * (Rii[None, :, None, None] * input.imag + Rri[None, :, None, None] * input.real).type(CDTYPE)

        if self.affine:
            input = (self.weight[None, :, 0, None, None] * input.real + self.weight[None, :, 2, None, None] * input.imag
                     + self.bias[None, :, 0, None, None]).type(CDTYPE) \
                + 1j * (self.weight[None, :, 2, None, None] * input.real + self.weight[None, :, 1, None, None] * input.imag
                        + self.bias[None, :, 1, None, None]).type(CDTYPE)
        return input
