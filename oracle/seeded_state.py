"""ORACLE — TEST INFRASTRUCTURE ONLY.

Deterministic, construction-order-independent fill of a DCS-Net state_dict.
Every tensor is generated from a generator seeded by crc32(key name) ^ seed, so
the reference net (in make_golden.py), the oracle net and the HIP net receive
bit-identical parameters without shipping an 11.6 MB checkpoint as a fixture.
BatchNorm affine weights, biases and running statistics are moved off their
defaults so that parity tests exercise them.
"""
import zlib
import torch


def _gen(key, seed):
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _u(shape, g, lo=-1.0, hi=1.0):
    return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo


def seeded_tensor(key, ref, seed):
    """Value for state_dict entry ``key`` shaped/dtyped like ``ref``."""
    g = _gen(key, seed)
    leaf = key.split('.')[-1]
    shape = tuple(ref.shape)
    if leaf == 'num_batches_tracked':
        return torch.zeros_like(ref)
    if leaf == 'running_mean':                       # complex64 [C]
        return torch.complex(_u(shape, g, -0.2, 0.2), _u(shape, g, -0.2, 0.2))
    if leaf == 'running_covar':                      # [C,3] positive definite
        out = torch.empty(shape)
        out[:, 0] = _u(shape[:1], g, 0.8, 1.8)
        out[:, 1] = _u(shape[:1], g, 0.8, 1.8)
        out[:, 2] = _u(shape[:1], g, -0.3, 0.3)
        return out
    if leaf == 'weight' and len(shape) == 2 and shape[1] == 3 and 'fc_' not in key:   # CBN weight
        out = torch.empty(shape)
        out[:, 0] = 1.4142135 + _u(shape[:1], g, -0.3, 0.3)
        out[:, 1] = 1.4142135 + _u(shape[:1], g, -0.3, 0.3)
        out[:, 2] = _u(shape[:1], g, -0.3, 0.3)
        return out
    if leaf == 'bias' and len(shape) == 2 and shape[1] == 2:                          # CBN bias
        return _u(shape, g, -0.2, 0.2)
    if 'lstm' in key:
        k = 1.0 / 8.0                                # 1/sqrt(hidden=64)
        return _u(shape, g, -k, k)
    if leaf == 'bias':
        return _u(shape, g, -0.05, 0.05)
    # conv / convT / linear weights: xavier-uniform-like bound from fan_in + fan_out
    rf = 1
    for s in shape[2:]:
        rf *= s
    fan = (shape[0] + shape[1]) * rf if len(shape) >= 2 else shape[0]
    bound = (6.0 / fan) ** 0.5
    return _u(shape, g, -bound, bound)


@torch.no_grad()
def fill_state(module, seed=0):
    sd = module.state_dict()
    new = {k: seeded_tensor(k, v, seed).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


def seeded_input(B, F, T, seed=0, scale=1.0):
    g = torch.Generator(device='cpu')
    g.manual_seed(1000 + seed)
    re = torch.randn((B, F, T), generator=g) * scale
    im = torch.randn((B, F, T), generator=g) * scale
    return torch.complex(re, im)


@torch.no_grad()
def fill_state_stream(module, seed=5):
    """Stock torch.nn models (the real-valued R_NETWORK): every parameter / buffer drawn from ONE seeded
    stream in sorted-key order; running variances kept positive."""
    g = torch.Generator().manual_seed(seed)
    for k, v in sorted(module.state_dict().items()):
        if k.endswith('num_batches_tracked'):
            continue
        if k.endswith('running_var'):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        else:
            fan = v[0].numel() if v.dim() > 1 else 16
            v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * (1.5 / max(fan, 1)) ** 0.5)
    return module
