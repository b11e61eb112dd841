"""Mask math, complex element-wise layers and step functions around the hot path, with the
names and argument meaning of the reference's ``network_functions.py`` so ``c_network`` /
``train.py`` / ``test.py`` find what they star-import.

On the hot path (HIP, libdcsnet_hip.so):  bound_cRM (:77-88), cRM (:62-75), the bound + complex
multiply + subtract of the step functions (:240-243), complex_lrelu / complex_sigmoid
(:98-112), and the waveform synthesis of mag_phase_2_wave around the inverse FFT (:140-150: polar
spectrum, 512-point inverse real FFT (csrc/fft512.hip), window / overlap-add / envelope), SiSNR and the assembly of
the configured loss pair (csrc/synth.hip).  On PyTorch-ROCm: the non-configured loss types (wSDR, L1, MSE reductions)
and, for n_fft != 512 only, the inverse FFT (rocFFT via torch.fft.irfft).
"""
import sys

import torch

from . import functional as F
from . import ops

try:                                     # metric packages are absent from this image
    from pypesq import pesq
except ImportError:                      # pragma: no cover
    pesq = None
try:
    from pystoi import stoi
except ImportError:                      # pragma: no cover
    from .metrics import stoi            # the published algorithm restated (parity unpinned: dcsnet/metrics.py)


def _mode():
    """The reference reads the network choice from sys.argv[1] deep inside the model code
    (network_functions.py:170,223); default to 'dcs' when a harness has not set it."""
    return sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ('dcs', 'drs', 'dc', 'dr') else 'dcs'


def check_inf_neginf_nan(tensor, error_msg):
    t = torch.view_as_real(tensor) if tensor.is_complex() else tensor
    assert bool(torch.isfinite(t).all()), error_msg


class SiSNR(object):
    """network_functions.py:30-42.  [B, L] float32 device signals take the fused HIP kernels (two launches forward
    including the batch mean, one backward; gradient to the estimate — the clean signal is data in every call site of
    the reference, :186-199); any other input runs the reference's formula as written."""

    def __call__(self, clean, estimate, eps=1e-8):
        if (estimate.is_cuda and estimate.dim() == 2 and estimate.dtype == torch.float32 and clean.shape == estimate.shape
                and not clean.requires_grad):
            return F.sisnr_mean(clean, estimate, eps)
        dot = torch.sum(estimate * clean, -1, keepdim=True)
        energy = torch.sum(clean * clean, -1, keepdim=True)
        target = dot * clean / (energy + eps)
        resid = estimate - target
        t = torch.sum(target * target, -1, keepdim=True)
        r = torch.sum(resid * resid, -1, keepdim=True)
        return torch.mean(10 * torch.log10(t / (r + eps) + eps))


class wSDR(object):
    """network_functions.py:45-60."""

    def __call__(self, mixed, clean, clean_est, eps=2e-8):
        def neg_cos(a, b):
            return -(a * b).sum(1) / (torch.norm(a, p=2, dim=1) * torch.norm(b, p=2, dim=1) + eps)

        noise, noise_est = mixed - clean, mixed - clean_est
        e_c, e_n = (clean ** 2).sum(1), (noise ** 2).sum(1)
        alpha = e_c / (e_c + e_n + eps)
        return torch.mean(alpha * neg_cos(clean, clean_est) + (1 - alpha) * neg_cos(noise, noise_est))


# ---- hot-path element-wise math: HIP ----------------------------------------------------------

def cRM(S, Y, eps=1e-8):
    return F.crm_complex(S, Y, eps)


def bound_cRM(cRM, hparams):
    return F.bound_crm_complex(cRM, hparams['atan2_eps'])


def complex_mat_mult(A, B):
    """Element-wise complex product (network_functions.py:90-96).  The step functions below do
    not call this: they use the fused bound + multiply + subtract kernel."""
    return A * B


def bound_mask_apply(noisy_data, mask_out, hparams):
    """network_functions.py:240-243 in one kernel: returns (bounded mask, Y (.) M, Y - Y (.) M).  A batch of ONE reaches
    here with the network's output already squeezed to [F,T] (c_network.py:224); the reference's element-wise product
    then broadcasts it against noisy_data [1,F,T]: same shapes out here (mask [F,T], N_hat / S_hat [1,F,T])."""
    if mask_out.dim() + 1 == noisy_data.dim() and noisy_data.shape[0] == 1:
        M, N, S = F.bound_mask_apply_complex(noisy_data, mask_out.unsqueeze(0), hparams['atan2_eps'])
        return M.squeeze(0), N, S
    return F.bound_mask_apply_complex(noisy_data, mask_out, hparams['atan2_eps'])


def complex_lrelu(input):
    return F.complex_lrelu(input)


def complex_sigmoid(input):
    return F.complex_sigmoid(input)


class ComplexLReLU(torch.nn.Module):
    def forward(self, input):
        return complex_lrelu(input)


class ComplexSigmoid(torch.nn.Module):
    def forward(self, input):
        return complex_sigmoid(input)


def complex_adaptive_avg_pool2d(input, output_size=1):
    if output_size not in (1, (1, 1)):
        raise NotImplementedError('only the global pool the reference uses (c_network.py:56-57)')
    return torch.view_as_complex(torch.view_as_real(input).mean(dim=(2, 3), keepdim=True))


def complex_adaptive_max_pool2d(input, output_size=1):
    """The reference's "max" pool is an average pool (network_functions.py:135-138)."""
    return complex_adaptive_avg_pool2d(input, output_size)


class ComplexAdaptiveAvgPool2d(torch.nn.Module):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, input):
        return complex_adaptive_avg_pool2d(input, self.output_size)


class ComplexAdaptiveMaxPool2d(torch.nn.Module):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, input):
        return complex_adaptive_max_pool2d(input, self.output_size)


# ---- step functions (plumbing on PyTorch-ROCm around the HIP path) ------------------------------

_windows = {}


def _window_on(config, device):
    """The synthesis window on `device`, uploaded once per window TENSOR (a per-call host-to-device copy would also be illegal
    inside a hipGraph capture).  The entry holds a weak reference to the host tensor: a recycled id() of a dead window must not
    serve another config the old one's copy."""
    import weakref
    src = config.window
    if src.device == device:
        return src
    key = (id(src), device)
    ent = _windows.get(key)
    if ent is not None and ent[0]() is src and ent[2] == src._version:
        return ent[1]
    w = src.to(device)
    if not (device.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
        if len(_windows) > 64:
            _windows.clear()
        _windows[key] = (weakref.ref(src), w, src._version)
    return w


_envelopes = {}


def _inv_envelope(window, T, hop):
    """1 / squared-window envelope for T frames; depends only on (window, T, hop): computed once per device window tensor (HIP).
    The entry keeps the window alive, so its address cannot be handed to another tensor while the entry exists."""
    key = (window.data_ptr(), T, hop, window.device)
    ent = _envelopes.get(key)
    if ent is not None and ent[0] is window and ent[2] == window._version:
        return ent[1]
    env = ops.istft_envelope(window, T, hop)
    if not (window.is_cuda and torch.cuda.is_current_stream_capturing()):
        if len(_envelopes) > 64:
            _envelopes.clear()
        _envelopes[key] = (window, env, window._version)
    return env


def _frames_to_wave(comp_t, n_fft, hop, window, normalized):
    """comp_t: complex [B, T, n_fft/2 + 1] frame-major one-sided spectra -> waveform [B, hop (T-1)]: one contiguous
    batched inverse real FFT, then window / overlap-add / envelope / centre trim in ONE HIP pass
    (dcs_istft_ola_fwd).  torch.istft's NOLA check — a host read-back per call, illegal under hipGraph capture — is
    not made: the configured Hann window at hop <= n_fft/2 satisfies it by construction."""
    T = comp_t.shape[1]
    frames = torch.fft.irfft(comp_t, n=n_fft, dim=-1)                                # [B, T, n_fft]
    scale = float(n_fft) ** 0.5 if normalized else 1.0
    return F.istft_ola(frames, window, _inv_envelope(window, T, hop), hop, scale)


def istft(comp, n_fft, hop, window, normalized):
    """torch.istft(comp [B, n_fft/2+1, T], center=True, onesided, length=None) on the HIP synthesis path."""
    return _frames_to_wave(comp.transpose(1, 2).contiguous(), n_fft, hop, window, normalized)


def mag_phase_2_wave(mag, phase, config):
    """network_functions.py:140-150; the window follows the tensor's device instead of the
    reference's hard-coded cuda index (:147)."""
    comp = torch.complex(mag * torch.cos(phase), mag * torch.sin(phase))
    comp = torch.nn.functional.pad(comp, (0, 0, 0, 1))
    return istft(comp, config.fft_size, config.hop_length, _window_on(config, comp.device), config.normalise_stft)


def _polar_wave(z, eps, config):
    """mag_phase_2_wave(|z|, atan2(z_i, z_r + eps)) of the step functions (network_functions.py:213-221,
    :244-247): polar round trip + zero bin + frame-major transpose in one HIP pass, contiguous irfft, fused
    overlap-add."""
    window = _window_on(config, z.device)
    n_fft, hop = config.fft_size, config.hop_length
    if z.shape[1] + 1 != n_fft // 2 + 1:                  # not the configured geometry: the three-node spelling
        comp_t = F.polar_frames_complex(z, 1, eps)
        return _frames_to_wave(comp_t, n_fft, hop, window, config.normalise_stft)
    scale = float(n_fft) ** 0.5 if config.normalise_stft else 1.0
    return F.polar_wave(z, window, _inv_envelope(window, z.shape[2], hop), n_fft, hop, scale, eps)


def calc_metric(clean_audio, predict_audio, config, metric):
    if metric is None:
        return float('nan')
    vals = []
    for i in range(predict_audio.shape[0]):
        v = metric(clean_audio[i, :].cpu().numpy(), predict_audio[i, :].cpu().numpy(), config.sr)
        if v == v:
            vals.append(v)
    return float(sum(vals)) / max(len(vals), 1)


def calc_loss(self, **kw):
    """network_functions.py:168-208: every noise_loss_type (0-6) with speech_loss_type 0.  The configured pair (type 6 +
    type 0, config.py:38-39) on [B, L] device signals runs as fused HIP launches; the other types are plain torch
    reductions around the same HIP network / synthesis outputs."""
    mode = _mode()
    alpha = self.hparams['speech_alpha']
    if (mode in ('dcs', 'drs') and self.hparams['noise_loss_type'] == 6 and isinstance(self.config.SiSNR, SiSNR) and
            all(kw[k].is_cuda and kw[k].dim() == 2 and kw[k].dtype == torch.float32
                for k in ('clean_audio', 'predict_clean_audio', 'noise_audio', 'predict_noise_audio')) and
            not (kw['clean_audio'].requires_grad or kw['noise_audio'].requires_grad)):
        # the configured pair of SiSNR losses and their assembly in five HIP launches (forward + backward) — three when
        # the step handed over the stacked signals (and the train step's NaN flag rides the assembly launch)
        if '_pair' in kw:
            tw, ew = kw['_pair']
            skip = getattr(self, '_dcs_skip_flag', None)
            if skip is not None:
                self._dcs_skip_written = True
            return F.sisnr_losses_pair(tw, ew, alpha, skip=skip)
        return F.sisnr_losses(kw['clean_audio'], kw['predict_clean_audio'], kw['noise_audio'], kw['predict_noise_audio'],
                              alpha)
    speech_loss = alpha * (-self.config.SiSNR(kw['clean_audio'], kw['predict_clean_audio']))
    if mode in ('dc', 'dr'):
        return speech_loss
    t = self.hparams['noise_loss_type']
    # network_functions.py:171-193.  nn.L1Loss of two COMPLEX masks is the mean complex modulus |a - b| (not the mean
    # of |re| and |im| parts); wSDR and the audio-domain L1 / the mask MSE terms are plain torch reductions.
    l1_mask = lambda: (kw['target_noise_mask'] - kw['predict_noise_mask']).abs().mean()
    wsdr = lambda: self.config.wSDR(kw['noisy_audio'], kw['noise_audio'], kw['predict_noise_audio'])
    l1_audio = lambda: self.config.L1(kw['noise_audio'], kw['predict_noise_audio'])
    if t == 6:
        raw = -self.config.SiSNR(kw['noise_audio'], kw['predict_noise_audio'])
    elif t == 0:
        raw = l1_mask()
    elif t == 1:
        raw = wsdr()
    elif t == 2:
        raw = l1_mask() + l1_audio()
    elif t == 3:
        raw = wsdr() + l1_audio()
    elif t == 4:
        raw = wsdr() + l1_mask()
    elif t == 5:
        tm, pm = kw['target_noise_mask'], kw['predict_noise_mask']
        if tm.is_complex():
            raw = wsdr() + self.config.mse(tm.real, pm.real) + self.config.mse(tm.imag, pm.imag)
        else:
            raw = wsdr() + self.config.mse(tm, pm)
    else:
        raise ValueError(f'noise_loss_type {t}: the reference defines types 0-6 (network_functions.py:171-193)')
    noise_loss = 1 - alpha * raw                       # QUIRK kept: network_functions.py:196
    return noise_loss, speech_loss, noise_loss + speech_loss


def _stacked(a, b):
    """[2B, ...] tensor of two equally shaped batches: a zero-copy view when b sits right behind a in one allocation (the
    captured train step's input buffers do), else one concatenation."""
    if (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.device == b.device and
            a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and
            b.storage_offset() == a.storage_offset() + a.numel() and not (a.requires_grad or b.requires_grad)):
        return torch.as_strided(a, (2 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
    return torch.cat((a, b), 0)


def _complex_step_pair(self, noise_data, noisy_data, clean_data):
    """The 'dcs' train step with the two targets and the two estimates each synthesised as ONE batch of 2B signals (rows
    [0,B) noise, [B,2B) speech): 3 + 3 synthesis launches instead of 6 + 6, one SiSNR launch instead of two, and the
    same halved counts backward.  Same arithmetic per signal as _complex_step."""
    eps = self.hparams['atan2_eps']
    cfg = self.config
    B = noisy_data.shape[0]
    tw = _polar_wave(_stacked(noise_data, clean_data), eps, cfg)
    # the network's own bound_cRM and the second one of network_functions.py:240 run as ONE kernel on the raw output when
    # the module offers it (this build's C_NETWORK: forward(x, bound=False)); any other module gets the two-step form
    fused = getattr(self, 'supports_unbounded_forward', False)
    mask_out = self(noisy_data, bound=False) if fused else self(noisy_data)
    squeezed = mask_out.dim() + 1 == noisy_data.dim() and B == 1           # the B = 1 squeeze quirk (c_network.py:224)
    if fused and cfg.fft_size == 512:
        # Round 5: mask application and the synthesis' polar round trip in one kernel each way — the estimates make no round trip
        # through HBM (F.bound2_apply_polar_wave_pair).  The twice-bounded mask itself is read by the mask-domain losses only
        # (noise_loss_type 0, 2, 4, 5), never on this path (type 6): it is not stored.
        drop = self.__dict__.pop('_pending_dropout', (0.0, 0))
        window = _window_on(cfg, noisy_data.device)
        scale = float(cfg.fft_size) ** 0.5 if cfg.normalise_stft else 1.0
        mask, ew = F.bound2_apply_polar_wave_pair(noisy_data, mask_out.unsqueeze(0) if squeezed else mask_out, window,
                                                  _inv_envelope(window, noisy_data.shape[2], cfg.hop_length), cfg.fft_size,
                                                  cfg.hop_length, scale, eps, drop, want_mask=False)
        return {'noise_audio': tw[:B], 'clean_audio': tw[B:], 'predict_noise_mask': mask,
                'predict_noise_audio': ew[:B], 'predict_clean_audio': ew[B:], '_pair': (tw, ew)}
    if fused:
        drop = self.__dict__.pop('_pending_dropout', (0.0, 0))
        apply_pair = lambda y_, m_, e_: F.bound2_mask_apply_pair_complex(y_, m_, e_, drop)
    else:
        apply_pair = F.bound_mask_apply_pair_complex
    if squeezed:
        mask, NS = apply_pair(noisy_data, mask_out.unsqueeze(0), eps)
        mask = mask.squeeze(0)
    else:
        mask, NS = apply_pair(noisy_data, mask_out, eps)
    ew = _polar_wave(NS.reshape((2 * B,) + tuple(NS.shape[2:])), eps, cfg)
    return {'noise_audio': tw[:B], 'clean_audio': tw[B:], 'predict_noise_mask': mask,
            'predict_noise_audio': ew[:B], 'predict_clean_audio': ew[B:], '_pair': (tw, ew)}


def _complex_step(self, noise_data, noisy_data, clean_data, need_noisy_audio=False):
    eps = self.hparams['atan2_eps']
    cfg = self.config
    if (_mode() in ('dcs', 'drs') and not need_noisy_audio and self.hparams.get('noise_loss_type') == 6 and
            noisy_data.is_cuda and noise_data.shape == noisy_data.shape == clean_data.shape and
            noisy_data.shape[1] + 1 == cfg.fft_size // 2 + 1):
        return _complex_step_pair(self, noise_data, noisy_data, clean_data)
    audio = {'noise_audio': _polar_wave(noise_data, eps, cfg), 'clean_audio': _polar_wave(clean_data, eps, cfg)}
    if need_noisy_audio or self.hparams.get('noise_loss_type') in (1, 3, 4, 5):
        # only the wSDR losses and the evaluation outputs read it; the reference always builds it
        audio['noisy_audio'] = _polar_wave(noisy_data, eps, cfg)
    mask_out = self(noisy_data)
    if _mode() in ('dcs', 'drs'):
        if need_noisy_audio or self.hparams.get('noise_loss_type') in (0, 2, 4, 5):
            # only the mask-domain L1 / MSE losses read it (the reference always builds it: network_functions.py:237-239)
            audio['target_noise_mask'] = bound_cRM(cRM(noise_data, noisy_data), self.hparams)
        mask, noise_hat, clean_hat = bound_mask_apply(noisy_data, mask_out, self.hparams)
        audio['predict_noise_mask'] = mask
        audio['predict_noise_audio'] = _polar_wave(noise_hat, eps, cfg)
        audio['predict_clean_audio'] = _polar_wave(clean_hat, eps, cfg)
    else:                                               # 'dc': the mask is applied, not subtracted
        _, clean_hat, _ = bound_mask_apply(noisy_data, mask_out, self.hparams)
        audio['predict_clean_audio'] = _polar_wave(clean_hat, eps, cfg)
    return audio


def _real_step(self, noise_data, noisy_data, clean_data, need_noisy_audio=False):
    """dtype == "real" (DRS / DR-Net, network_functions.py:224-232, :261-267): the network sees |noisy| and returns a
    sigmoid mask; the estimates keep the noisy phase."""
    eps = self.hparams['atan2_eps']
    cfg = self.config
    audio = {'noise_audio': _polar_wave(noise_data, eps, cfg), 'clean_audio': _polar_wave(clean_data, eps, cfg)}
    if need_noisy_audio or self.hparams.get('noise_loss_type') in (1, 3, 4, 5):
        audio['noisy_audio'] = _polar_wave(noisy_data, eps, cfg)
    noisy_mag = torch.abs(noisy_data)
    noisy_phase = torch.atan2(noisy_data.imag, noisy_data.real + eps)
    mask = self(noisy_mag)
    if _mode() in ('dcs', 'drs'):
        if need_noisy_audio or self.hparams.get('noise_loss_type') in (0, 2, 4, 5):
            audio['target_noise_mask'] = torch.sigmoid(torch.abs(noise_data) / noisy_mag)
        noise_mag_hat = noisy_mag * mask
        audio['predict_noise_mask'] = mask
        audio['predict_noise_audio'] = mag_phase_2_wave(noise_mag_hat, noisy_phase, cfg)
        audio['predict_clean_audio'] = mag_phase_2_wave(noisy_mag - noise_mag_hat, noisy_phase, cfg)
    else:                                               # 'dr': the mask is applied, not subtracted
        audio['predict_clean_audio'] = mag_phase_2_wave(noisy_mag * mask, noisy_phase, cfg)
    return audio


def _step(self, dtype, *args, **kw):
    if dtype == 'complex':
        return _complex_step(self, *args, **kw)
    if dtype == 'real':
        return _real_step(self, *args, **kw)
    raise ValueError(f'dtype {dtype!r}: the reference passes "complex" (C_NETWORK) or "real" (R_NETWORK)')


def train_batch_2_loss(self, train_batch, batch_idx, dtype):
    noise_data, noisy_data, clean_data = train_batch[:3]
    return calc_loss(self, **_step(self, dtype, noise_data, noisy_data, clean_data))


def val_batch_2_metric_loss(self, val_batch, val_idx, dtype):
    noise_data, noisy_data, clean_data = val_batch[:3]
    a = _step(self, dtype, noise_data, noisy_data, clean_data, need_noisy_audio=True)
    pesq_av = calc_metric(a['clean_audio'], a['predict_clean_audio'], self.config, pesq)
    stoi_av = calc_metric(a['clean_audio'], a['predict_clean_audio'], self.config, stoi)
    losses = calc_loss(self, **a)
    if _mode() in ('dcs', 'drs'):
        return (*losses, pesq_av, stoi_av, a['predict_noise_audio'], a['predict_clean_audio'],
                a['noise_audio'], a['noisy_audio'], a['clean_audio'])
    return losses, pesq_av, stoi_av, a['predict_clean_audio'], a['noise_audio'], a['noisy_audio'], a['clean_audio']


def test_batch_2_metric_loss(self, test_batch, test_idx, dtype):
    out = val_batch_2_metric_loss(self, test_batch[:4], test_idx, dtype)
    if _mode() in ('dcs', 'drs'):
        return (*out, test_batch[3], test_batch[4])
    return out


def epoch_end(self, outputs, type):
    """Audio-sample logging (network_functions.py:450-498) needs the trainer's TensorBoard
    logger; without one there is nothing to write."""
    logger = getattr(self, 'logger', None)
    if logger is None or not outputs:
        return
    exp = logger.experiment
    last = outputs[-1]
    for name, wav in last.items():
        exp.add_audio(f'{type}_{name}', torch.as_tensor(wav[0]).unsqueeze(0), self.current_epoch, self.config.sr)
