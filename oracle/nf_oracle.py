"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path.

CPU restatement (pure PyTorch, fp32) of the element-wise complex math and the
step function that surround ``C_NETWORK.forward`` in the reference's
``network_functions.py``.  Each function names the reference lines it follows.

PARITY STATUS: pinned.  ``oracle/make_golden.py`` imports the reference's own
``network_functions.py`` in the build container and stores its outputs on
seeded inputs under ``tests/golden/nf_*.npz``; ``tests/test_oracle.py`` checks
this file against those vectors.
"""
import torch

ATAN2_EPS = 10e-7          # config.py:45  hparams['atan2_eps']

# complexPyTorch casts with a literal torch.complex64; a calibration run (tools/full_size_grad_probe.py) sets this to
# complex128 to get an fp64 ground truth of the same arithmetic.  Every test leaves it at complex64.
CDTYPE = torch.complex64


def _cplx(re, im):
    return torch.complex(re, im)


def bound_cRM(mask, hparams=None):
    """network_functions.py:77-88.  m = tanh|M|, phase taken twice through atan2
    with eps added to the real part each time."""
    eps = ATAN2_EPS if hparams is None else hparams['atan2_eps']
    mag = torch.tanh(torch.abs(mask))
    phi1 = torch.atan2(mask.imag, mask.real + eps)
    re1 = mag * torch.cos(phi1)
    im1 = mag * torch.sin(phi1)
    phi2 = torch.atan2(im1, re1 + eps)
    return _cplx(mag * torch.cos(phi2), mag * torch.sin(phi2))


def cRM(S, Y, eps=1e-8):
    """network_functions.py:62-75.  M = S conj(Y) / (|Y|^2 + eps)."""
    den = torch.square(Y.real) + torch.square(Y.imag) + eps
    mr = (Y.real * S.real + Y.imag * S.imag) / den
    mi = (Y.real * S.imag - Y.imag * S.real) / den
    return _cplx(mr, mi)


def complex_mat_mult(A, B):
    """network_functions.py:90-96.  Element-wise complex product (not a matmul)."""
    return _cplx(A.real * B.real - A.imag * B.imag, A.real * B.imag + A.imag * B.real)


def mask_apply_subtract(Y, M_raw, hparams=None):
    """network_functions.py:240-243 (train), :313-316 (val), :394-397 (test):
    bound the network output a second time, multiply, subtract."""
    M = bound_cRM(M_raw, hparams)
    N_hat = complex_mat_mult(Y, M)
    return M, N_hat, Y - N_hat


def complex_lrelu(x):
    """network_functions.py:103-105, slope 0.01 (torch default)."""
    return _cplx(torch.nn.functional.leaky_relu(x.real), torch.nn.functional.leaky_relu(x.imag))


def complex_sigmoid(x):
    """network_functions.py:111-112."""
    return torch.sigmoid(x.real).type(CDTYPE) + 1j * torch.sigmoid(x.imag).type(CDTYPE)


def complex_adaptive_avg_pool2d(x, output_size):
    """network_functions.py:122-125."""
    r = torch.nn.functional.adaptive_avg_pool2d(x.real, output_size)
    i = torch.nn.functional.adaptive_avg_pool2d(x.imag, output_size)
    return r.type(CDTYPE) + 1j * i.type(CDTYPE)


def complex_adaptive_max_pool2d(x, output_size):
    """network_functions.py:135-138.  QUIRK kept on purpose: the reference's
    "max" pool calls adaptive_AVG_pool2d, so this is the average pool again."""
    return complex_adaptive_avg_pool2d(x, output_size)


class ComplexLReLU(torch.nn.Module):
    def forward(self, x):
        return complex_lrelu(x)


class ComplexSigmoid(torch.nn.Module):
    def forward(self, x):
        return complex_sigmoid(x)


class ComplexAdaptiveAvgPool2d(torch.nn.Module):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, x):
        return complex_adaptive_avg_pool2d(x, self.output_size)


class ComplexAdaptiveMaxPool2d(torch.nn.Module):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, x):
        return complex_adaptive_max_pool2d(x, self.output_size)


def si_snr(clean, estimate, eps=1e-8):
    """network_functions.py:30-42."""
    dot = torch.sum(estimate * clean, -1, keepdim=True)
    energy = torch.sum(clean * clean, -1, keepdim=True)
    target = dot * clean / (energy + eps)
    resid = estimate - target
    t = torch.sum(target * target, -1, keepdim=True)
    r = torch.sum(resid * resid, -1, keepdim=True)
    return torch.mean(10 * torch.log10(t / (r + eps) + eps))


def mag_phase_2_wave(mag, phase, n_fft=512, hop=32, window=None):
    """network_functions.py:140-150 (device pin at :147 dropped: CPU oracle).
    QUIRK kept: the zero row is padded at the END of the bin axis, so bins
    1..256 of the STFT (data.py:118) land on bins 0..255 of the iSTFT."""
    comp = _cplx(mag * torch.cos(phase), mag * torch.sin(phase))
    comp = torch.nn.functional.pad(comp, (0, 0, 0, 1))
    if window is None:
        window = torch.hann_window(n_fft)
    return torch.istft(comp, n_fft=n_fft, hop_length=hop, win_length=n_fft,
                       window=window.to(comp.device), normalized=True)


def _polar(x, eps):
    return torch.abs(x), torch.atan2(x.imag, x.real + eps)


def dcs_train_losses(net, noise, noisy, clean, speech_alpha=0.7, eps=ATAN2_EPS):
    """network_functions.py:210-258 in 'dcs' mode with calc_loss :168-208,
    noise_loss_type 6 / speech_loss_type 0 (config.py:38-39).
    QUIRK kept: noise_loss = 1 - alpha * L  (network_functions.py:196)."""
    noise_audio = mag_phase_2_wave(*_polar(noise, eps))
    clean_audio = mag_phase_2_wave(*_polar(clean, eps))
    M_raw = net(noisy)
    _, n_hat, s_hat = mask_apply_subtract(noisy, M_raw, {'atan2_eps': eps})
    n_hat_audio = mag_phase_2_wave(*_polar(n_hat, eps))
    s_hat_audio = mag_phase_2_wave(*_polar(s_hat, eps))
    noise_loss = 1 - speech_alpha * (-si_snr(noise_audio, n_hat_audio))
    speech_loss = speech_alpha * (-si_snr(clean_audio, s_hat_audio))
    return noise_loss, speech_loss, noise_loss + speech_loss


def drs_train_losses(net, noise, noisy, clean, speech_alpha=0.7, eps=ATAN2_EPS):
    """network_functions.py:210-232, :249-258 in 'drs' mode (dtype "real": R_NETWORK sees |noisy|, returns a sigmoid mask;
    the estimates keep the noisy phase) with calc_loss :168-208, noise_loss_type 6 / speech_loss_type 0."""
    noise_audio = mag_phase_2_wave(*_polar(noise, eps))
    clean_audio = mag_phase_2_wave(*_polar(clean, eps))
    noisy_mag, noisy_phase = _polar(noisy, eps)
    mask = net(noisy_mag)
    n_mag = noisy_mag * mask
    n_hat_audio = mag_phase_2_wave(n_mag, noisy_phase)
    s_hat_audio = mag_phase_2_wave(noisy_mag - n_mag, noisy_phase)
    noise_loss = 1 - speech_alpha * (-si_snr(noise_audio, n_hat_audio))
    speech_loss = speech_alpha * (-si_snr(clean_audio, s_hat_audio))
    return noise_loss, speech_loss, noise_loss + speech_loss


def stft_frontend(clean_wave, noisy_wave, n_fft=512, hop=32, window=None, normalized=True):
    """data.py:104-134 for a batch of cropped [B, L] waveforms: noise = noisy - clean in the time domain, then
    torch.stft (center=True, reflect padding) of clean, noise and noisy, keeping bins 1..n_fft/2.
    Returns (noise, noisy, clean) complex64 [B, n_fft/2, T] — the order of the reference's batches."""
    window = torch.hann_window(n_fft) if window is None else window

    def one(x):
        return torch.stft(x, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window, return_complex=True,
                          normalized=normalized)[:, 1:n_fft // 2 + 1, :]
    noise_wave = noisy_wave - clean_wave
    return one(noise_wave), one(noisy_wave), one(clean_wave)
