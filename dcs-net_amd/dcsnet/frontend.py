"""On-device STFT front end: what the reference's Dataset does per item on DataLoader CPU
workers (data.py:84-134: crop to integer_win_size - hop samples, noise = noisy - clean, three torch.stft calls, keep
bins 1..n_fft/2), for a whole batch of waveforms already on the GPU.

    frames (HIP: window + reflect padding, all three signals)  ->  one batched contiguous real FFT (fft512.hip at n_fft = 512, else rocFFT via torch.fft)
    ->  bins (HIP: drop the DC bin, 1/sqrt(n_fft), transpose to the network's [B, 256, T] layout)

Resampling (data.py:84-85, torchaudio) and file decoding stay with the loader; they are out of scope."""
import torch

from . import ops
from .network_functions import _window_on


def crop_batch(clean, noisy, length, generator=None):
    """data.py:90-104 for a batch: zero-pad items shorter than `length`, otherwise one random start point PER ITEM
    (torch.randint(0, len - length) as the reference draws it); clean / noisy: lists of 1-D tensors or [B, L] tensors
    of equal item lengths.  Returns two [B, length] tensors."""
    out_c, out_n = [], []
    for c, n in zip(clean, noisy):
        if c.shape[0] != n.shape[0]:
            raise Exception('clean_data and noisy_data are not the same length')        # data.py:87-88
        if length > c.shape[0]:
            c = torch.nn.functional.pad(c, (0, length - c.shape[0]))
            n = torch.nn.functional.pad(n, (0, length - n.shape[0]))
            start = 0
        else:
            span = c.shape[0] - length
            start = int(torch.randint(0, span, (1,), generator=generator)) if span > 0 else 0
        out_c.append(c[start:start + length])
        out_n.append(n[start:start + length])
    return torch.stack(out_c), torch.stack(out_n)


def stft_batch(clean_wave, noisy_wave, config):
    """[B, L] float32 device waveforms (L = hop * (T - 1), data.py:91) -> (noise, noisy, clean) complex64 [B, 256, T]:
    exactly the tuple order of the reference's batches (train_batch[:3], network_functions.py:225-228)."""
    if not clean_wave.is_cuda:
        raise ops._lib.DcsHipError('stft_batch: device tensors expected (the HIP path has no CPU fallback)')
    n_fft, hop = config.fft_size, config.hop_length
    L = clean_wave.shape[1]
    T = L // hop + 1
    w = _window_on(config, clean_wave.device)
    frames = ops.stft_frames(clean_wave.contiguous(), noisy_wave.contiguous(), w, T, hop)          # [3,B,T,n_fft]
    if n_fft == 512:                                     # hand-written 512-point real FFT (fft512.hip)
        spec = ops.rfft512(frames)                                                                  # [3,B,T,257,2]
    else:
        spec = torch.view_as_real(torch.fft.rfft(frames, dim=-1))                                  # [3,B,T,n_fft/2+1,2]
    scale = float(n_fft) ** -0.5 if config.normalise_stft else 1.0
    out = torch.view_as_complex(ops.stft_bins(spec.contiguous(), scale))                           # [3,B,F,T]
    clean, noise, noisy = out[0], out[1], out[2]
    return noise, noisy, clean
