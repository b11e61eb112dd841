"""Evaluation metrics of the reference's validation / test steps (network_functions.py:152-166, test.py:18-27).

The reference scores every utterance on the CPU with two third-party packages, `pystoi.stoi` (pystoi==0.3.3,
requirements.txt:166) and `pypesq.pesq` (pypesq==1.2.4, requirements.txt:162).  Neither is in this image and neither
can be installed.

STOI.  `stoi()` below restates the PUBLISHED algorithm — C. H. Taal, R. C. Hendriks, R. Heusdens, J. Jensen, "An
Algorithm for Intelligibility Prediction of Time-Frequency Weighted Noisy Speech", IEEE TASLP 19(7), 2011 — with the
constants and processing order of pystoi 0.3.3 (10 kHz internal rate through its Octave-style polyphase resampler,
256-sample Hann frames at 50 % overlap, removal of frames more than 40 dB below the loudest clean frame, 512-point FFT,
15 one-third-octave bands from 150 Hz, 30-frame segments, clipping at -15 dB SDR, mean correlation).  It is host-side
numpy, as in the reference (which calls `.cpu().numpy()` per utterance), and is used when `pystoi` itself cannot be
imported.  PARITY UNPINNED: there is no pystoi here to compare with and the reference holds no STOI vectors; the
tests check the algorithm's defining properties only (tests/test_host_cpu.py).

PESQ.  ITU-T P.862 is ~2 k lines of reference C with psychoacoustic tables; it is not restated.  `pesq` stays the
imported package when present, else None (calc_metric then reports NaN, as in round 1)."""
import numpy as np

FS = 10000            # internal sample rate
N_FRAME = 256         # window length
NFFT = 512
NUMBAND = 15          # one-third-octave bands
MINFREQ = 150         # centre frequency of the first band
N_SEG = 30            # frames per intermediate-intelligibility segment (384 ms)
BETA = -15.0          # lower SDR bound
DYN_RANGE = 40        # speech dynamic range kept by the silent-frame removal
EPS = np.finfo('float').eps


def _resample_window_oct(p, q):
    """Octave / Matlab `resample` anti-aliasing window (Kaiser-windowed sinc), as pystoi.utils._resample_window_oct."""
    g = np.gcd(p, q)
    p, q = p // g, q // g
    log10_rejection = -3.0
    stopband_cutoff_f = 1.0 / (2 * max(p, q))
    roll_off_width = stopband_cutoff_f / 10
    rejection_db = -20 * log10_rejection
    L = np.ceil((rejection_db - 8) / (28.714 * roll_off_width))
    t = np.arange(-L, L + 1)
    ideal = 2 * p * stopband_cutoff_f * np.sinc(2 * stopband_cutoff_f * t)
    if 21 <= rejection_db <= 50:
        beta = 0.5842 * (rejection_db - 21) ** 0.4 + 0.07886 * (rejection_db - 21)
    elif rejection_db > 50:
        beta = 0.1102 * (rejection_db - 8.7)
    else:
        beta = 0.0
    return np.kaiser(int(2 * L + 1), beta) * ideal


def resample_oct(x, p, q):
    from scipy.signal import resample_poly
    h = _resample_window_oct(p, q)
    return resample_poly(x, p, q, window=h / np.sum(h))


def thirdoct(fs, nfft, num_bands, min_freq):
    """One-third-octave band matrix [num_bands, nfft/2 + 1] and the centre frequencies."""
    f = np.linspace(0, fs, nfft + 1)[:nfft // 2 + 1]
    k = np.arange(num_bands, dtype=float)
    cf = np.power(2.0 ** (1.0 / 3), k) * min_freq
    lo = min_freq * np.power(2.0, (2 * k - 1) / 6)
    hi = min_freq * np.power(2.0, (2 * k + 1) / 6)
    obm = np.zeros((num_bands, len(f)))
    for i in range(num_bands):
        a = int(np.argmin(np.square(f - lo[i])))
        b = int(np.argmin(np.square(f - hi[i])))
        obm[i, a:b] = 1
    return obm, cf


def _hann(n):
    return np.hanning(n + 2)[1:-1]


def _frames(x, framelen, hop):
    idx = range(0, len(x) - framelen, hop)
    return np.array([x[i:i + framelen] for i in idx]) if len(x) > framelen else np.zeros((0, framelen))


def remove_silent_frames(x, y, dyn_range, framelen, hop):
    """Drop the frames whose CLEAN energy is more than dyn_range dB below the loudest one; overlap-add the rest."""
    w = _hann(framelen)
    xf, yf = _frames(x, framelen, hop) * w, _frames(y, framelen, hop) * w
    if len(xf) == 0:
        return x[:0], y[:0]
    e = 20 * np.log10(np.linalg.norm(xf, axis=1) + EPS)
    keep = (np.max(e) - dyn_range - e) < 0
    xf, yf = xf[keep], yf[keep]
    n = (len(xf) - 1) * hop + framelen if len(xf) else 0
    xs, ys = np.zeros(n), np.zeros(n)
    for i in range(len(xf)):
        xs[i * hop:i * hop + framelen] += xf[i]
        ys[i * hop:i * hop + framelen] += yf[i]
    return xs, ys


def _stft(x, win, nfft, overlap):
    hop = win // overlap
    return np.array([np.fft.rfft(_hann(win) * x[i:i + win], n=nfft) for i in range(0, len(x) - win, hop)])


def stoi(x, y, fs_sig, extended=False):
    """Short-Time Objective Intelligibility of the processed signal y against the clean signal x (1-D, equal length).
    Same call as pystoi.stoi; extended=True (ESTOI) is not restated."""
    if extended:
        raise NotImplementedError('extended STOI is not restated; the reference calls stoi(clean, estimate, sr)')
    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    if x.shape != y.shape or x.ndim != 1:
        raise ValueError(f'stoi: x {x.shape} and y {y.shape} must be 1-D signals of equal length')
    if fs_sig != FS:
        x, y = resample_oct(x, FS, fs_sig), resample_oct(y, FS, fs_sig)
    x, y = remove_silent_frames(x, y, DYN_RANGE, N_FRAME, N_FRAME // 2)
    xs, ys = _stft(x, N_FRAME, NFFT, 2), _stft(y, N_FRAME, NFFT, 2)
    if xs.ndim != 2 or xs.shape[0] < N_SEG:
        return 1e-5                                        # pystoi: "Not enough STFT frames to compute intermediate intelligibility"
    obm, _ = thirdoct(FS, NFFT, NUMBAND, MINFREQ)
    xt = np.sqrt(obm @ (np.abs(xs.T) ** 2))                # [bands, frames]
    yt = np.sqrt(obm @ (np.abs(ys.T) ** 2))
    M = xt.shape[1]
    xseg = np.stack([xt[:, m - N_SEG:m] for m in range(N_SEG, M + 1)])      # [segments, bands, N_SEG]
    yseg = np.stack([yt[:, m - N_SEG:m] for m in range(N_SEG, M + 1)])
    norm = np.linalg.norm(xseg, axis=2, keepdims=True) / (np.linalg.norm(yseg, axis=2, keepdims=True) + EPS)
    yn = yseg * norm
    clip = 10 ** (-BETA / 20)
    yp = np.minimum(yn, xseg * (1 + clip))
    yp = yp - yp.mean(axis=2, keepdims=True)
    xz = xseg - xseg.mean(axis=2, keepdims=True)
    yp = yp / (np.linalg.norm(yp, axis=2, keepdims=True) + EPS)
    xz = xz / (np.linalg.norm(xz, axis=2, keepdims=True) + EPS)
    return float(np.sum(yp * xz) / (xseg.shape[0] * xseg.shape[1]))
