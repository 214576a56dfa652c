"""Oracle front-ends: waveform -> CMVN'ed log-mel / MFCC features.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Restates ``sidekit/nnet/preprocessor.py:212-285`` (MelSpecFrontEnd) and
``:61-124`` (MfccFrontEnd) plus ``sidekit/nnet/augmentation.py:49-74``
(PreEmphasis).  The spectrogram / mel / DCT arithmetic itself is third-party
``torchaudio==0.8.2`` (``install.sh:36``), absent from the reference tree:
its published algorithm is restated here -> **parity unpinned** at that
boundary.  ``stft_power_dft`` is an independent float64 direct DFT used to
cross-check the ``torch.stft`` based path.
"""
import math

import numpy
import torch


def pre_emphasis(x, coef=0.97):
    """augmentation.py:63-74 -- y[t] = x[t] - coef*x[t-1], reflect pad 1 on the left
    (so y[0] = x[0] - coef*x[1])."""
    assert x.dim() == 2, 'The number of dimensions of input tensor must be 2!'
    prev = torch.cat([x[:, 1:2], x[:, :-1]], dim=1)
    return x - coef * prev


def hann_window(win_length, dtype=torch.float32):
    """torch.hann_window(win_length) (periodic=True), preprocessor.py:222."""
    n = torch.arange(win_length, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * math.pi * n / win_length)).to(dtype)


def mel_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio 0.8.2 functional.create_fb_matrix(norm=None), HTK mel scale.
    Returns (n_freqs, n_mels) float32 -- the `mel_scale.fb` buffer."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def dct_matrix(n_mfcc, n_mels):
    """torchaudio 0.8.2 functional.create_dct(norm='ortho') -> (n_mels, n_mfcc)."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t().contiguous()


def stft_power(x, n_fft, hop, win_length, window=None):
    """torchaudio Spectrogram(power=2): torch.stft(center=True, reflect, onesided) -> |.|^2.
    x (B, L) -> (B, n_fft//2+1, 1+L//hop)."""
    if window is None:
        window = hann_window(win_length, x.dtype)
    spec = torch.stft(x, n_fft, hop, win_length, window.to(x.dtype), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    return spec.real ** 2 + spec.imag ** 2


def stft_power_dft(x, n_fft, hop, win_length):
    """Independent float64 direct-DFT statement of the same spectrogram (numpy)."""
    x = numpy.asarray(x, dtype=numpy.float64)
    B, L = x.shape
    pad = n_fft // 2
    xp = numpy.pad(x, ((0, 0), (pad, pad)), mode="reflect")
    T = 1 + L // hop
    left = (n_fft - win_length) // 2
    n = numpy.arange(win_length)
    w = 0.5 - 0.5 * numpy.cos(2.0 * numpy.pi * n / win_length)
    k = numpy.arange(n_fft // 2 + 1)
    ang = -2.0 * numpy.pi * numpy.outer(k, n + left) / n_fft
    basis = numpy.cos(ang) + 1j * numpy.sin(ang)
    out = numpy.empty((B, n_fft // 2 + 1, T))
    for t in range(T):
        fr = xp[:, t * hop + left: t * hop + left + win_length] * w
        out[:, :, t] = numpy.abs(fr @ basis.T) ** 2
    return out


def cmvn(x, eps=1e-5):
    """torch.nn.InstanceNorm1d (no affine): per (b, c) over T, biased variance.
    preprocessor.py:263,281 / :111,123."""
    mean = x.mean(dim=2, keepdim=True)
    var = x.var(dim=2, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


MELSPEC_CFG = dict(pre_emphasis=0.97, sample_rate=16000, n_fft=1024, f_min=90, f_max=7600,
                   win_length=400, hop_length=160, n_mels=80)
MFCC_CFG = dict(pre_emphasis=0.97, sample_rate=16000, n_fft=2048, f_min=133.333, f_max=6855.4976,
                win_length=1024, hop_length=512, n_mels=100, n_mfcc=80)


def melspec_frontend(x, fb=None, window=None, cfg=MELSPEC_CFG):
    """MelSpecFrontEnd.forward(is_eval=True), preprocessor.py:267-285.
    x (L,) or (B, L) -> (B, 80, 1+L//160)."""
    if x.dim() == 1:
        x = x.unsqueeze(0)
    out = pre_emphasis(x, cfg["pre_emphasis"])
    spec = stft_power(out, cfg["n_fft"], cfg["hop_length"], cfg["win_length"], window)
    if fb is None:
        fb = mel_filterbank(cfg["n_fft"] // 2 + 1, cfg["f_min"], cfg["f_max"], cfg["n_mels"], cfg["sample_rate"])
    mel = torch.matmul(spec.transpose(1, 2), fb.to(spec.dtype)).transpose(1, 2)
    out = torch.log(mel + 1e-6)
    return cmvn(out)


def mfcc_frontend(x, fb=None, dct=None, window=None, cfg=MFCC_CFG):
    """MfccFrontEnd.forward, preprocessor.py:113-124 (torchaudio MFCC, log_mels=True).
    x (B, L) -> (B, 80, 1+L//512)."""
    if x.dim() == 1:
        x = x.unsqueeze(0)
    out = pre_emphasis(x, cfg["pre_emphasis"])
    spec = stft_power(out, cfg["n_fft"], cfg["hop_length"], cfg["win_length"], window)
    if fb is None:
        fb = mel_filterbank(cfg["n_fft"] // 2 + 1, cfg["f_min"], cfg["f_max"], cfg["n_mels"], cfg["sample_rate"])
    if dct is None:
        dct = dct_matrix(cfg["n_mfcc"], cfg["n_mels"])
    mel = torch.matmul(spec.transpose(1, 2), fb.to(spec.dtype)).transpose(1, 2)
    logmel = torch.log(mel + 1e-6)
    mfcc = torch.matmul(logmel.transpose(1, 2), dct.to(spec.dtype)).transpose(1, 2)
    return cmvn(mfcc)


def resample_sinc(waveform, orig_freq, new_freq, lowpass_filter_width=6):
    """``torchaudio.transforms.Resample(orig_freq, new_freq)`` as ``sidekit/bin/extract_xvectors.py:141-143`` applies it.  PARITY
    UNPINNED: torchaudio is pinned at 0.8.2 (``install.sh:36``) and not vendored; this restates its published algorithm
    (``torchaudio.compliance.kaldi.resample_waveform`` of that release: windowed-sinc interpolation, roll-off 0.99, evaluated as a
    strided ``conv1d`` with one filter per output phase) in float64 -- the noise floor of the comparison is then the float32
    arithmetic of the kernel under test, not this restatement's.  ``waveform``: ``(n,)`` or ``(B, n)``."""
    import math
    x = torch.as_tensor(waveform, dtype=torch.float64)
    squeeze = x.dim() == 1
    if squeeze:
        x = x[None]
    g = math.gcd(int(orig_freq), int(new_freq))
    O, N = int(orig_freq) // g, int(new_freq) // g
    base = min(O, N) * 0.99
    width = math.ceil(lowpass_filter_width * O / base)
    idx = torch.arange(-width, width + O, dtype=torch.float64)
    kernels = []
    for i in range(N):
        t = ((-i / N + idx / O) * base).clamp(-lowpass_filter_width, lowpass_filter_width) * math.pi
        window = torch.cos(t / lowpass_filter_width / 2) ** 2
        kernels.append(torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window)
    kernel = torch.stack(kernels).view(N, 1, -1) * (base / O)
    n = x.shape[1]
    padded = torch.nn.functional.pad(x, (width, width + O))
    out = torch.nn.functional.conv1d(padded[:, None], kernel, stride=O).transpose(1, 2).reshape(x.shape[0], -1)
    out = out[..., :int(math.ceil(N * n / O))]
    return out[0] if squeeze else out
