"""Front-end constants of the x-vector extractors (host side).

Mirrors the constructor surface of ``sidekit/nnet/preprocessor.py:61-111`` (MfccFrontEnd) and
``:212-265`` (MelSpecFrontEnd): the objects hold the same hyper-parameters and the three
torchaudio buffers that appear in reference checkpoints (``spectrogram.window``,
``mel_scale.fb``, ``dct_mat``).  The arithmetic itself (pre-emphasis, STFT-as-DFT, mel / DCT
projection, log, CMVN) runs in the HIP library -- see ``csrc/gemm.hip`` (A_FRAMES / A_POWER
loaders) and ``csrc/pool.hip`` (cmvn_kernel).
"""
import math

import torch


def hann_window(win_length):
    """``torch.hann_window(win_length)`` (periodic), as torchaudio's Spectrogram registers it."""
    return torch.hann_window(win_length, periodic=True, dtype=torch.float32)


def mel_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """Triangular HTK mel filterbank without area normalisation: what torchaudio 0.8.2
    ``MelScale`` stores in ``mel_scale.fb`` for (f_min, f_max, n_mels, sample_rate).  (n_freqs, n_mels)."""
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    lo = 2595.0 * math.log10(1.0 + f_min / 700.0)
    hi = 2595.0 * math.log10(1.0 + f_max / 700.0)
    edges = 700.0 * (10 ** (torch.linspace(lo, hi, n_mels + 2) / 2595.0) - 1.0)
    width = edges[1:] - edges[:-1]
    dist = edges[None, :] - freqs[:, None]
    falling = -dist[:, :-2] / width[:-1]
    rising = dist[:, 2:] / width[1:]
    return torch.clamp(torch.minimum(falling, rising), min=0.0)


def dct_matrix(n_mfcc, n_mels):
    """Orthonormal DCT-II matrix, torchaudio ``create_dct(norm='ortho')`` layout (n_mels, n_mfcc)."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc))[:, None]
    m = torch.cos(math.pi / n_mels * (n + 0.5) * k)
    m[0] *= 1.0 / math.sqrt(2.0)
    m *= math.sqrt(2.0 / n_mels)
    return m.t().contiguous()


class _FrontEnd:
    """Callable view of the extractor's front-end: ``model.preprocessor(x, is_eval=True)``."""

    def __init__(self, owner):
        self._owner = owner

    def __call__(self, x, is_eval=True):
        if not is_eval:
            raise NotImplementedError("training-time masking is out of scope: call with is_eval=True")
        return self._owner.features(x)

    forward = __call__


class MelSpecFrontEnd(_FrontEnd):
    """Hyper-parameters of ``sidekit/nnet/preprocessor.py:216-226``."""

    def __init__(self, owner=None, pre_emphasis=0.97, sample_rate=16000, n_fft=1024, f_min=90, f_max=7600, win_length=400,
                 hop_length=160, power=2.0, n_mels=80):
        super().__init__(owner)
        self.pre_emphasis, self.sample_rate, self.n_fft, self.f_min, self.f_max = pre_emphasis, sample_rate, n_fft, f_min, f_max
        self.win_length, self.hop_length, self.power, self.n_mels = win_length, hop_length, power, n_mels

    def buffers(self):
        return {
            "PreEmphasis.flipped_filter": torch.tensor([[[-self.pre_emphasis, 1.0]]], dtype=torch.float32),
            "MelSpec.spectrogram.window": hann_window(self.win_length),
            "MelSpec.mel_scale.fb": mel_filterbank(self.n_fft // 2 + 1, self.f_min, self.f_max, self.n_mels, self.sample_rate),
        }


class MfccFrontEnd(_FrontEnd):
    """Hyper-parameters of ``sidekit/nnet/preprocessor.py:65-76``."""

    def __init__(self, owner=None, pre_emphasis=0.97, sample_rate=16000, n_fft=2048, f_min=133.333, f_max=6855.4976,
                 win_length=1024, hop_length=512, power=2.0, n_mels=100, n_mfcc=80):
        super().__init__(owner)
        self.pre_emphasis, self.sample_rate, self.n_fft, self.f_min, self.f_max = pre_emphasis, sample_rate, n_fft, f_min, f_max
        self.win_length, self.hop_length, self.power, self.n_mels, self.n_mfcc = win_length, hop_length, power, n_mels, n_mfcc

    def buffers(self):
        return {
            "PreEmphasis.flipped_filter": torch.tensor([[[-self.pre_emphasis, 1.0]]], dtype=torch.float32),
            "MFCC.dct_mat": dct_matrix(self.n_mfcc, self.n_mels),
            "MFCC.MelSpectrogram.spectrogram.window": hann_window(self.win_length),
            "MFCC.MelSpectrogram.mel_scale.fb": mel_filterbank(self.n_fft // 2 + 1, self.f_min, self.f_max, self.n_mels,
                                                               self.sample_rate),
        }
