"""CPU oracle for the mel front-end (reference mel_processing.py:51-112).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference's mel_processing.py cannot be imported in the build container (it needs librosa,
which is absent, and calls torch.stft without return_complex, which torch >= 2.0 rejects), so no golden vector could
be taken from the reference for this row.  This restatement follows the reference line by line with today's
torch.stft signature, and restates librosa 0.8.0 ``filters.mel`` (htk=False, norm='slaney') in numpy from its
published algorithm; the HIP path (an independent C++ restatement of the same filterbank) is checked against it.
"""
import numpy as np
import torch


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa 0.8.0 filters.mel defaults (Slaney scale, 'slaney' area normalisation), float32 [n_mels, 1+n_fft/2]."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def spectrogram(y, n_fft, hop, win, center=False):
    """mel_processing.py:51-70 (center is forwarded to torch.stft, :66-67)"""
    pad = int((n_fft - hop) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    s = torch.stft(y, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win, dtype=y.dtype), center=center,
                   pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    s = torch.view_as_real(s)
    return torch.sqrt(s.pow(2).sum(-1) + 1e-6)


def spec_to_mel(spec, n_fft, n_mels, sr, fmin, fmax):
    basis = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin or 0.0, fmax))
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))


def mel_spectrogram(y, n_fft, n_mels, sr, hop, win, fmin, fmax, center=False):
    return spec_to_mel(spectrogram(y, n_fft, hop, win, center), n_fft, n_mels, sr, fmin, fmax)
