"""Mel front-end with the reference's function names (reference mel_processing.py), computed on the GPU by
libsvoc_hip.so: the windowed DFT and the mel projection both run on the MFMA convolution kernel
(csrc/melspec.hip).  The mel basis restates librosa 0.8.0 ``filters.mel`` (Slaney scale and normalisation), the
third-party call at reference mel_processing.py:77 — librosa is not needed at run time.
"""
import torch

try:
    from . import _native as N
except ImportError:
    import _native as N

MAX_WAV_VALUE = 32768.0
_handles = {}


def dynamic_range_compression_torch(x, C=1, clip_val=1e-5):
    """reference mel_processing.py:19-25 (elementwise, kept in torch for API completeness; the fused GPU path
    applies it in the mel projection's epilogue)"""
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression_torch(x, C=1):
    return torch.exp(x) / C


def spectral_normalize_torch(magnitudes):
    return dynamic_range_compression_torch(magnitudes)


def spectral_de_normalize_torch(magnitudes):
    return dynamic_range_decompression_torch(magnitudes)


def _handle(n_fft, hop_size, win_size, num_mels, sampling_rate, fmin, fmax, device):
    key = (n_fft, hop_size, win_size, num_mels, sampling_rate, float(fmin or 0.0), float(fmax or 0.0), str(device))
    h = _handles.get(key)
    if h is None:
        h = N.Handle("svoc_melspec_destroy")
        with torch.cuda.device(device):
            N.check(N.lib().svoc_melspec_create(h.out(), n_fft, hop_size, win_size, num_mels, sampling_rate,
                                                float(fmin or 0.0), float(fmax or 0.0)))
        _handles[key] = h
    return h


def _to_gpu(t):
    """The reference notebook (inference.ipynb cell 4) calls these functions on CPU tensors and moves the mel to the GPU
    afterwards.  The DFT / mel kernels run only on the GPU: host tensors are moved to the current device for the call and
    the result is returned on the caller's device, so the notebook cell runs unchanged."""
    N.require_gpu()
    src = t.device
    if not t.is_cuda:
        t = t.cuda()
    return N.f32(t), src


def _check_range(y):
    if torch.min(y) < -1.:
        print('min value is ', torch.min(y))
    if torch.max(y) > 1.:
        print('max value is ', torch.max(y))


def spectrogram_torch(y, n_fft, sampling_rate, hop_size, win_size, center=False):
    """y [B, samples] in [-1, 1] -> magnitude spectrogram [B, n_fft/2+1, frames] (reference mel_processing.py:51-70).
    `center` is torch.stft's flag, forwarded as the reference does (mel_processing.py:66-67): True frames the signal after a
    second reflect padding of n_fft/2."""
    y, src = _to_gpu(y)
    _check_range(y)
    B, Lw = y.shape
    h = _handle(n_fft, hop_size, win_size, 80, sampling_rate, 0.0, None, y.device)
    c = 1 if center else 0
    F = N.lib().svoc_melspec_frames_center(h.h, Lw, c)
    spec = torch.empty(B, n_fft // 2 + 1, F, dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        N.check(N.lib().svoc_melspec_spectrogram_center(h.h, N.stream_ptr(y.device), N.ptr(y), B, Lw, c, N.ptr(spec)))
    return spec.to(src)


def spec_to_mel_torch(spec, n_fft, num_mels, sampling_rate, fmin, fmax):
    """spec [B, n_fft/2+1, F] -> log-mel [B, num_mels, F] (reference mel_processing.py:73-82)."""
    spec, src = _to_gpu(spec)
    B, nb, F = spec.shape
    if nb != n_fft // 2 + 1:
        raise ValueError(f"expected {n_fft // 2 + 1} frequency bins, got {nb}")
    h = _handle(n_fft, n_fft // 4, n_fft, num_mels, sampling_rate, fmin, fmax, spec.device)
    mel = torch.empty(B, num_mels, F, dtype=torch.float32, device=spec.device)
    with torch.cuda.device(spec.device):
        N.check(N.lib().svoc_melspec_mel(h.h, N.stream_ptr(spec.device), N.ptr(spec), B, F, N.ptr(mel)))
    return mel.to(src)


def mel_spectrogram_torch(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """reference mel_processing.py:85-112"""
    y, src = _to_gpu(y)
    _check_range(y)
    B, Lw = y.shape
    h = _handle(n_fft, hop_size, win_size, num_mels, sampling_rate, fmin, fmax, y.device)
    c = 1 if center else 0
    F = N.lib().svoc_melspec_frames_center(h.h, Lw, c)
    spec = torch.empty(B, n_fft // 2 + 1, F, dtype=torch.float32, device=y.device)
    mel = torch.empty(B, num_mels, F, dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        N.check(N.lib().svoc_melspec_spectrogram_center(h.h, N.stream_ptr(y.device), N.ptr(y), B, Lw, c, N.ptr(spec)))
        N.check(N.lib().svoc_melspec_mel(h.h, N.stream_ptr(y.device), N.ptr(spec), B, F, N.ptr(mel)))
    return mel.to(src)


def mel_filterbank(sampling_rate, n_fft, num_mels, fmin=0.0, fmax=None):
    """The [num_mels, n_fft/2+1] basis the library uses (host tensor), for inspection."""
    import ctypes
    out = torch.empty(num_mels, n_fft // 2 + 1, dtype=torch.float32)
    N.check(N.lib().svoc_mel_filterbank(sampling_rate, n_fft, num_mels, float(fmin or 0.0), float(fmax or 0.0),
                                        ctypes.c_void_p(out.data_ptr())))
    return out
