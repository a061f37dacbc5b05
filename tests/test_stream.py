"""Incremental inference (SynthesizerTrn.stream / models.InferStream): chunks pushed as they arrive, the
concatenated pieces must equal one infer over the whole utterance."""
import numpy as np
import pytest
import torch

import cases
from cases import sw


class _SmoothNet:
    """CPU stand-in with a NON-causal receptive field of 5 frames on either side (bookkeeping test only)."""
    RECEPTIVE_FRAMES = 5
    inter_channels = 3

    class dec:
        hop = 4

    def infer(self, mel, ln, noise_scale=1, eps=None, max_len=None):
        x = mel[:, :1] + noise_scale * eps[:, :1]
        y = torch.nn.functional.conv1d(x, torch.ones(1, 1, 11) / 11, padding=5)
        return y.repeat_interleave(4, dim=2), None, None

    def _dev(self):
        return torch.device("cpu")


@pytest.mark.parametrize("chunks", [(1, 3, 20, 2, 40, 31), (97,), (4, 4, 4), (60, 1)])
def test_stream_bookkeeping_cpu(monkeypatch, chunks):
    from smart_vocoder_amd import models, _native
    monkeypatch.setattr(_native, "f32", lambda t: t.float().contiguous())       # the device check is not what is tested here
    net = _SmoothNet()
    T = sum(chunks)
    g = torch.Generator().manual_seed(3)
    mel = torch.randn(2, 80, T, generator=g); eps = torch.randn(2, 3, T, generator=g)
    ref = net.infer(mel, None, 0.5, eps)[0]
    s = models.InferStream(net, noise_scale=0.5)
    out, pos = [], 0
    for t in chunks:
        out.append(s.push(mel[:, :, pos:pos + t], eps[:, :, pos:pos + t])); pos += t
        assert s.end - s.start <= 2 * net.RECEPTIVE_FRAMES + max(chunks)       # only the halo is carried between calls
    out.append(s.flush())
    got = torch.cat(out, 2)
    assert got.shape == ref.shape and torch.allclose(got, ref, atol=1e-6)
    assert s.flush().shape[2] == 0


@pytest.mark.gpu
def test_stream_equals_one_shot_infer_gpu():
    from smart_vocoder_amd import models
    net = models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in cases.full_model_weights(skip_enc_q=True).items()}, strict=False)
    net = net.cuda().eval()
    B, T = 2, 700
    mel = torch.from_numpy(sw.synthetic_mel(515, B, T)).cuda(); eps = torch.from_numpy(sw.synthetic_eps(515, B, T)).cuda()
    ln = torch.full((B,), T, dtype=torch.int64).cuda()
    ref = net.infer(mel, ln, noise_scale=0.667, eps=eps)[0]
    s = net.stream(noise_scale=0.667)
    out, pos = [], 0
    for t in (37, 200, 129, 5, 329):
        out.append(s.push(mel[:, :, pos:pos + t], eps[:, :, pos:pos + t])); pos += t
    out.append(s.flush())
    got = torch.cat(out, 2)
    assert got.shape == ref.shape
    d = (got - ref).abs().max().item()
    print(f"stream vs one-shot: max |diff| {d:.2e}; pieces {[o.shape[2] // 256 for o in out]} frames")
    assert d <= 1e-5
