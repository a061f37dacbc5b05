"""Config and checkpoint I/O with the reference's names and on-disk formats (reference utils.py).

In scope: HParams, get_hparams_from_file/_from_dir, load_checkpoint /
save_checkpoint ({'model','iteration','optimizer','learning_rate'}), latest_checkpoint_path,
load_wav_to_torch.  TensorBoard/matplotlib/logging helpers of the reference are training-side and absent.
"""
import glob
import json
import logging
import os
import sys

import numpy as np
import torch

logging.basicConfig(stream=sys.stdout, level=logging.INFO)
logger = logging.getLogger("smart_vocoder_amd")


class HParams():
    """Recursive attribute dict (reference utils.py:229-258): attr access, [] access and ** via keys()."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            if type(v) == dict:
                v = HParams(**v)
            self[k] = v

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def __len__(self):
        return len(self.__dict__)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        return setattr(self, key, value)

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        return self.__dict__.__repr__()


def get_hparams_from_file(config_path):
    """reference utils.py:185-191"""
    with open(config_path, "r") as f:
        config = json.loads(f.read())
    return HParams(**config)


def get_hparams_from_dir(model_dir):
    """reference utils.py:174-182"""
    hparams = get_hparams_from_file(os.path.join(model_dir, "config.json"))
    hparams.model_dir = model_dir
    return hparams


def load_checkpoint(checkpoint_path, model, optimizer=None):
    """reference utils.py:18-43: tolerant load — keys missing from the file keep the model's value."""
    assert os.path.isfile(checkpoint_path)
    checkpoint_dict = torch.load(checkpoint_path, map_location="cpu")
    iteration = checkpoint_dict["iteration"]
    learning_rate = checkpoint_dict["learning_rate"]
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint_dict["optimizer"])
    saved_state_dict = checkpoint_dict["model"]
    target = model.module if hasattr(model, "module") else model
    new_state_dict = {}
    for k, v in target.state_dict().items():
        if k in saved_state_dict:
            new_state_dict[k] = saved_state_dict[k]
        else:
            logger.info("%s is not in the checkpoint" % k)
            new_state_dict[k] = v
    target.load_state_dict(new_state_dict)
    logger.info("Loaded checkpoint '{}' (iteration {})".format(checkpoint_path, iteration))
    return model, optimizer, learning_rate, iteration


def save_checkpoint(model, optimizer, learning_rate, iteration, checkpoint_path):
    """reference utils.py:46-56"""
    logger.info("Saving model and optimizer state at iteration {} to {}".format(iteration, checkpoint_path))
    target = model.module if hasattr(model, "module") else model
    torch.save({"model": target.state_dict(), "iteration": iteration,
                "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "learning_rate": learning_rate}, checkpoint_path)


def remove_all_weight_norm(model):
    """Fold every (weight_g, weight_v) pair of `model` into a plain `weight` on the device: the whole-model version of
    Generator.remove_weight_norm / WN.remove_weight_norm / ResBlock*.remove_weight_norm (reference models.py:162-167,
    modules.py:178-184, 225-229, 254-256).  The model must live on the GPU (the fold is a HIP kernel)."""
    try:
        from .modules import _fold_in_place
    except ImportError:      # imported as a top-level module (drop-in mode)
        from modules import _fold_in_place
    n = 0
    for m in model.modules():
        if "weight_v" in m._parameters and "weight_g" in m._parameters:
            _fold_in_place(m)
            n += 1
    return n


def folded_state_dict(model):
    """state_dict of `model` as it would be after remove_all_weight_norm, computed without touching the model."""
    try:
        from .modules import _fold_tensor
    except ImportError:
        from modules import _fold_tensor
    out, folded = {}, set()
    for name, m in model.named_modules():
        if "weight_v" in m._parameters and "weight_g" in m._parameters:
            p = name + "." if name else ""
            out[p + "weight"] = _fold_tensor(m.weight_v, m.weight_g)
            folded.update((p + "weight_v", p + "weight_g"))
    for k, v in model.state_dict().items():
        if k not in folded:
            out[k] = v
    return out


def export_folded(model, path, iteration=0):
    """Folded-weight cache file: what the reference obtains by remove_weight_norm() + torch.save.
    The source model is left untouched."""
    target = model.module if hasattr(model, "module") else model
    torch.save({"model_folded": {k: v.detach().cpu() for k, v in folded_state_dict(target).items()}, "iteration": iteration,
                "format": "svoc-folded-1"}, path)
    return path


def load_folded(path, model):
    """Load a file written by export_folded into `model` (on the GPU): the model's weight-norm pairs are folded first so
    that the key sets agree; keys missing from the file keep the model's value, as in load_checkpoint."""
    d = torch.load(path, map_location="cpu")
    if d.get("format") != "svoc-folded-1":
        raise ValueError(f"{path} is not a folded-weight file")
    target = model.module if hasattr(model, "module") else model
    remove_all_weight_norm(target)
    saved = d["model_folded"]
    new_sd = {k: (saved[k] if k in saved else v) for k, v in target.state_dict().items()}
    target.load_state_dict(new_sd)
    return model, d.get("iteration", 0)


def latest_checkpoint_path(dir_path, regex="G_*.pth"):
    """reference utils.py:70-75: newest by the digits in the file name"""
    f_list = glob.glob(os.path.join(dir_path, regex))
    f_list.sort(key=lambda f: int("".join(filter(str.isdigit, f))))
    return f_list[-1]


def load_wav_to_torch(full_path):
    """reference utils.py:133-135"""
    from scipy.io.wavfile import read
    sampling_rate, data = read(full_path)
    return torch.FloatTensor(data.astype(np.float32)), sampling_rate


def load_filepaths(filename):
    """One stripped line per entry of a file list (reference utils.py:138-141; used to enumerate the wavs / mels to synthesise)."""
    with open(filename, encoding="utf-8") as f:
        return [line.strip() for line in f]


def save_wav(path, sampling_rate, audio):
    """The notebook's (commented-out) last line, ``write('./generated_files/'+f_name, 22050, audio_)`` (inference.ipynb
    cell 4), for tensors straight out of ``infer``: `audio` is [samples], [1, samples] or [1, 1, samples] (any device)
    or a numpy array; written as 32-bit float PCM exactly as scipy writes the notebook's float32 array."""
    from scipy.io.wavfile import write
    if isinstance(audio, torch.Tensor):
        on_gpu = audio.is_cuda
        audio = audio.detach().float().cpu().numpy()
        if on_gpu:
            # the copy synchronised the stream: if a persistent WN launch of the call that produced this audio gave up a wait, say so HERE, for
            # the audio that is affected (it is NaN where that happened), not at the next call (include/svoc.h svoc_check_async_error)
            from . import _native
            _native.check_async_error()
    audio = np.asarray(audio, dtype=np.float32)
    while audio.ndim > 1 and audio.shape[0] == 1:
        audio = audio[0]
    if audio.ndim != 1:
        raise ValueError(f"expected one utterance, got shape {audio.shape}")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    write(path, int(sampling_rate), audio)
    return path


def infer_to_wavs(net_g, mel, lengths, out_paths, sampling_rate=22050, chunk_frames=None, **infer_kw):
    """Long-form convenience: run ``infer`` (or ``infer_chunked`` when `chunk_frames` is given) and write
    every utterance, cut to its own length, to out_paths[b]."""
    hop = net_g.dec.hop
    if chunk_frames:
        o = net_g.infer_chunked(mel, lengths, chunk_frames=chunk_frames, **infer_kw)
    else:
        o = net_g.infer(mel, lengths, **infer_kw)[0]
    ln = lengths.cpu().tolist()
    for b, path in enumerate(out_paths):
        save_wav(path, sampling_rate, o[b, 0, :int(ln[b]) * hop])
    return o
