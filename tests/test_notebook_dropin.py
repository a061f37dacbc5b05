"""The reference's inference notebook as the drop-in test (/root/reference/inference.ipynb cells 0-4).

The notebook runs from the repository root of the reference with TOP-LEVEL imports (`import commons`, `import utils`,
`from models import SynthesizerTrn`, ...) and reads `./configs/iitp_base.json`.  A user switches by pointing the
working directory / sys.path at `smart-vocoder_amd/`; these tests do exactly that in a fresh interpreter (top-level
module names would collide with the test process's own imports otherwise).  matplotlib / IPython (display only) are
not installed in this image and are left out of cell 0; everything else is the notebook's statement order.
"""
import json
import os
import subprocess
import sys

import pytest

import cases

PKG_DIR = os.path.join(cases.ROOT, "smart-vocoder_amd")

CELLS_0_TO_3 = r'''
import os
import json
import math
import torch
from torch import nn
from torch.nn import functional as F
from torch.utils.data import DataLoader
import time
import commons
from utils import load_wav_to_torch
import utils
from data_utils import AudioSpecLoader, AudioSpecCollate
from models import SynthesizerTrn

from glob import glob
from scipy.io.wavfile import write
from mel_processing import mel_spectrogram_torch, spec_to_mel_torch, spectrogram_torch

hps = utils.get_hparams_from_file("./configs/iitp_base.json")

net_g = SynthesizerTrn(
    hps.data.filter_length // 2 + 1,
    hps.train.segment_size // hps.data.hop_length,
    n_speakers=hps.data.n_speakers,
    **hps.model)@CUDA@
_ = net_g.eval()
'''

REPORT = r'''
import json as _j
sd = net_g.state_dict()
print("REPORT " + _j.dumps({"keys": len(sd), "layout": [[k, list(v.shape)] for k, v in sd.items()], "model": dict(**hps.model),
                            "spec_channels": hps.data.filter_length // 2 + 1, "segment": hps.train.segment_size // hps.data.hop_length,
                            "n_speakers": hps.data.n_speakers, "file": os.path.abspath(utils.__file__)}))
'''


def _run(code, timeout=600):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)                      # nothing but the working directory may resolve the top-level names
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, '')\n" + code], cwd=PKG_DIR, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def _report(out):
    line = [l for l in out.splitlines() if l.startswith("REPORT ")][-1]
    return json.loads(line[len("REPORT "):])


def test_shipped_configs_are_the_reference_configurations():
    """configs/iitp_base.json and iitp_base_ms.json ship as data; the model section is the one every
    test builds from (cases.IITP_MODEL), the data section carries the mel front-end's parameters."""
    for name, tag in (("iitp_base", "ss"), ("iitp_base_ms", "ms")):
        cfg = json.load(open(os.path.join(PKG_DIR, "configs", name + ".json")))
        assert cfg["model"] == cases.IITP_MODEL
        assert cfg["data"]["training_files"] == f"filelists/train_files_{tag}.txt"
        d = cfg["data"]
        assert (d["sampling_rate"], d["filter_length"], d["hop_length"], d["win_length"], d["n_mel_channels"], d["mel_fmin"], d["mel_fmax"],
                d["n_speakers"]) == (22050, 1024, 256, 1024, 80, 0.0, None, 109)
        assert cfg["train"]["segment_size"] == 8192


def test_notebook_cells_0_to_3_top_level_imports_cpu():
    """Cells 0, 2, 3 without `.cuda()` (no GPU in the build container): imports resolve to smart-vocoder_amd/, the shipped
    JSON feeds the cell-3 constructor, and the module holds the reference's 659-tensor layout."""
    out = _run(CELLS_0_TO_3.replace("@CUDA@", "") + REPORT)
    rep = _report(out)
    assert rep["file"].startswith(PKG_DIR)
    assert rep["keys"] == 659
    assert rep["layout"] == json.load(open(os.path.join(cases.GOLDEN_DIR, "state_dict_layout.json")))
    assert rep["model"] == cases.IITP_MODEL and (rep["spec_channels"], rep["segment"], rep["n_speakers"]) == (513, 32, 109)


CELL_3_LOAD_AND_4 = r'''
import numpy as np
sys.path.insert(0, os.path.join(@ROOT@, "tests")); sys.path.insert(0, @ROOT@)
import cases as _cases
# a G_*.pth in the reference's checkpoint format (utils.py:46-56), written from the synthetic weights
_sd = {k: torch.from_numpy(v) for k, v in _cases.full_model_weights(skip_enc_q=False).items()}
os.makedirs(@TMP@ + "/checkpoint_dir", exist_ok=True); os.makedirs(@TMP@ + "/test_wav_dir/spk", exist_ok=True)
torch.save({"model": _sd, "iteration": 12, "optimizer": None, "learning_rate": 2e-4}, @TMP@ + "/checkpoint_dir/G_12.pth")
_t = np.arange(20000) / 22050.0
write(@TMP@ + "/test_wav_dir/spk/a.wav", 22050, (0.4 * np.sin(2 * np.pi * 220 * _t) * 32767).astype(np.int16))

_ = utils.load_checkpoint(utils.latest_checkpoint_path(@TMP@ + "/checkpoint_dir", "G_*.pth"), net_g, None)

filenames=glob(@TMP@ + '/test_wav_dir/**/*.wav',recursive=True)
for filename in filenames:
    audio, sampling_rate = load_wav_to_torch(filename)
    audio_norm = audio / 32768.0
    audio_norm = audio_norm.unsqueeze(0)
    spec = spectrogram_torch(audio_norm, 1024,
            22050, 256, 1024,
            center=False)
    f_name=os.path.basename(filename)
    mel = spec_to_mel_torch(
        spec,
        hps.data.filter_length,
        hps.data.n_mel_channels,
        hps.data.sampling_rate,
        hps.data.mel_fmin,
        hps.data.mel_fmax)
    with torch.no_grad():
        mel = mel.cuda()
        spec_lengths = torch.LongTensor([mel.size(2)]).cuda()
        sid = None
        start=time.time()
        audio_ = net_g.infer(mel, spec_lengths, sid=sid, noise_scale=.667, noise_scale_w=0.8, length_scale=1)[0][0,0].data.cpu().float().numpy()
        end = time.time()
        write(@TMP@ + '/' + f_name, 22050, audio_)
print("REPORT " + json.dumps({"n": len(filenames), "samples": int(audio_.shape[0]), "frames": int(mel.size(2)), "finite": bool(np.isfinite(audio_).all()),
                              "peak": float(np.abs(audio_).max())}))
'''


@pytest.mark.gpu
def test_notebook_cells_0_to_4_verbatim_gpu(tmp_path):
    """Cells 0-4 with `.cuda()` on the device: constructor from the shipped JSON, checkpoint through utils.load_checkpoint,
    wav -> spectrogram_torch -> spec_to_mel_torch -> infer -> scipy write."""
    code = (CELLS_0_TO_3.replace("@CUDA@", ".cuda()") + CELL_3_LOAD_AND_4).replace("@ROOT@", repr(cases.ROOT)).replace("@TMP@", repr(str(tmp_path)))
    rep = _report(_run(code))
    assert rep["n"] == 1 and rep["finite"] and rep["samples"] == rep["frames"] * 256 and 0 < rep["peak"] <= 1.0
    assert os.path.getsize(str(tmp_path / "a.wav")) > rep["samples"] * 4
