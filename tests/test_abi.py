"""CPU checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
include/svoc.h declares, and the host modules mirror the reference's state_dict layout."""
import json
import os
import re
import ctypes

import pytest
import torch

import cases


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from smart_vocoder_amd import _native
    return _native


def test_header_symbols_exported(built):
    hdr = open(os.path.join(cases.ROOT, "include", "svoc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(svoc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = ctypes.CDLL(built.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(built.SIGNATURES), declared ^ set(built.SIGNATURES)


def test_library_identity(built):
    lib = built.lib()
    assert lib.svoc_abi_version() == built.ABI_VERSION
    assert lib.svoc_build_arch() == b"gfx950"


def test_errors_do_not_throw(built):
    lib = built.lib()
    # null arguments are reported through the return code and svoc_last_error, never by crashing
    rc = lib.svoc_sequence_mask(None, None, None, 1, 1)
    assert rc == -1 and b"svoc_sequence_mask" in lib.svoc_last_error()
    h = ctypes.c_void_p(0)
    rc = lib.svoc_wn_create(ctypes.byref(h), 192, 5, 1, 2, 0, None, 0, b"")
    assert rc == -1 and not h.value


def test_state_dict_layout_matches_reference():
    from smart_vocoder_amd import models
    net = models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
    got = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    want = json.load(open(os.path.join(cases.GOLDEN_DIR, "state_dict_layout.json")))
    assert len(got) == 659
    assert got == want
    assert not any(k.startswith("emb_g") for k, _ in got)


def test_no_silent_cpu_fallback():
    from smart_vocoder_amd import models, modules, _native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rb = modules.ResBlock1(32, 3)
    with pytest.raises(_native.SvocError):
        rb(torch.zeros(1, 32, 16))
    net = models.Generator(32, "1", [3], [[1, 3, 5]], [2], 32, [4])
    with pytest.raises(_native.SvocError):
        net(torch.zeros(1, 32, 8))


def test_module_level_state_dicts():
    from smart_vocoder_amd import models, modules
    c = cases.WN_CASES["wn_h64_k3_dr2_n4_g"]
    m = modules.WN(c["H"], c["k"], c["dr"], c["n"], gin_channels=c["gin"])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == cases.wn_shapes(c["H"], c["k"], c["n"], c["gin"])
    assert {k: tuple(v.shape) for k, v in modules.ResBlock1(64, 7).state_dict().items()} == cases.resblock1_shapes(64, 7)
    assert {k: tuple(v.shape) for k, v in modules.ResBlock2(64, 3).state_dict().items()} == cases.resblock2_shapes(64, 3)
    assert {k: tuple(v.shape) for k, v in modules.ResidualCouplingLayer(64, 96, 5, 1, 2, gin_channels=32).state_dict().items()} \
        == cases.coupling_shapes(64, 96, 5, 2, 32, False)
    assert {k: tuple(v.shape) for k, v in modules.DDSConv(64, 5, 2).state_dict().items()} == cases.dds_shapes(64, 5, 2)
    assert {k: tuple(v.shape) for k, v in modules.ConvFlow(2, 192, 3, 3).state_dict().items()} == cases.convflow_shapes(2, 192, 3, 3)
    g = cases.GENERATOR_CASES["gen_small_g"]
    gm = models.Generator(g["initial_channel"], g["resblock"], g["rks"], g["rds"], g["ur"], g["uic"], g["uks"], gin_channels=g["gin"])
    assert {k: tuple(v.shape) for k, v in gm.state_dict().items()} == cases.generator_shapes(g)


def test_hparams_and_checkpoint_roundtrip(tmp_path):
    from smart_vocoder_amd import utils, modules
    cfg = tmp_path / "c.json"
    cfg.write_text(json.dumps({"train": {"segment_size": 8192}, "data": {"hop_length": 256}, "model": {"a": [1, 2]}}))
    hps = utils.get_hparams_from_file(str(cfg))
    assert hps.train.segment_size // hps.data.hop_length == 32 and dict(**hps.model) == {"a": [1, 2]} and "model" in hps
    m = modules.ResBlock2(32, 3)
    ck = tmp_path / "G_100.pth"
    utils.save_checkpoint(m, None, 2e-4, 100, str(ck))
    m2 = modules.ResBlock2(32, 3)
    _, _, lr, it = utils.load_checkpoint(str(ck), m2, None)
    assert it == 100 and lr == 2e-4
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    (tmp_path / "G_20.pth").write_bytes(b"")
    assert utils.latest_checkpoint_path(str(tmp_path)).endswith("G_100.pth")


def test_every_kernel_switch_has_a_variant_run():
    """Every environment switch the library reads that changes WHICH kernel runs must be exercised by
    tests/test_gpu_variants.py (and the variant suite must not name switches the library no longer reads)."""
    import glob
    import test_gpu_variants as V
    read = set()
    for f in glob.glob(os.path.join(cases.ROOT, "smart-vocoder_amd", "csrc", "*.hip")) + glob.glob(os.path.join(cases.ROOT, "smart-vocoder_amd", "csrc", "*.h")):
        read |= set(re.findall(r'getenv\("(SVOC_[A-Z0-9_]+)"\)', open(f).read()))
    # not kernel choices: sizes / thresholds of the plan cache, the variant-batch preset, diagnostics
    tunables = {"SVOC_CT_MIN_TILES", "SVOC_MRF_MIN_TILES", "SVOC_WN_SMALL_TILES", "SVOC_GRAPH_MAX_FRAMES", "SVOC_GRAPH_MIN_SEEN", "SVOC_VARIANT_BATCH", "SVOC_DBG_WALL", "SVOC_DBG_DUMP", "SVOC_DBG_ABL", "SVOC_RB_LDS_MIN", "SVOC_PERSIST_TIMEOUT_MS", "SVOC_WN_PREFETCH"}
    tested = set()
    for env, _slice in V.VARIANTS.values():
        tested |= set(env)
    assert read - tunables - tested == set(), f"switches without a variant run: {sorted(read - tunables - tested)}"
    assert tested - read == set(), f"variant runs of switches the library does not read: {sorted(tested - read)}"
