"""The alternate kernel paths are selected by environment variables that the library reads once per process, so each
variant runs a slice of the parity suite in a subprocess:
  SVOC_WS=1                      experimental persistent wave-specialised convolution kernel (csrc/conv_ws.hip)
  SVOC_WS=2                      ... with two consumer sets per workgroup
  SVOC_FUSE=0 SVOC_FUSE_WN=0     unfused fallbacks (two convolutions per ResBlock iteration / WN layer)
  SVOC_STREAMS=0                 single-stream MRF
  SVOC_WN_KSPLIT=0               6-wave WN layer kernel (one wave per row pair) instead of the 12-wave K-split one
  SVOC_FUSE_WS=1                 persistent fused-ResBlock kernel with loader waves (experiment)
  SVOC_XCD=0                     natural workgroup -> tile order instead of the XCD-aware one
  SVOC_GROUP=0                   MRF chains on separate streams for every stage (no grouped launches)
  SVOC_TILE_256=1                256x128 tile for the C=256 stage
"""
import os
import subprocess
import sys

import pytest

import cases

pytestmark = pytest.mark.gpu

SLICE = "test_infer_vs_reference_golden or test_resblock1 or test_wn or test_generator or test_coupling"


@pytest.mark.parametrize("env", [{"SVOC_WS": "1"}, {"SVOC_WS": "2"}, {"SVOC_FUSE": "0", "SVOC_FUSE_WN": "0"}, {"SVOC_STREAMS": "0"}, {"SVOC_GROUP": "0"}, {"SVOC_WN_KSPLIT": "0"}, {"SVOC_FUSE_WS": "1"}, {"SVOC_XCD": "0"},
                                 {"SVOC_TILE_256": "1"}], ids=["ws", "ws2", "unfused", "single_stream", "ungrouped", "wn_no_ksplit", "fused_rb_loader_waves", "natural_tile_order", "tile256"])
def test_variant(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(cases.ROOT, "tests", "test_gpu_parity.py"), "-q", "-x",
                        "-m", "gpu", "-p", "no:cacheprovider", "-k", SLICE], env=e, cwd=cases.ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
