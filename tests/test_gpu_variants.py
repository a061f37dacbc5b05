"""The alternate kernel paths that still ship are selected by environment variables the library reads once per process, so
each variant runs a slice of the parity suite in a subprocess.  Every `getenv` in csrc/ that changes a kernel choice is
listed here (tests/test_abi.py::test_every_kernel_switch_has_a_variant_run checks the two lists against each other):

  SVOC_FUSE=0 SVOC_FUSE_WN=0     unfused fallbacks: two convolutions per ResBlock iteration / WN layer
  SVOC_FUSE_V=1                  generic fused-ResBlock kernel instead of the compile-time-specialised one
  SVOC_FUSE64=1                  C=64 stage on the fused direct-form ResBlock kernel instead of Winograd conv by conv
  SVOC_STREAMS=0                 single-stream MRF
  SVOC_GROUP=0                   MRF chains on separate streams for every stage (no grouped launches)
  SVOC_WINO=0                    direct-form grouped kernel (conv_group_kernel) instead of Winograd F(2,3)
  SVOC_WINO_F4=0                 Winograd F(2,3) kernels (wave-specialised, two workgroups per CU) instead of F(4,3)
  SVOC_WINO_F4=0 SVOC_WINO_WS=0  four-wave F(2,3) Winograd kernels
  SVOC_W4_PRIO=0                 F(4,3) producers at the consumers' priority
  SVOC_W4_F44=0                  k = 7 / 11 in F(4,3) form (six-product groups + left-over taps) instead of F(4,4)
  SVOC_W4_ACC3=0                 the accumulate launch as three read-modify-write members instead of one set of accumulators
  SVOC_W4_PAIR64=0               C = 64 stage: the undilated ResBlock iteration conv by conv as well (the pair kernel at C = 32 only)
  SVOC_W4_PAIR=0                 C = 32 stage: c1 and c2 of a ResBlock iteration as two grouped launches instead of one (conv_wino4_pair.hip)
  SVOC_W4_C32=0                  C = 32 MRF stage on the fused direct-form ResBlock kernel instead of F(4,3) conv by conv
  SVOC_W4_PERM=0                 dilated F(4,3) convolutions store their rows in natural order (four scattered dwords per lane)
  SVOC_WN_SMALL_F25=0            short inputs: WN layers as two K-split convolutions instead of one launch per layer (wn_small.hip)
  SVOC_WN_F25=0                  WN in_layers in direct form (K-split layer kernel) instead of Winograd F(2,5)
  SVOC_W4_ACCUM=0                the chains' last convolutions as three launches instead of one accumulate launch
  SVOC_WINO_WM=2                 2x2 wave layout (64-row tiles) for the Winograd kernels at C >= 128
  SVOC_WN_CT=0                   generic (runtime-geometry) MFMA loops in the K-split WN layer kernel
  SVOC_WN_KSPLIT=0               6-wave WN layer kernel (one wave per row pair) instead of the 12-wave K-split one
  SVOC_KSPLIT=0 SVOC_WN_SMALL=0 SVOC_MRF_SMALL=0    short inputs on the throughput kernels (no K-split convolutions, fused WN
                                 layers, grouped MRF launches)
  SVOC_GRAPH=0                   short inputs as direct launches (no captured hipGraph plans)
  SVOC_XCD=0                     natural workgroup -> tile order instead of the XCD-aware one
  SVOC_LN_V2=0                   round-1 LayerNorm / DDSConv tile kernel
  SVOC_CT_WINO=0                 upsamplers on the direct polyphase kernel instead of the Winograd F(4,2) one (convt_wino.hip)
  SVOC_CT_TAIL=0                 F(4,2) upsamplers: the column q = L inside the window tiles (no separate tail launch)
  SVOC_CT_ROWS256=0              F(4,2) upsamplers: 128-row blocks (four consumer waves) where 256-row blocks would be used
"""
import os
import subprocess
import sys

import pytest

import cases

pytestmark = pytest.mark.gpu

# slices of tests/test_gpu_parity.py per kind of switch (each variant runs in its own process: the library reads the
# environment once); every slice reaches the kernels its switches select (the long-form and full-size cases are the ones
# large enough for the grouped / fused / persistent kernels)
DEC = ("test_conv1d_winograd or test_resblock1 or test_generator or test_infer_vs_reference_golden or test_infer_long_form_tiling")
WNS = "test_wn or test_coupling or test_infer_vs_reference_golden or test_full_size_properties"
SMALL = "test_infer_vs_reference_golden or test_small_shape_graph_replay or test_generator or test_wn or test_resblock1"
OFFG = "test_dds or test_layer_norm or test_convflow"
UPS = "test_conv_transpose or test_generator or test_infer_vs_reference_golden or test_c2_full_size_vs_oracle"

VARIANTS = {
    "unfused": ({"SVOC_FUSE": "0", "SVOC_FUSE_WN": "0"}, DEC + " or " + WNS),
    "generic_fused_resblock": ({"SVOC_FUSE_V": "1"}, DEC),
    "fused_c64": ({"SVOC_FUSE64": "1"}, DEC),
    "single_stream": ({"SVOC_STREAMS": "0"}, DEC),
    "ungrouped": ({"SVOC_GROUP": "0"}, DEC),
    "no_winograd": ({"SVOC_WINO": "0"}, DEC),
    "winograd_f23": ({"SVOC_WINO_F4": "0"}, DEC),
    "winograd_f23_4wave": ({"SVOC_WINO_F4": "0", "SVOC_WINO_WS": "0"}, DEC),
    "winograd_f43_equal_priority": ({"SVOC_W4_PRIO": "0"}, DEC),
    "winograd_f43_for_k7_k11": ({"SVOC_W4_F44": "0"}, DEC),
    "mrf_accumulate_one_by_one": ({"SVOC_W4_ACCUM": "0"}, DEC),
    "mrf_accumulate_three_members": ({"SVOC_W4_ACC3": "0"}, DEC),
    "c32_conv_by_conv": ({"SVOC_W4_PAIR": "0"}, DEC),
    "c64_conv_by_conv": ({"SVOC_W4_PAIR64": "0"}, DEC),
    "c32_fused_direct": ({"SVOC_W4_C32": "0"}, DEC),
    "winograd_f43_natural_rows": ({"SVOC_W4_PERM": "0"}, DEC),
    "wn_direct_form": ({"SVOC_WN_F25": "0"}, WNS),
    "wn_short_inputs_two_convolutions": ({"SVOC_WN_SMALL_F25": "0"}, SMALL + " or test_coupling or test_flow"),
    "winograd_2x2": ({"SVOC_WINO_F4": "0", "SVOC_WINO_WM": "2"}, DEC),
    "wn_no_ksplit": ({"SVOC_WN_KSPLIT": "0"}, WNS),
    "wn_generic_loops": ({"SVOC_WN_CT": "0"}, WNS),
    "no_small_shape_kernels": ({"SVOC_KSPLIT": "0", "SVOC_WN_SMALL": "0", "SVOC_MRF_SMALL": "0"}, SMALL),
    "no_graph": ({"SVOC_GRAPH": "0"}, SMALL),
    "natural_tile_order": ({"SVOC_XCD": "0"}, DEC),
    "layernorm_v1": ({"SVOC_LN_V2": "0"}, OFFG),
    "upsamplers_direct": ({"SVOC_CT_WINO": "0"}, UPS),
    "upsamplers_f42_no_tail_launch": ({"SVOC_CT_TAIL": "0"}, UPS),
    "upsamplers_f42_128_row_blocks": ({"SVOC_CT_ROWS256": "0"}, UPS),
}


def _jobs():
    return max(1, int(os.environ.get("SVOC_VARIANT_JOBS", "4")))


def _run_variant(name):
    e = dict(os.environ)
    env, sl = VARIANTS[name]
    e.update(env)
    # the host threads are shared out between the concurrent variant processes: _jobs() x (all cores) OpenMP threads spin on each
    # other in the oracle's convolutions (a first parallel run of this suite did not finish in 25 minutes)
    nthr = str(max(2, (os.cpu_count() or 8) // (2 * _jobs())))
    e.setdefault("OMP_NUM_THREADS", nthr)
    e.setdefault("MKL_NUM_THREADS", nthr)
    return subprocess.run([sys.executable, "-m", "pytest", os.path.join(cases.ROOT, "tests", "test_gpu_parity.py"), "-q", "-x",
                           "-m", "gpu", "-p", "no:cacheprovider", "-k", sl], env=e, cwd=cases.ROOT, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=1200)


# The variant processes spend most of their time on the host (interpreter start, weight generation, the oracle's float64
# convolutions), so SVOC_VARIANT_JOBS of them (default 4) run side by side: the first test starts them all, every test collects
# its own result.  Parity only - nothing here is timed, so sharing the GPU is harmless.
_POOL = None
_FUTURES = {}


def _futures():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=_jobs())
        for n in VARIANTS:
            _FUTURES[n] = _POOL.submit(_run_variant, n)
    return _FUTURES


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant(name):
    r = _futures()[name].result(timeout=3600)
    assert r.returncode == 0, r.stdout[-3000:]
