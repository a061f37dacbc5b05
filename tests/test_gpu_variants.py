"""The fall-back kernel paths a SHAPE can reach are selected, at the shapes of the parity suite, by environment variables the library
reads once per process, so each variant runs a slice of the parity suite in a subprocess.  Every `getenv` in csrc/ that changes a kernel
choice is listed here (tests/test_abi.py::test_every_kernel_switch_has_a_variant_run checks the two lists against each other).  Round 5
removed the switches (and the code) that only an environment variable could reach - A/B arms whose verdict is recorded in DESIGN.md:
SVOC_W4_{ACCUM,ACC3,PERM,PRIO,PAIR,PAIR64}, SVOC_FUSE_V, SVOC_FUSE64, SVOC_WINO_WS, SVOC_WINO_WM, SVOC_WN_CT, SVOC_WN_KSPLIT, SVOC_LN_V2,
SVOC_XCD, SVOC_CT_ROWS256, SVOC_CT_TAIL, SVOC_STREAMS, SVOC_W4_F44 (kernels those arms shared with a shape-reachable path are covered by shape:
tests/test_gpu_parity.py::test_fallback_shapes).

  SVOC_FUSE=0 SVOC_FUSE_WN=0     unfused fallbacks: two convolutions per ResBlock iteration / WN layer (any shape the fused kernels refuse)
  SVOC_GROUP=0                   MRF chains on separate streams for every stage (chains that disagree on the dilation order; short inputs)
  SVOC_WINO=0                    direct-form kernels instead of Winograd (kernel sizes other than 3 / 7 / 11, dilations other than 1 / 3 / 5)
  SVOC_WINO_F4=0                 Winograd F(2,3) kernels instead of F(4,3) / F(4,4) (odd row-block counts, unaligned rows, L % 4 != 0)
  SVOC_W4_C32=0                  C = 32 MRF stage on the fused direct-form ResBlock kernel instead of F(4,3) conv by conv (short inputs)
  SVOC_W4_TAIL=0                 dilated Winograd rows keep their partial last tile instead of handing its windows to the direct-form tail launch (every length
                                 whose last q block does not sit alone in a tile, or where dropping it saves no round of the persistent workgroups)
  SVOC_WN_SMALL_F25=0            short inputs: WN layers as two K-split convolutions instead of one launch per layer (wn_small.hip)
  SVOC_WN_SMALL_PP2=0            mid-size batches: six workgroups per 32-column tile (one row pair each) instead of three (fewer than 43 or more than 85 tiles)
  SVOC_WN_STACK=0                WN stacks one launch per layer instead of one persistent launch per stack (more 32-column tiles than CUs; conditioning input)
  SVOC_WN_F25=0                  WN in_layers in direct form (K-split layer kernel) instead of Winograd F(2,5) (H != 192, k != 5)
  SVOC_KSPLIT=0 SVOC_WN_SMALL=0 SVOC_MRF_SMALL=0    short inputs on the throughput kernels (no K-split convolutions, fused WN
                                 layers, grouped MRF launches)
  SVOC_GRAPH=0                   short inputs as direct launches (no captured hipGraph plans: a caller that is itself capturing)
  SVOC_CT_WINO=0                 upsamplers on the direct polyphase kernel instead of the Winograd F(4,2) one (other k / stride; short inputs)
"""
import os
import subprocess
import sys
import threading

import pytest

import cases

pytestmark = pytest.mark.gpu

# slices of tests/test_gpu_parity.py per kind of switch (each variant runs in its own process: the library reads the
# environment once); every slice reaches the kernels its switches select (the long-form and full-size cases are the ones
# large enough for the grouped / fused / persistent kernels)
DEC = ("test_conv1d_winograd or test_resblock1 or test_generator or test_infer_vs_reference_golden or test_infer_long_form_tiling")
WNS = "test_wn or test_coupling or test_infer_vs_reference_golden or test_full_size_properties"
SMALL = "test_infer_vs_reference_golden or test_small_shape_graph_replay or test_generator or test_wn or test_resblock1"
UPS = "test_conv_transpose or test_generator or test_infer_vs_reference_golden or test_full_size_vs_reference_fixture"

VARIANTS = {
    "unfused": ({"SVOC_FUSE": "0", "SVOC_FUSE_WN": "0"}, DEC + " or " + WNS),
    "ungrouped": ({"SVOC_GROUP": "0"}, DEC),
    "no_winograd": ({"SVOC_WINO": "0"}, DEC),
    "winograd_f23": ({"SVOC_WINO_F4": "0"}, DEC),
    "c32_fused_direct": ({"SVOC_W4_C32": "0"}, DEC),
    "dilated_rows_keep_their_last_tile": ({"SVOC_W4_TAIL": "0"}, "test_generator_dilated_row_tails or test_c2_full_size or test_wn_mid_size"),
    # the merged accumulate launch refuses (fewer than 1000 tiles) where the grouped ones still run: the last dilated launch of a stage then writes its rows in
    # natural order, and so does its tail launch (the window-major position is the other branch of conv_wino4_tail_kernel's store)
    "row_tails_in_natural_order": ({"SVOC_MRF_MIN_TILES": "1000"}, "test_generator_dilated_row_tails"),
    "wn_direct_form": ({"SVOC_WN_F25": "0"}, WNS),
    "wn_one_launch_per_layer": ({"SVOC_WN_STACK": "0"}, WNS),
    "wn_short_inputs_one_launch_per_layer": ({"SVOC_WN_MESH": "0"}, "test_infer_vs_reference_golden or test_small_shape_graph_replay or test_wn or test_coupling or test_flow"),
    "wn_short_inputs_two_convolutions": ({"SVOC_WN_SMALL_F25": "0"}, SMALL + " or test_coupling or test_flow"),
    "wn_mid_size_six_workgroups_per_tile": ({"SVOC_WN_SMALL_PP2": "0"}, "test_wn_mid_size or test_wn_edge"),
    "no_small_shape_kernels": ({"SVOC_KSPLIT": "0", "SVOC_WN_SMALL": "0", "SVOC_MRF_SMALL": "0"}, SMALL),
    "no_graph": ({"SVOC_GRAPH": "0"}, SMALL),
    "upsamplers_direct": ({"SVOC_CT_WINO": "0"}, UPS),
}


def _jobs():
    return max(1, int(os.environ.get("SVOC_VARIANT_JOBS", "2")))


def _run_variant(name):
    e = dict(os.environ)
    env, sl = VARIANTS[name]
    e.update(env)
    # the host threads are shared out between the concurrent variant processes: _jobs() x (all cores) OpenMP threads spin on each
    # other in the oracle's convolutions (a first parallel run of this suite did not finish in 25 minutes)
    # (and never more than 16 each: oneDNN's small convolutions get SLOWER beyond that on the 128-core hosts - bench.py's thread probe)
    nthr = str(max(2, min(16, (os.cpu_count() or 8) // (2 * _jobs()))))
    e.setdefault("OMP_NUM_THREADS", nthr)
    e.setdefault("MKL_NUM_THREADS", nthr)
    with _GATE:
        return _run_gated(name, e, sl)


def _run_gated(name, e, sl):
    return subprocess.run([sys.executable, "-m", "pytest", os.path.join(cases.ROOT, "tests", "test_gpu_parity.py"), "-q", "-x",
                           "-m", "gpu", "-p", "no:cacheprovider", "-k", sl], env=e, cwd=cases.ROOT, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=1200)


# The variant processes spend most of their time on the host (interpreter start, weight generation, the oracle's float64
# convolutions), so SVOC_VARIANT_JOBS of them run side by side: the first test starts them all, every test collects its own result.
# Parity only - nothing here is timed.  TWO at a time since round 5 (four until then): the persistent WN launches (csrc/wn_stack.hip,
# csrc/wn_mesh.hip) are validated for two processes on one GPU; with four, launches of the 16-layer stack can hold each other's CUs until their
# bounded waits give up (profiles/r05_z_persistent_launches_shared_gpu.txt) - reported, but a failed test all the same.
# Round 6 (suite <= 200 s): the pool is started at the BEGINNING of the session (tests/conftest.py calls start_pool() once the
# collection holds a test of this module) and works through the variants ONE at a time while the main process runs tests/test_gpu_parity.py - two
# processes on the GPU, the validated regime; when the main process reaches this module it has nothing else to do and a second variant may run beside
# the first (the gate below gets its second permit).
_POOL = None
_FUTURES = {}
_GATE = threading.Semaphore(1)
_GATE_OPEN = False


def start_pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=_jobs())
        for n in VARIANTS:
            _FUTURES[n] = _POOL.submit(_run_variant, n)
    return _FUTURES


def _futures():
    global _GATE_OPEN
    f = start_pool()
    if not _GATE_OPEN:
        _GATE_OPEN = True
        for _ in range(_jobs() - 1):
            _GATE.release()
    return f


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant(name):
    r = _futures()[name].result(timeout=3600)
    assert r.returncode == 0, r.stdout[-3000:]
