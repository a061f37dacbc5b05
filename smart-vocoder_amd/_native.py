"""ctypes binding of libsvoc_hip.so (C ABI: include/svoc.h).

There is deliberately no fallback: if the library has not been built
(``python -c "import __graft_entry__ as g; g.build()"``) or no GPU is present,
calling into this module raises.  PyTorch is used only for device memory and
streams; every op of the inference path runs in the HIP library.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVOC_LIB selects another build of the same library (A/B comparisons of kernel variants on one GPU box)
LIB_PATH = os.environ.get("SVOC_LIB") or os.path.join(_HERE, "csrc", "libsvoc_hip.so")

ABI_VERSION = 6
_lib = None


class SvocError(RuntimeError):
    pass


class svoc_tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class svoc_generator_config(C.Structure):
    _fields_ = [("initial_channel", C.c_int32), ("resblock_kind", C.c_int32), ("n_kernels", C.c_int32),
                ("resblock_kernel_sizes", C.c_int32 * 8), ("n_dilations", C.c_int32 * 8),
                ("resblock_dilation_sizes", (C.c_int32 * 8) * 8), ("n_upsamples", C.c_int32),
                ("upsample_rates", C.c_int32 * 8), ("upsample_kernel_sizes", C.c_int32 * 8),
                ("upsample_initial_channel", C.c_int32), ("gin_channels", C.c_int32)]


class svoc_synth_config(C.Structure):
    _fields_ = [("n_mel", C.c_int32), ("inter_channels", C.c_int32), ("hidden_channels", C.c_int32),
                ("enc_n_layers", C.c_int32), ("enc_kernel_size", C.c_int32), ("enc_dilation_rate", C.c_int32),
                ("flow_n_layers", C.c_int32), ("flow_kernel_size", C.c_int32), ("flow_dilation_rate", C.c_int32),
                ("flow_n_flows", C.c_int32), ("gin_channels", C.c_int32), ("dec", svoc_generator_config)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
_TAB = (C.POINTER(svoc_tensor), _I)

# name -> (restype, argtypes); must list every symbol include/svoc.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "svoc_abi_version": (_I, []),
    "svoc_last_error": (C.c_char_p, []),
    "svoc_build_arch": (C.c_char_p, []),
    "svoc_stats_reset": (_I, []),
    "svoc_stats_get": (_I, [C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(_L)]),
    "svoc_stats_convolutions": (_L, []),
    "svoc_stats_executed_flops": (C.c_double, []),
    "svoc_set_variant_batch": (_I, [_I]),
    "svoc_check_async_error": (_I, []),
    "svoc_debug_raise_async_error": (_I, []),
    "svoc_persist_state": (_I, [_P, _P]),
    "svoc_debug_persist_control": (_I, [_I, _I, _I]),
    "svoc_profile_enable": (_I, [_I]),
    "svoc_profile_report": (_I, [C.c_char_p, _I]),
    "svoc_debug_set_stamp_buffer": (_I, [_P]),
    "svoc_debug_conv_timing": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(C.c_double)]),
    "svoc_wn_create": (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, *_TAB, C.c_char_p]),
    "svoc_wn_forward": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I]),
    "svoc_wn_destroy": (None, [_P]),
    "svoc_resblock_create": (_I, [C.POINTER(_P), _I, _I, _I, C.POINTER(C.c_int), _I, *_TAB, C.c_char_p]),
    "svoc_resblock_forward": (_I, [_P, _P, _P, _P, _P, _I, _I]),
    "svoc_resblock_destroy": (None, [_P]),
    "svoc_coupling_create": (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, _I, _I, *_TAB, C.c_char_p]),
    "svoc_coupling_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I]),
    "svoc_coupling_destroy": (None, [_P]),
    "svoc_flow_create": (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, _I, _I, *_TAB, C.c_char_p]),
    "svoc_flow_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _I, _I]),
    "svoc_flow_destroy": (None, [_P]),
    "svoc_generator_create": (_I, [C.POINTER(_P), C.POINTER(svoc_generator_config), *_TAB, C.c_char_p]),
    "svoc_generator_forward": (_I, [_P, _P, _P, _I, _L, _P, _L, _P, _P, _I, _I]),
    "svoc_generator_destroy": (None, [_P]),
    "svoc_synth_create": (_I, [C.POINTER(_P), C.POINTER(svoc_synth_config), *_TAB]),
    "svoc_synth_infer": (_I, [_P, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _P, _I, _I]),
    "svoc_synth_workspace_bytes": (_L, [_P, _I, _I]),
    "svoc_synth_reserve": (_I, [_P, _I, _I]),
    "svoc_synth_hop": (_I, [_P]),
    "svoc_synth_plan_stats": (_I, [_P, C.POINTER(_L)]),
    "svoc_synth_destroy": (None, [_P]),
    "svoc_posterior_create": (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, _I, _I, *_TAB, C.c_char_p]),
    "svoc_posterior_forward": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I]),
    "svoc_posterior_destroy": (None, [_P]),
    "svoc_dds_create": (_I, [C.POINTER(_P), _I, _I, _I, *_TAB, C.c_char_p]),
    "svoc_mel_encoder_create": (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, _I, _I, *_TAB, C.c_char_p]),
    "svoc_mel_encoder_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I]),
    "svoc_mel_encoder_destroy": (None, [_P]),
    "svoc_layer_norm": (_I, [_P, _P, _P, _P, _F, _P, _I, _I, _I]),
    "svoc_dds_forward": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I]),
    "svoc_dds_destroy": (None, [_P]),
    "svoc_convflow_create": (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, _F, *_TAB, C.c_char_p]),
    "svoc_convflow_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I]),
    "svoc_convflow_destroy": (None, [_P]),
    "svoc_melspec_create": (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, C.c_double, C.c_double]),
    "svoc_melspec_frames": (_I, [_P, _L]),
    "svoc_melspec_spectrogram": (_I, [_P, _P, _P, _I, _I, _P]),
    "svoc_melspec_frames_center": (_I, [_P, _L, _I]),
    "svoc_melspec_spectrogram_center": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "svoc_melspec_mel": (_I, [_P, _P, _P, _I, _I, _P]),
    "svoc_mel_filterbank": (_I, [_I, _I, _I, C.c_double, C.c_double, _P]),
    "svoc_melspec_destroy": (None, [_P]),
    "svoc_rq_spline": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _F, _F, _F, _P, _P]),
    "svoc_sequence_mask": (_I, [_P, _P, _P, _I, _I]),
    "svoc_fused_add_tanh_sigmoid_multiply": (_I, [_P, _P, _P, _P, _I, _I, _I]),
    "svoc_flip_channels": (_I, [_P, _P, _P, _I, _I, _I]),
    "svoc_fold_weight_norm": (_I, [_P, _P, _P, _P, _L, _L]),
    "svoc_conv1d": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F]),
    "svoc_conv1d_winograd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F]),
    "svoc_conv_transpose1d": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F]),
}


def lib():
    """Load (once) and return the ctypes handle of libsvoc_hip.so; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise SvocError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C smart-vocoder_amd/csrc). "
                "There is no CPU or eager-PyTorch fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.svoc_abi_version() != ABI_VERSION:
            raise SvocError(f"libsvoc_hip.so ABI {h.svoc_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
        _lib = h
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().svoc_last_error()
        raise SvocError(f"libsvoc_hip error {rc}: {msg.decode() if msg else '?'}")


def require_gpu(t=None):
    if not torch.cuda.is_available():
        raise SvocError("no GPU visible: the smart-vocoder_amd ops run only as HIP kernels on a gfx950 device "
                        "(there is no CPU fallback)")
    if t is not None and not t.is_cuda:
        raise SvocError(f"expected a CUDA/HIP tensor, got device {t.device}; call .cuda() on the module and inputs")


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def f32(t):
    """Contiguous fp32 view/copy of a device tensor (the ABI takes fp32 NCW)."""
    require_gpu(t)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class TensorTable:
    """ctypes array of svoc_tensor built from {name: device tensor}; keeps the tensors alive."""

    def __init__(self, named):
        self.keep = []
        items = []
        for name, t in named.items():
            t = f32(t.detach())
            self.keep.append(t)
            if t.dim() > 4:
                raise SvocError(f"{name}: more than 4 dims")
            st = svoc_tensor()
            nm = name.encode()
            self.keep.append(nm)
            st.name = nm
            st.data = t.data_ptr()
            st.ndim = t.dim()
            for i, s in enumerate(t.shape):
                st.shape[i] = s
            items.append(st)
        self.n = len(items)
        self.arr = (svoc_tensor * max(1, self.n))(*items)


class Handle:
    """Owns one opaque library handle; rebuilt by the owning module when its parameters change."""

    def __init__(self, destroy_name):
        self.h = C.c_void_p(0)
        self._destroy = destroy_name

    def out(self):
        return C.byref(self.h)

    def close(self):
        if self.h and self.h.value and _lib is not None:
            getattr(_lib, self._destroy)(self.h)
        self.h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stats_reset():
    lib().svoc_stats_reset()


def stats_get():
    a, b, c = _L(0), C.c_double(0), _L(0)
    lib().svoc_stats_get(C.byref(a), C.byref(b), C.byref(c))
    return dict(conv_launches=a.value, conv_flops=b.value, other_launches=c.value, convolutions=lib().svoc_stats_convolutions(),
                executed_flops=lib().svoc_stats_executed_flops())


def set_variant_batch(n):
    """Kernel variants are chosen as if the batch held `n` utterances (0: the real batch); returns the previous value.
    See include/svoc.h svoc_set_variant_batch."""
    return lib().svoc_set_variant_batch(int(n))


def check_async_error():
    """Raises when a persistent WN launch of an earlier call gave up one of its bounded waits (include/svoc.h svoc_check_async_error)."""
    check(lib().svoc_check_async_error())


def persist_state():
    """(disabled, timeout_ms) of the persistent WN launches: `disabled` is True once a launch gave up a wait and the process fell back to one
    launch per WN layer; `timeout_ms` is the bound of a wait (SVOC_PERSIST_TIMEOUT_MS).  include/svoc.h svoc_persist_state."""
    d, t = C.c_int(0), C.c_int(0)
    check(lib().svoc_persist_state(C.byref(d), C.byref(t)))
    return bool(d.value), t.value


def debug_persist_control(fault_tile=-1, timeout_ms=0, reenable=False):
    """Diagnostics (tests): see include/svoc.h svoc_debug_persist_control."""
    check(lib().svoc_debug_persist_control(int(fault_tile), int(timeout_ms), 1 if reenable else 0))


class variant_batch:
    """``with variant_batch(n): ...`` - scoped svoc_set_variant_batch."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.prev = set_variant_batch(self.n)
        return self

    def __exit__(self, *exc):
        set_variant_batch(self.prev)
        return False


def profile_enable(on=True):
    lib().svoc_profile_enable(1 if on else 0)


def profile_report():
    buf = C.create_string_buffer(1 << 16)
    lib().svoc_profile_report(buf, len(buf))
    return buf.value.decode()
