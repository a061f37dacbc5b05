"""MI355X-native SMART-Vocoder inference path (SynthesizerTrn.infer and its modules).

Host side: PyTorch-ROCm modules with the reference's names, constructor
signatures and state_dict layout (reference models.py / modules.py).  Every
tensor op runs in hand-written HIP kernels for gfx950 behind the C ABI declared
in ``include/svoc.h`` (``csrc/libsvoc_hip.so``, loaded with ctypes).  There is no
CPU or eager-PyTorch fallback: using a module without the built library or
without a GPU raises.
"""
__version__ = "0.1.0"
