"""omniserve_amd -- MI355X (gfx950) implementation of the OmniServe quantized-inference hot path.

Layout:
  csrc/      hand-written HIP kernels + the C ABI (include/omniserve_hip.h)
  _lib.py    ctypes binding of libomniserve_hip.so (fails loudly if it is missing)
  backend/   host-side mirror of the reference's ``omniserve_backend.*`` extension modules
             (same module names, function names and positional arguments)
  runtime.py minimal decode-step driver that wires the kernels like
             omniserve/modeling/models/llama_w4a8_unpad.py:406-438 (used by bench.py / smoke)
"""
__version__ = "0.1.0"
