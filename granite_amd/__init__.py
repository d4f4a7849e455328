"""granite_amd — MI355X (gfx950) executor for Granite's image-space chain.

csrc/  hand-written HIP kernels + the C ABI (include/granite_hip.h) + the C++ host layer that restates Granite's
       RenderGraph pass API; capi.py binds the C ABI with ctypes; synth.py generates the synthetic inputs.
"""
__all__ = ["capi", "synth"]
