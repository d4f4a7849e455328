"""EASU / RCAS launch times, 2560x1440 -> 3840x2160 (hipEvents around each launch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import capi
gr = capi.Context(0)
src = np.random.default_rng(0).integers(0, 256, (1440, 2560, 4), dtype=np.uint8)
a = capi.DeviceImage(gr, 2560, 1440, capi.FORMAT_R8G8B8A8_SRGB).upload(src)
b = capi.DeviceImage(gr, 3840, 2160, capi.FORMAT_R8G8B8A8_UNORM)
c = capi.DeviceImage(gr, 3840, 2160, capi.FORMAT_R8G8B8A8_SRGB)
d = capi.DeviceImage(gr, 3840, 2160, capi.FORMAT_R8G8B8A8_UNORM)
for fp16 in (0, 1):
    for _ in range(5):
        gr.fsr_upscale(a, b, fp16)
    gr.sync(); gr.timing_reset(); gr.timing_enable(True)
    for _ in range(50):
        gr.fsr_upscale(a, b, fp16); gr.fsr_sharpen(b, c if fp16 else d, 0.7071)
    gr.sync(); q = gr.timing_query(); gr.timing_enable(False)
    print("EASU fp16" if fp16 else "EASU fp32", {k: round(1e3 * ms / n, 1) for k, (n, ms) in q.items()}, "us; sharpen target", "sRGB" if fp16 else "UNORM")
