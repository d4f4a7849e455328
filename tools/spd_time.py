"""Single-pass downsampler launch time and achieved HBM rate (hipEvents around the launches): an RGBA16F image's mips 1.. from
its level 0.  Algorithmic bytes = the source once + the chain once."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import capi
for w, h in ((2048, 2048), (4096, 4096), (3840, 2160)):
    gr = capi.Context(0)
    src = np.random.default_rng(0).random((h, w, 4), dtype=np.float32).astype(np.float16).view(np.uint16)
    img = capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16B16A16_SFLOAT).upload(src)
    w0, h0 = w // 2, h // 2
    mips = min(12, max(w0, h0).bit_length())
    chain = gr.spd_downsample(img, w0, h0, mips, 3)
    for _ in range(5):
        gr.spd_downsample(img, w0, h0, mips, 3, chain=chain)
    gr.sync()
    gr.timing_reset(); gr.timing_enable(True)
    for _ in range(50):
        gr.spd_downsample(img, w0, h0, mips, 3, chain=chain)
    gr.sync()
    t = gr.timing_query()["spd"]
    gr.timing_enable(False)
    us = 1e3 * t[1] / t[0]
    bytes_ = w * h * 8 + chain.nbytes
    print(f"{w}x{h} -> {mips} mips: {us:.1f} us (tiles + tail), {bytes_ / 1e6:.1f} MB algorithmic -> {bytes_ / us / 1e6:.2f} TB/s")
    gr.close()
