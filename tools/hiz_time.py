"""Depth-hierarchy launch time and achieved HBM rate at a given size (hipEvents around each launch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import capi
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
gr = capi.Context(0)
depth = np.random.default_rng(0).random((h, w), dtype=np.float32) * 0.9 + 0.05
img = capi.DeviceImage(gr, w, h, capi.FORMAT_D32_SFLOAT).upload(depth)
zt = [0.0, -4.995, 1.0, 5.005]
for ds in (False, True):
    chain, counter, lay = gr.hiz(img, zt, ds)
    for _ in range(5):
        gr.hiz(img, zt, ds, chain=chain, counter=counter)
    gr.sync()
    gr.timing_reset(); gr.timing_enable(True)
    for _ in range(50):
        gr.hiz(img, zt, ds, chain=chain, counter=counter)
    gr.sync()
    t = gr.timing_query()["hiz"]
    gr.timing_enable(False)
    us = 1e3 * t[1] / t[0]
    bytes_ = w * h * 4 + chain.nbytes
    print(f"{w}x{h} output_downsample={int(ds)}: {us:.1f} us/launch, {bytes_ / 1e6:.1f} MB algorithmic -> {bytes_ / us / 1e6:.2f} TB/s")
