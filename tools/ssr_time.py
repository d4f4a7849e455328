"""Screen-space reflection launch times at a given size (hipEvents around each launch group): classify (count + scan + emit),
trace_primary, apply -- on the close-up trough scene of tests/test_gpu_ssr.py, where about a third of the frame shoots rays."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from granite_amd import capi
from granite_amd.data import expand_sssr_dither, load_brdf_lut, load_sssr_noise_base
from util import close_up_scene

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
gr = capi.Context(0)
cam, depth, normal, pbr, albedo, light = close_up_scene(w, h)
rp = cam.render_params()
m = np.asarray(rp[48:64], np.float32).reshape(4, 4)  # inverse projection, m[c] = column c
zt = [-float(m[2][2]), float(m[2][3]), -float(m[3][2]), float(m[3][3])]  # spd.cpp:164-165
F16 = capi.FORMAT_R16G16B16A16_SFLOAT
ddepth = capi.DeviceImage(gr, w, h, capi.FORMAT_D32_SFLOAT).upload(depth)
chain, _, layout = gr.hiz(ddepth, zt)
noise = expand_sssr_dither(load_sssr_noise_base())
imgs = dict(pbr=capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8_UNORM).upload(pbr),
            normal=capi.DeviceImage(gr, w, h, capi.FORMAT_A2B10G10R10_UNORM_PACK32).upload(normal),
            light=capi.DeviceImage(gr, w, h, F16).upload(light), albedo=capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8B8A8_SRGB).upload(albedo),
            lut=capi.DeviceImage(gr, 256, 256, capi.FORMAT_R16G16_SFLOAT).upload(load_brdf_lut()), hdr=capi.DeviceImage(gr, w, h, F16).upload(light))
dither = capi.DeviceBuffer(gr, noise.nbytes).upload(noise)


def frame(i):
    out = gr.ssr_trace(chain, layout, imgs["pbr"], imgs["normal"], imgs["light"], dither, i & 63, rp[32:48], rp[80:96], rp[96:99])
    gr.ssr_apply(imgs["hdr"], out["output"], imgs["albedo"], imgs["normal"], imgs["pbr"], ddepth, imgs["lut"], rp[80:96], rp[96:99])
    return out


out = frame(0)
gr.sync()
rays = int(out["ray_counter"].download(np.uint32)[5])
gr.timing_reset(); gr.timing_enable(True)
N = 20
for i in range(N):
    frame(i)
    gr.sync()
t = gr.timing_query()
gr.timing_enable(False)
print(f"{w}x{h}: {rays} rays ({100.0 * rays / (w * h):.1f} % of the pixels)")
total = 0.0
for name in ("ssr_classify", "ssr_trace", "ssr_apply"):
    us = 1e3 * t[name][1] / t[name][0]
    total += us
    print(f"  {name}: {us:.1f} us/launch")
# algorithmic bytes: classify reads depth level 0 (4) + pbr (2), clears output (8) + confidence (1); a ray reads normal, pbr, the lit
# texel it hits (4 + 2 + 8) and writes 8 + 2 + 1 for itself and its copies; apply reads reflected 8 + albedo 4 + normal 4 + pbr 2 + depth 4
# and read-modify-writes hdr (16)
px = w * h
print(f"  sum {total:.1f} us = {px / total / 1e3:.1f} Gpx/s; apply alone moves {px * 38 / 1e6:.0f} MB algorithmic")
