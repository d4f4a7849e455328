"""Each anti-aliasing kernel alone at a given size (hipEvents around every launch, nothing else on the GPU): FXAA, the three
SMAA passes at every preset, the TAA resolve at every quality.  Two inputs: the AA test card (flat regions, steps, diagonals,
circles, a noisy band: edges on a few per cent of the pixels, as in a rendered frame) and the tonemapped white-noise HDR frame
of the benchmark scene (an edge on nearly every pixel: the worst case for the SMAA weight pass).
usage: python tools/aa_time.py [width height]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from granite_amd import app as gapp, capi, synth
from granite_amd.data import load_smaa_luts

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
gr = capi.Context(0)
gr.smaa_set_luts(*load_smaa_luts())
RGBA8, F16 = capi.FORMAT_R8G8B8A8_UNORM, capi.FORMAT_R16G16B16A16_SFLOAT


def timed(label, names, fn, algo_bytes, reps=int(os.environ.get("AA_TIME_REPS", "20"))):  # AA_TIME_REPS: fewer launches under a counter pass
    for _ in range(min(3, reps)):
        fn()
    gr.sync()
    gr.timing_reset(); gr.timing_enable(True)
    for _ in range(reps):
        fn()
    gr.sync()
    t = gr.timing_query()
    gr.timing_enable(False)
    for n in names:
        us = 1e3 * t[n][1] / t[n][0]
        print(f"  {label:34s} {n:22s} {us:8.1f} us  {algo_bytes[n] * w * h / us / 1e6:6.2f} TB/s algorithmic ({algo_bytes[n]} B/px)")


# the benchmark frame, tonemapped: run the application once and read its backbuffer
cam = synth.Camera(w, h)
a = gapp.Application(w, h, lighting=False)
a.upload_hdr(synth.make_hdr(w, h))
a.render_frames(2, sync=True)
noisy = a.read_backbuffer().copy()
a.close()
inputs = {"test card": synth.make_ldr_pattern(w, h), "tonemapped noise": noisy}
# AA_TIME_INPUT=card|noise: one input only (for profiler runs, whose per-kernel means would otherwise mix the two)
only = os.environ.get("AA_TIME_INPUT")
if only:
    inputs = {k: v for k, v in inputs.items() if only in k}
print(f"{w}x{h}")
for title, ldr in inputs.items():
    src = capi.DeviceImage(gr, w, h, RGBA8).upload(ldr)
    out = capi.DeviceImage(gr, w, h, RGBA8)
    edges = capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8_UNORM)
    weights = capi.DeviceImage(gr, w, h, RGBA8)
    print(title)
    timed("FXAA", ["fxaa"], lambda: gr.fxaa(src, out), {"fxaa": 8})
    for q, preset in enumerate(("Low", "Medium", "High", "Ultra")):
        def smaa():
            gr.smaa_edge_detection(src, edges, q)
            gr.smaa_blend_weight(edges, weights, q)
            gr.smaa_neighbor_blend(src, weights, out)
        timed(f"SMAA {preset}", ["smaa_edge_detection", "smaa_blend_weight", "smaa_neighbor_blend"], smaa,
              {"smaa_edge_detection": 6, "smaa_blend_weight": 6, "smaa_neighbor_blend": 12})
        e = edges.download()
        if q == 3:
            print(f"    edge pixels: {100.0 * (e != 0).any(axis=2).mean():.1f} %")

gbuf = synth.make_gbuffer(cam)
cur = capi.DeviceImage(gr, w, h, F16).upload(gbuf["emissive"])
depth = capi.DeviceImage(gr, w, h, capi.FORMAT_D32_SFLOAT).upload(gbuf["depth"])
mv = capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16_SFLOAT).upload(synth.make_motion_vectors(w, h))
hist = [capi.DeviceImage(gr, w, h, F16), capi.DeviceImage(gr, w, h, F16)]
color = capi.DeviceImage(gr, w, h, F16)
reproj = np.eye(4, dtype=np.float32)
reproj[0, 0] = reproj[1, 1] = reproj[0, 3] = reproj[1, 3] = 0.5
reproj = np.ascontiguousarray(reproj.T).reshape(-1)
gr.taa_resolve(cur, depth, mv, None, color, hist[0], reproj, 2)
print("TAA resolve (current 8 + depth 4 + mv 4 + history 8 read, colour 8 + history 8 written)")
for q, name in enumerate(("Low", "Medium", "High")):
    state = {"i": 0}
    def taa():
        i = state["i"]; state["i"] ^= 1
        gr.taa_resolve(cur, depth, mv, hist[i], color, hist[i ^ 1], reproj, q)
    timed(f"TAA {name}", ["taa_resolve"], taa, {"taa_resolve": 40})
