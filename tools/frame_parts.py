"""Frame period of the 4K executor in a few configurations (which part of the frame bounds it?).
usage: python tools/frame_parts.py full|postonly|hdr10|config4|ssr [frames]
config4 = TAA High in front of the post chain + SMAA Ultra behind the tonemap (BASELINE config 4); ssr = config 3 + the SSR pass.
config4 / ssr also print the per-kernel launch times (hipEvents, serialising -- read them as "alone", not as shares of the frame)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import app as gapp, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
w, h = 3840, 2160
cam = synth.Camera(w, h); gbuf = synth.make_gbuffer(cam); descs = synth.make_lights(cam, 4096)
if mode == "postonly":
    a = gapp.Application(w, h, lighting=False); a.upload_hdr(gbuf["emissive"])
elif mode == "config4":
    import numpy as np
    a = gapp.Application(w, h, pre_aa=gapp.POST_AA_TAA_HIGH, post_aa=gapp.POST_AA_SMAA_ULTRA)
    a.set_camera(np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16))
    a.set_lights(descs); a.upload_gbuffer(gbuf, synth.make_motion_vectors(w, h))
elif mode == "ssr":
    a = gapp.Application(w, h, ssr=True); a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
else:
    a = gapp.Application(w, h, hdr10=(mode == "hdr10")); a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
a.render_frames(30, sync=True)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); a.render_frames(frames, sync=True); t = time.perf_counter() - t0
    best = min(best, t / frames)
hs = a.host_stats()
print(mode, "WGS", os.environ.get("GR_LIGHTING_WGS_PER_CU", "-"), "frame us %.1f" % (1e6 * best))
if mode in ("config4", "ssr"):
    k = a.kernel_context()
    k.timing_set_sampling(1); k.timing_set_filter(None); k.timing_enable(True); k.timing_reset()
    a.render_frames(20, sync=True)
    for name, (count, ms) in sorted(k.timing_query().items(), key=lambda e: -e[1][1]):
        print("  %-28s %3d launches / 20 frames  %8.1f us each" % (name, count, 1e3 * ms / count))
    k.timing_enable(False)
a.close()
