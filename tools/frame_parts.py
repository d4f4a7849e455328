"""Frame period of the 4K executor in a few configurations (which part of the frame bounds it?).
usage: python tools/frame_parts.py full|postonly|hdr10 [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import app as gapp, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
w, h = 3840, 2160
cam = synth.Camera(w, h); gbuf = synth.make_gbuffer(cam); descs = synth.make_lights(cam, 4096)
if mode == "postonly":
    a = gapp.Application(w, h, lighting=False); a.upload_hdr(gbuf["emissive"])
else:
    a = gapp.Application(w, h, hdr10=(mode == "hdr10")); a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
a.render_frames(30, sync=True)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); a.render_frames(frames, sync=True); t = time.perf_counter() - t0
    best = min(best, t / frames)
hs = a.host_stats()
print(mode, "WGS", os.environ.get("GR_LIGHTING_WGS_PER_CU", "-"), "frame us %.1f" % (1e6 * best))
a.close()
