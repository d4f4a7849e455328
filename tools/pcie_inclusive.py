"""PCIe-inclusive frame rate: the time to hand a 4K G-buffer over from host memory (pageable numpy arrays, as the harness
boundary takes them) plus one frame, next to the HBM-resident frame time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import app as gapp, synth
w, h = 3840, 2160
cam = synth.Camera(w, h); gbuf = synth.make_gbuffer(cam); descs = synth.make_lights(cam, 4096)
a = gapp.Application(w, h)
a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
a.render_frames(8, sync=True)
t0 = time.perf_counter(); a.render_frames(100, sync=True); frame = (time.perf_counter() - t0) / 100
ups = []
for _ in range(5):
    t0 = time.perf_counter(); a.upload_gbuffer(gbuf); ups.append(time.perf_counter() - t0)
up = min(ups)
nbytes = sum(v.nbytes for v in gbuf.values())
print(f"frame {1e3 * frame:.3f} ms; G-buffer upload {nbytes / 1e6:.0f} MB in {1e3 * up:.2f} ms = {nbytes / up / 1e9:.1f} GB/s; PCIe-inclusive {w * h / (frame + up) / 1e6:.0f} Mpx/s vs resident {w * h / frame / 1e6:.0f} Mpx/s")
