"""Cross-stream GPU timeline of pipelined frames: every launch bracketed with hipEvents (GR_TIMING_DUMP=<file> makes the kernel library
write name, start, stop per launch).  Usage (GPU box): GR_TIMING_DUMP=gpurun_out/timeline.txt python tools/gpu_timeline.py [config4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from granite_amd import app as gapp, synth
w,h=3840,2160
cam=synth.Camera(w,h); gbuf=synth.make_gbuffer(cam); descs=synth.make_lights(cam,4096, spot_fraction=0.25)
if len(sys.argv) > 1 and sys.argv[1] == "config4":
    a=gapp.Application(w,h, pre_aa=gapp.POST_AA_TAA_HIGH, post_aa=gapp.POST_AA_SMAA_ULTRA)
    a.set_camera(np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16))
    a.set_lights(descs); a.upload_gbuffer(gbuf, synth.make_motion_vectors(w, h)); a.set_camera_motion((0.01, 0.0, 0.0))
else:
    a=gapp.Application(w,h); a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
k=a.kernel_context()
a.render_frames(30, sync=True)
k.timing_enable(True); k.timing_set_filter(None); k.timing_reset()
a.render_frames(12, sync=True)
k.timing_query()
