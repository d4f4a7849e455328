import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import app as gapp, synth
w,h=3840,2160
cam=synth.Camera(w,h); gbuf=synth.make_gbuffer(cam); descs=synth.make_lights(cam,4096)
a=gapp.Application(w,h); a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
k=a.kernel_context()
a.render_frames(30, sync=True)
for mode in ("none","lighting","none","lighting"):
    k.timing_enable(mode!="none"); k.timing_set_filter(None if mode=="none" else mode); k.timing_reset()
    t0=time.perf_counter(); a.render_frames(300, sync=True); t=time.perf_counter()-t0
    print(mode, "frame us", round(1e6*t/300,1))
