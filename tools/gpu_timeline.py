import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import app as gapp, synth
w,h=3840,2160
cam=synth.Camera(w,h); gbuf=synth.make_gbuffer(cam); descs=synth.make_lights(cam,4096)
a=gapp.Application(w,h); a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
k=a.kernel_context()
a.render_frames(30, sync=True)
k.timing_enable(True); k.timing_set_filter(None); k.timing_reset()
a.render_frames(12, sync=True)
k.timing_query()
