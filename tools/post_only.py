import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import app as gapp, synth
w,h=3840,2160
a=gapp.Application(w,h,lighting=False); a.upload_hdr(synth.make_hdr(w,h))
k=a.kernel_context()
a.render_frames(20, sync=True)
t0=time.perf_counter(); a.render_frames(200, sync=True); t=time.perf_counter()-t0
print("post-only frame us", 1e6*t/200)
k.timing_enable(True); k.timing_set_filter(None); k.timing_reset()
a.render_frames(20, sync=True)
q=k.timing_query()
print({n:round(1000*ms/c,1) for n,(c,ms) in q.items()})
