import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import app as gapp, synth
W,H=3840,2160
cam=synth.Camera(W,H); gbuf=synth.make_gbuffer(cam); descs=synth.make_lights(cam,4096)
a=gapp.Application(W,H); a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
for n in (1,8):
    a.render_frames(n, sync=True)
    bb=a.read_backbuffer()
    print(n, 'unique', np.unique(bb[::16,::16,:3])[:20], 'mean', bb[...,:3].mean(), 'lum', a.read('average-luminance').view(np.float32))
    hdr=a.read('HDR-main').view(np.float16).astype(np.float32)
    print(' hdr mean', np.nanmean(hdr[...,:3]), 'max', np.nanmax(hdr[...,:3]), 'nan', np.isnan(hdr).sum(), 'tm', a.read('tonemapped')[::500,::500,:3].reshape(-1)[:12] if False else '')
    u0=a.read('upsample-0').view(np.float16).astype(np.float32); print(' u0 mean', np.nanmean(u0[...,:3]), np.isnan(u0).sum())
