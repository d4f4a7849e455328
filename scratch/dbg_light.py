import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from granite_amd import capi, synth
from gpu_scene import Scene
gr=capi.Context(0)
out={}
for (w,h,n) in ((320,180,200),(480,270,256)):
    sc=Scene(w,h,n)
    dev=sc.build_clusters_gpu(gr)
    args,imgs=sc.lighting_args(gr,dev,capi.LIGHTING_CLUSTERED_BIT)
    gr.check(gr.lib.gr_lighting(gr.handle,None,args)); gr.sync()
    out[f'hdr_{w}_{h}_{n}']=imgs['hdr'].download()
np.savez(os.path.join(ROOT,'gpurun_out','dbg_light.npz'),**out)
print('saved')
