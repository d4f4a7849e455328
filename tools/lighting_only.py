import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np
from granite_amd import capi, synth
from gpu_scene import Scene
gr=capi.Context(0)
W,H=(int(sys.argv[1]),int(sys.argv[2])) if len(sys.argv)>2 else (3840,2160)
SCENE=sys.argv[3] if len(sys.argv)>3 else os.environ.get('GR_SCENE','default')
sc=Scene(W,H,4096,scene=SCENE); dev=sc.build_clusters_gpu(gr)
flags=capi.LIGHTING_DIRECTIONAL_BIT|capi.LIGHTING_CLUSTERED_BIT|capi.LIGHTING_AMBIENT_FALLBACK_BIT
args,imgs=sc.lighting_args(gr,dev,flags,alias_emissive=False)
for _ in range(5): gr.check(gr.lib.gr_lighting(gr.handle,None,args))
gr.sync()
gr.timing_enable(True); gr.timing_reset()
for _ in range(30): gr.check(gr.lib.gr_lighting(gr.handle,None,args))
gr.sync()
q=gr.timing_query()
print(f'{W}x{H}', SCENE, os.environ.get('GR_LIGHTING_PX','2'), os.environ.get('GR_LIGHTING_WGS_PER_CU','-'), {k:round(1000*ms/c,1) for k,(c,ms) in q.items()})
