"""Lighting launches back to back on ONE stream against the same launches dealt to TWO streams in turn (so that launch N + 1 may start while launch N
drains), the launch alone on the machine: is the sum of two launches in flight the sum of their rates?  (profiles/r06_front_stream_alternation.txt)
Usage (GPU box): python tools/lighting_two_streams.py [W H]   -- GR_LIGHTING_WGS_PER_CU selects the residency cap as for tools/lighting_only.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from granite_amd import capi, synth
from gpu_scene import Scene
hip = C.CDLL("libamdhip64.so")
gr = capi.Context(0)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
sc = Scene(W, H, 4096); dev = sc.build_clusters_gpu(gr)
flags = capi.LIGHTING_DIRECTIONAL_BIT | capi.LIGHTING_CLUSTERED_BIT | capi.LIGHTING_AMBIENT_FALLBACK_BIT
sets = [sc.lighting_args(gr, dev, flags, alias_emissive=False) for _ in range(2)]  # two targets: launches in flight together never share one
streams = []
for _ in range(2):
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0  # hipStreamNonBlocking
    streams.append(s)
N = 40


def run(two):
    for i in range(6):
        gr.check(gr.lib.gr_lighting(gr.handle, streams[i & 1] if two else streams[0], sets[i & 1][0]))
    for s in streams:
        hip.hipStreamSynchronize(s)
    t0 = time.perf_counter()
    for i in range(N):
        gr.check(gr.lib.gr_lighting(gr.handle, streams[i & 1] if two else streams[0], sets[i & 1][0]))
    for s in streams:
        hip.hipStreamSynchronize(s)
    return (time.perf_counter() - t0) / N * 1e6


for rnd in range(3):
    print("%dx%d wgs/CU %s: one stream %.1f us per launch, two streams in turn %.1f us per launch" % (W, H, os.environ.get('GR_LIGHTING_WGS_PER_CU', '-'), run(False), run(True)))
gr.close()
