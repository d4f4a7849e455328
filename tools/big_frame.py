"""Config-5-sized frame (7680x4320, 4096 lights) on ONE GPU: memory, grid limits, frame time; plus a 2-band emulation of the
same frame (two executor instances, local exchange) compared bit for bit -- what each rank of a 2-GPU run would do."""
import os, sys, time, threading
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import app as gapp, capi, multigpu, synth
w, h = int(sys.argv[1]) if len(sys.argv) > 1 else 7680, int(sys.argv[2]) if len(sys.argv) > 2 else 4320
t0 = time.time()
cam = synth.Camera(w, h); gbuf = synth.make_gbuffer(cam); descs = synth.make_lights(cam, 4096)
print(f"synth {w}x{h}: {time.time() - t0:.1f} s")

def make(**kw):
    a = gapp.Application(w, h, **kw)
    a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
    return a

a = make()
a.render_frames(6, sync=True)
t0 = time.perf_counter(); a.render_frames(50, sync=True); t = (time.perf_counter() - t0) / 50
print(f"whole frame on one GPU: {1e6 * t:.0f} us/frame = {w * h / t / 1e9:.1f} Gpx/s")
want = a.read_backbuffer().copy()
a.close()

world = 2
lib = capi.load_library()
lib.gr_copy.argtypes = [C.c_void_p] * 4 + [C.c_size_t]; lib.gr_sync.argtypes = [C.c_void_p, C.c_void_p]
apps = [make(strip_index=r, strip_count=world) for r in range(world)]
ctx = apps[0].lib.gra_get_kernel_context(apps[0].handle)
local = multigpu.LocalExchange(world, lambda d, s, n, st: lib.gr_copy(ctx, st, d, s, n), lambda st: lib.gr_sync(ctx, st))
got = [None] * world
def run(r):
    apps[r].set_exchange_callback(local.for_rank(r))
    for _ in range(56):
        apps[r].render_frames(1)
    got[r] = apps[r].read_backbuffer().copy()
ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
for r in range(world):
    print(f"rank {r} of {world}: bands identical to the whole frame:", bool(np.array_equal(got[r], want)), apps[r].strip_plan()["lighting"])
