"""The reference's z-binning timing vehicle (tests/z_binning_test.cpp:56-96) on the HIP executor: 4096 light z-intervals, all empty
(uvec2(1000000000, 0)), 4096 ranges, 1000 dispatches of the z-range kernel back to back on one stream; prints the time per
iteration like the reference's "Time per iteration: %.3f ms."  A second line uses the benchmark scene's real intervals."""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from granite_amd import capi

NUM_INPUTS = NUM_RANGES = 4 * 1024
ITERATIONS = 1000
gr = capi.Context(0)


def run(label, inputs):
    src = capi.DeviceBuffer(gr, inputs.nbytes).upload(inputs)
    out = capi.DeviceBuffer(gr, NUM_RANGES * 8)
    push = capi.PushZRange(NUM_INPUTS, (NUM_INPUTS + 127) // 128, NUM_RANGES)
    for _ in range(10):
        gr.check(gr.lib.gr_cluster_z_range(gr.handle, None, src.ptr, out.ptr, push))
    gr.sync()
    t0 = time.perf_counter()
    for _ in range(ITERATIONS):
        gr.check(gr.lib.gr_cluster_z_range(gr.handle, None, src.ptr, out.ptr, push))
    gr.sync()
    t = time.perf_counter() - t0
    print(f"{label}: Time per iteration: {1e3 * t / ITERATIONS:.3f} ms.")


empty = np.empty((NUM_INPUTS, 2), np.uint32)
empty[:, 0], empty[:, 1] = 1000000000, 0
run("empty intervals (the reference's input)", empty)
rng = np.random.default_rng(1)
lo = rng.integers(0, NUM_RANGES - 64, NUM_INPUTS)
real = np.stack([lo, lo + rng.integers(1, 64, NUM_INPUTS)], axis=1).astype(np.uint32)
run("random intervals up to 64 slices long", real)
