"""Per-tile timeline of the lighting kernel: where the launch time goes across tiles, XCDs and the tail.

Needs the measurement build (records are written only there):
    make -C granite_amd/csrc OUT=../lib_stamp EXTRA_lighting=-DLV_STAMP
    GRANITE_LIB_DIR=lib_stamp python tools/lighting_stamps.py [out.txt]      
One record per wave tile (16 x 8 pixels): {start, end} on the 100 MHz s_memrealtime counter, shader cycles in between, XCC_ID | HW_ID.
The stamps cost a few s_memtime / s_waitcnt per tile; launch times quoted elsewhere come from the un-stamped build."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from granite_amd import capi
from gpu_scene import Scene

w, h, nl = 3840, 2160, 4096
gr = capi.Context(0)
sc = Scene(w, h, nl); dev = sc.build_clusters_gpu(gr)
flags = capi.LIGHTING_DIRECTIONAL_BIT | capi.LIGHTING_CLUSTERED_BIT | capi.LIGHTING_AMBIENT_FALLBACK_BIT
args, imgs = sc.lighting_args(gr, dev, flags, alias_emissive=False)
tiles_x, tiles_y = (w + 15) // 16, (h + 7) // 8
# the static grid pads the row of workgroups to 4 waves: index space = blocks * 4
records = ((w + 63) // 64) * 4 * tiles_y
buf = capi.DeviceBuffer(gr, 2 * records * 16)  # two records per tile: {start, end, cycles, where} and the cycles to each phase mark
fn = gr.lib.gr_debug_lighting_stamps
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; fn.restype = C.c_int
for _ in range(5):
    gr.check(gr.lib.gr_lighting(gr.handle, None, args))
gr.sync()
out = []
def emit(*a):
    line = ' '.join(str(x) for x in a); print(line); out.append(line)
emit('lighting tile timeline, %dx%d, %d lights, static grid in screen order' % (w, h, nl))
for rep in range(3):
    buf.upload(np.zeros(2 * records * 4, np.uint32))
    gr.check(fn(gr.handle, buf.ptr, records)); gr.check(gr.lib.gr_lighting(gr.handle, None, args)); gr.sync(); gr.check(fn(gr.handle, None, 0))
    both = buf.download(np.uint32).reshape(-1, 2, 4)
    if os.environ.get('LV_STAMP_DUMP'):  # raw records for tools/lighting_residency.py (offline: wave slots over time)
        np.savez_compressed('%s_%d.npz' % (os.environ['LV_STAMP_DUMP'], rep), records=both)
    r, marks = both[:, 0], both[:, 1]
    valid = (r[:, 0] != 0) | (r[:, 1] != 0)
    r, marks = r[valid], marks[valid]
    t0 = r[:, 0].astype(np.int64); t1 = r[:, 1].astype(np.int64)
    base = t0.min(); t0 -= base; t1 -= base
    dur = (t1 - t0) * 0.01  # us
    span = t1.max() * 0.01
    xcc = r[:, 3] >> 28
    ghz = r[:, 2] / np.maximum(t1 - t0, 1) / 10.0 * 1e-0 / 100.0  # cycles per 10 ns tick -> GHz
    emit('--- launch %d: %d tiles, first start -> last end %.1f us; shader clock (median over tiles) %.2f GHz' % (rep, len(r), span, np.median(r[:, 2] / np.maximum((t1 - t0), 1)) / 10.0))
    emit('  tile duration us: min %.1f  p5 %.1f  p25 %.1f  median %.1f  p75 %.1f  p95 %.1f  max %.1f  mean %.2f' %
         (dur.min(), *np.percentile(dur, [5, 25, 50, 75, 95]), dur.max(), dur.mean()))
    hist, edges = np.histogram(dur, bins=[0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 32, 48, 1e9])
    emit('  histogram (us bucket: tiles): ' + '  '.join('%g-%s: %d' % (edges[i], ('%g' % edges[i + 1]) if edges[i + 1] < 1e8 else 'inf', hist[i]) for i in range(len(hist))))
    ends = np.sort(t1) * 0.01
    for frac in (0.5, 0.9, 0.95, 0.99):
        emit('  %4.0f %% of the tiles done at %.1f us (%.1f us before the end)' % (100 * frac, ends[int(frac * len(ends)) - 1], span - ends[int(frac * len(ends)) - 1]))
    emit('  per XCD: tiles, busy wave-us, first start, last end')
    for x in range(16):
        m = xcc == x
        if m.any():
            emit('    xcc %2d: %6d tiles  %9.0f wave-us  start %6.1f  end %6.1f' % (x, m.sum(), dur[m].sum(), t0[m].min() * 0.01, t1[m].max() * 0.01))
    walked = marks[:, 1] != 0
    if walked.any():
        total = r[walked, 2].astype(np.float64); m = marks[walked].astype(np.float64)
        parts = [('loads, position, range[] requested, cells, sphere, N, V', m[:, 0]),
                 ('window, words, material terms, records requested, directional quad', m[:, 1] - m[:, 0]),
                 ('cull + stage, all chunks (first: wait for the records)', m[:, 2]), ('walks, all chunks', m[:, 3]),
                 ('blend + store', total - m[:, 1] - m[:, 2] - m[:, 3])]
        emit('  shader cycles per tile by phase (tiles with lights: %d; mean %.0f cycles):' % (walked.sum(), total.mean()))
        for name, c in parts:
            emit('    %-72s mean %6.0f  (%4.1f %%)  median %6.0f' % (name, c.mean(), 100.0 * c.mean() / total.mean(), np.median(c)))
    slots = len(np.unique(r[:, 3]))
    emit('  wave slots seen (distinct XCC | HW_ID): %d; sum of tile time / (slots x span) = %.3f' % (slots, dur.sum() / (slots * span)))
if len(sys.argv) > 1:
    open(sys.argv[1], 'w').write('\n'.join(out) + '\n')
