"""GPU, BASELINE.json's full size (3840x2160, 4096 clustered point + spot lights): size-independent properties instead of
the CPU oracle (which needs seconds per frame at this size).

  * render areas: lighting the frame in uneven row bands (gr_lighting_args.rows) gives the very bytes of the whole-frame
    launch -- every pixel is shaded from full-image coordinates, tile-level light supersets only ever add exact zeros;
  * clustered == unclustered: the clustered result equals a launch whose cluster bitmask has every light set in every cell
    (brute force over all 4096 lights), i.e. the conservative culling removes nothing but zeros, at full size;
  * the frame is reproducible run to run (no order-dependent accumulation anywhere on the path).
"""
import numpy as np
import pytest

from granite_amd import app as gapp, capi, synth
from gpu_scene import Scene

pytestmark = pytest.mark.gpu

ALL = capi.LIGHTING_DIRECTIONAL_BIT | capi.LIGHTING_CLUSTERED_BIT | capi.LIGHTING_AMBIENT_FALLBACK_BIT
W, H, LIGHTS = 3840, 2160, 4096


@pytest.fixture(scope="module")
def scene4k(gr):
    sc = Scene(W, H, LIGHTS)
    dev = sc.build_clusters_gpu(gr)
    return sc, dev


def light(gr, sc, dev, rows=None, flags=ALL, dev_override=None):
    args, imgs = sc.lighting_args(gr, dev_override or dev, flags, alias_emissive=False)
    if rows is None:
        gr.check(gr.lib.gr_lighting(gr.handle, None, args))
    else:
        for first, count in rows:
            args.rows[0], args.rows[1] = first, count
            gr.check(gr.lib.gr_lighting(gr.handle, None, args))
    gr.sync()
    return imgs["hdr"].download()


def test_4k_row_bands_equal_whole_frame(gr, scene4k):
    sc, dev = scene4k
    whole = light(gr, sc, dev)
    assert not (whole == 0x7e00).all(axis=2).any(), "every pixel of the NaN-filled target must have been written"
    bands = [(0, 1), (1, 6), (7, 534), (541, 1079), (1620, 540)]  # uneven, not multiples of the 8-row tile
    assert sum(c for _, c in bands) == H
    np.testing.assert_array_equal(light(gr, sc, dev, rows=bands), whole)


def test_4k_clustered_equals_all_lights_everywhere(gr, scene4k):
    sc, dev = scene4k
    clustered = light(gr, sc, dev, flags=capi.LIGHTING_CLUSTERED_BIT)
    n32 = (sc.n + 31) // 32
    words = np.full(sc.res[0] * sc.res[1] * n32, 0xffffffff, np.uint32)
    if sc.n % 32:
        words.reshape(-1, n32)[:, -1] = (1 << (sc.n % 32)) - 1
    full_mask = capi.DeviceBuffer(gr, words.nbytes).upload(words)
    full_range = capi.DeviceBuffer(gr, sc.res[2] * 8).upload(np.tile(np.array([0, sc.n - 1], np.uint32), sc.res[2]))
    # a band of 64 rows keeps the brute-force launch (4096 candidates per tile) short
    band = [(1000, 64)]
    brute = light(gr, sc, dev, rows=band, flags=capi.LIGHTING_CLUSTERED_BIT, dev_override={**dev, "bitmask": full_mask, "range": full_range})
    np.testing.assert_array_equal(brute[1000:1064], clustered[1000:1064])


def test_4k_frames_are_reproducible():
    cam = synth.Camera(W, H)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, LIGHTS)
    outs = []
    for _ in range(2):
        a = gapp.Application(W, H)
        a.set_render_parameters(cam.render_params())
        a.set_lights(descs)
        a.upload_gbuffer(gbuf)
        a.render_frames(8, sync=False)
        a.sync()
        outs.append((a.read_backbuffer().copy(), a.read("average-luminance").copy()))
        a.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert len(np.unique(outs[0][0][::16, ::16, :3])) > 8  # a real (if bright: exposure has not adapted in 8 frames) image


def test_4k_post_launchers_in_row_bands_equal_whole_image(gr):
    """Render areas of the post launchers (gr_*_rows): threshold, 2:1 and generic downsample, 1:2 upsample and tonemap run
    in uneven row bands must write the very bytes of the whole-image launch, at 3840x2160 (all levels exact 2x) and at an
    odd size (generic tent path)."""
    from test_gpu_post import F16
    for w, h in ((W, H), (1366, 769)):
        hdr_bits = synth.make_hdr(w, h)
        hdr = capi.DeviceImage(gr, w, h, F16).upload(hdr_bits)
        sz = [orc_level(w, h, s) for s in (0.5, 0.25, 0.125)]

        def bands(height):
            cuts = [0, 1, height // 3 + 1, (2 * height) // 3 - 2, height]
            return [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(4) if cuts[i + 1] > cuts[i]]

        def both(launch, out_w, out_h, fmt=F16):
            whole = capi.DeviceImage(gr, out_w, out_h, fmt)
            launch(whole, None)
            split = capi.DeviceImage(gr, out_w, out_h, fmt)
            for b in bands(out_h):
                launch(split, b)
            gr.sync()
            np.testing.assert_array_equal(split.download(), whole.download())
            return whole

        t = both(lambda out, rows: gr.bloom_threshold(hdr, out, None, rows=rows), *sz[0])
        d0 = both(lambda out, rows: gr.bloom_downsample(t, out, rows=rows), *sz[1])
        d1 = both(lambda out, rows: gr.bloom_downsample(d0, out, rows=rows), *sz[2])
        u0 = both(lambda out, rows: gr.bloom_upsample(d1, out, rows=rows), *sz[1])
        both(lambda out, rows: gr.tonemap(hdr, u0, out, None, rows=rows), w, h, capi.FORMAT_R8G8B8A8_SRGB)


def orc_level(w, h, scale):
    from oracle import oracle as orc
    return orc.level_size(w, h, scale)


def test_upload_batch_writes_every_range(gr):
    """gr_upload_batch: several pinned-host -> HBM ranges in one kernel (aligned and unaligned to 16 bytes, odd dword
    counts) land byte for byte; other bytes of the destination stay untouched."""
    import ctypes as C

    lib = gr.lib
    rng = np.random.default_rng(7)
    sizes = [196608, 4 * 1021, 512, 32768, 4]
    dst = capi.DeviceBuffer(gr, sum(sizes) + 64 * len(sizes))
    dst.upload(np.full(dst.nbytes, 0xEE, np.uint8))
    ranges = (capi.UploadRange * len(sizes))()
    keep, expect, off = [], np.full(dst.nbytes, 0xEE, np.uint8), 4
    for i, n in enumerate(sizes):
        hptr = C.c_void_p()
        gr.check(lib.gr_alloc_host(gr.handle, n, C.byref(hptr)))
        data = rng.integers(0, 256, n, dtype=np.uint8)
        C.memmove(hptr.value, data.ctypes.data, n)
        keep.append(hptr)
        ranges[i] = capi.UploadRange(dst.ptr + off, hptr.value, n)
        expect[off:off + n] = data
        off += n + 60  # keeps dword alignment, breaks 16-byte alignment
    gr.check(lib.gr_upload_batch(gr.handle, None, ranges, len(sizes)))
    gr.sync()
    np.testing.assert_array_equal(dst.download(np.uint8), expect)
    for hptr in keep:
        gr.check(lib.gr_free_host(gr.handle, hptr))
