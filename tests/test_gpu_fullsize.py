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
