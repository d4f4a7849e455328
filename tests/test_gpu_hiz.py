"""GPU parity: the single-launch depth hierarchy (gr_hiz) vs the CPU restatement of hiz.comp, bit for bit."""
import numpy as np
import pytest

from granite_amd import capi, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gr():
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


def depth_image(w, h, seed=7):
    rng = np.random.default_rng(seed)
    # smooth ramp + noise + a few far / near outliers, strictly inside (0, 1)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    d = 0.2 + 0.6 * (0.5 + 0.5 * np.sin(x * 0.013 + y * 0.007)) + 0.05 * rng.random((h, w), dtype=np.float32)
    d[rng.integers(0, h, 64), rng.integers(0, w, 64)] = 0.999
    d[rng.integers(0, h, 64), rng.integers(0, w, 64)] = 0.001
    return np.clip(d, 1e-4, 0.9999).astype(np.float32)


def z_transform(w, h):
    cam = synth.Camera(w, h)
    return orc.hiz_z_transform(cam.render_params()[48:64])


def run(gr, depth, zt, output_downsample=False):
    h, w = depth.shape
    img = capi.DeviceImage(gr, w, h, capi.FORMAT_D32_SFLOAT).upload(depth)
    chain, counter, layout = gr.hiz(img, zt, output_downsample)
    gr.sync()
    assert counter.download(np.uint32)[0] == 0, "the pass must leave its counter at zero for the next frame"
    # second launch into the same chain + counter: same result, counter reset worked
    first = [l.copy() for l in gr.read_mip_chain(chain, layout)]
    gr.hiz(img, zt, output_downsample, chain=chain, counter=counter)
    gr.sync()
    second = gr.read_mip_chain(chain, layout)
    for a, b in zip(first, second):
        np.testing.assert_array_equal(a, b)
    return first, layout


@pytest.mark.parametrize("size", [(64, 64), (100, 60), (256, 64), (257, 131), (1000, 600), (1920, 1080), (1001, 333)])
@pytest.mark.parametrize("output_downsample", [False, True])
def test_chain_matches_oracle_bit_for_bit(gr, size, output_downsample):
    w, h = size
    depth = depth_image(w, h)
    zt = z_transform(w, h)
    want = orc.hiz(depth, zt, output_downsample)
    got, layout = run(gr, depth, zt, output_downsample)
    lay = orc.hiz_layout(w, h, output_downsample)
    assert (layout["chain_w"], layout["chain_h"], layout["levels"]) == (lay["chain_w"], lay["chain_h"], lay["levels"])
    assert len(got) == len(want)
    for level, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape, level
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=f"level {level} of {size}")


def test_full_size_properties_4k(gr):
    """3840 x 2160 (chain 3840 x 2176, 11 levels): level 0 is the transformed input with the edge repeated; every texel of
    every coarser level bounds its footprint; the top of the chain is the global maximum."""
    w, h = 3840, 2160
    depth = depth_image(w, h, seed=11)
    zt = z_transform(w, h)
    got, layout = run(gr, depth, zt)
    assert (layout["chain_w"], layout["chain_h"], layout["levels"]) == (3840, 2176, 11)
    num = zt[0] * depth + zt[2]
    den = zt[1] * depth + zt[3]
    lin = np.minimum(num / den, np.float32(1e30)).astype(np.float32)
    np.testing.assert_array_equal(got[0][:h, :w], lin)
    np.testing.assert_array_equal(got[0][h:, :w], np.broadcast_to(lin[-1], (16, w)))
    for l in range(1, 7):
        fine = got[l - 1]
        want = fine.reshape(fine.shape[0] // 2, 2, fine.shape[1] // 2, 2).max(axis=(1, 3))
        np.testing.assert_array_equal(got[l], want)
    top = got[-1]
    assert top.shape == (2, 3) and top.max() == lin.max()
    for l in range(7, 11):
        assert got[l].max() == lin.max() and got[l].min() >= lin.min()
    want = orc.hiz(depth, zt)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("mode", [1, 2])
def test_depth_hierarchy_pass_in_the_frame_graph(mode):
    """setup_depth_hierarchy_pass on the executor: the pass runs on the frame front after lighting (which publishes
    "depth-main"), several pipelined frames in a row, and the chain equals the oracle's for the uploaded depth."""
    from granite_amd import app as gapp
    w, h = 480, 270
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    a = gapp.Application(w, h, depth_hierarchy=mode)
    a.set_render_parameters(cam.render_params())
    a.set_lights(synth.make_lights(cam, 300))
    a.upload_gbuffer(gbuf)
    g = a.graph()
    order = [p["name"] for p in g["passes"]]
    assert order.index("lighting-main") < order.index("depth-hiz") < order.index("tonemap")
    a.render_frames(5)
    a.sync()
    plain = gapp.Application(w, h)
    plain.set_render_parameters(cam.render_params())
    plain.set_lights(synth.make_lights(cam, 300))
    plain.upload_gbuffer(gbuf)
    plain.render_frames(5)
    np.testing.assert_array_equal(a.read_backbuffer(), plain.read_backbuffer())
    plain.close()
    got = a.read_mip_chain("depth-hiz")
    want = orc.hiz(gbuf["depth"], orc.hiz_z_transform(cam.render_params()[48:64]), output_downsample=(mode == 2))
    assert len(got) == len(want) == 8 - (mode == 2)
    for level, (x, y) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32), err_msg=f"level {level}")
    assert a.read("depth-hiz-counter").view(np.uint32)[0] == 0
    a.close()
