"""GPU parity for the spatial upscaling passes (FSR 1.0 EASU + RCAS as setup_after_post_chain_upscaling records them).

EASU is built from correctly rounded fp32 / fp16 operations and integer bit tricks, so both shader variants are required to
be bit-exact against the oracle; RCAS into an *_SRGB target carries the +-1 LSB of the device's pow in the encode."""
import numpy as np
import pytest

from granite_amd import capi
from oracle import oracle as orc
from util import assert_rgba8_close

pytestmark = pytest.mark.gpu

UNORM, SRGB = capi.FORMAT_R8G8B8A8_UNORM, capi.FORMAT_R8G8B8A8_SRGB


@pytest.fixture(scope="module")
def gr():
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


def image(w, h, kind, seed=3):
    r = np.random.default_rng(seed)
    if kind == "noise":
        img = r.integers(0, 256, (h, w, 4), dtype=np.uint8)
    elif kind == "blocks":  # flat areas (zero gradients: 0 * inf in the half path), hard edges, black and white
        coarse = r.choice(np.array([0, 0, 255, 17, 128, 200], np.uint8), ((h + 4) // 5, (w + 6) // 7, 4))
        img = np.repeat(np.repeat(coarse, 5, axis=0), 7, axis=1)[:h, :w].copy()
    else:  # smooth ramps + a diagonal edge
        y, x = np.mgrid[0:h, 0:w]
        img = np.zeros((h, w, 4), np.uint8)
        img[..., 0] = (x * 255) // max(w - 1, 1)
        img[..., 1] = (y * 255) // max(h - 1, 1)
        img[..., 2] = np.where(x * h > y * w, 220, 30)
    img[..., 3] = 255
    return img


def upscale(gr, src, ow, oh, fp16, out_fmt=UNORM):
    h, w = src.shape[:2]
    a = capi.DeviceImage(gr, w, h, SRGB).upload(src)
    b = capi.DeviceImage(gr, ow, oh, out_fmt)
    gr.fsr_upscale(a, b, fp16)
    gr.sync()
    return b.download()


@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("case", [(160, 90, 320, 180, "ramp"), (160, 90, 240, 135, "noise"), (200, 120, 333, 201, "blocks"),
                                  (64, 64, 64, 64, "noise"), (7, 5, 30, 22, "noise"), (320, 180, 256, 144, "blocks")])
def test_easu_bit_exact(gr, fp16, case):
    w, h, ow, oh, kind = case
    src = image(w, h, kind)
    want = orc.fsr_easu(src, ow, oh, fp16)
    got = upscale(gr, src, ow, oh, fp16)
    np.testing.assert_array_equal(got, want)
    # an *_SRGB output attachment receives the same gamma-space bytes (decode in the shader, encode in the store)
    assert_rgba8_close(upscale(gr, src, ow, oh, fp16, SRGB), orc.fsr_easu(src, ow, oh, fp16, target_srgb=True), 1, what="easu srgb target")


@pytest.mark.parametrize("kind", ["ramp", "noise", "blocks"])
@pytest.mark.parametrize("srgb", [False, True])
def test_rcas(gr, kind, srgb):
    w, h = 253, 127
    src = image(w, h, kind, seed=9)
    sharp = orc.fsr_rcas_sharpness(0.5)
    a = capi.DeviceImage(gr, w, h, UNORM).upload(src)
    b = capi.DeviceImage(gr, w, h, SRGB if srgb else UNORM)
    gr.fsr_sharpen(a, b, sharp)
    gr.sync()
    want = orc.fsr_rcas(src, sharp, srgb)
    if srgb:
        assert_rgba8_close(b.download(), want, 1, what="rcas srgb")
    else:
        np.testing.assert_array_equal(b.download(), want)


def test_full_size_1440p_to_4k_properties(gr):
    """2560x1440 -> 3840x2160 (the sizes a resolution_scale of 2/3 gives the viewer): a constant image stays constant through
    both passes, every output stays inside the range of its 12-tap footprint (de-ringing), the two EASU variants agree to
    fp16 precision, and a sample of rows equals the oracle."""
    w, h, ow, oh = 2560, 1440, 3840, 2160
    src = image(w, h, "ramp", seed=1)
    g32, g16 = upscale(gr, src, ow, oh, False), upscale(gr, src, ow, oh, True)
    diff = np.abs(g32.astype(int) - g16.astype(int))
    assert diff[..., :2].max() <= 1            # smooth ramps: half precision costs at most one LSB
    assert diff.max() <= 16 and (diff > 1).mean() < 1e-3  # along the hard diagonal edge the half variant may land elsewhere
    band = orc.fsr_easu(src, ow, oh, True)[1000:1016]
    np.testing.assert_array_equal(g16[1000:1016], band)
    flat = np.full((h, w, 4), 255, np.uint8)
    flat[..., :3] = (10, 128, 250)
    up = upscale(gr, flat, ow, oh, True)
    assert (up.reshape(-1, 4) == up[0, 0]).all() and tuple(up[0, 0]) == (10, 128, 250, 255)
    lo, hi = src[..., :3].min(), src[..., :3].max()
    assert g16[..., :3].min() >= lo and g16[..., :3].max() <= hi


def test_argument_validation(gr):
    a = capi.DeviceImage(gr, 16, 16, UNORM)
    b = capi.DeviceImage(gr, 32, 32, UNORM)
    f = capi.DeviceImage(gr, 32, 32, capi.FORMAT_R16G16B16A16_SFLOAT)
    with pytest.raises(capi.GraniteHipError):
        gr.fsr_upscale(a, f)
    with pytest.raises(capi.GraniteHipError):
        gr.fsr_sharpen(a, b, 0.7)  # sizes differ
    with pytest.raises(capi.GraniteHipError):
        gr.fsr_upscale(a, a)


@pytest.mark.parametrize("sharpen,fp16,post_aa", [(True, True, 0), (False, True, 0), (True, False, 1)])
def test_resolution_scale_in_the_frame_graph(sharpen, fp16, post_aa):
    """viewer_config resolutionScale: G-buffer, lighting, post chain (and FXAA) at 2/3 size, then
    setup_after_post_chain_upscaling to the backbuffer.  The low-resolution part equals a plain application of that size;
    the upscaled frame equals the oracle's EASU (+ RCAS) of that application's output."""
    from granite_amd import app as gapp, synth
    W, H = 720, 405
    a = gapp.Application(W, H, resolution_scale=2.0 / 3.0, resolution_scale_sharpen=sharpen, fsr_fp16=fp16,
                         post_aa=gapp.POST_AA_FXAA if post_aa else gapp.POST_AA_NONE)
    w, h = a.render_size()
    assert (w, h) == (480, 270)
    cam = synth.Camera(w, h)
    gbuf, descs = synth.make_gbuffer(cam), synth.make_lights(cam, 400)
    low = gapp.Application(w, h, post_aa=gapp.POST_AA_FXAA if post_aa else gapp.POST_AA_NONE)
    for app in (a, low):
        app.set_render_parameters(cam.render_params())
        app.set_lights(descs)
        app.upload_gbuffer(gbuf)
        app.render_frames(5)
    low_frame = low.read_backbuffer()
    source = "post-aa-output" if post_aa else "tonemapped"
    np.testing.assert_array_equal(a.read(source), low_frame)
    up = orc.fsr_easu(low_frame, W, H, fp16)
    if sharpen:
        np.testing.assert_array_equal(a.read("post-scale-output-scale"), up)
        assert_rgba8_close(a.read_backbuffer(), orc.fsr_rcas(up, orc.fsr_rcas_sharpness(0.5), srgb=True), 1, what="sharpened frame")
    else:
        np.testing.assert_array_equal(a.read_backbuffer(), up)
    a.close()
    low.close()
