"""GPU parity for the HDR10 output encode (pq10_encode.frag) and the HDR10 frame graph."""
import numpy as np
import pytest

from granite_amd import app as gapp, capi, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def channels(words):
    return np.stack([(words >> s) & 1023 for s in (0, 10, 20)], axis=-1).astype(int)


def test_pq10_kernel_matches_oracle():
    gr = capi.Context(0)
    w, h = 333, 77
    hdr = synth.make_hdr(w, h, 5)
    ui = np.random.default_rng(2).integers(0, 256, (h, w, 4), dtype=np.uint8)
    ui[: h // 2] = (0, 0, 0, 255)
    conversion = orc.rec709_to_display()
    dh = capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16B16A16_SFLOAT).upload(hdr)
    du = capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8B8A8_SRGB).upload(ui)
    do = capi.DeviceImage(gr, w, h, capi.FORMAT_A2B10G10R10_UNORM_PACK32)
    for mll in (1000.0, 400.0):
        gr.pq10_encode(dh, du, do, conversion, 500.0, 400.0, mll)
        gr.sync()
        got, want = do.download(), orc.pq10_encode(hdr, ui, conversion, 500.0, 400.0, mll)
        assert ((got >> 30) == 3).all()
        diff = np.abs(channels(got) - channels(want))
        assert diff.max() <= 1, diff.max()      # 10-bit LSB: device log / exp in the two pow() of the PQ curve
        assert (diff == 0).mean() > 0.9
    bad = capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8B8A8_UNORM)
    with pytest.raises(capi.GraniteHipError):
        gr.pq10_encode(dh, du, bad, conversion)
    gr.close()


def test_hdr10_frame():
    """lighting -> ui (cleared layer) -> pq10 on an A2B10G10R10 backbuffer == the oracle's lighting fed through the oracle's
    encoder with the host's conversion matrix (ST.2020 primaries, 1000 nits, pre-exposures 500 / 400)."""
    w, h = 480, 270
    cam = synth.Camera(w, h)
    gbuf, descs = synth.make_gbuffer(cam), synth.make_lights(cam, 500)
    a = gapp.Application(w, h, hdr10=True, hdr_bloom=False)
    a.set_render_parameters(cam.render_params())
    a.set_lights(descs)
    a.upload_gbuffer(gbuf)
    a.render_frames(3)
    got = a.read_backbuffer()
    assert got.dtype == np.uint32 and got.shape == (h, w)
    hdr = a.read("HDR-main")
    ui = np.zeros((h, w, 4), np.uint8)
    ui[..., 3] = 255
    np.testing.assert_array_equal(a.read("ui-temporary"), ui)
    want = orc.pq10_encode(hdr, ui, orc.rec709_to_display(), 500.0, 400.0, 1000.0)  # from the executor's own HDR target
    assert np.abs(channels(got) - channels(want)).max() <= 1
    a.close()
