"""GPU: frames rendered from a G-buffer that arrives as .gtx files equal frames rendered from the same arrays; what the
executor writes back as .gtx is what it rendered."""
import numpy as np
import pytest

from granite_amd import app as gapp, capi, gtx, synth

pytestmark = pytest.mark.gpu


def test_gbuffer_in_and_frame_out_as_gtx(tmp_path):
    w, h = 480, 270
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, 500)
    formats = {"emissive": capi.FORMAT_R16G16B16A16_SFLOAT, "albedo": capi.FORMAT_R8G8B8A8_SRGB,
               "normal": capi.FORMAT_A2B10G10R10_UNORM_PACK32, "pbr": capi.FORMAT_R8G8_UNORM, "depth": capi.FORMAT_D32_SFLOAT}
    paths = {}
    for k, fmt in formats.items():
        a = np.ascontiguousarray(gbuf[k])
        paths[k] = str(tmp_path / f"{k}.gtx")
        gtx.write(paths[k], fmt, [a.view(np.uint8).reshape(h, w, -1)])

    def make(**kw):
        a = gapp.Application(w, h, depth_hierarchy=1, **kw)
        a.set_render_parameters(cam.render_params())
        a.set_lights(descs)
        return a

    a = make()
    a.upload_gbuffer(gbuf)
    a.render_frames(4)
    want = a.read_backbuffer().copy()
    want_hiz = [l.copy() for l in a.read_mip_chain("depth-hiz")]
    a.close()

    b = make()
    b.upload_gbuffer_gtx(**paths)
    b.render_frames(4)
    np.testing.assert_array_equal(b.read_backbuffer(), want)

    out = str(tmp_path / "frame.gtx")
    b.save_gtx(out)
    f = gtx.read(out)
    assert (f.info.format, f.info.width, f.info.height, f.info.levels) == (capi.FORMAT_R8G8B8A8_SRGB, w, h, 1)
    np.testing.assert_array_equal(f.level(0)[0], want)

    b.save_gtx(str(tmp_path / "hdr.gtx"), "HDR-main")
    hdr = gtx.read(str(tmp_path / "hdr.gtx"))
    np.testing.assert_array_equal(hdr.level(0)[0].view(np.uint16).reshape(h, w, 4), b.read("HDR-main"))

    # a mip chain: the executor packs levels back to back, GTX aligns each to 16 bytes
    b.save_gtx(str(tmp_path / "hiz.gtx"), "depth-hiz")
    chain = gtx.read(str(tmp_path / "hiz.gtx"))
    assert (chain.info.format, chain.info.width, chain.info.height, chain.info.levels) == (capi.FORMAT_R32_SFLOAT, 512, 320, 8)
    for l, lv in enumerate(want_hiz):
        np.testing.assert_array_equal(chain.level(l)[0].view(np.float32).reshape(lv.shape), lv)

    # wrong size / wrong format are refused with the file named
    gtx.write(str(tmp_path / "small.gtx"), capi.FORMAT_D32_SFLOAT, [np.zeros((8, 8, 4), np.uint8)])
    with pytest.raises(capi.GraniteHipError, match="does not match"):
        b.upload_gbuffer_gtx(depth=str(tmp_path / "small.gtx"))
    with pytest.raises(capi.GraniteHipError, match="wrong format"):
        b.upload_gbuffer_gtx(albedo=paths["depth"])
    b.close()
