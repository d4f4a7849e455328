"""GPU parity, bit for bit, for the single-pass downsampler (gr_spd_downsample: emit_single_pass_downsample + spd.comp) through the
C ABI, and for the host-layer emit_single_pass_downsample through the harness."""
import numpy as np
import pytest

from granite_amd import app as gapp, capi, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def source(iw, ih, seed=11):
    src = synth.make_hdr(iw, ih, seed).copy()
    src.view(np.float16)[..., 3] = np.random.default_rng(3).uniform(0.0, 4.0, (ih, iw)).astype(np.float16)
    return src


@pytest.mark.parametrize("iw,ih,w0,h0,mips,components,depth,mods", [
    (128, 128, 64, 64, 7, 4, False, False),
    (200, 120, 100, 60, 7, 4, False, False),
    (256, 128, 128, 64, 7, 3, False, True),
    (256, 128, 128, 64, 7, 2, False, False),
    (128, 128, 128, 128, 8, 1, True, False),
    (512, 512, 256, 256, 9, 4, False, False),
    (2048, 1024, 1024, 512, 11, 4, False, True),
    (4096, 4096, 2048, 2048, 12, 3, False, False),   # the largest chain one tail workgroup covers
    (333, 77, 150, 40, 8, 4, False, False),          # outputs smaller than half the source, odd sizes
    (64, 64, 32, 32, 3, 4, False, False),
    (64, 64, 32, 32, 1, 4, False, False),
])
def test_spd_matches_oracle_bit_for_bit(gr, iw, ih, w0, h0, mips, components, depth, mods):
    src = source(iw, ih)
    fm = None
    if mods:
        fm = np.ones((mips, 4), np.float32)
        fm[mips - 1] = (0.0, 1.0, 1.0, 1.0)
        fm[2] = (0.5, 2.0, 1.0, 1.0)
    want = orc.spd(src, w0, h0, mips, components, depth, fm, fill=0x3c00)
    dev = capi.DeviceImage(gr, iw, ih, capi.FORMAT_R16G16B16A16_SFLOAT).upload(src)
    texels = orc.spd_chain_texels(w0, h0, mips)
    chain = capi.DeviceBuffer(gr, texels * 8).upload(np.full(texels * 4, 0x3c00, np.uint16))
    gr.spd_downsample(dev, w0, h0, mips, components, depth, fm, chain=chain)
    gr.sync()
    got = orc.spd_split(chain.download(np.uint16), w0, h0, mips)
    for level, (a, b) in enumerate(zip(want, got)):
        np.testing.assert_array_equal(b, a, err_msg=f"level {level}")


def test_spd_argument_checks(gr):
    dev = capi.DeviceImage(gr, 64, 64, capi.FORMAT_R16G16B16A16_SFLOAT).upload(source(64, 64))
    for kw in (dict(mips=0), dict(mips=13), dict(components=0), dict(components=5)):
        args = dict(width=32, height=32, mips=3, components=4)
        args.update(kw)
        with pytest.raises(capi.GraniteHipError):
            gr.spd_downsample(dev, args["width"], args["height"], args["mips"], args["components"])
    with pytest.raises(capi.GraniteHipError):
        gr.spd_downsample(dev, 4096, 32, 8)          # beyond the single tail workgroup
    bad = capi.DeviceImage(gr, 64, 64, capi.FORMAT_R8G8B8A8_UNORM)
    with pytest.raises(capi.GraniteHipError):
        gr.spd_downsample(bad, 32, 32, 3)


def test_host_emit_single_pass_downsample_generates_the_mips_of_an_image():
    """The reference's use (renderer/ocean.cpp:579-601): level 0 of an RGBA16F image is the source, levels 1.. come out of the
    downsampler, three components, the last level's first channel forced to 0 by filter_mod."""
    w, h, levels = 512, 256, 9
    src = source(w, h, 4)
    a = gapp.Application(64, 64, lighting=False)
    mods = np.ones((levels - 1, 4), np.float32)
    mods[-1] = (0.0, 1.0, 1.0, 1.0)
    chain = a.generate_mipmaps(src, levels, components=3, filter_mods=mods)
    got = orc.spd_split(chain, w, h, levels)
    np.testing.assert_array_equal(got[0], src)
    want = orc.spd(src, w // 2, h // 2, levels - 1, 3, False, mods)
    for level, (x, y) in enumerate(zip(want, got[1:])):
        np.testing.assert_array_equal(y, x, err_msg=f"level {level + 1}")
    assert (got[-1][..., 0] == 0).all() and (got[-1][..., 1] != 0).any()
    with pytest.raises(capi.GraniteHipError):
        a.generate_mipmaps(src, 1)
    a.close()
