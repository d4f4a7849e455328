"""CPU: pin the FSR 1.0 restatement (oracle/oracle_fsr.cpp) with known answers derived from the algorithm's definition --
the reference holds no test or golden data for these passes ("parity unpinned")."""
import numpy as np
import pytest

from oracle import oracle as orc


def flat(w, h, rgb):
    img = np.full((h, w, 4), 255, np.uint8)
    img[..., :3] = rgb
    return img


@pytest.mark.parametrize("fp16", [False, True])
def test_easu_is_the_identity_on_flat_colour_and_at_scale_one(fp16):
    # weights are normalised (sum of taps / sum of weights) and the result is clamped to the 2x2 ring: flat in, flat out,
    # including black and white where the half path divides by a zero gradient
    for rgb in ((0, 0, 0), (255, 255, 255), (200, 100, 50), (1, 2, 3)):
        out = orc.fsr_easu(flat(24, 16, rgb), 37, 29, fp16)
        assert (out[..., :3] == np.array(rgb, np.uint8)).all() and (out[..., 3] == 255).all()
    # scale 1: the resolve point is the centre of texel f, every other tap gets the window weight of an integer distance;
    # de-ringing keeps the result inside [min, max] of the 2x2 that contains f
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (20, 30, 4), dtype=np.uint8)
    out = orc.fsr_easu(img, 30, 20, fp16).astype(int)
    pad = np.pad(img.astype(int), ((0, 1), (0, 1), (0, 0)), mode="edge")
    quad = np.stack([pad[:-1, :-1], pad[:-1, 1:], pad[1:, :-1], pad[1:, 1:]])
    assert (out[..., :3] >= quad.min(0)[..., :3]).all() and (out[..., :3] <= quad.max(0)[..., :3]).all()


def test_easu_resolve_position_and_edge_clamp():
    """Output pixel x maps to input position (x + 0.5) * iw / ow - 0.5 (FsrEasuCon): at 2x, output pixels 2k and 2k + 1
    straddle input texel k; a vertical step edge stays a step edge (no ringing beyond the two sides)."""
    img = flat(16, 8, (20, 20, 20))
    img[:, 8:, :3] = 220
    for fp16 in (False, True):
        out = orc.fsr_easu(img, 32, 16, fp16)
        assert set(np.unique(out[..., 0])) <= set(range(20, 221))
        assert (out[:, :14, 0] == 20).all() and (out[:, 18:, 0] == 220).all()
        row = out[8, :, 0].astype(int)
        assert (np.diff(row) >= 0).all()  # monotone across the edge
    # taps outside the image repeat the border texel: the corner pixel of a flat-bordered image is that colour
    assert tuple(orc.fsr_easu(img, 32, 16, True)[0, 0, :3]) == (20, 20, 20)


def test_fp16_and_fp32_variants_agree_to_half_precision():
    y, x = np.mgrid[0:48, 0:64]
    img = np.zeros((48, 64, 4), np.uint8)
    img[..., 0], img[..., 1], img[..., 2], img[..., 3] = x * 4, y * 5, 128 + 64 * np.sin(x * 0.3) * np.cos(y * 0.2), 255
    a, b = orc.fsr_easu(img, 96, 72, False).astype(int), orc.fsr_easu(img, 96, 72, True).astype(int)
    assert np.abs(a - b).max() <= 2


def test_rcas_known_answers():
    """sharpen.frag + FsrRcasF: on flat colour the lobe is the limit -(0.25 - 1/16) * sharpness and the output is the input up
    to the medium-precision reciprocal; an isolated bright pixel gets brighter relative to its ring, never out of range."""
    s = orc.fsr_rcas_sharpness(0.5)
    assert abs(s - 2.0 ** -0.5) < 1e-7
    for srgb in (False, True):
        for v in (0, 255, 64, 200):
            out = orc.fsr_rcas(flat(9, 7, (v, v, v)), s, srgb)
            assert np.abs(out[..., :3].astype(int) - v).max() <= 1, (v, srgb)
    img = flat(9, 9, (100, 100, 100))
    img[4, 4, :3] = 160
    out = orc.fsr_rcas(img, s, srgb=False)
    assert out[4, 4, 0] > 160 and out[4, 3, 0] < 100 and out[0, 0, 0] in (99, 100, 101)
    # the bit-trick reciprocal: b = asfloat(0x7ef19fff - asuint(a)); b * (2 - b * a), relative error below 0.4 %
    a = np.linspace(0.3, 1.0, 1000, dtype=np.float32)
    b = (np.uint32(0x7EF19FFF) - a.view(np.uint32)).view(np.float32)
    assert np.abs(b * (-b * a + np.float32(2.0)) * a - 1.0).max() < 4e-3
