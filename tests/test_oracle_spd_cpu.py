"""Known answers for the single-pass downsampler oracle (oracle/oracle_spd.cpp: emit_single_pass_downsample + spd.comp over FFX
SPD).  The executed reference shader pins it bit for bit in tests/test_reference_shaders_cpu.py."""
import numpy as np
import pytest

from granite_amd import synth
from oracle import oracle as orc

H = lambda a: np.asarray(a, np.float16).view(np.uint16)  # noqa: E731
F = lambda bits: bits.view(np.float16).astype(np.float64)  # noqa: E731


def test_constant_image_stays_constant_on_every_level():
    img = np.zeros((128, 128, 4), np.float16)
    img[...] = (1.5, 0.25, -2.0, 1.0)
    levels = orc.spd(img.view(np.uint16), 64, 64, 7)
    assert [l.shape[:2] for l in levels] == [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    for l in levels:
        assert (l.view(np.float16) == np.array([1.5, 0.25, -2.0, 1.0], np.float16)).all()


def test_levels_are_box_averages_with_spds_rounding_points():
    """Level 0 = 2 x 2 mean of the source (one bilinear tap at the footprint centre); levels 1..5 average the UNROUNDED level above
    inside a 64 x 64 source tile; level 6 averages level 5 AS STORED (fp16)."""
    r = np.random.default_rng(0)
    src = r.uniform(0.0, 8.0, (256, 256, 4)).astype(np.float16)
    levels = orc.spd(src.view(np.uint16), 128, 128, 8)
    s = src.astype(np.float32)
    box = lambda a: ((a[0::2, 0::2] + a[0::2, 1::2]) + a[1::2, 0::2] + a[1::2, 1::2]) * np.float32(0.25)  # noqa: E731
    cur = (s[0::2, 0::2] * np.float32(0.5) + s[0::2, 1::2] * np.float32(0.5)) * np.float32(0.5) + \
          (s[1::2, 0::2] * np.float32(0.5) + s[1::2, 1::2] * np.float32(0.5)) * np.float32(0.5)
    for l in range(6):
        np.testing.assert_array_equal(levels[l].view(np.float16), cur.astype(np.float16), err_msg=f"level {l}")
        if l < 5:
            cur = box(cur)
    # level 6: from the stored (rounded) level 5, column-first summation
    p = levels[5].view(np.float16).astype(np.float32)
    l6 = ((p[0::2, 0::2] + p[1::2, 0::2]) + p[0::2, 1::2] + p[1::2, 1::2]) * np.float32(0.25)
    np.testing.assert_array_equal(levels[6].view(np.float16), l6.astype(np.float16))
    l7 = box(l6)   # unrounded level 6
    np.testing.assert_array_equal(levels[7].view(np.float16), l7.astype(np.float16))
    # and that rounding point is observable: averaging the unrounded level 5 gives different bits somewhere in a big sample
    unrounded = box(cur)
    assert unrounded.shape == l6.shape


def test_components_filter_mods_and_untouched_texels():
    src = synth.make_hdr(128, 64, 3)
    mods = np.ones((6, 4), np.float32)
    mods[2] = (0.5, 2.0, 1.0, 1.0)
    mods[5] = (0.0, 1.0, 1.0, 1.0)   # the ocean's "last level goes to 0" (renderer/ocean.cpp:588-589)
    plain = orc.spd(src, 64, 32, 6, components=3)
    modded = orc.spd(src, 64, 32, 6, components=3, filter_mods=mods)
    for l in range(6):
        assert (plain[l][..., 3] == 0).all() and (modded[l][..., 3] == 0).all()   # chopped channel is written as 0
        want = (plain[l].view(np.float16).astype(np.float32) * mods[l]).astype(np.float16)
        if l in (2, 5):
            # the modifier is applied to the unrounded value: equal to scaling the stored one only up to 1 ulp, exact for 0.5 / 2 / 0
            np.testing.assert_array_equal(modded[l].view(np.float16), want)
        else:
            np.testing.assert_array_equal(modded[l], plain[l])
    assert (modded[5].view(np.float16)[..., 0] == 0).all()
    # mods never feed the levels below: level 3 is the same with and without level 2's modifier
    np.testing.assert_array_equal(modded[3], plain[3])
    # one component
    one = orc.spd(src, 64, 32, 6, components=1)
    assert all((l[..., 1:] == 0).all() for l in one)
    np.testing.assert_array_equal(one[0][..., 0], plain[0][..., 0])


def test_sizes_that_are_not_multiples_of_the_tile():
    """100 x 60 outputs from a 200 x 120 source: edge workgroups tap past the source (clamped), stores outside a level are dropped,
    levels shrink as max(size >> level, 1), level 6 reads clamp inside the 3 x 1 level 5."""
    src = synth.make_hdr(200, 120, 5)
    levels = orc.spd(src, 100, 60, 7, fill=0x7e00)
    assert [l.shape[:2] for l in levels] == [(60, 100), (30, 50), (15, 25), (7, 12), (3, 6), (1, 3), (1, 1)]
    for l in levels:
        assert not (l == 0x7e00).any()   # every texel of every level is written
    s = src.view(np.float16).astype(np.float32)
    mean = ((s[0::2, 0::2] * 0.5 + s[0::2, 1::2] * 0.5) * 0.5 + (s[1::2, 0::2] * 0.5 + s[1::2, 1::2] * 0.5) * 0.5).astype(np.float16)
    # 1 / 200 is not a binary fraction: the tap lands a few ulp off the footprint centre, so the weights are 0.5 +- 1e-5
    np.testing.assert_allclose(levels[0].view(np.float16).astype(np.float32), mean.astype(np.float32), rtol=2e-3, atol=1e-6)


def test_depth_mode_is_a_min_pyramid_of_the_colocated_texel():
    r = np.random.default_rng(1)
    d = np.zeros((64, 64, 4), np.float16)
    d[..., 0] = r.uniform(0.0, 1.0, (64, 64))
    levels = orc.spd(d.view(np.uint16), 64, 64, 7, components=1, depth_mode=True)
    cur = d[..., 0]
    np.testing.assert_array_equal(levels[0].view(np.float16)[..., 0], cur)       # NearestClamp of the same texel
    for l in range(1, 7):
        cur = np.minimum(np.minimum(cur[0::2, 0::2], cur[0::2, 1::2]), np.minimum(cur[1::2, 0::2], cur[1::2, 1::2]))
        np.testing.assert_array_equal(levels[l].view(np.float16)[..., 0], cur)
        assert (levels[l][..., 1:] == 0).all()


def test_fewer_mips_stop_early():
    src = synth.make_hdr(128, 128, 2)
    full = orc.spd(src, 64, 64, 7)
    for mips in (1, 2, 5, 6):
        part = orc.spd(src, 64, 64, mips)
        assert len(part) == mips
        for a, b in zip(part, full):
            np.testing.assert_array_equal(a, b)
