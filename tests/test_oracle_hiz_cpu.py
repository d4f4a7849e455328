"""CPU: pin the depth-hierarchy restatement (oracle/oracle_hiz.cpp) with known answers derived from hiz.comp's definition --
the reference holds no test or golden data for this pass ("parity unpinned")."""
import numpy as np
import pytest

from granite_amd import synth
from oracle import oracle as orc


def footprints(res, mips):
    """[start, end) of every texel of every mip along one axis, in mip-0 texels, from the definition alone: mip m halves
    mip m - 1 (floor, at least 1) and its last texel also takes the texel an odd size would drop (hiz.comp:38-41,178-247)."""
    spans = [[(i, i + 1) for i in range(res)]]
    for m in range(1, mips):
        prev = spans[-1]
        n = max(res >> m, 1)
        cur = []
        for x in range(n):
            last = 2 * x + 1
            if x + 1 == n and (len(prev) & 1):
                last = 2 * x + 2
            last = min(last, len(prev) - 1)
            cur.append((prev[min(2 * x, len(prev) - 1)][0], prev[last][1]))
        spans.append(cur)
    return spans


@pytest.mark.parametrize("size", [(64, 64), (300, 100), (257, 131), (1000, 600), (70, 1030)])
def test_every_texel_is_the_maximum_over_its_footprint(size):
    w, h = size
    rng = np.random.default_rng(w * 7919 + h)
    depth = rng.random((h, w), dtype=np.float32)
    ident = np.array([1, 0, 0, 1], np.float32)  # num = z, den = 1
    lay = orc.hiz_layout(w, h)
    levels = orc.hiz(depth, ident)
    assert len(levels) == lay["levels"] == int(np.floor(np.log2(max(w, h))))
    pad = np.pad(depth, ((0, lay["res_h"] - h), (0, lay["res_w"] - w)), mode="edge")  # NearestClamp
    np.testing.assert_array_equal(levels[0], pad)
    sx, sy = footprints(lay["res_w"], lay["mips"]), footprints(lay["res_h"], lay["mips"])
    for m in range(1, lay["mips"]):
        lv = levels[m]
        assert lv.shape == (len(sy[m]), len(sx[m])), m
        for y, (y0, y1) in enumerate(sy[m]):
            for x, (x0, x1) in enumerate(sx[m]):
                assert lv[y, x] == pad[y0:y1, x0:x1].max(), (m, x, y)
    # every mip-0 texel is covered by the last level: nothing is lost to rounding down
    assert sx[-1][0][0] == 0 and sx[-1][-1][1] == lay["res_w"] and sy[-1][-1][1] == lay["res_h"]
    assert levels[-1].max() == depth.max()


def test_output_downsample_drops_only_the_top_level():
    depth = np.random.default_rng(3).random((200, 330), dtype=np.float32)
    ident = np.array([1, 0, 0, 1], np.float32)
    full = orc.hiz(depth, ident)
    half = orc.hiz(depth, ident, output_downsample=True)
    lay = orc.hiz_layout(330, 200, True)
    assert (lay["chain_w"], lay["chain_h"], lay["levels"], lay["mips"]) == (192, 128, 7, 8)
    assert len(half) == len(full) - 1
    for a, b in zip(half, full[1:]):
        np.testing.assert_array_equal(a, b)


def test_z_transform_recovers_view_distance():
    """spd.cpp:164-165 feeds rows 2,3 of inv_projection: stored values are positive view-space distances."""
    cam = synth.Camera(320, 192)
    zt = orc.hiz_z_transform(cam.render_params()[48:64])
    dist = np.linspace(0.5, 90.0, 320 * 192, dtype=np.float32).reshape(192, 320)
    depth = cam.depth_from_view_distance(dist)
    top = orc.hiz(depth, zt)[0]
    np.testing.assert_allclose(top, dist, rtol=2e-3)
    # a zero denominator saturates instead of producing inf: min(num / den, 1e30)
    sat = orc.hiz(np.zeros((64, 64), np.float32), np.array([0, 1, 1, 0], np.float32))  # num = 1, den = z = 0
    assert sat[0].max() == np.float32(1e30) and sat[-1].max() == np.float32(1e30)
