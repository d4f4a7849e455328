"""An outside pin for the sampler model the oracle, the reference-shader runner (glsl_cpu.hpp) and the kernels share
(oracle_common.h linear_axis: exact fp32 weights, a coordinate within 2^-8 of a texel centre reads that texel alone).

Nothing here calls that code.  For every bilinear tap the chain produces along one axis -- threshold, the four downsamples with their
feedback tap, the three upsamples, the tonemap's two taps, the centre taps of FXAA / SMAA / TAA -- at 1080p, 4K and 8K, the tap's
position is evaluated twice: in exact rational arithmetic from the shader text (bloom_threshold.comp:29, bloom_downsample.comp:27-38,
bloom_upsample.comp:22-33, tonemap.frag:57-58), and as the model does it (fp32 coordinate, then the snap rule restated below from the
comment of oracle_common.h:327-346, not imported).  Checked:
  * a tap whose exact position is a texel centre gets weight exactly 0 and the exact texel (what any fixed-point sampler returns);
  * a tap further than the snap radius from a centre keeps its texel pair and a weight within 2^-9 of the exact one -- SURVEY 8a's
    "differs by <= 1/512 in weight", the slack its 2 ulp + 1e-4 tolerance is stated to absorb (odd levels, 135 -> 68, put taps anywhere);
  * a tap inside the radius reads the centre texel alone, which is what a truncating sampler with 8 fractional bits returns;
  * so the model's position always lies inside the cell [a - 2^-8, a + 2^-8] an 8-bit sub-texel sampler (subTexelPrecisionBits = 8:
    lavapipe, every desktop driver) can return for the exact position a."""
import math
from fractions import Fraction

import numpy as np
import pytest

SNAP = np.float32(1.0 / 256.0)


def level(size, scale):
    """RenderGraph size resolution of an InputRelative attachment: ceil(input * scale) (render_graph.cpp:1022-1036)."""
    return max(1, int(math.ceil(size * scale)))


def model_axis(out_n, in_n, offset_texels):
    """fp32 evaluation as in the shaders + the model: vUV = (x + 0.5) * inv_out (+ offset * inv_in); f = vUV * in - 0.5; snap."""
    f32 = np.float32
    x = np.arange(out_n, dtype=np.float32)
    inv_out, inv_in = f32(1.0) / f32(out_n), f32(1.0) / f32(in_n)
    uv = (x + f32(0.5)) * inv_out
    if offset_texels != 0.0:
        uv = uv + f32(offset_texels) * inv_in
    f = uv * f32(in_n) - f32(0.5)
    fl = np.floor(f + SNAP)
    a = f - fl
    a = np.where(a < SNAP, f32(0.0), a)
    return fl.astype(np.int64), a.astype(np.float64)


def exact_axis(out_n, in_n, offset_texels):
    """Exact rational position of the same taps with exact 1 / size: F = (x + 1/2) in / out + offset - 1/2 = index + weight."""
    eighths = int(round(offset_texels * 8))
    assert eighths == offset_texels * 8
    x = np.arange(out_n, dtype=np.int64)
    den = 16 * out_n                                    # F = num / den
    num = 8 * (2 * x + 1) * in_n + (eighths - 4) * 2 * out_n
    index = np.floor_divide(num, den)
    return index, (num - index * den).astype(np.float64) / den, den


def passes(w):
    """(name, output length, input length, tap offsets in input texels) of every pass along an axis of length w."""
    t, d0, d1, d2, d3 = (level(w, s) for s in (0.5, 0.25, 0.125, 0.0625, 0.03125))
    down, up = (-1.75, 0.0, 1.75), (-0.875, 0.0, 0.875)
    return [("threshold", t, w, (0.0,)), ("downsample-0", d0, t, down), ("downsample-1", d1, d0, down), ("downsample-2", d2, d1, down),
            ("downsample-3", d3, d2, down), ("downsample-3 feedback", d3, d3, (0.0,)), ("upsample-2", d2, d3, up), ("upsample-1", d1, d2, up),
            ("upsample-0", d0, d1, up), ("tonemap hdr", w, w, (0.0,)), ("tonemap bloom", w, d0, (0.0,)), ("aa centre taps", w, w, (0.0,))]


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160), (7680, 4320), (1918, 1078), (253, 127)])
def test_model_weights_against_exact_positions(w, h):
    worst = 0.0
    for n in (w, h):
        for name, out_n, in_n, offsets in passes(n):
            for off in offsets:
                index, weight = model_axis(out_n, in_n, off)
                want_index, want_weight, den = exact_axis(out_n, in_n, off)
                centre = want_weight == 0.0
                # taps on texel centres (all 2:1 / 1:2 / 1:1 / 4:1 levels of even-sized targets): that texel, weight 0
                assert (weight[centre] == 0.0).all() and (index[centre] == want_index[centre]).all(), (name, off, n)
                position, want_position = index + weight, want_index + want_weight
                err = np.abs(position - want_position)
                # everywhere: inside the cell [a - 2^-8, a + 2^-8] that an 8-bit sub-texel sampler can return for the exact position
                assert err.max() <= 2.0 ** -8, (name, off, n, err.max())
                # odd levels (135 -> 68) put taps anywhere; those further than the snap radius (+ the fp32 coordinate error) from a
                # centre keep their texel pair and a weight within 2^-9 of the exact one
                gap = np.minimum(want_weight, 1.0 - want_weight)
                clear = gap >= 2.0 ** -8 + 2.0 ** -9
                if clear.any():
                    assert (index[clear] == want_index[clear]).all(), (name, off, n)
                    worst = max(worst, float(err[clear].max()))
                    assert err[clear].max() <= 2.0 ** -9, (name, off, n, err[clear].max())
                # inside the radius the model reads the centre texel alone, like a truncating 8-bit sampler
                inside = (gap < 2.0 ** -8 - 2.0 ** -9) & ~centre
                assert (weight[inside] == 0.0).all(), (name, off, n)
    # (informative) the largest weight error over the chain at this size: fp32 rounding of a coordinate of up to `w` texels
    assert worst <= 2.0 ** -9


def test_the_snap_is_what_an_8_bit_sampler_does_near_a_centre():
    """Positions within 2^-8 of a centre: every fixed-point sampler with 8 fractional bits that truncates returns the centre texel alone
    (weight 0); one that rounds returns weight 0 or 2^-8.  The model's 0 is inside both cells; just outside the radius it returns the
    exact weight, which is inside [a - 2^-8, a + 2^-8] trivially."""
    f32 = np.float32
    for centre in (0.0, 17.0, 4095.0, 7679.0):
        for delta in (0.0, 2.0 ** -12, 2.0 ** -9, 2.0 ** -8 - 2.0 ** -11):
            f = f32(centre + delta)
            fl = np.floor(f + SNAP); a = f - fl
            a = f32(0.0) if a < SNAP else a
            truncating = math.floor((float(f) - math.floor(float(f))) * 256.0) / 256.0
            assert int(fl) == int(centre) and a == 0.0 and truncating == 0.0, (centre, delta)
        f = f32(centre + 2.0 ** -8 + 2.0 ** -10)
        fl = np.floor(f + SNAP); a = float(f - fl)
        assert int(fl) == int(centre) and abs(a - (float(f) - centre)) == 0.0
