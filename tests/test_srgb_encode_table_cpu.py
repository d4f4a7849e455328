"""The linear -> sRGB8 staircase table of the *_SRGB stores (csrc/device_common.hpp: encode_srgb8_lut, built by
gr_srgb_encode_table on the host) against the oracle's transfer function, value by value -- no GPU involved: the table
lookup is restated in numpy exactly as the device function does it."""
import ctypes as C

import numpy as np

from granite_amd import capi
from oracle import oracle as orc

ENTRIES, MIN_BITS, SHIFT = 13 * 128 + 1, 0x39000000, 16


def table():
    lib = capi.load_library()
    lib.gr_srgb_encode_table.restype = C.c_int
    lib.gr_srgb_encode_table.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
    t = np.zeros(2 * ENTRIES, np.uint32)
    assert lib.gr_srgb_encode_table(t.ctypes.data_as(C.POINTER(C.c_uint32)), ENTRIES) == 0
    return t.reshape(ENTRIES, 2)


def encode_with_table(t, values):
    """encode_srgb8_lut: med3(c, 2^-13, 1) (NaN -> 2^-13), bucket by the top bits, one compare against the threshold."""
    c = np.asarray(values, np.float32)
    c = np.where(np.isnan(c), np.float32(2.0 ** -13), np.clip(c, np.float32(2.0 ** -13), np.float32(1.0))).astype(np.float32)
    index = (c.view(np.uint32) - np.uint32(MIN_BITS)) >> np.uint32(SHIFT)
    thr = t[index, 0].view(np.float32)
    return (t[index, 1] + (c >= thr)).astype(np.uint8)


def test_table_shape_and_monotonicity():
    t = table()
    assert t[0, 1] == 0 and t[-1, 1] == 255 and t[-1, 0] == 0x7f800000
    assert (np.diff(t[:, 1].astype(np.int64)) >= 0).all() and (np.diff(t[:, 1].astype(np.int64)) <= 1).all()
    # a threshold lies inside its own bucket
    lo = MIN_BITS + (np.arange(ENTRIES, dtype=np.int64) << SHIFT)
    finite = t[:, 0] != 0x7f800000
    assert ((t[finite, 0] > lo[finite]) & (t[finite, 0] < lo[finite] + (1 << SHIFT))).all()
    assert finite.sum() == 255 - 0  # each of the 255 steps is some bucket's threshold


def test_table_equals_the_transfer_function_everywhere_it_can_differ():
    t = table()
    rng = np.random.default_rng(11)
    bits = [rng.integers(0x30000000, 0x3f800001, 4_000_000, dtype=np.uint32)]      # random floats in [2^-31, 1]
    lo = (MIN_BITS + (np.arange(ENTRIES, dtype=np.int64) << SHIFT)).astype(np.uint32)
    for d in range(-3, 4):                                                          # around every bucket edge and threshold
        bits.append((lo.astype(np.int64) + d).astype(np.uint32))
        fin = t[:, 0] != 0x7f800000
        bits.append((t[fin, 0].astype(np.int64) + d).astype(np.uint32))
    values = np.concatenate(bits).view(np.float32)
    special = np.array([0.0, -0.0, -1.0, 1.0, 1.5, 65504.0, np.inf, -np.inf, np.nan, 1e-30, 2.0 ** -13, 0.0031308, 0.00313081], np.float32)
    values = np.concatenate([values, special])
    np.testing.assert_array_equal(encode_with_table(t, values), orc.float_to_srgb8(values))


# ---- the fused tonemap staircase (tonemap_srgb8_lut / gr_tonemap_srgb8_table) ---------------------------------------------
T_ENTRIES, T_MIN_BITS, T_SHIFT = 16 * 64 + 1, 0x39800000, 17


def tonemap_table():
    lib = capi.load_library()
    lib.gr_tonemap_srgb8_table.restype = C.c_int
    lib.gr_tonemap_srgb8_table.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
    t = np.zeros(2 * T_ENTRIES, np.uint32)
    assert lib.gr_tonemap_srgb8_table(t.ctypes.data_as(C.POINTER(C.c_uint32)), T_ENTRIES) == 0
    return t.reshape(T_ENTRIES, 2)


def test_tonemap_staircase_equals_the_oracle_tonemap_for_every_finite_non_negative_colour():
    """One channel through orc.tonemap (hdr = x, bloom = 0, no exposure) vs the table lookup the kernel does."""
    t = tonemap_table()
    assert t[0, 1] == 0 and t[-1, 1] == 255 and (np.diff(t[:, 1].astype(np.int64)) >= 0).all() and (np.diff(t[:, 1].astype(np.int64)) <= 1).all()
    rng = np.random.default_rng(12)
    # fp16 HDR values are what the pass sees (x = fp16 + bilinear fp16 bloom, times an fp32 scale): take all non-negative finite
    # halves as hdr, then arbitrary scales, and compare table(x) with the oracle's arithmetic on the same x
    halves = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float32)
    scales = np.array([1.0, 0.37, 2.9, 0.011, 17.0], np.float32)
    xs = [halves * s for s in scales]
    lo = (T_MIN_BITS + (np.arange(T_ENTRIES, dtype=np.int64) << T_SHIFT)).astype(np.uint32)
    fin = t[:, 0] != 0x7f800000
    for d in range(-3, 4):
        xs.append((lo.astype(np.int64) + d).astype(np.uint32).view(np.float32))
    xs.append(rng.integers(0x30000000, 0x42000000, 4_000_000, dtype=np.uint32).view(np.float32))
    probes = np.concatenate([(t[fin, 0].astype(np.int64) + d).astype(np.uint32).view(np.float32) for d in range(-48, 49)])
    x = np.concatenate(xs + [probes]).astype(np.float32)
    near_threshold = np.zeros(x.size, bool)
    near_threshold[-probes.size:] = True
    near_threshold = near_threshold[np.isfinite(x) & (x >= 0)]
    x = x[np.isfinite(x) & (x >= 0)]
    # the oracle's arithmetic on x: uncharted2(x) * white_scale -> srgb8 (oracle_post.cpp orc_tonemap with bloom = 0 would
    # round x to fp16 first, so the curve is restated here in float32 numpy and only the encode goes through the oracle)
    A, B, Cc, D, E, F, W = (np.float32(v) for v in (0.15, 0.50, 0.10, 0.20, 0.02, 0.30, 11.2))
    u2 = lambda v: ((v * (A * v + Cc * B) + D * E) / (v * (A * v + B) + D * F)) - E / F
    want = orc.float_to_srgb8((u2(x) * (np.float32(1.0) / u2(W))).astype(np.float32))
    c = np.clip(x, np.float32(2.0 ** -12), np.float32(16.0)).astype(np.float32)
    index = (c.view(np.uint32) - np.uint32(T_MIN_BITS)) >> np.uint32(T_SHIFT)
    got = (t[index, 1] + (c >= t[index, 0].view(np.float32))).astype(np.uint8)
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    # fp32 rounding noise of the evaluated curve (a few ulps of x) makes it non-monotone right next to a step: within some
    # tens of ulps of a threshold a float may sit on the other side of it.  Never more than one byte value; away from the
    # thresholds (random colours, every fp16 value times a scale, bucket edges) a few in a million.
    assert diff.max() <= 1
    assert (diff[~near_threshold] != 0).mean() < 1e-5, (diff[~near_threshold] != 0).sum()
    assert (diff[near_threshold] != 0).mean() < 0.05
