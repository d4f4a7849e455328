"""Comparators that carry the stated tolerances (SURVEY.md §8a tolerance note)."""
import numpy as np


def half_bits_to_f32(bits):
    return np.asarray(bits, np.uint16).view(np.float16).astype(np.float32)


def ulp_fp16(v):
    """Spacing of fp16 at magnitude |v| (2^-24 in the subnormal range)."""
    a = np.maximum(np.abs(v).astype(np.float64), 2.0 ** -14)
    e = np.floor(np.log2(a))
    return np.exp2(e - 10.0)


def rgba16f_mismatch(a_bits, b_bits, ulps=2.0, abs_tol=1e-4):
    """Boolean mask of channels violating |a-b| <= ulps*ulp_fp16(max(|a|,|b|)) + abs_tol."""
    a = half_bits_to_f32(a_bits).astype(np.float64)
    b = half_bits_to_f32(b_bits).astype(np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    tol = ulps * ulp_fp16(np.maximum(np.abs(a), np.abs(b))) + abs_tol
    with np.errstate(invalid="ignore"):
        bad = ~(np.abs(a - b) <= tol)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    return bad & ~both_nan & ~same_inf


def assert_rgba16f_close(a_bits, b_bits, ulps=2.0, abs_tol=1e-4, what=""):
    bad = rgba16f_mismatch(a_bits, b_bits, ulps, abs_tol)
    if bad.any():
        idx = np.argwhere(bad)[0]
        a = half_bits_to_f32(a_bits)[tuple(idx)]
        b = half_bits_to_f32(b_bits)[tuple(idx)]
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} channels out of tolerance "
                             f"({ulps} ulp fp16 + {abs_tol}); first at {tuple(idx)}: {a} vs {b}")


def assert_rgba8_close(a, b, lsb=1, what=""):
    d = np.abs(np.asarray(a, np.int16) - np.asarray(b, np.int16))
    if (d > lsb).any():
        idx = np.argwhere(d > lsb)[0]
        raise AssertionError(f"{what}: {(d > lsb).sum()} of {d.size} bytes differ by more than {lsb} LSB; "
                             f"first at {tuple(idx)}: {np.asarray(a)[tuple(idx)]} vs {np.asarray(b)[tuple(idx)]} (max {d.max()})")
