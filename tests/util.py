"""Comparators that carry the stated tolerances (SURVEY.md §8a tolerance note)."""
import numpy as np

from granite_amd import synth


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


def assert_rgba16f_close_but_for_ill_conditioned_pixels(a_bits, b_bits, ulps=2.0, abs_tol=1e-4, max_pixels=4, outer_ulps=4.0, what=""):
    """The stated tolerance everywhere except at up to `max_pixels` pixels, which must still be within `outer_ulps`.  For lit
    frames of millions of pixels under a moving camera: a surface seen edge-on (N.V at its 0.001 clamp) with a light almost
    exactly behind it along the view ray (|V + L| -> 0) has a condition number of ~1e4 in its specular term -- one rounding of
    fp32 in the half vector is a per-mille of the pixel; two correct fp32 evaluations differ there."""
    bad = rgba16f_mismatch(a_bits, b_bits, ulps, abs_tol)
    pixels = bad.any(axis=-1).sum()
    if pixels > max_pixels:
        raise AssertionError(f"{what}: {pixels} pixels out of tolerance ({ulps} ulp fp16 + {abs_tol}); at most {max_pixels} ill-conditioned ones are expected")
    if pixels:
        assert_rgba16f_close(a_bits, b_bits, outer_ulps, abs_tol, what=what + " (ill-conditioned pixels)")


def assert_rgba8_close(a, b, lsb=1, what=""):
    d = np.abs(np.asarray(a, np.int16) - np.asarray(b, np.int16))
    if (d > lsb).any():
        idx = np.argwhere(d > lsb)[0]
        raise AssertionError(f"{what}: {(d > lsb).sum()} of {d.size} bytes differ by more than {lsb} LSB; "
                             f"first at {tuple(idx)}: {np.asarray(a)[tuple(idx)]} vs {np.asarray(b)[tuple(idx)]} (max {d.max()})")


def close_up_scene(w, h, seed=0):
    """A trough within one unit of the camera (the pass only reflects where the hierarchy's level 0 is below 1: sssr_util.h
    IsReflective on the view-space depth chain the reference feeds it): floor in the middle, walls rising towards the camera at
    both sides and at the top, normals derived from that geometry so that reflected rays do find it again.  Left half glossy
    (denoised: one ray per quad phase + copies), a mirror strip (one ray per pixel), the rest too rough to reflect."""
    cam = synth.Camera(w, h)
    g = synth.make_gbuffer(cam, synth.SEED + seed)
    u, v = (np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h
    view_z = 0.95 - 0.6 * np.abs(2 * u - 1)[None, :] ** 2 * np.ones((h, 1)) - 0.25 * (1 - v)[:, None] ** 2
    view_z = np.clip(view_z, 0.25, 0.97)
    depth = cam.depth_from_view_distance(view_z).astype(np.float32)
    ys, xs = np.mgrid[0:h, 0:w]
    ndc = np.stack([2 * (xs + 0.5) / w - 1, 2 * (ys + 0.5) / h - 1, depth.astype(np.float64), np.ones((h, w))], 0).reshape(4, -1)
    clip = cam.invVP @ ndc
    pos = (clip[:3] / clip[3]).T.reshape(h, w, 3)
    n = np.cross(np.gradient(pos, axis=1), np.gradient(pos, axis=0))
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    n[(n * (cam.position[None, None, :] - pos)).sum(2) < 0] *= -1
    q = np.clip(np.rint((0.5 * n + 0.5) * 1023.0), 0, 1023).astype(np.uint32)
    normal = (q[..., 0] | (q[..., 1] << 10) | (q[..., 2] << 20) | np.uint32(3 << 30)).astype(np.uint32)
    pbr = g["pbr"].copy()
    pbr[:, :w // 2] = (pbr[:, :w // 2] & 0xff) | (np.uint16(25) << 8)   # roughness 0.098: glossy, denoised
    pbr[:, :w // 8] &= 0xff                                              # roughness 0: mirror
    light = synth.make_hdr(w, h, synth.SEED + seed)
    return cam, depth, normal, pbr, g["albedo"], light
