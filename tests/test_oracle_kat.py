"""CPU: pin the oracle with analytic known-answer cases derived from the shader math (SURVEY.md §8c — the reference
holds no golden vectors for this path, "parity unpinned"), plus the committed fixtures under tests/golden/."""
import math
import os

import numpy as np
import pytest

from granite_amd import synth
from oracle import oracle as orc

L = orc.lib()


def f16(a):
    return np.asarray(a, np.float32).astype(np.float16).view(np.uint16)


def f32(bits):
    return np.asarray(bits, np.uint16).view(np.float16).astype(np.float32)


# ---- storage formats --------------------------------------------------------------------------------------------------
def test_half_roundtrip_every_bit_pattern():
    for h in range(0, 65536, 1):
        f = L.orc_half_to_float(h)
        if (h & 0x7c00) == 0x7c00 and (h & 0x3ff):
            assert math.isnan(f)
            continue
        assert L.orc_float_to_half(f) == h
        assert f == float(np.uint16(h).view(np.float16))


def test_half_rne_matches_ieee_and_muglm_rounds_ties_up():
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.normal(size=2000) * 10.0 ** rng.integers(-8, 5, 2000), [65504.0, 65519.9, 65520.0, 1e-8, 6e-8, 2.0 ** -25]])
    for v in vals.astype(np.float32):
        assert L.orc_float_to_half(float(v)) == int(np.float32(v).astype(np.float16).view(np.uint16))
    # exact tie between 1.0 (0x3c00) and next (0x3c01): RNE -> even (0x3c00), muglm::floatToHalf -> up (0x3c01)
    tie = float(np.float32(1.0 + 2.0 ** -11))
    assert L.orc_float_to_half(tie) == 0x3C00
    assert L.orc_float_to_half_muglm(tie) == 0x3C01
    # tie above an odd mantissa rounds up in both
    tie2 = float(np.float32(1.0 + 2.0 ** -10 + 2.0 ** -11))
    assert L.orc_float_to_half(tie2) == 0x3C02 and L.orc_float_to_half_muglm(tie2) == 0x3C02


def test_srgb_roundtrip_all_bytes():
    for i in range(256):
        assert L.orc_float_to_srgb8(L.orc_srgb8_to_float(i)) == i
    assert L.orc_srgb8_to_float(0) == 0.0 and L.orc_srgb8_to_float(255) == 1.0
    assert abs(L.orc_srgb8_to_float(128) - ((128 / 255 + 0.055) / 1.055) ** 2.4) < 1e-6


def test_bilinear_sampler_centres_and_midpoints():
    img = f16(np.arange(4 * 3 * 4, dtype=np.float32).reshape(3, 4, 4))
    out = np.zeros(4, np.float32)
    import ctypes as C
    L.orc_sample_linear_rgba16f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
    L.orc_sample_linear_rgba16f(img.ctypes.data, 4, 3, (1 + 0.5) / 4, (2 + 0.5) / 3, out.ctypes.data)
    np.testing.assert_allclose(out, f32(img)[2, 1], rtol=1e-6)
    L.orc_sample_linear_rgba16f(img.ctypes.data, 4, 3, 2.0 / 4, 1.0 / 3, out.ctypes.data)  # corner between 4 texels
    np.testing.assert_allclose(out, f32(img)[0:2, 1:3].mean(axis=(0, 1)), rtol=1e-6)
    L.orc_sample_linear_rgba16f(img.ctypes.data, 4, 3, -1.0, 5.0, out.ctypes.data)  # clamp to edge
    np.testing.assert_allclose(out, f32(img)[2, 0], rtol=1e-6)


# ---- post chain ------------------------------------------------------------------------------------------------------
def test_threshold_known_answer():
    c = np.array([2.0, 12.0, 4.0, 1.0], np.float32)
    hdr = np.broadcast_to(f16(c), (8, 8, 4)).copy()
    lum3 = np.array([0.0, 1.0, 1.0], np.float32)
    t = f32(orc.bloom_threshold(hdr, 4, 4, lum3))
    lum = 12.0 + 1e-4
    np.testing.assert_allclose(t[..., :3], np.broadcast_to(c[:3] / lum * (lum - 8.0), t[..., :3].shape), rtol=2e-3)
    np.testing.assert_allclose(t[..., 3], math.log2(lum), rtol=1e-3)
    # below the knee everything clamps to zero
    t0 = f32(orc.bloom_threshold(hdr, 4, 4, np.array([0.0, 2.0, 0.5], np.float32)))
    assert (t0[..., :3] == 0).all()
    # DYNAMIC_EXPOSURE off: fixed knee of 8
    t1 = f32(orc.bloom_threshold(hdr, 4, 4, None))
    np.testing.assert_allclose(t1[..., :3], np.broadcast_to(c[:3] / lum * (lum - 8.0), t1[..., :3].shape), rtol=2e-3)


def test_downsample_impulse_response_is_separable_tent():
    """2:1 even sizes: the 9 bilinear taps at 0 / +-1.75 texels collapse to the separable 6-tap [1,3,4,4,3,1]/16."""
    src = np.zeros((16, 16, 4), np.float32)
    src[8, 8] = 1024.0
    out = f32(orc.bloom_downsample(f16(src), 8, 8))[..., 0] / 1024.0
    w = np.array([1, 3, 4, 4, 3, 1], np.float64) / 16.0
    # output x collects input texels 2x-2 .. 2x+3 with weights w  =>  input 8 contributes to x = 3,4,5 with w[4],w[2],w[0]
    col = {3: w[4], 4: w[2], 5: w[0]}
    for y in range(8):
        for x in range(8):
            expect = col.get(x, 0.0) * col.get(y, 0.0)
            assert abs(out[y, x] - expect) < 1e-3, (x, y, out[y, x], expect)
    assert abs(out.sum() - 0.25) < 1e-3  # energy of one texel spread over a 4x smaller image


def test_tent_weights_sum_to_one_and_feedback_mix():
    c = np.array([0.75, 3.0, 0.5, -1.25], np.float32)
    img = np.broadcast_to(f16(c), (20, 28, 4)).copy()
    for fn, size in ((orc.bloom_downsample, (14, 10)), (orc.bloom_upsample, (56, 40))):
        np.testing.assert_allclose(f32(fn(img, *size)), np.broadcast_to(c, (size[1], size[0], 4)), rtol=1e-3)
    hist = np.broadcast_to(f16(np.array([4.0, 4.0, 4.0, 9.0], np.float32)), (10, 14, 4)).copy()
    mixed = f32(orc.bloom_downsample(img, 14, 10, hist, 0.25))
    np.testing.assert_allclose(mixed[0, 0, :3], 4.0 * 0.75 + c[:3] * 0.25, rtol=1e-3)
    np.testing.assert_allclose(mixed[0, 0, 3], c[3], rtol=1e-3)  # alpha lerp factor is 1: history ignored


def test_luminance_known_answer_and_clamp():
    d3 = np.zeros((8, 8, 4), np.float32)
    d3[..., 3] = 1.5
    lum = orc.luminance(f16(d3), np.array([0.25, 0, 0], np.float32), 0.5)
    assert abs(lum[0] - (0.25 * 0.5 + 1.5 * 0.5)) < 1e-6
    assert abs(lum[1] - 2.0 ** lum[0]) < 1e-6 and abs(lum[2] - 2.0 ** -lum[0]) < 1e-6
    d3[..., 3] = 7.0
    assert abs(orc.luminance(f16(d3), np.zeros(3, np.float32), 1.0)[0] - 2.0) < 1e-6  # clamp max
    d3[..., 3] = -9.0
    assert abs(orc.luminance(f16(d3), np.zeros(3, np.float32), 1.0)[0] + 3.0) < 1e-6  # clamp min


def test_tonemap_white_point_and_black():
    def run(value, exposure=1.0, lum=None):
        hdr = np.broadcast_to(f16(np.array([value, value, value, 1.0], np.float32)), (4, 4, 4)).copy()
        bloom = np.zeros((1, 1, 4), np.uint16)
        return orc.tonemap(hdr, bloom, lum, exposure)[0, 0]
    assert tuple(run(11.2)[:3]) == (255, 255, 255)  # filmic(W)/filmic(W) = 1
    assert tuple(run(0.0)) == (0, 0, 0, 255)
    # exposure scaling: x=2 with exposure 0.5 equals x=1
    np.testing.assert_array_equal(run(2.0, 0.5), run(1.0))
    np.testing.assert_array_equal(run(4.0, 1.0, np.array([2.0, 4.0, 0.25], np.float32)), run(1.0))
    # closed form at x = 1
    A, B, C_, D, E, F = 0.15, 0.5, 0.1, 0.2, 0.02, 0.3
    u = lambda x: ((x * (A * x + C_ * B) + D * E) / (x * (A * x + B) + D * F)) - E / F
    lin = u(1.0) / u(11.2)
    srgb = 1.055 * lin ** (1 / 2.4) - 0.055
    assert abs(int(run(1.0)[0]) - round(srgb * 255)) <= 1


# ---- lights ------------------------------------------------------------------------------------------------------------
def one_light_scene(light_type, color=(10.0, 10.0, 10.0), light_pos_view=(0.0, 0.0, -3.0), w=64, h=36):
    cam = synth.Camera(w, h)
    rp = cam.render_params()
    d = 4.0
    depth = np.full((h, w), cam.depth_from_view_distance(np.array(d))[()], np.float32)
    n_world = -cam.front  # facing the camera
    q = np.clip(np.rint((0.5 * n_world + 0.5) * 1023.0), 0, 1023).astype(np.uint32)
    normal = np.full((h, w), q[0] | (q[1] << 10) | (q[2] << 20) | (3 << 30), np.uint32)
    albedo = np.full((h, w), 188 | (188 << 8) | (188 << 16) | (255 << 24), np.uint32)  # sRGB 188 ~ 0.5 linear
    pbr = np.full((h, w), 0 | (128 << 8), np.uint16)
    gbuf = {"emissive": np.zeros((h, w, 4), np.uint16), "albedo": albedo, "normal": normal, "pbr": pbr, "depth": depth}
    descs = np.zeros(1, synth.LIGHT_DESC_DTYPE)
    world = (cam.invV @ np.array([*light_pos_view, 1.0]))[:3]
    descs["type"] = light_type
    descs["color"] = color
    descs["inner_cone"], descs["outer_cone"] = math.cos(math.radians(20)), math.cos(math.radians(30))
    descs["cutoff_range"] = 4.0
    tr = np.zeros((3, 4))
    fwd = cam.front  # spot looks along the view direction, at the surface
    z = -fwd
    x = np.cross([0, 1, 0], z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3] = x, y, z, world
    descs["transform"] = tr
    return cam, rp, gbuf, descs, world


@pytest.mark.parametrize("light_type", [1, 0])
def test_single_light_closed_form(light_type):
    cam, rp, gbuf, descs, lpos = one_light_scene(light_type)
    n, lights, model, tmask, _ = orc.pack_lights(descs, rp[99:102])
    assert n == 1 and (tmask[0] & 1) == light_type
    prm = orc.cluster_params(rp, 128, 64, 4096, n)
    cb = orc.cluster_build(rp, prm, lights, model, tmask, n, 4096)
    hdr = orc.lighting(gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], (0, 0, 0), (0, 1, 0), directional=False)
    h, w = gbuf["depth"].shape
    px, py = w // 2, h // 2
    got = f32(hdr)[py, px, :3]

    # float64 closed form at that pixel
    ndc = np.array([2 * (px + 0.5) / w - 1, 2 * (py + 0.5) / h - 1, gbuf["depth"][py, px], 1.0])
    clip = cam.invVP @ ndc
    pos = clip[:3] / clip[3]
    N = -cam.front
    Lv = lpos - pos
    dist = np.linalg.norm(Lv); Lv /= dist
    radius = 4.0
    t = np.clip((dist / radius - 0.9) / 0.1, 0, 1)
    atten = 1 - t * t * (3 - 2 * t)
    if light_type == 0:
        cone = np.dot(-Lv, cam.front)
        scale = 1 / (math.cos(math.radians(20)) - math.cos(math.radians(30)))
        bias = -math.cos(math.radians(30)) * scale
        atten *= np.clip(cone * scale + bias, 0, 1) ** 2
    color = np.array([10.0] * 3) * atten / dist ** 2
    V = cam.position - pos; V /= np.linalg.norm(V)
    H = V + Lv; H /= np.linalg.norm(H)
    NoV, NoL, HoV, NoH = (np.clip(np.dot(a, b), 0.001, 1) for a, b in ((N, V), (N, Lv), (H, V), (N, H)))
    base = ((188 / 255 + 0.055) / 1.055) ** 2.4
    rough = (128 / 255) * 0.75 + 0.25
    F0 = 0.04
    F = F0 + (1 - F0) * (1 - HoV) ** 5
    PI = 3.1415628
    m2 = rough ** 4
    dd = (NoH * m2 - NoH) * NoH + 1
    D = m2 / (PI * dd * dd)
    k = (rough + 1) ** 2 / 8
    G = 0.25 / max((NoV * (1 - k) + k) * (NoL * (1 - k) + k), 0.001)
    expect = color * NoL * (F * G * D + (1 - F) / PI * base)
    np.testing.assert_allclose(got, expect, rtol=4e-3, atol=1e-4)
    assert got[0] > 0.01


def test_clustering_is_conservative_equals_bruteforce_and_wave_union():
    cam = synth.Camera(160, 90)
    rp = cam.render_params()
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, 600)
    n, lights, model, tmask, _ = orc.pack_lights(descs, rp[99:102])
    prm = orc.cluster_params(rp, 128, 64, 4096, n)
    for tile_h in (8, 4, 0):  # wave64 / wave32 subgroup variants and the per-cell fallback of binning.comp
        cb = orc.cluster_build(rp, prm, lights, model, tmask, n, 4096, subgroup_tile_h=tile_h)
        a = orc.lighting(gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], (0, 0, 0), (0, 1, 0), directional=False)
        b = orc.lighting(gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], (0, 0, 0), (0, 1, 0), directional=False, bruteforce=True)
        c = orc.lighting(gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], (0, 0, 0), (0, 1, 0), directional=False, wave_tile=8)
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, c)


def test_light_packing_known_values():
    cam = synth.Camera(64, 36)
    descs = np.zeros(2, synth.LIGHT_DESC_DTYPE)
    descs["type"] = [0, 1]
    descs["color"] = [[10.0, 5.0, 2.0], [0.05, 0.1, 0.02]]
    descs["inner_cone"], descs["outer_cone"] = math.cos(math.radians(20)), math.cos(math.radians(30))
    descs["cutoff_range"] = 4.0
    tr = np.zeros((2, 3, 4)); tr[:, 0, 0] = tr[:, 1, 1] = tr[:, 2, 2] = 1.0
    tr[0, :, 3] = (0, 0, -5); tr[1, :, 3] = (0, 0, -1)
    descs["transform"] = tr
    n, lights, model, tmask, order = orc.pack_lights(descs, np.array([0, 0, -1], np.float32))
    assert n == 2 and list(order) == [1, 0]  # sorted front-to-back along camera_front
    assert tmask[0] == 0b01  # first (nearest) is the point light
    # point: falloff range sqrt(0.1/0.1) = 1 < cutoff 4
    assert abs(lights["inv_radius"][0] - 1.0) < 1e-6
    assert lights["offset_radius"][0] >> 16 == int(f16(1.0)) and lights["offset_radius"][0] & 0xffff == 0
    # spot: range min(sqrt(10/0.1)=10, 4) = 4; scale = 1/(cos20-cos30), bias = -cos30*scale
    assert abs(lights["inv_radius"][1] - 0.25) < 1e-7
    scale = 1.0 / (math.cos(math.radians(20)) - math.cos(math.radians(30)))
    sb = np.array([lights["spot_scale_bias"][1] & 0xffff, lights["spot_scale_bias"][1] >> 16], np.uint16).view(np.float16)
    assert abs(float(sb[0]) - scale) / scale < 1e-3 and abs(float(sb[1]) + math.cos(math.radians(30)) * scale) < 1e-2
    np.testing.assert_allclose(lights["direction"][1], [0, 0, -1], atol=1e-7)
    # model = transform * scale(tan(30deg)*4, tan(30deg)*4, 4)
    np.testing.assert_allclose(model[1][:, :3].diagonal(), [math.tan(math.radians(30)) * 4] * 2 + [4.0], rtol=1e-6)


def test_z_range_and_uint_ranges_known_answers():
    import ctypes as C
    light_ranges = np.array([[0, 2], [1, 1], [3, 5], [0xffffffff, 0]], np.uint32)
    out = np.zeros((8, 2), np.uint32)
    L.orc_cluster_z_range(light_ranges.ctypes.data_as(C.c_void_p), 4, 8, out.ctypes.data_as(C.c_void_p))
    assert out.tolist() == [[0, 0], [0, 1], [0, 0], [2, 2], [2, 2], [2, 2], [0xffffffff, 0], [0xffffffff, 0]]
    cam = synth.Camera(64, 36)
    rp = cam.render_params()
    descs = np.zeros(3, synth.LIGHT_DESC_DTYPE)
    descs["type"] = 1
    descs["color"] = 1000.0
    descs["cutoff_range"] = [1.0, 2.0, 0.5]
    tr = np.zeros((3, 3, 4)); tr[:, 0, 0] = tr[:, 1, 1] = tr[:, 2, 2] = 1.0
    for i, dist in enumerate((3.0, 10.0, -2.0)):  # the last one is behind the camera
        tr[i, :, 3] = cam.position + cam.front * dist
    descs["transform"] = tr
    n, lights, model, tmask, order = orc.pack_lights(descs, rp[99:102])
    zr = orc.light_z_ranges(rp, lights, model, tmask, n, 4096)
    extent = 100.0 / 4096
    by_src = {int(o): zr[i] for i, o in enumerate(order)}
    assert by_src[2].tolist() == [0xffffffff, 0]  # entirely behind the camera
    assert abs(int(by_src[0][0]) - int(2.0 / extent)) <= 1 and abs(int(by_src[0][1]) - int(4.0 / extent)) <= 1
    assert abs(int(by_src[1][0]) - int(8.0 / extent)) <= 1 and abs(int(by_src[1][1]) - int(12.0 / extent)) <= 1


def test_cluster_params_layout_and_values():
    cam = synth.Camera(1920, 1080)
    rp = cam.render_params()
    prm = orc.cluster_params(rp, 128, 64, 4096, 100)
    assert prm["num_lights_32"][0] == 4 and prm["z_max_index"][0] == 4095
    assert abs(prm["z_scale"][0] - 4096 / 100.0) < 1e-3  # 1 / min(0.5, z_far / res_z)
    np.testing.assert_allclose(prm["camera_base"][0], cam.position, rtol=1e-6)
    # transform = T(.5,.5,0) S(.5,.5,1) VP maps the view centre to (0.5, 0.5)
    p = cam.position + cam.front * 5.0
    t = prm["transform"][0].reshape(4, 4).T @ np.array([*p, 1.0])
    np.testing.assert_allclose(t[:2] / t[3], [0.5, 0.5], atol=1e-5)


# ---- committed fixtures --------------------------------------------------------------------------------------------------
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "oracle_golden_v1.npz")


def test_oracle_reproduces_committed_fixtures():
    from golden.make_golden import compute
    ref = np.load(GOLDEN)
    now = compute()
    assert set(ref.files) == set(now)
    for k in ref.files:
        if now[k].dtype == np.uint16 and now[k].ndim == 3:  # RGBA16F: libm differences across hosts stay within 1 ulp
            d = np.abs(now[k].view(np.int16).astype(np.int32) - ref[k].view(np.int16).astype(np.int32))
            assert d.max() <= 1, k
        elif now[k].dtype == np.uint8:
            assert np.abs(now[k].astype(np.int16) - ref[k].astype(np.int16)).max() <= 1, k
        elif now[k].dtype == np.float32:
            np.testing.assert_allclose(now[k], ref[k], rtol=1e-5, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_array_equal(now[k], ref[k], err_msg=k)


def test_unorm8_decode_without_a_division_is_exact():
    """device_common.hpp unorm8_to_float: fma(v, hi, fl(v * lo)) with hi = fl(1/255), lo = fl(1/255 - hi) equals the IEEE
    quotient v / 255 for every byte (exact rational arithmetic, one rounding per operation, ties to even), so kernels may
    use it where the oracle divides."""
    from fractions import Fraction
    f32 = np.float32

    def rounded(q: Fraction):
        near = f32(float(q))  # within one ulp: pick the best of it and its neighbours exactly
        cands = [np.nextafter(near, f32(-np.inf)), near, np.nextafter(near, f32(np.inf))]
        return min(cands, key=lambda c: (abs(Fraction(float(c)) - q), int(f32(c).view(np.uint32)) & 1))

    hi, lo = f32(float.fromhex("0x1.010102p-8")), f32(float.fromhex("-0x1.fdfdfep-33"))
    assert hi == f32(1.0) / f32(255.0)
    for v in range(256):
        p = rounded(Fraction(v) * Fraction(float(lo)))
        got = rounded(Fraction(v) * Fraction(float(hi)) + Fraction(float(p)))
        assert got == f32(v) / f32(255.0), v


def test_ambient_occlusion_scales_only_the_fallback_ambient_term():
    """directional.frag:52-64 under LIGHTING_NO_AMBIENT + VOLUMETRIC_DIFFUSE_FALLBACK: FragColor += ao * base_color * 0.05.
    White AO = the plain variant; black AO = the variant without the fallback term; grey is linear in between (before the
    fp16 store)."""
    cam = synth.Camera(96, 54)
    gbuf = synth.make_gbuffer(cam)
    rp = cam.render_params()
    n, lights, model, tmask, _ = orc.pack_lights(synth.make_lights(cam, 40), rp[99:102])
    prm = orc.cluster_params(rp, *synth.CLUSTER_RESOLUTION, n)
    cb = orc.cluster_build(rp, prm, lights, model, tmask, n, synth.CLUSTER_RESOLUTION[2])

    def run(**kw):
        return orc.lighting(gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION,
                            clustered=False, **kw)

    plain, none = run(), run(ambient_fallback=False)
    np.testing.assert_array_equal(run(ambient_occlusion=np.full((54, 96), 255, np.uint8)), plain)
    np.testing.assert_array_equal(run(ambient_occlusion=np.zeros((54, 96), np.uint8)), none)
    half = f32(run(ambient_occlusion=np.full((27, 48), 51, np.uint8)))  # 51 / 255 = 0.2, any size: constant image
    lit = gbuf["depth"] != 0.0
    want = f32(none)[lit] + 0.2 * (f32(plain)[lit] - f32(none)[lit])
    np.testing.assert_allclose(half[lit][:, :3], want[:, :3], rtol=4e-3, atol=2e-4)
    assert np.abs(f32(plain)[lit][:, :3] - f32(none)[lit][:, :3]).max() > 1e-3


def test_pq10_known_answers():
    """SMPTE ST 2084: 100 nits -> 0.508, 1000 nits -> 0.7518, 10000 -> 1, 0 -> ~0 (c1^m2); the encoder soft-clips above 75 % of
    maxContentLightLevel (x -> 4x / (1 + 4x)) and the UI layer's alpha is the scene's visibility."""
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
    hdr = np.zeros((1, 6, 4), np.float16)
    hdr[0, :, :3] = np.array([0.0, 0.2, 1.0, 1.4, 2.0, 200.0])[:, None]  # x 500 nits pre-exposure
    ui = np.zeros((1, 6, 4), np.uint8)
    ui[..., 3] = 255
    out = orc.pq10_encode(hdr.view(np.uint16), ui, ident, 500.0, 400.0, 1000.0)[0]
    r = (out & 1023).astype(int)
    assert ((out >> 10) & 1023 == out & 1023).all() and ((out >> 30) == 3).all()
    def pq(nits):
        y = nits / 10000.0
        return ((0.8359375 + 18.8515625 * y ** 0.1593017578125) / (1 + 18.6875 * y ** 0.1593017578125)) ** 78.84375
    clip = lambda x: 4 * x / (1 + 4 * x) if x > 0.75 else x
    want = [round(1023 * pq(1000.0 * clip(v * 500.0 / 1000.0))) for v in (0.0, 0.2, 1.0, 1.4, 2.0, 200.0)]
    assert np.abs(r - np.array(want)).max() <= 1, (r, want)
    assert r[0] == 0 and abs(r[1] - 520) <= 1 and r[5] < 1023 * pq(1000.0) + 1
    # alpha 0 hides the scene, the UI colour alone remains: white UI = 400 nits
    ui[...] = (255, 255, 255, 0)
    out = orc.pq10_encode(hdr.view(np.uint16), ui, ident, 500.0, 400.0, 1000.0)[0]
    assert np.abs((out & 1023).astype(int) - round(1023 * pq(400.0))).max() <= 1


def test_product_side_packed_float_encoder_equals_the_oracle_on_every_half():
    """granite_amd/synth.py packs the synthetic emissive image for renderTargetFp16 = false set-ups (bench, headless runner) without the
    oracle: every half-float bit pattern, in each channel position, must give the oracle's packed word."""
    h = np.arange(65536, dtype=np.uint16)
    img = np.zeros((65536, 1, 4), np.uint16)
    img[:, 0, 0], img[:, 0, 1], img[:, 0, 2] = h, h[::-1], np.roll(h, 12345)
    np.testing.assert_array_equal(synth.pack_b10g11r11(img), orc.pack_b10g11r11(img))
