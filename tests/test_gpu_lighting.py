"""GPU parity: cluster build (bit-exact) and the fused lighting kernel (fp16 tolerance) vs the CPU oracle."""
import numpy as np
import pytest

from granite_amd import capi, synth
from oracle import oracle as orc
from gpu_scene import Scene
from util import assert_rgba16f_close, rgba16f_mismatch

pytestmark = pytest.mark.gpu

ALL = capi.LIGHTING_DIRECTIONAL_BIT | capi.LIGHTING_CLUSTERED_BIT | capi.LIGHTING_AMBIENT_FALLBACK_BIT


@pytest.mark.parametrize("num_lights", [0, 1, 31, 256, 1000, 4096, 5000])
def test_cluster_build_bit_exact(gr, num_lights):
    sc = Scene(480, 270, num_lights)
    assert sc.n == min(num_lights, 4096)
    ref = orc.cluster_build(sc.rp, sc.prm, sc.lights, sc.model, sc.type_mask, sc.n, sc.res[2], subgroup_tile_h=8)
    dev = sc.build_clusters_gpu(gr)
    got_range = dev["range"].download(np.uint32).reshape(-1, 2)
    np.testing.assert_array_equal(got_range, ref["range"])
    if sc.n == 0:
        return
    got_spots = dev["spots"].download(np.float32).reshape(4096, 24)[:sc.n]
    np.testing.assert_array_equal(got_spots.view(np.uint32), ref["spots"][:sc.n].view(np.uint32))
    got_setup = dev["setup"].download(np.uint32).reshape(4096, 128)[:sc.n]
    np.testing.assert_array_equal(got_setup, ref["setup"][:sc.n].view(np.uint32))
    n32 = (sc.n + 31) // 32
    got_mask = dev["bitmask"].download(np.uint32)[:sc.res[0] * sc.res[1] * n32]
    np.testing.assert_array_equal(got_mask, ref["bitmask"])


@pytest.mark.parametrize("num_lights,num_ranges,case", [(4096, 4096, "all_empty"), (1, 64, "sorted"), (127, 128, "mixed"), (128, 128, "wide"), (129, 320, "sorted"),
                                                        (1000, 1024, "mixed"), (4096, 4096, "sorted"), (4096, 1024, "wide"), (4096, 4096, "mixed")])
def test_z_range_equals_the_executed_opt_shader(gr, num_lights, num_ranges, case):
    """gr_cluster_z_range over arbitrary slice intervals -- from one light to the 4096-light maximum, unsorted and empty intervals, and
    the all-empty input of the reference's tests/z_binning_test.cpp:56-96 -- bit for bit against the oracle and, where the library
    of executed reference shaders travelled to this box (oracle/_ref), against clusterer_bindless_z_range_opt.comp itself run at
    subgroup size 64 (the shader a subgroup-capable device dispatches, clusterer.cpp:1305-1314)."""
    import ctypes as C
    import os
    from test_reference_shaders_cpu import REF_LIB, z_range_inputs
    zr = z_range_inputs(case, num_lights, num_ranges, seed=num_lights * 7 + num_ranges)
    want = np.zeros((num_ranges, 2), np.uint32)
    orc.lib().orc_cluster_z_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    orc.lib().orc_cluster_z_range(zr.ctypes.data, num_lights, num_ranges, want.ctypes.data)
    zr_buf = capi.DeviceBuffer(gr, zr.nbytes).upload(zr)
    out = capi.DeviceBuffer(gr, num_ranges * 8).upload(np.full(num_ranges * 2, 0xdeadbeef, np.uint32))
    gr.check(gr.lib.gr_cluster_z_range(gr.handle, None, zr_buf.ptr, out.ptr, capi.PushZRange(num_lights, (num_lights + 127) // 128, num_ranges)))
    gr.sync()
    got = out.download(np.uint32).reshape(-1, 2)
    np.testing.assert_array_equal(got, want, err_msg="kernel vs oracle")
    if os.path.exists(REF_LIB):
        ref = C.CDLL(REF_LIB)
        ref.ref_cluster_z_range_opt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        shader = np.zeros((num_ranges, 2), np.uint32)
        ref.ref_cluster_z_range_opt(zr.ctypes.data, num_lights, num_ranges, shader.ctypes.data, 64)
        np.testing.assert_array_equal(got, shader, err_msg="kernel vs clusterer_bindless_z_range_opt.comp, executed")


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("num_lights", [1, 300, 4096])
def test_cluster_front_as_one_launch_equals_the_separate_launches(gr, num_lights, pinned):
    """gr_cluster_front (the four uploads + spot_transform + setup + z_range in one grid) + gr_cluster_binning: every buffer
    bit-identical to the launch-by-launch build, whether the packed light data is read from the transforms buffer or from pinned
    host memory (and then copied into the buffers by the launch itself)."""
    sc = Scene(480, 270, num_lights)
    a, b = sc.build_clusters_gpu(gr), sc.build_clusters_gpu_fused(gr, pinned)
    for key in ("spots", "setup", "bitmask", "range", "transforms"):
        np.testing.assert_array_equal(b[key].download(np.uint32), a[key].download(np.uint32), err_msg=key)
    zr = np.ascontiguousarray(b["zr"]).view(np.uint32).reshape(-1)
    np.testing.assert_array_equal(b["zr_buf"].download(np.uint32)[:zr.size], zr)  # the intervals reached their buffer either way


@pytest.mark.parametrize("w,h,num_lights", [(480, 270, 256), (480, 270, 4096), (333, 77, 700), (1920, 1080, 256)])
def test_lighting_matches_oracle(gr, w, h, num_lights):
    sc = Scene(w, h, num_lights)
    ref_c = orc.cluster_build(sc.rp, sc.prm, sc.lights, sc.model, sc.type_mask, sc.n, sc.res[2])
    dev = sc.build_clusters_gpu(gr)
    for flags, kw in ((ALL, {}), (capi.LIGHTING_CLUSTERED_BIT, dict(directional=False)),
                      (capi.LIGHTING_DIRECTIONAL_BIT, dict(clustered=False, ambient_fallback=False))):
        ref = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, ref_c["bitmask"], ref_c["range"],
                           synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION, **kw)
        args, imgs = sc.lighting_args(gr, dev, flags)
        gr.check(gr.lib.gr_lighting(gr.handle, None, args))
        gr.sync()
        got = imgs["hdr"].download()
        # alpha and sky pixels must be bit-identical to the input
        np.testing.assert_array_equal(got[..., 3], sc.gbuf["emissive"][..., 3])
        sky = sc.gbuf["depth"] == 0.0
        np.testing.assert_array_equal(got[sky], sc.gbuf["emissive"][sky])
        # 2 ulp fp16 + 1e-4 (two blend roundings, each at most one ulp apart; histogram: tools/ulp_hist.py).
        assert_rgba16f_close(got, ref, ulps=2.0, abs_tol=1e-4, what=f"lighting {w}x{h} n={num_lights} flags={flags}")
        exact = (got == ref).mean()
        assert exact > 0.95, f"only {exact:.3f} of channels bit-identical"


@pytest.mark.parametrize("w,h", [(480, 270), (333, 77), (1280, 720)])
def test_lighting_wide_light_index_windows_match_oracle(gr, w, h):
    """Scene "depth_split" (granite_amd/synth.py): half of the surface is a wall 30 units from the camera, where a Z slice's light range
    spans some 1400 indices of the 4096-light scene (22 chunks of 64), and a third of the tiles straddles the 1-7 / 30 unit discontinuity.
    These tiles take the kernel's wide-window path (the window's cluster words read once, their set bits compacted into a list, 64 list
    entries per turn; the gap between the foreground's and the background's ranges trimmed) -- two-pixel form at the even widths, one-pixel
    form (16-word spans) at 333.  Against the oracle's exact per-pixel light sets, and against its unclustered sum over all lights.  The
    allowance: tests/test_gpu_fullsize.py::test_worst_case_scenes_match_oracle_at_4k (lights within 0.1 units of a surface 30 to 43 units away)."""
    from util import assert_rgba16f_close_but_for_ill_conditioned_pixels
    sc = Scene(w, h, 4096, scene="depth_split")
    ref_c = orc.cluster_build(sc.rp, sc.prm, sc.lights, sc.model, sc.type_mask, sc.n, sc.res[2])
    dev = sc.build_clusters_gpu(gr)
    ref = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, ref_c["bitmask"], ref_c["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
    args, imgs = sc.lighting_args(gr, dev, ALL)
    gr.check(gr.lib.gr_lighting(gr.handle, None, args))
    gr.sync()
    got = imgs["hdr"].download()
    assert_rgba16f_close_but_for_ill_conditioned_pixels(got, ref, ulps=2.0, abs_tol=1e-4, max_pixels=2, outer_ulps=8.0, what=f"depth_split {w}x{h}")
    assert (got == ref).mean() > 0.95
    if w == 480:
        brute = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, np.zeros(1, np.uint32), np.zeros((1, 2), np.uint32),
                             synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION, directional=False, bruteforce=True)
        args, imgs = sc.lighting_args(gr, dev, capi.LIGHTING_CLUSTERED_BIT)
        gr.check(gr.lib.gr_lighting(gr.handle, None, args))
        gr.sync()
        assert_rgba16f_close_but_for_ill_conditioned_pixels(imgs["hdr"].download(), brute, ulps=2.0, max_pixels=2, outer_ulps=8.0, what="wide windows vs brute force")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lighting_with_three_depth_layers_and_holes_matches_oracle(gr, seed):
    """The window logic of the wide path under its worst inputs: every 16 x 8 tile mixes up to three depth layers (2-6, 14-16 and 30-36 units, in
    blocks of 1 to 5 pixels, so that pixel PAIRS of one lane straddle layers too) with sky holes (pixels without a range) -- the gap between the
    foreground's ranges and the rest is trimmed, the middle layer sits inside what a two-layer assumption would have dropped.  4096 lights, both
    pixel-per-lane forms; against the oracle's exact per-pixel light sets."""
    from util import assert_rgba16f_close_but_for_ill_conditioned_pixels
    for w, h in ((320, 180), (251, 96)):
        sc = Scene(w, h, 4096)
        r = np.random.default_rng(seed * 100 + w)
        layer = np.zeros((h, w), np.int32)
        y = 0
        while y < h:
            bh = int(r.integers(1, 6))
            x = 0
            while x < w:
                bw = int(r.integers(1, 6))
                layer[y:y + bh, x:x + bw] = r.integers(0, 4)
                x += bw
            y += bh
        view_z = np.choose(np.minimum(layer, 2), [r.uniform(2.0, 6.0, (h, w)), r.uniform(14.0, 16.0, (h, w)), r.uniform(30.0, 36.0, (h, w))])
        depth = sc.cam.depth_from_view_distance(view_z).astype(np.float32)
        depth[layer == 3] = 0.0  # sky
        sc.gbuf["depth"] = depth
        ref_c = orc.cluster_build(sc.rp, sc.prm, sc.lights, sc.model, sc.type_mask, sc.n, sc.res[2])
        dev = sc.build_clusters_gpu(gr)
        ref = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, ref_c["bitmask"], ref_c["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
        args, imgs = sc.lighting_args(gr, dev, ALL)
        gr.check(gr.lib.gr_lighting(gr.handle, None, args))
        gr.sync()
        got = imgs["hdr"].download()
        sky = depth == 0.0
        np.testing.assert_array_equal(got[sky], sc.gbuf["emissive"][sky])
        assert_rgba16f_close_but_for_ill_conditioned_pixels(got, ref, ulps=2.0, abs_tol=1e-4, max_pixels=3, outer_ulps=8.0, what=f"three layers {w}x{h} seed {seed}")


def test_lighting_separate_emissive_equals_aliased(gr):
    """emissive as a distinct input attachment gives bit-identical HDR to the aliased read-modify-write form and leaves
    the G-buffer untouched."""
    sc = Scene(333, 77, 700)
    dev = sc.build_clusters_gpu(gr)
    a1, i1 = sc.lighting_args(gr, dev, ALL, alias_emissive=True)
    gr.check(gr.lib.gr_lighting(gr.handle, None, a1))
    a2, i2 = sc.lighting_args(gr, dev, ALL, alias_emissive=False)
    gr.check(gr.lib.gr_lighting(gr.handle, None, a2))
    gr.sync()
    np.testing.assert_array_equal(i1["hdr"].download(), i2["hdr"].download())
    np.testing.assert_array_equal(i2["emissive"].download(), sc.gbuf["emissive"])


def test_lighting_equals_bruteforce_sum(gr):
    """Size-independent property: the clustered result equals the unclustered sum over ALL lights (culling is
    conservative), evaluated by the oracle's brute-force loop."""
    sc = Scene(320, 180, 1500)
    dev = sc.build_clusters_gpu(gr)
    ref = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, np.zeros(1, np.uint32), np.zeros((1, 2), np.uint32),
                       synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION, directional=False, bruteforce=True)
    args, imgs = sc.lighting_args(gr, dev, capi.LIGHTING_CLUSTERED_BIT)
    gr.check(gr.lib.gr_lighting(gr.handle, None, args))
    gr.sync()
    assert_rgba16f_close(imgs["hdr"].download(), ref, ulps=2.0, what="clustered vs brute force")


def test_lighting_linearity_in_light_colour(gr):
    """Doubling every light colour doubles the added radiance (up to fp16 rounding of the sum)."""
    sc = Scene(256, 128, 300)
    dev = sc.build_clusters_gpu(gr)
    sc.gbuf["emissive"][...] = 0
    args, imgs = sc.lighting_args(gr, dev, capi.LIGHTING_CLUSTERED_BIT)
    gr.check(gr.lib.gr_lighting(gr.handle, None, args))
    gr.sync()
    base = imgs["hdr"].download().view(np.float16).astype(np.float32)
    sc.lights["color"] *= 2.0
    dev2 = {**dev, "transforms": sc.upload_transforms(gr)}
    args2, imgs2 = sc.lighting_args(gr, dev2, capi.LIGHTING_CLUSTERED_BIT)
    gr.check(gr.lib.gr_lighting(gr.handle, None, args2))
    gr.sync()
    dbl = imgs2["hdr"].download().view(np.float16).astype(np.float32)
    np.testing.assert_allclose(dbl[..., :3], 2.0 * base[..., :3], rtol=2e-3, atol=1e-4)


def ambient_occlusion_image(w, h, seed=5):
    r = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    ao = 128 + 100 * np.sin(x * 0.11) * np.cos(y * 0.07) + r.integers(-20, 21, (h, w))
    return np.clip(ao, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("ao_size,lights,scene", [("full", 700, "default"), ("half", 700, "default"), ("half", 4096, "depth_split")])
def test_lighting_ambient_occlusion_variant(gr, ao_size, lights, scene):
    """AMBIENT_OCCLUSION (renderer.cpp:1050-1051, directional.frag:52-64): the SSAO texture, sampled LinearClamp at the pixel
    centre, scales the 0.05 fallback ambient term.  Full-size and half-size (bilinear weights in play) inputs; scene "depth_split" with 4096
    lights takes the AO instantiation through its wide-window path."""
    w, h = 480, 270
    sc = Scene(w, h, lights, scene=scene)
    ref_c = orc.cluster_build(sc.rp, sc.prm, sc.lights, sc.model, sc.type_mask, sc.n, sc.res[2])
    dev = sc.build_clusters_gpu(gr)
    aw, ah = (w, h) if ao_size == "full" else (w // 2, h // 2)
    ao = ambient_occlusion_image(aw, ah)
    ref = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, ref_c["bitmask"], ref_c["range"],
                       synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION, ambient_occlusion=ao)
    args, imgs = sc.lighting_args(gr, dev, ALL | capi.LIGHTING_AMBIENT_OCCLUSION_BIT)
    ao_img = capi.DeviceImage(gr, aw, ah, capi.FORMAT_R8_UNORM).upload(ao)
    args.ambient_occlusion = ao_img.desc
    gr.check(gr.lib.gr_lighting(gr.handle, None, args))
    gr.sync()
    got = imgs["hdr"].download()
    if scene == "default":
        assert_rgba16f_close(got, ref, ulps=2.0, abs_tol=1e-4, what=f"lighting with {ao_size}-size AO")
    else:
        from util import assert_rgba16f_close_but_for_ill_conditioned_pixels
        assert_rgba16f_close_but_for_ill_conditioned_pixels(got, ref, ulps=2.0, abs_tol=1e-4, max_pixels=2, outer_ulps=8.0, what=f"lighting with {ao_size}-size AO, {scene}")
    # it matters: the result differs from the un-occluded one, and white AO reproduces it exactly
    args2, imgs2 = sc.lighting_args(gr, dev, ALL)
    gr.check(gr.lib.gr_lighting(gr.handle, None, args2))
    white = capi.DeviceImage(gr, aw, ah, capi.FORMAT_R8_UNORM).upload(np.full((ah, aw), 255, np.uint8))
    args3, imgs3 = sc.lighting_args(gr, dev, ALL | capi.LIGHTING_AMBIENT_OCCLUSION_BIT)
    args3.ambient_occlusion = white.desc
    gr.check(gr.lib.gr_lighting(gr.handle, None, args3))
    gr.sync()
    plain = imgs2["hdr"].download()
    assert (plain != got).mean() > 0.2
    np.testing.assert_array_equal(imgs3["hdr"].download(), plain)
    # the bit without an image, or without the fallback term it scales, is refused
    bad, _ = sc.lighting_args(gr, dev, ALL | capi.LIGHTING_AMBIENT_OCCLUSION_BIT)
    with pytest.raises(capi.GraniteHipError):
        gr.check(gr.lib.gr_lighting(gr.handle, None, bad))


def test_lights_smaller_than_the_distance_clamp(gr):
    """Lights of radius < 1/9 sitting on surfaces: pixels closer than 0.1 to the light exercise light_dist = max(0.1, dist)
    (point.h:36-38 / spot.h:38-40) with a non-zero smoothstep argument -- the kernel compiles that clamp only into the walk
    used for chunks that hold such a light -- and the 1 / max(dist, 0.1)^2 ceiling.  Mixed with ordinary lights so that both
    walks run in one launch."""
    w, h = 480, 270
    sc = Scene(w, h, 200)
    cam, depth = sc.cam, sc.gbuf["depth"]
    rng = np.random.default_rng(5)
    ys, xs = rng.integers(8, h - 8, 120), rng.integers(8, w - 8, 120)
    keep = depth[ys, xs] != 0.0
    ys, xs = ys[keep], xs[keep]
    ndc = np.stack([2.0 * (xs + 0.5) / w - 1.0, 2.0 * (ys + 0.5) / h - 1.0, depth[ys, xs].astype(np.float64), np.ones(len(xs))])
    clip = cam.invVP @ ndc
    pos = (clip[:3] / clip[3]).T
    tiny = np.zeros(len(xs), synth.LIGHT_DESC_DTYPE)
    tiny["type"] = np.where(np.arange(len(xs)) % 3 == 0, 0, 1)   # every third one a spot
    tiny["color"] = rng.uniform(20.0, 60.0, (len(xs), 3))
    tiny["inner_cone"], tiny["outer_cone"] = np.cos(np.radians(20.0)), np.cos(np.radians(40.0))
    tiny["cutoff_range"] = rng.uniform(0.04, 0.105, len(xs))
    tr = np.zeros((len(xs), 3, 4))
    tr[:, :, :3] = np.eye(3)
    toward_camera = cam.position[None, :] - pos
    toward_camera /= np.linalg.norm(toward_camera, axis=1, keepdims=True)
    tr[:, :, 3] = pos + toward_camera * rng.uniform(0.0, 0.05, (len(xs), 1))
    # spots look along -Z of the node: point them at the surface (away from the camera)
    z = toward_camera
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    tr[:, :, 0], tr[:, :, 1], tr[:, :, 2] = x, np.cross(z, x), z
    tiny["transform"] = tr.astype(np.float32)
    sc.descs = np.concatenate([sc.descs, tiny])
    sc.n, sc.lights, sc.model, sc.type_mask, sc.order = orc.pack_lights(sc.descs, sc.rp[99:102])
    sc.prm = orc.cluster_params(sc.rp, sc.res[0], sc.res[1], sc.res[2], sc.n)
    inv_radius = sc.lights["inv_radius"][:sc.n]
    assert (inv_radius > 9.0).sum() >= 60, "the scene must hold lights with radius < 1/9"

    ref_c = orc.cluster_build(sc.rp, sc.prm, sc.lights, sc.model, sc.type_mask, sc.n, sc.res[2])
    dev = sc.build_clusters_gpu(gr)
    base = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, np.zeros_like(ref_c["bitmask"]), ref_c["range"],
                        synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
    ref = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, ref_c["bitmask"], ref_c["range"],
                       synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
    args, imgs = sc.lighting_args(gr, dev, ALL)
    gr.check(gr.lib.gr_lighting(gr.handle, None, args))
    gr.sync()
    got = imgs["hdr"].download()
    assert_rgba16f_close(got, ref, ulps=2.0, abs_tol=1e-4, what="tiny lights")
    # the tiny lights do reach pixels: around their centres the frame differs from the frame without positional lights
    lit = (ref != base).any(axis=2)
    near = np.zeros((h, w), bool)
    for yy, xx in zip(ys, xs):
        near[yy - 2:yy + 3, xx - 2:xx + 3] = True
    assert (lit & near).sum() > 200


@pytest.mark.parametrize("packed", [False, True])
def test_fog_quad_behind_the_clustered_quad(gr, packed):
    """render_light's fog quad (renderer.cpp:1179-1196, fog.frag): the third blend fused behind the lighting store, into an
    RGBA16F target (colour + alpha) and into a B10G11R11 one; sky pixels untouched; falloff 0 = no fog."""
    w, h = 256, 144
    sc = Scene(w, h, 300)
    if packed:
        sc.gbuf["emissive"] = orc.quantize_b10g11r11(sc.gbuf["emissive"])
    ref_c = orc.cluster_build(sc.rp, sc.prm, sc.lights, sc.model, sc.type_mask, sc.n, sc.res[2])
    dev = sc.build_clusters_gpu(gr)
    fog = ((0.35, 0.4, 0.55), 0.012)
    want = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, ref_c["bitmask"], ref_c["range"], synth.DIRECTIONAL_COLOR,
                        synth.DIRECTIONAL_DIRECTION, b10g11r11=packed, fog=fog)
    args, imgs = sc.lighting_args(gr, dev, ALL)
    if packed:
        target = capi.DeviceImage(gr, w, h, capi.FORMAT_B10G11R11_UFLOAT_PACK32).upload(orc.pack_b10g11r11(sc.gbuf["emissive"]))
        args.hdr = target.desc
        args.emissive = target.desc
    else:
        target = imgs["hdr"]
    args.fog_color[:] = fog[0]
    args.fog_falloff = fog[1]
    gr.check(gr.lib.gr_lighting(gr.handle, None, args))
    gr.sync()
    got = target.download()
    if packed:
        want_words = orc.pack_b10g11r11(want)
        for shift, bits in ((0, 11), (11, 11), (22, 10)):
            a, b = (got >> shift) & ((1 << bits) - 1), (want_words >> shift) & ((1 << bits) - 1)
            assert np.abs(a.astype(np.int64) - b.astype(np.int64)).max() <= 1
        assert (got == want_words).mean() > 0.995
    else:
        assert_rgba16f_close(got, want, ulps=2.0, what="fogged HDR")
        sky = sc.gbuf["depth"] == 0.0
        np.testing.assert_array_equal(got[sky], sc.gbuf["emissive"][sky])
        assert (got[~sky][:, 3] != sc.gbuf["emissive"][~sky][:, 3]).mean() > 0.9  # the blend factors apply to alpha as well


def test_the_share_registers_hint_does_not_change_a_byte(gr):
    """GR_LIGHTING_SHARE_REGISTERS_BIT only caps the launch's residency (four workgroups per CU instead of five: room for the
    register-heavy passes of other streams); which workgroup runs when does not enter any pixel."""
    sc = Scene(640, 360, 900)
    dev = sc.build_clusters_gpu(gr)
    base = capi.LIGHTING_DIRECTIONAL_BIT | capi.LIGHTING_CLUSTERED_BIT | capi.LIGHTING_AMBIENT_FALLBACK_BIT
    out = []
    for flags in (base, base | capi.LIGHTING_SHARE_REGISTERS_BIT):
        args, imgs = sc.lighting_args(gr, dev, flags, alias_emissive=False)
        gr.check(gr.lib.gr_lighting(gr.handle, None, args))
        gr.sync()
        out.append(imgs["hdr"].download().copy())
    np.testing.assert_array_equal(out[0], out[1])
