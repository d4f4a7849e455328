"""CPU: the oracle against the REFERENCE's OWN SHADER SOURCES, executed.

oracle/ref_build/ re-spells the GLSL of /root/reference/assets/shaders (glsl2cpp.py: strips layout()/precision/interface-block
syntax, suffixes float literals -- statements, expressions and constants untouched) into a scratch directory and compiles it
as C++ against a small GLSL environment (glsl_cpu.hpp: vectors with swizzles, built-ins, texture / image types).  The shader
text that runs here is the reference's; the environment (bilinear filtering, format conversion of stores, fp32 without
contraction) is stated once and shared with the oracle.  Equality below is therefore a statement about the oracle's reading
of every expression, association order and constant of these shaders -- it is required BIT FOR BIT."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from granite_amd import synth
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_shaders.so")
P = C.c_void_p


@pytest.fixture(scope="module")
def ref():
    if os.path.isdir("/root/reference/assets/shaders"):
        try:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_build")])
        except (subprocess.CalledProcessError, OSError) as e:  # keep going with a prebuilt library if there is one
            print("oracle/ref_build did not build:", e)
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libref_shaders.so not built (needs /root/reference)")
    lib = C.CDLL(REF_LIB)
    lib.ref_bloom_threshold.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int, P]
    lib.ref_bloom_downsample.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int, P, C.c_float]
    lib.ref_bloom_upsample.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int]
    lib.ref_luminance.argtypes = [P, C.c_int, C.c_int, P, C.c_float, C.c_float, C.c_float]
    lib.ref_tonemap.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int, P, C.c_float, P]
    return lib


def ptr(a):
    return None if a is None else a.ctypes.data


SIZES = [(256, 256), (200, 120), (333, 77), (65, 129), (16, 16)]


@pytest.mark.parametrize("w,h", SIZES)
def test_post_chain_shaders_bit_for_bit(ref, w, h):
    """bloom_threshold.comp (both DYNAMIC_EXPOSURE variants), bloom_downsample.comp (plain + FEEDBACK), bloom_upsample.comp,
    luminance.comp (one 8x8 workgroup of 64 real threads with barriers: the shader's own reduction tree) and tonemap.frag,
    chained over the pyramid exactly as hdr.cpp records them, two frames so that the feedback paths carry state."""
    hdr = synth.make_hdr(w, h)
    lum_lerp, fb_lerp = orc.frame_lerps(0.01)
    sz = [orc.level_size(w, h, s) for s in (0.5, 0.25, 0.125, 0.0625, 0.03125)]
    lum = np.array([0.2, 1.1, 0.9], np.float32)
    history = None
    for frame in range(2):
        for l in (None, lum):
            want = orc.bloom_threshold(hdr, *sz[0], lum3=l)
            got = np.zeros_like(want)
            ref.ref_bloom_threshold(ptr(hdr), w, h, ptr(got), sz[0][0], sz[0][1], ptr(l))
            np.testing.assert_array_equal(got, want, err_msg="bloom_threshold")
        levels = [want]
        for i in range(1, 5):
            hist = history if i == 4 else None
            want = orc.bloom_downsample(levels[-1], *sz[i], history=hist, lerp=fb_lerp if hist is not None else 0.0)
            got = np.zeros_like(want)
            src = levels[-1]
            ref.ref_bloom_downsample(ptr(src), src.shape[1], src.shape[0], ptr(got), sz[i][0], sz[i][1], ptr(hist), fb_lerp)
            np.testing.assert_array_equal(got, want, err_msg=f"bloom_downsample level {i} frame {frame}")
            levels.append(want)
        history = levels[-1]
        d3 = levels[-1]
        if d3.shape[0] >= 2 and d3.shape[1] >= 2:
            want_lum = orc.luminance(d3, lum.copy(), lum_lerp)
            got_lum = lum.copy()
            ref.ref_luminance(ptr(d3), d3.shape[1], d3.shape[0], ptr(got_lum), lum_lerp, -3.0, 2.0)
            np.testing.assert_array_equal(got_lum.view(np.uint32), want_lum.view(np.uint32), err_msg="luminance")
            lum = want_lum
        up = d3
        for i in (3, 2, 1):
            want = orc.bloom_upsample(up, *sz[i])
            got = np.zeros_like(want)
            ref.ref_bloom_upsample(ptr(up), up.shape[1], up.shape[0], ptr(got), sz[i][0], sz[i][1])
            np.testing.assert_array_equal(got, want, err_msg=f"bloom_upsample to level {i}")
            up = want
        for l, exposure in ((None, 1.0), (lum, 1.0), (lum, 1.7)):
            want = orc.tonemap(hdr, up, l, exposure)
            got = np.zeros((h, w, 4), np.uint8)
            ref.ref_tonemap(ptr(hdr), w, h, ptr(up), up.shape[1], up.shape[0], ptr(l), exposure, ptr(got))
            np.testing.assert_array_equal(got, want, err_msg="tonemap")


def test_whole_chain_equals_hdr_chain_helper(ref):
    """The same shaders driven in orc.hdr_chain's order reproduce its outputs: the recorded order is part of the contract."""
    w, h = 192, 108
    hdr = synth.make_hdr(w, h)
    state = {}
    out = None
    for _ in range(3):
        out = orc.hdr_chain(hdr, state)
    got = np.zeros((h, w, 4), np.uint8)
    u0 = out["u0"]
    ref.ref_tonemap(ptr(hdr), w, h, ptr(u0), u0.shape[1], u0.shape[0], ptr(out["lum"]), 1.0, ptr(got))
    np.testing.assert_array_equal(got, out["tonemapped"])


# ---- deferred lighting: lights/directional.frag + lights/clustering.frag and everything they include -----------------------
def lighting_scene(w, h, num_lights, seed_offset=0):
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    rp = cam.render_params()
    n, lights, model, tmask, _ = orc.pack_lights(synth.make_lights(cam, num_lights), rp[99:102])
    prm = orc.cluster_params(rp, *synth.CLUSTER_RESOLUTION, n)
    cb = orc.cluster_build(rp, prm, lights, model, tmask, n, synth.CLUSTER_RESOLUTION[2])
    return gbuf, rp, prm, lights, tmask, cb


@pytest.mark.parametrize("w,h,num_lights", [(160, 90, 300), (97, 61, 64), (64, 36, 4096), (48, 27, 0), (33, 19, 1)])
def test_lighting_shaders_bit_for_bit(ref, w, h, num_lights):
    """render_light's two quads with the shader variants of this path (renderer.cpp:1020-1056,1125-1147), one fragment per
    subgroup (the exact per-pixel light set = the oracle's wave_tile 0 form): directional with / without the fallback ambient
    term, the AMBIENT_OCCLUSION variant with a half-size AO texture, clustered point + spot lights, and both blended."""
    gbuf, rp, prm, lights, tmask, cb = lighting_scene(w, h, num_lights)
    ao = np.random.default_rng(w).integers(0, 256, (max(h // 2, 1), max(w // 2, 1)), dtype=np.uint8)

    def both(**kw):
        args = (gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
        return orc.lighting(*args, **kw), orc.lighting(*args, entry=ref.ref_lighting, **kw)

    for kw in (dict(clustered=False, ambient_fallback=False), dict(clustered=False), dict(directional=False), dict(),
               dict(ambient_occlusion=ao)):
        want, got = both(**kw)
        np.testing.assert_array_equal(got, want, err_msg=str({k: (v if k != "ambient_occlusion" else "ao") for k, v in kw.items()}))
    # sky pixels (depth 0) are untouched by both quads: depth test NOT_EQUAL
    sky = gbuf["depth"] == 0.0
    np.testing.assert_array_equal(got[sky], gbuf["emissive"][sky])
    if num_lights:
        plain, _ = both(clustered=False)
        assert (want != plain).any(), "clustered lights must contribute"


# ---- cluster build: lights/clusterer_bindless_{spot_transform,setup,binning,z_range}.comp -------------------------------------
@pytest.mark.parametrize("num_lights,res,forms", [(96, (16, 8, 64), (0, 64, 32)), (300, (32, 16, 256), (64, 32)), (33, (8, 8, 64), (0, 64)),
                                                  (1, (8, 8, 64), (0, 64))])
def test_cluster_build_shaders_bit_for_bit(ref, num_lights, res, forms):
    """The four compute shaders of LightClusterer's GPU cluster build with the push constants / buffers of clusterer.cpp:1277-1346,
    1463-1562.  Binning runs in all three forms the reference can take: plain (32 threads per cell and 32-light chunk, shared
    mask, barriers) and SUBGROUPS with 64- and 32-lane subgroups (ballot of the coarse 8 x N tile test, then per-cell tests);
    workgroups are teams of real threads.  Every buffer -- transformed spots, cull set-up (incl. what the shader leaves
    untouched), cell bitmasks, per-slice light ranges -- must equal the oracle's bit for bit."""
    for name, argtypes in (("ref_cluster_spot_transform", [P, P, C.c_int, P]), ("ref_cluster_setup", [P, P, P, P, P, C.c_int, P]),
                           ("ref_cluster_binning", [P, P, P, P, C.c_int]), ("ref_cluster_z_range", [P, C.c_int, C.c_int, P])):
        getattr(ref, name).argtypes = argtypes
    cam = synth.Camera(320, 180)
    rp = cam.render_params()
    n, lights, model, tmask, _ = orc.pack_lights(synth.make_lights(cam, num_lights), rp[99:102])
    prm = orc.cluster_params(rp, *res, n)
    for subgroup in forms:
        cb = orc.cluster_build(rp, prm, lights, model, tmask, n, res[2], subgroup_tile_h=subgroup // 8)
        spots, setup = np.zeros_like(cb["spots"]), np.zeros_like(cb["setup"])
        bitmask, ranges = np.zeros_like(cb["bitmask"]), np.zeros_like(cb["range"])
        ref.ref_cluster_spot_transform(ptr(rp), ptr(model), n, ptr(spots))
        ref.ref_cluster_setup(ptr(rp), ptr(prm), ptr(lights), ptr(tmask), ptr(spots), n, ptr(setup))
        ref.ref_cluster_binning(ptr(prm), ptr(tmask), ptr(setup), ptr(bitmask), subgroup)
        ref.ref_cluster_z_range(ptr(cb["light_ranges"]), n, res[2], ptr(ranges))
        np.testing.assert_array_equal(spots.view(np.uint32), cb["spots"].view(np.uint32), err_msg="transformed spots")
        np.testing.assert_array_equal(setup.view(np.uint32), cb["setup"].view(np.uint32), err_msg="cull set-up")
        np.testing.assert_array_equal(bitmask, cb["bitmask"], err_msg=f"cell bitmask, subgroup size {subgroup}")
        np.testing.assert_array_equal(ranges, cb["range"], err_msg="slice ranges")
        assert bitmask.any()
    # ... and the shader a subgroup-capable device really dispatches for the ranges (clusterer.cpp:1305-1314): the 128-thread
    # bit-matrix transpose, at every subgroup size the reference allows (set_subgroup_size_log2(true, 5, 7)).
    ref.ref_cluster_z_range_opt.argtypes = [P, C.c_int, C.c_int, P, C.c_int]
    for subgroup in (32, 64, 128):
        ranges = np.full_like(cb["range"], 0xdeadbeef)
        ref.ref_cluster_z_range_opt(ptr(cb["light_ranges"]), n, res[2], ptr(ranges), subgroup)
        np.testing.assert_array_equal(ranges, cb["range"], err_msg=f"slice ranges, z_range_opt at subgroup size {subgroup}")


def z_range_inputs(case, num_lights, num_ranges, seed):
    """Slice intervals per light as compute_uint_range produces them (clusterer.cpp:1265-1275), plus the degenerate inputs."""
    r = np.random.default_rng(seed)
    if case == "all_empty":  # tests/z_binning_test.cpp:56-96: 4096 inputs of (1000000000, 0) over 4096 ranges
        return np.tile(np.array([[1000000000, 0]], np.uint32), (num_lights, 1))
    lo = r.integers(0, num_ranges, num_lights)
    if case == "sorted":  # the clusterer sorts lights front to back: increasing first slice, short intervals
        lo = np.sort(lo)
        hi = np.minimum(lo + r.integers(0, max(num_ranges // 16, 2), num_lights), num_ranges - 1)
    elif case == "wide":  # long, overlapping intervals incl. whole-range and single-slice ones; the last slice is resolution_z - 1 at most
        # (compute_uint_range clamps it, clusterer.cpp:1273), the first may lie beyond it (a light behind the far plane: empty)
        hi = np.minimum(lo + r.integers(0, 2 * num_ranges, num_lights), num_ranges - 1)
        lo = np.where(r.random(num_lights) < 0.1, lo + num_ranges, lo)
        whole = r.random(num_lights) < 0.05
        lo, hi = np.where(whole, 0, lo), np.where(whole, num_ranges - 1, hi)
    else:  # "mixed": random order, one light in eight empty (x > y)
        hi = np.minimum(lo + r.integers(0, max(num_ranges // 4, 2), num_lights), num_ranges - 1)
        empty = r.random(num_lights) < 0.125
        hi = np.where(empty, 0, hi)
        lo = np.where(empty, 1000000000, lo)
    return np.stack([lo, hi], axis=1).astype(np.uint32)


@pytest.mark.parametrize("num_lights,num_ranges,case", [(4096, 4096, "all_empty"), (0, 256, "mixed"), (1, 64, "sorted"), (127, 128, "mixed"), (128, 128, "wide"),
                                                        (129, 300, "sorted"), (1000, 1000, "mixed"), (4096, 4096, "sorted"), (4096, 1024, "wide")])
def test_z_range_opt_shader_bit_for_bit(ref, num_lights, num_ranges, case):
    """clusterer_bindless_z_range_opt.comp (the shader the reference dispatches when subgroup shuffles exist, clusterer.cpp:1305-1314;
    assets/shaders/lights/clusterer_bindless_z_range_opt.comp:16-179) executed as 128-thread teams with subgroups of 32, 64 and 128
    lanes, against the oracle's range buffer and against the naive shader, bit for bit -- from no lights to the 4096-light maximum,
    including the all-empty input of the reference's own tests/z_binning_test.cpp."""
    ref.ref_cluster_z_range.argtypes = [P, C.c_int, C.c_int, P]
    ref.ref_cluster_z_range_opt.argtypes = [P, C.c_int, C.c_int, P, C.c_int]
    zr = z_range_inputs(case, num_lights, num_ranges, seed=num_lights * 7 + num_ranges)
    if num_lights == 0:
        zr = np.zeros((1, 2), np.uint32)  # a binding needs an address; never read
    want = np.zeros((num_ranges, 2), np.uint32)
    orc.lib().orc_cluster_z_range.argtypes = [P, C.c_int, C.c_int, P]
    orc.lib().orc_cluster_z_range(ptr(zr), num_lights, num_ranges, ptr(want))
    naive = np.zeros((num_ranges, 2), np.uint32)
    ref.ref_cluster_z_range(ptr(zr), num_lights, num_ranges, ptr(naive))
    np.testing.assert_array_equal(naive, want, err_msg="naive shader vs oracle")
    for subgroup in (32, 64, 128):
        got = np.full((num_ranges, 2), 0xdeadbeef, np.uint32)
        ref.ref_cluster_z_range_opt(ptr(zr), num_lights, num_ranges, ptr(got), subgroup)
        np.testing.assert_array_equal(got, want, err_msg=f"z_range_opt at subgroup size {subgroup}")
    if case == "all_empty":
        assert (want[:, 0] == 0xffffffff).all() and (want[:, 1] == 0).all()


# ---- anti-aliasing: post/fxaa.frag, post/taa_resolve.frag (+ reprojection.h, reprojection_color_space.h) -----------------------
def blocky(w, h, seed=7):
    r = np.random.default_rng(seed)
    coarse = r.integers(0, 256, ((h + 2) // 3, (w + 2) // 3, 4), dtype=np.uint8)
    img = np.repeat(np.repeat(coarse, 3, axis=0), 3, axis=1)[:h, :w].copy()
    fine = r.integers(0, 256, (h, w, 4), dtype=np.uint8)
    mask = r.random((h, w)) < 0.3
    img[mask] = fine[mask]
    img[..., 3] = 255
    return img


@pytest.mark.parametrize("w,h", [(120, 67), (64, 64), (9, 7)])
def test_fxaa_shader_bit_for_bit(ref, w, h):
    ref.ref_fxaa.argtypes = [P, C.c_int, C.c_int, P, C.c_int]
    img = blocky(w, h)
    for srgb in (False, True):
        want = orc.fxaa(img, srgb)
        got = np.zeros_like(want)
        ref.ref_fxaa(ptr(img), w, h, ptr(got), int(srgb))
        np.testing.assert_array_equal(got, want, err_msg=f"fxaa, FXAA_TARGET_SRGB={int(srgb)}")


@pytest.mark.parametrize("quality", [0, 1, 2])
def test_taa_resolve_shader_bit_for_bit(ref, quality):
    """TAA_QUALITY 0 / 1 / 2 (clamp vs AABB clip, 5-tap vs rounded-corner vs variance neighbourhood, bilinear vs Catmull-Rom
    history, 5-tap vs 3x3 nearest depth), the first frame without history and two frames with it (moving camera + explicit
    motion vectors)."""
    ref.ref_taa_resolve.argtypes = [P, P, P, P, C.c_int, C.c_int, P, C.c_int, P, P]
    w, h = 120, 68
    cam = synth.Camera(w, h)
    depth = np.ascontiguousarray(synth.make_gbuffer(cam, 3)["depth"], np.float32)
    mv = np.ascontiguousarray(synth.make_motion_vectors(w, h), np.uint16)
    view_prev = synth.look_at((0.01, 2.0, 8.0), (0.01, 1.0, 0.0))
    T = np.eye(4)
    T[0, 0] = T[1, 1] = 0.5
    T[0, 3] = T[1, 3] = 0.5
    reproj = np.ascontiguousarray((T @ (cam.P @ view_prev) @ cam.invVP).T, np.float32).reshape(16)
    history = None
    for frame in range(3):
        cur = synth.make_hdr(w, h, seed=3 + 8 * frame)
        want_c, want_h = orc.taa_resolve(cur, depth, mv, history, reproj, quality)
        got_c, got_h = np.zeros_like(want_c), np.zeros_like(want_h)
        ref.ref_taa_resolve(ptr(cur), ptr(depth), ptr(mv), ptr(history), w, h, ptr(reproj), quality, ptr(got_c), ptr(got_h))
        np.testing.assert_array_equal(got_c, want_c, err_msg=f"colour, frame {frame}")
        np.testing.assert_array_equal(got_h, want_h, err_msg=f"history, frame {frame}")
        history = want_h


# ---- SMAA 1x: post/smaa_{edge_detection,blend_weight,neighbor_blend}.{vert,frag} + post/SMAA.hlsl (GLSL 4 spelling) -----------
@pytest.mark.parametrize("quality", [0, 1, 2, 3])
@pytest.mark.parametrize("w,h,seed", [(160, 90, 7), (61, 47, 2)])
def test_smaa_passes_bit_for_bit(ref, quality, w, h, seed):
    """SMAA_PRESET_LOW .. ULTRA (threshold, search steps, diagonal and corner detection on from HIGH): edge detection with its
    discards, blend-weight calculation masked to edge pixels (area / search lookup tables, every texture LinearClamp) and the
    neighbourhood blend into UNORM and sRGB targets.  One translation unit per preset, because SMAA.hlsl keeps the preset as
    macro state."""
    from granite_amd.data import load_smaa_luts
    area, search = load_smaa_luts()
    img = blocky(w, h, seed)
    edges_fn, weights_fn = getattr(ref, f"ref_smaa_edges_q{quality}"), getattr(ref, f"ref_smaa_weights_q{quality}")
    edges_fn.argtypes = [P, C.c_int, C.c_int, P]
    weights_fn.argtypes = [P, C.c_int, C.c_int, P, P, P]
    ref.ref_smaa_blend.argtypes = [P, P, C.c_int, C.c_int, P, C.c_int]
    want_e = orc.smaa_edges(img, quality)
    got_e = np.full_like(want_e, 0xAA)
    edges_fn(ptr(img), w, h, ptr(got_e))
    np.testing.assert_array_equal(got_e, want_e, err_msg="edges")
    assert (want_e != 0).any()
    want_w = orc.smaa_weights(want_e, area, search, quality)
    got_w = np.full_like(want_w, 0xAA)
    weights_fn(ptr(want_e), w, h, ptr(area), ptr(search), ptr(got_w))
    np.testing.assert_array_equal(got_w, want_w, err_msg="blend weights")
    for srgb in (False, True):
        want = orc.smaa_blend(img, want_w, srgb)
        got = np.zeros_like(want)
        ref.ref_smaa_blend(ptr(img), ptr(want_w), w, h, ptr(got), int(srgb))
        np.testing.assert_array_equal(got, want, err_msg=f"neighbourhood blend, SMAA_TARGET_SRGB={int(srgb)}")


# ---- depth hierarchy: post/hiz.comp -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(128, 128), (200, 100), (257, 131), (520, 70), (1000, 300)])
@pytest.mark.parametrize("output_downsample", [False, True])
def test_depth_hierarchy_shader_bit_for_bit(ref, w, h, output_downsample):
    """hiz.comp with the bindings of HiZPassState::build_render_pass (spd.cpp:141-194): workgroups of 256 real threads (subgroups
    of 64, quad swaps, shared memory, barriers), run one after the other so the last one takes the ticket and reduces mips 7 and
    up with the odd-size folds.  Every level of the chain equals the oracle's."""
    ref.ref_hiz.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, C.c_int, P, C.c_int, C.c_int, C.c_int]
    depth = (np.random.default_rng(w * 31 + h).random((h, w), dtype=np.float32) * 0.9 + 0.05).astype(np.float32)
    zt = orc.hiz_z_transform(synth.Camera(w, h).render_params()[48:64])
    lay = orc.hiz_layout(w, h, output_downsample)
    want = orc.hiz(depth, zt, output_downsample)
    chain = np.zeros(sum(l.size for l in want), np.float32)
    ref.ref_hiz(ptr(depth), w, h, lay["res_w"], lay["res_h"], lay["mips"], ptr(zt), int(not output_downsample), ptr(chain),
                lay["chain_w"], lay["chain_h"], lay["levels"])
    offset = 0
    for level, lv in enumerate(want):
        got = chain[offset:offset + lv.size].reshape(lv.shape)
        offset += lv.size
        np.testing.assert_array_equal(got.view(np.uint32), lv.view(np.uint32), err_msg=f"level {level}")


# ---- spatial upscaling: post/ffx-fsr/{upscale,sharpen}.frag + the vendored FidelityFX headers ---------------------------------
def fsr_image(w, h, kind, seed=3):
    r = np.random.default_rng(seed)
    if kind == "noise":
        img = r.integers(0, 256, (h, w, 4), dtype=np.uint8)
    else:  # flat areas, hard edges, black and white
        coarse = r.choice(np.array([0, 0, 255, 17, 128, 200], np.uint8), ((h + 4) // 5, (w + 6) // 7, 4))
        img = np.repeat(np.repeat(coarse, 5, axis=0), 7, axis=1)[:h, :w].copy()
    img[..., 3] = 255
    return img


@pytest.mark.parametrize("case", [(160, 90, 240, 135, "noise"), (200, 120, 333, 201, "blocks"), (64, 64, 64, 64, "noise"), (7, 5, 30, 22, "noise"),
                                  (320, 180, 256, 144, "blocks")])
def test_fsr_easu_shader_bit_for_bit(ref, case):
    """upscale.frag with FP16 = 0 (FsrEasuF from ffx_fsr1.h, helpers from ffx_a.h, constants from the header's own FsrEasuCon)
    against the oracle's fp32 path, and FsrEasuH from the same headers on emulated GLSL float16 types (every operation rounds
    to half; gather callbacks as upscale.frag:8-14) against the oracle's half path -- the variant the reference selects on
    fp16-capable hardware."""
    ref.ref_fsr_easu.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int]
    ref.ref_fsr_easu_fp16.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int]
    w, h, ow, oh, kind = case
    src = fsr_image(w, h, kind)
    for fp16, entry in ((False, ref.ref_fsr_easu), (True, ref.ref_fsr_easu_fp16)):
        want = orc.fsr_easu(src, ow, oh, fp16=fp16)
        got = np.zeros_like(want)
        entry(ptr(src), w, h, ptr(got), ow, oh)
        np.testing.assert_array_equal(got, want, err_msg="FP16" if fp16 else "FP32")


@pytest.mark.parametrize("kind", ["noise", "blocks"])
@pytest.mark.parametrize("srgb", [False, True])
def test_fsr_rcas_shader_bit_for_bit(ref, kind, srgb):
    """sharpen.frag (FsrRcasF, FsrRcasCon(0.5)), reading through the sRGB or the UNORM view as aa.cpp:147-151 selects; black
    pixels exercise the 0 * inf paths, decided by minNum / maxNum on both sides."""
    ref.ref_fsr_rcas.argtypes = [P, C.c_int, C.c_int, P, C.c_float, C.c_int]
    src = fsr_image(253, 127, kind, seed=9)
    want = orc.fsr_rcas(src, orc.fsr_rcas_sharpness(0.5), srgb)
    got = np.zeros_like(want)
    ref.ref_fsr_rcas(ptr(src), 253, 127, ptr(got), 0.5, int(srgb))
    np.testing.assert_array_equal(got, want)


# ---- HDR10 output: post/pq10_encode.frag ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("max_light_level", [1000.0, 400.0])
def test_pq10_encode_shader_bit_for_bit(ref, max_light_level):
    ref.ref_pq10_encode.argtypes = [P, P, C.c_int, C.c_int, P, C.c_float, C.c_float, C.c_float, P]
    w, h = 96, 54
    hdr = synth.make_hdr(w, h, 5)
    ui = np.random.default_rng(1).integers(0, 256, (h, w, 4), dtype=np.uint8)
    ui[:20] = (0, 0, 0, 255)  # the cleared layer: scene fully visible
    conversion = orc.rec709_to_display()
    want = orc.pq10_encode(hdr, ui, conversion, 500.0, 400.0, max_light_level)
    got = np.zeros_like(want)
    ref.ref_pq10_encode(ptr(hdr), ptr(ui), w, h, ptr(conversion), 500.0, 400.0, max_light_level, ptr(got))
    np.testing.assert_array_equal(got, want)
    assert len(np.unique(want & 1023)) > 50


# ---- single-pass downsampler: post/ffx-spd/spd.comp + ffx_spd.h --------------------------------------------------------------------
@pytest.mark.parametrize("iw,ih,w0,h0,mips,components,depth,mods", [
    (128, 128, 64, 64, 7, 4, False, False),
    (200, 120, 100, 60, 7, 4, False, False),     # edge workgroups, clamped taps, clamped level-5 reads
    (256, 128, 128, 64, 7, 3, False, True),      # COMPONENTS 3 + FILTER_MOD: the reference's ocean use
    (128, 128, 128, 128, 8, 1, True, False),     # REDUCTION_MODE depth
    (512, 512, 256, 256, 9, 4, False, False),    # 64 workgroups, levels 6, 7, 8 in the last one
    (64, 64, 32, 32, 3, 4, False, False),
])
def test_spd_shader_bit_for_bit(ref, iw, ih, w0, h0, mips, components, depth, mods):
    ref.ref_spd.restype = C.c_int
    src = synth.make_hdr(iw, ih, 11).copy()
    src.view(np.float16)[..., 3] = np.random.default_rng(3).uniform(0.0, 4.0, (ih, iw)).astype(np.float16)
    fm = None
    if mods:
        fm = np.ones((mips, 4), np.float32)
        fm[mips - 1] = (0.0, 1.0, 1.0, 1.0)
        fm[2] = (0.5, 2.0, 1.0, 1.0)
    want = orc.spd(src, w0, h0, mips, components, depth, fm, fill=0x3c00)
    got = orc.spd(src, w0, h0, mips, components, depth, fm, entry=ref.ref_spd, fill=0x3c00)
    for level, (a, b) in enumerate(zip(want, got)):
        np.testing.assert_array_equal(b, a, err_msg=f"level {level}")
    assert len({int(v) for v in want[0][..., 0].reshape(-1)[:4096]}) > 100


@pytest.mark.parametrize("w,h,frame", [(96, 64, 0), (75, 41, 5), (192, 108, 63)])
def test_sssr_shaders_bit_for_bit(ref, w, h, frame):
    """post/ffx-sssr: classify.comp + build_indirect.comp + trace_primary.comp (64 real threads per workgroup: quad swaps,
    the ordered append, subgroupBallot as a rendezvous of the lanes still traversing) and apply.frag, on the close-up trough
    scene of the GPU tests: ray list, counters, traced colour, ray length, confidence and the blended target, bit for bit with
    oracle_ssr.cpp.  What the shader leaves to the hardware (append order, copy conflicts) is fixed as ref_ssr.cpp's header
    says -- the same statement oracle_ssr.cpp makes."""
    from granite_amd.data import expand_sssr_dither, load_brdf_lut, load_sssr_noise_base
    from util import close_up_scene
    cam, depth, normal, pbr, albedo, light = close_up_scene(w, h, seed=2)
    rp = cam.render_params()
    levels = orc.hiz(depth, orc.hiz_z_transform(rp[48:64]))
    noise = expand_sssr_dither(load_sssr_noise_base())
    want = orc.ssr_trace(levels, pbr, normal, light, noise, frame, rp[32:48], rp[80:96], rp[96:99])
    ref.ref_ssr_classify.argtypes = ref.ref_ssr_trace.argtypes = [P]
    got = orc.ssr_trace(levels, pbr, normal, light, noise, frame, rp[32:48], rp[80:96], rp[96:99],
                        entry=(ref.ref_ssr_classify, ref.ref_ssr_trace))
    np.testing.assert_array_equal(got["ray_counter"], want["ray_counter"])
    assert len(want["ray_list"]) > 0.1 * w * h
    np.testing.assert_array_equal(got["ray_list"], want["ray_list"])
    assert (want["confidence"] > 0).sum() > 10, "some rays must hit with confidence"
    copies = (want["ray_list"] >> 28) & 7
    assert (copies & 1).any() and (copies & 2).any(), "the scene must exercise horizontal and vertical copies"
    np.testing.assert_array_equal(got["confidence"], want["confidence"])
    np.testing.assert_array_equal(got["ray_length"], want["ray_length"])
    np.testing.assert_array_equal(got["output"], want["output"])

    lut = load_brdf_lut()
    real_depth = depth.copy()
    real_depth[::7, ::5] = 1.0
    want_hdr = orc.ssr_apply(light, want["output"], albedo, normal, pbr, real_depth, lut, rp[80:96], rp[96:99])
    got_hdr = np.array(light, np.uint16, copy=True)
    ivp, cam_pos = np.ascontiguousarray(rp[80:96], np.float32), np.ascontiguousarray(rp[96:99], np.float32)
    ref.ref_ssr_apply.argtypes = [C.c_int, C.c_int, P, P, P, P, P, P, C.c_int, C.c_int, P, P, P]
    keep = [np.ascontiguousarray(want["output"]), np.ascontiguousarray(albedo, np.uint32), np.ascontiguousarray(normal, np.uint32),
            np.ascontiguousarray(pbr, np.uint16), np.ascontiguousarray(real_depth, np.float32), np.ascontiguousarray(lut, np.uint16)]
    ref.ref_ssr_apply(w, h, *[ptr(k) for k in keep], lut.shape[1], lut.shape[0], ptr(ivp), ptr(cam_pos), ptr(got_hdr))
    np.testing.assert_array_equal(got_hdr, want_hdr)
    assert (got_hdr != light).any(axis=2).mean() > 0.2


@pytest.mark.parametrize("src_fmt,dst_fmt,linear,sw,sh,dw,dh", [
    ("rgba8_srgb", "rgba16f", True, 100, 60, 167, 39), ("rgba16f", "rgba8_srgb", False, 167, 39, 167, 39),
    ("rgba16f", "rgba8_srgb", False, 83, 19, 167, 39), ("rgba8_unorm", "rgba8_unorm", True, 64, 64, 100, 60),
    ("rgba16f", "rgba16f", True, 96, 54, 192, 108)])
def test_blit_shader_bit_for_bit(ref, src_fmt, dst_fmt, linear, sw, sh, dw, dh):
    """assets/shaders/blit.frag (the full-screen copy of tools/aa_bench.cpp) executed == orc.blit, for every format pair and both
    samplers the AA benchmark graph uses."""
    src = synth.make_hdr(sw, sh) if src_fmt == "rgba16f" else synth.make_ldr_pattern(sw, sh)
    want = orc.blit(src, src_fmt, dw, dh, dst_fmt, linear)
    got = np.zeros_like(want)
    src = np.ascontiguousarray(src)
    ref.ref_blit.argtypes = [P, C.c_int, C.c_int, C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_int]
    ref.ref_blit(ptr(src), sw, sh, orc.BLIT_FORMATS[src_fmt], ptr(got), dw, dh, orc.BLIT_FORMATS[dst_fmt], int(linear))
    np.testing.assert_array_equal(got, want)


# ---- B10G11R11_UFLOAT_PACK32 HDR targets: the reference's default (renderTargetFp16 = false) ------------------------------------------

def test_packed_float_codec_rounds_to_the_closest_finite_value():
    """float -> unsigned 11 / 10-bit float as the Vulkan / OpenGL packed-float rules state it (oracle_common.h: float_to_ufloat):
    closest representable finite value (brute force over the format's value set in float64), negatives -> 0, +inf -> +inf,
    NaN -> NaN, and every packed value survives the trip through RGBA16F exactly."""
    def values(mant_bits):
        v = [0.0]
        for e in range(31):
            for m in range(1 << mant_bits):
                v.append(m * 2.0 ** (-14 - mant_bits) if e == 0 else (1 + m / (1 << mant_bits)) * 2.0 ** (e - 15))
        return np.unique(np.array(v))
    r = np.random.default_rng(0)
    x = np.concatenate([np.exp2(r.uniform(-26, 17, 100000)), r.uniform(0, 2, 20000), [0.0, 65024.0, 65279.9, 65280.0, 1e9, 64512.0]]).astype(np.float32)
    rgb = np.repeat(x[:, None], 3, axis=1).copy()
    words = np.zeros(x.size, np.uint32)
    orc.lib().orc_pack_b10g11r11_from_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    orc.lib().orc_pack_b10g11r11_from_f32(ptr(rgb), ptr(words), x.size)
    back = orc.half_to_float(orc.unpack_b10g11r11(words.reshape(-1, 1))).reshape(-1, 4).astype(np.float64)
    for ch, mant_bits in ((0, 6), (1, 6), (2, 5)):
        vs = values(mant_bits)
        xv = np.minimum(x.astype(np.float64), vs[-1])
        idx = np.clip(np.searchsorted(vs, xv), 1, len(vs) - 1)
        lo, hi = vs[idx - 1], vs[idx]
        tie = np.abs(xv - lo) == np.abs(hi - xv)
        best = np.where(np.abs(xv - lo) <= np.abs(hi - xv), lo, hi)
        assert ((back[:, ch] == best) | tie).all()
        assert ((back[tie, ch] == lo[tie]) | (back[tie, ch] == hi[tie])).all()
    assert (back[:, 3] == 1.0).all()
    special = orc.half_to_float(orc.quantize_b10g11r11(np.array([[[0xbc00, 0x7c00, 0x7e00, 0], [0xfc00, 0x8000, 0x0001, 0]]], np.uint16)))
    assert special[0, 0, 0] == 0.0 and np.isinf(special[0, 0, 1]) and np.isnan(special[0, 0, 2])
    assert special[0, 1, 0] == 0.0 and special[0, 1, 1] == 0.0
    # all 2048 x 1024 packed patterns are their own fixed points (the format -> RGBA16F -> format)
    every = (np.arange(2048, dtype=np.uint32)[:, None] | (np.arange(2048, dtype=np.uint32)[:, None] << 11) | (np.arange(1024, dtype=np.uint32)[None, :] << 22))
    finite = ((every & 0x7ff) < 0x7c0) & ((every >> 22) < 0x3e0)
    again = orc.pack_b10g11r11(orc.unpack_b10g11r11(every))
    assert np.array_equal(again[finite], every[finite])


def test_lighting_into_a_b10g11r11_target_reference_shaders(ref):
    """directional.frag + clustering.frag blended ONE / ONE into a B10G11R11_UFLOAT_PACK32 attachment: the executed shaders with
    the environment's packed store == the oracle, bit for bit; and the packed target really differs from the RGBA16F one."""
    from granite_amd import synth
    w, h, n = 64, 36, 96
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam, seed=5)
    gbuf["emissive"] = orc.quantize_b10g11r11(gbuf["emissive"])
    rp = cam.render_params()
    descs = synth.make_lights(cam, n, seed=5)
    count, lights, model, tmask, _ = orc.pack_lights(descs, rp[99:102])
    prm = orc.cluster_params(rp, *synth.CLUSTER_RESOLUTION, count)
    cb = orc.cluster_build(rp, prm, lights, model, tmask, count, synth.CLUSTER_RESOLUTION[2])
    args = (gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
    want = orc.lighting(*args, b10g11r11=True)
    got = orc.lighting(*args, b10g11r11=True, entry=ref.ref_lighting)
    np.testing.assert_array_equal(got, want)
    assert np.array_equal(orc.quantize_b10g11r11(want), want)       # every texel is a packed value
    assert (want != orc.lighting(*args)).any()                       # and not what the fp16 target holds


@pytest.mark.parametrize("quality", [0, 2])
def test_taa_resolve_into_a_b10g11r11_colour_target_reference_shader(ref, quality):
    """taa_resolve.frag with its colour output in B10G11R11_UFLOAT_PACK32 (temporal.cpp:211-213), history RGBA16F."""
    P = C.c_void_p
    ref.ref_taa_resolve_fmt.argtypes = [P, P, P, P, C.c_int, C.c_int, P, C.c_int, P, P, C.c_int]
    from granite_amd import synth
    w, h = 48, 27
    cam = synth.Camera(w, h)
    depth = synth.make_gbuffer(cam, 3)["depth"]
    cur = orc.quantize_b10g11r11(synth.make_hdr(w, h, 3))
    mv = synth.make_motion_vectors(w, h)
    reproj = np.eye(4, dtype=np.float32)
    reproj[0, 0] = reproj[1, 1] = reproj[0, 3] = reproj[1, 3] = 0.5
    reproj = np.ascontiguousarray(reproj.T).reshape(-1)
    history = None
    for frame in range(2):
        want_c, want_h = orc.taa_resolve(cur, depth, mv, history, reproj, quality, color_b10g11r11=True)
        got_c, got_h = np.zeros_like(want_c), np.zeros_like(want_h)
        ref.ref_taa_resolve_fmt(ptr(cur), ptr(depth), ptr(mv), ptr(history) if history is not None else None, w, h, ptr(reproj), quality,
                                ptr(got_c), ptr(got_h), 1)
        np.testing.assert_array_equal(got_c, want_c)
        np.testing.assert_array_equal(got_h, want_h)
        assert np.array_equal(orc.quantize_b10g11r11(want_c), want_c)
        history = want_h


@pytest.mark.parametrize("packed", [False, True])
def test_fog_quad_behind_the_lighting_quads_reference_shader(ref, packed):
    """render_light's last quad (renderer.cpp:1179-1196): lights/fog.{vert,frag} + fog.h executed and blended ONE_MINUS_SRC_ALPHA /
    SRC_ALPHA (colour and alpha) == the oracle, bit for bit, into an RGBA16F and into a B10G11R11 target; sky pixels untouched."""
    w, h, n = 64, 36, 96
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam, seed=5)
    if packed:
        gbuf["emissive"] = orc.quantize_b10g11r11(gbuf["emissive"])
    rp = cam.render_params()
    count, lights, model, tmask, _ = orc.pack_lights(synth.make_lights(cam, n, seed=5), rp[99:102])
    prm = orc.cluster_params(rp, *synth.CLUSTER_RESOLUTION, count)
    cb = orc.cluster_build(rp, prm, lights, model, tmask, count, synth.CLUSTER_RESOLUTION[2])
    args = (gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
    fog = ((0.35, 0.4, 0.55), 0.012)
    want = orc.lighting(*args, b10g11r11=packed, fog=fog)
    got = orc.lighting(*args, b10g11r11=packed, fog=fog, entry=ref.ref_lighting)
    np.testing.assert_array_equal(got, want)
    plain = orc.lighting(*args, b10g11r11=packed)
    lit = gbuf["depth"] != 0.0
    assert (want[lit] != plain[lit]).any(axis=-1).mean() > 0.9
    np.testing.assert_array_equal(want[~lit], plain[~lit])
    # known answer: f = exp2(-|pos - eye|^2 falloff) pulls a lit pixel towards the fog colour
    f32 = orc.half_to_float
    towards = np.abs(f32(want)[lit][:, :3] - np.array(fog[0], np.float32)) <= np.abs(f32(plain)[lit][:, :3] - np.array(fog[0], np.float32)) + 2e-3
    assert towards.mean() > 0.99
