"""GPU parity for the anti-aliasing passes (FXAA, SMAA 1x, TAA resolve) against the CPU oracle.

The kernels are compiled without FMA contraction and mirror the oracle's association order, so the byte-valued
decisions (SMAA edges and blend weights) are required to be bit-exact; final colours carry the stated RGBA8 +-1 LSB
(the kernels store gamma-space bytes directly instead of decode_srgb -> attachment encode)."""
import os

import numpy as np
import pytest

from granite_amd import app as gapp, capi, synth
from granite_amd.data import load_smaa_luts
from oracle import oracle as orc
from util import assert_rgba16f_close, assert_rgba8_close

pytestmark = pytest.mark.gpu

RGBA8 = capi.FORMAT_R8G8B8A8_SRGB
F16 = capi.FORMAT_R16G16B16A16_SFLOAT


def tonemapped_like(w, h):
    """Worst case for the decision logic: blocky random gamma-space bytes (every pixel is an edge candidate)."""
    r = np.random.default_rng(7)
    coarse = r.integers(0, 256, ((h + 2) // 3, (w + 2) // 3, 4), dtype=np.uint8)
    img = np.repeat(np.repeat(coarse, 3, axis=0), 3, axis=1)[:h, :w].copy()
    fine = r.integers(0, 256, (h, w, 4), dtype=np.uint8)
    mask = r.random((h, w)) < 0.3
    img[mask] = fine[mask]
    img[..., 3] = 255
    return img


@pytest.fixture(scope="module")
def luts():
    return load_smaa_luts()


@pytest.mark.parametrize("w,h,kind", [(320, 180, "pattern"), (253, 127, "pattern"), (256, 144, "noise"), (8, 8, "pattern")])
def test_fxaa(gr, w, h, kind):
    src = synth.make_ldr_pattern(w, h) if kind == "pattern" else tonemapped_like(w, h)
    din = capi.DeviceImage(gr, w, h, RGBA8).upload(src)
    dout = capi.DeviceImage(gr, w, h, RGBA8)
    gr.fxaa(din, dout)
    gr.sync()
    got = dout.download()
    assert_rgba8_close(got, orc.fxaa(src, True), 1, what="fxaa srgb target")
    # UNORM target: same op sequence, same rounding => bit-exact
    dun = capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8B8A8_UNORM)
    gr.fxaa(din, dun)
    gr.sync()
    np.testing.assert_array_equal(dun.download(), orc.fxaa(src, False))
    assert (got != src).any() or kind == "noise"


@pytest.mark.parametrize("quality", [0, 1, 2, 3])
@pytest.mark.parametrize("w,h,kind", [(320, 180, "pattern"), (253, 127, "pattern"), (200, 120, "noise")])
def test_smaa_passes(gr, luts, quality, w, h, kind):
    area, search = luts
    gr.smaa_set_luts(area, search)
    src = synth.make_ldr_pattern(w, h) if kind == "pattern" else tonemapped_like(w, h)
    ref = orc.smaa(src, area, search, quality, True)
    dsrc = capi.DeviceImage(gr, w, h, RGBA8).upload(src)
    dedge = capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8_UNORM)
    dwt = capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8B8A8_UNORM)
    dout = capi.DeviceImage(gr, w, h, RGBA8)
    gr.smaa_edge_detection(dsrc, dedge, quality)
    gr.smaa_blend_weight(dedge, dwt, quality)
    gr.smaa_neighbor_blend(dsrc, dwt, dout)
    gr.sync()
    np.testing.assert_array_equal(dedge.download(), ref["edges"])
    np.testing.assert_array_equal(dwt.download(), ref["weights"])
    assert_rgba8_close(dout.download(), ref["out"], 1, what=f"smaa q{quality} blend")
    assert ref["edges"].any() and ref["weights"].any()
    if kind == "pattern" and quality >= 2:
        # stage-wise on identical inputs as well (weights from the oracle's edges)
        dedge.upload(ref["edges"])
        gr.smaa_blend_weight(dedge, dwt, quality)
        gr.sync()
        np.testing.assert_array_equal(dwt.download(), ref["weights"])


def taa_inputs(w, h, seed=3):
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam, seed)
    cur = synth.make_hdr(w, h, seed)
    mv = synth.make_motion_vectors(w, h)
    # VP_prev = camera translated by 0.01 along x
    V2 = synth.look_at((0.01, 2.0, 8.0), (0.01, 1.0, 0.0))
    T = np.eye(4); T[0, 0] = T[1, 1] = 0.5; T[0, 3] = T[1, 3] = 0.5
    reproj = T @ (cam.P @ V2) @ cam.invVP
    return cur, gbuf["depth"], mv, np.ascontiguousarray(reproj.T, np.float32).reshape(16)


@pytest.mark.parametrize("quality", [0, 1, 2])
def test_taa_resolve(gr, quality):
    w, h = 240, 135
    cur, depth, mv, reproj = taa_inputs(w, h)
    dcur = capi.DeviceImage(gr, w, h, F16).upload(cur)
    ddepth = capi.DeviceImage(gr, w, h, capi.FORMAT_D32_SFLOAT).upload(depth)
    dmv = capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16_SFLOAT).upload(mv)
    dcol = capi.DeviceImage(gr, w, h, F16)
    dh = [capi.DeviceImage(gr, w, h, F16), capi.DeviceImage(gr, w, h, F16)]
    # frame 0: no history
    gr.taa_resolve(dcur, ddepth, dmv, None, dcol, dh[0], reproj, quality)
    gr.sync()
    ref_c, ref_h = orc.taa_resolve(cur, depth, mv, None, reproj, quality)
    assert_rgba16f_close(dcol.download(), ref_c, what=f"taa q{quality} f0 colour")
    assert_rgba16f_close(dh[0].download(), ref_h, what=f"taa q{quality} f0 history")
    # frames 1..2: feed the ORACLE's history to both so errors are not carried
    cur2 = synth.make_hdr(w, h, seed=11)
    dcur.upload(cur2)
    hist = ref_h
    for f in range(2):
        dh[f & 1].upload(hist)
        gr.taa_resolve(dcur, ddepth, dmv, dh[f & 1], dcol, dh[(f & 1) ^ 1], reproj, quality)
        gr.sync()
        ref_c, ref_h2 = orc.taa_resolve(cur2, depth, mv, hist, reproj, quality)
        assert_rgba16f_close(dcol.download(), ref_c, ulps=2.0, abs_tol=1e-4, what=f"taa q{quality} f{f + 1} colour")
        assert_rgba16f_close(dh[(f & 1) ^ 1].download(), ref_h2, ulps=2.0, abs_tol=1e-4, what=f"taa q{quality} f{f + 1} history")
        hist = ref_h2


def test_taa_static_scene_is_identity(gr):
    """Size-independent property: no motion + history == current (in TAA space) => output == current up to the
    HDR -> tonemapped-YCgCo -> HDR round trip."""
    w, h = 128, 72
    cam = synth.Camera(w, h)
    depth = synth.make_gbuffer(cam)["depth"]
    cur = synth.make_hdr(w, h)
    mv = np.zeros((h, w, 2), np.uint16)
    T = np.eye(4); T[0, 0] = T[1, 1] = 0.5; T[0, 3] = T[1, 3] = 0.5
    reproj = np.ascontiguousarray((T @ cam.VP @ cam.invVP).T, np.float32).reshape(16)
    dcur = capi.DeviceImage(gr, w, h, F16).upload(cur)
    ddepth = capi.DeviceImage(gr, w, h, capi.FORMAT_D32_SFLOAT).upload(depth)
    dmv = capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16_SFLOAT).upload(mv)
    c0, h0, c1, h1 = (capi.DeviceImage(gr, w, h, F16) for _ in range(4))
    gr.taa_resolve(dcur, ddepth, dmv, None, c0, h0, reproj, 1)
    gr.taa_resolve(dcur, ddepth, dmv, h0, c1, h1, reproj, 1)
    gr.sync()
    a = c0.download().view(np.float16).astype(np.float32)[..., :3]
    b = c1.download().view(np.float16).astype(np.float32)[..., :3]
    src = cur.view(np.float16).astype(np.float32)[..., :3]
    ok = src.max(axis=2) < 10.0  # the max3 tonemapper saturates at 0.999: very bright pixels are clipped by design
    np.testing.assert_allclose(a[ok], src[ok], rtol=2e-2, atol=2e-3)
    # history is stored as fp16 in the compressed TAA space; inverting 1/(1-max) amplifies its quantisation for bright
    # pixels, so the frame-to-frame identity is asserted where max < 2 (compressed max < 0.94)
    ok2 = src.max(axis=2) < 2.0
    np.testing.assert_allclose(b[ok2], a[ok2], rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("post_aa,pre_aa", [(gapp.POST_AA_FXAA, 0), (gapp.POST_AA_SMAA_HIGH, 0), (gapp.POST_AA_SMAA_ULTRA, gapp.POST_AA_TAA_HIGH),
                                            (0, gapp.POST_AA_TAA_LOW)])
def test_application_with_aa_matches_oracle_pipeline(luts, post_aa, pre_aa):
    """Config 4 style graph: [TAA] -> bloom/tonemap -> [FXAA | SMAA], frame by frame against the oracle driven in graph order."""
    area, search = luts
    w, h = 320, 180
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, 200)
    mv = synth.make_motion_vectors(w, h)
    a = gapp.Application(w, h, post_aa=post_aa, pre_aa=pre_aa)
    P, V = np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16)
    a.set_camera(P, V)
    a.set_lights(descs)
    a.upload_gbuffer(gbuf, mv)

    state, taa_hist = {}, None
    taa_q = {gapp.POST_AA_TAA_LOW: 0, gapp.POST_AA_TAA_MEDIUM: 1, gapp.POST_AA_TAA_HIGH: 2}.get(pre_aa)
    smaa_q = {gapp.POST_AA_SMAA_LOW: 0, gapp.POST_AA_SMAA_MEDIUM: 1, gapp.POST_AA_SMAA_HIGH: 2, gapp.POST_AA_SMAA_ULTRA: 3}.get(post_aa)
    for frame in range(3):
        a.render_frames(1)
        rp = a.get_render_parameters()  # jittered when TAA is on
        n, lights, model, tmask, _ = orc.pack_lights(descs, rp[99:102])
        prm = orc.cluster_params(rp, *synth.CLUSTER_RESOLUTION, n)
        cb = orc.cluster_build(rp, prm, lights, model, tmask, n, synth.CLUSTER_RESOLUTION[2])
        hdr = orc.lighting(gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
        assert_rgba16f_close(a.read("HDR-main"), hdr, ulps=2.0, what=f"frame {frame} HDR-main")
        if taa_q is not None:
            # feed the device's lit HDR + previous device history to the oracle: this checks the TAA pass, not carried error
            cur = a.read("HDR-main").copy()
            ref_c, ref_h = orc.taa_resolve(cur, gbuf["depth"], mv, taa_hist, a.taa_reprojection(), taa_q)
            assert_rgba16f_close(a.read("HDR-resolved"), ref_c, ulps=2.0, abs_tol=1e-4, what=f"frame {frame} HDR-resolved")
            got_h = a.read("HDR-resolved-history").copy()
            assert_rgba16f_close(got_h, ref_h, ulps=2.0, abs_tol=1e-4, what=f"frame {frame} history")
            taa_hist = got_h
            chain_in = a.read("HDR-resolved").copy()
        else:
            chain_in = a.read("HDR-main").copy()
        chain = orc.hdr_chain(chain_in, state)
        tm = a.read("tonemapped") if post_aa else a.read_backbuffer()
        assert_rgba8_close(tm, chain["tonemapped"], 1, what=f"frame {frame} tonemapped")
        if post_aa == gapp.POST_AA_FXAA:
            assert_rgba8_close(a.read_backbuffer(), orc.fxaa(np.ascontiguousarray(tm), True), 1, what=f"frame {frame} fxaa")
        elif smaa_q is not None:
            ref = orc.smaa(np.ascontiguousarray(tm), area, search, smaa_q, True)
            np.testing.assert_array_equal(a.read("smaa-edge"), ref["edges"])
            np.testing.assert_array_equal(a.read("smaa-weights"), ref["weights"])
            assert_rgba8_close(a.read_backbuffer(), ref["out"], 1, what=f"frame {frame} smaa")
    if taa_q is not None:
        # reprojection matrix against a float64 evaluation of temporal.cpp:239-243 (static camera => pure jitter-free VP)
        T = np.eye(4); T[0, 0] = T[1, 1] = 0.5; T[0, 3] = T[1, 3] = 0.5
        expect = T @ cam.VP @ cam.invVP
        np.testing.assert_allclose(a.taa_reprojection().reshape(4, 4).T, expect, atol=2e-5)
    a.close()


@pytest.mark.parametrize("src_fmt,dst_fmt,linear,sw,sh,dw,dh", [
    ("rgba8_srgb", "rgba16f", True, 200, 120, 333, 77), ("rgba16f", "rgba8_srgb", False, 333, 77, 333, 77),
    ("rgba16f", "rgba8_srgb", False, 167, 39, 333, 77), ("rgba8_unorm", "rgba8_unorm", True, 64, 64, 200, 120),
    ("rgba16f", "rgba16f", True, 960, 540, 1920, 1080)])
def test_blit_matches_the_oracle(gr, src_fmt, dst_fmt, linear, sw, sh, dw, dh):
    """blit.frag (the copy between targets of different size / format in tools/aa_bench.cpp): decode by input format,
    LinearClamp / NearestClamp at the pixel centre, store by output format.  Same arithmetic as the oracle: exact."""
    fmt = {"rgba16f": capi.FORMAT_R16G16B16A16_SFLOAT, "rgba8_unorm": capi.FORMAT_R8G8B8A8_UNORM, "rgba8_srgb": capi.FORMAT_R8G8B8A8_SRGB}
    src = synth.make_hdr(sw, sh) if src_fmt == "rgba16f" else synth.make_ldr_pattern(sw, sh)
    want = orc.blit(src, src_fmt, dw, dh, dst_fmt, linear)
    out = capi.DeviceImage(gr, dw, dh, fmt[dst_fmt])
    gr.blit(capi.DeviceImage(gr, sw, sh, fmt[src_fmt]).upload(src), out, linear)
    gr.sync()
    np.testing.assert_array_equal(out.download(), want)
