"""GPU parity: HDR post-chain kernels through the C ABI vs the CPU oracle on identical seeded inputs.

Tolerances: RGBA16F levels |a-b| <= 2 ulp_fp16 + 1e-4 per channel; RGBA8 outputs +-1 LSB;
LuminanceData 1e-5 absolute on the log value (the HIP reduction uses a different, fixed, association order)."""
import numpy as np
import pytest

from granite_amd import capi, synth
from oracle import oracle as orc
from util import assert_rgba16f_close, assert_rgba8_close

pytestmark = pytest.mark.gpu

F16 = capi.FORMAT_R16G16B16A16_SFLOAT

SIZES = [(256, 256), (250, 130), (70, 40), (64, 64), (1920, 1080)]
# d3/2 has a zero dimension below 64 px: the reference's luminance pass divides by zero there (luminance.comp:31), so the
# tiny sizes run with dynamic exposure off and the C ABI rejects a zero-sized luminance grid.
TINY_SIZES = [(33, 17), (1, 1), (2, 3)]


def run_chain_gpu(gr, hdr_bits, state, frame_time=0.01, use_lum=True, fmt=capi.FORMAT_R8G8B8A8_SRGB):
    h, w = hdr_bits.shape[:2]
    lum_lerp, fb_lerp = orc.frame_lerps(frame_time)
    sz = [orc.level_size(w, h, s) for s in (0.5, 0.25, 0.125, 0.0625, 0.03125)]
    hdr = capi.DeviceImage(gr, w, h, F16).upload(hdr_bits)
    t = capi.DeviceImage(gr, *sz[0], F16)
    d0 = capi.DeviceImage(gr, *sz[1], F16)
    d1 = capi.DeviceImage(gr, *sz[2], F16)
    d2 = capi.DeviceImage(gr, *sz[3], F16)
    u2 = capi.DeviceImage(gr, *sz[3], F16)
    u1 = capi.DeviceImage(gr, *sz[2], F16)
    u0 = capi.DeviceImage(gr, *sz[1], F16)
    out = capi.DeviceImage(gr, w, h, fmt)
    if "lum_buf" not in state:
        state["lum_buf"] = capi.DeviceBuffer(gr, 12)
        state["d3"] = [capi.DeviceImage(gr, *sz[4], F16), capi.DeviceImage(gr, *sz[4], F16)]
        state["frame"] = 0
    lum = state["lum_buf"].ptr if use_lum else None
    # history swap (render_graph.cpp:2704-2708): current <-> history each frame; no history on frame 0
    cur = state["d3"][state["frame"] & 1]
    hist = state["d3"][(state["frame"] & 1) ^ 1] if state["frame"] > 0 else None
    gr.bloom_threshold(hdr, t, lum)
    gr.bloom_downsample(t, d0)
    gr.bloom_downsample(d0, d1)
    gr.bloom_downsample(d1, d2)
    gr.bloom_downsample(d2, cur, hist, fb_lerp)
    if use_lum:
        gr.luminance(cur, lum, lum_lerp)
    gr.bloom_upsample(cur, u2)
    gr.bloom_upsample(u2, u1)
    gr.bloom_upsample(u1, u0)
    gr.tonemap(hdr, u0, out, lum)
    gr.sync()
    state["frame"] += 1
    res = {"threshold": t.download(), "d0": d0.download(), "d1": d1.download(), "d2": d2.download(), "d3": cur.download(),
           "u2": u2.download(), "u1": u1.download(), "u0": u0.download(), "tonemapped": out.download()}
    if use_lum:
        res["lum"] = state["lum_buf"].download(np.float32)
    return res


@pytest.mark.parametrize("w,h", SIZES)
def test_chain_levels_match_oracle(gr, w, h):
    hdr = synth.make_hdr(w, h)
    ostate, gstate = {}, {}
    for frame in range(3):  # frame 0: no history, lum = 0; later frames exercise feedback + exposure
        ref = orc.hdr_chain(hdr, ostate)
        got = run_chain_gpu(gr, hdr, gstate)
        for name in ("threshold", "d0", "d1", "d2", "d3", "u2", "u1", "u0"):
            assert ref[name].shape == got[name].shape, name
            assert_rgba16f_close(got[name], ref[name], what=f"{w}x{h} frame {frame} {name}")
        np.testing.assert_allclose(got["lum"][0], ref["lum"][0], atol=1e-5, rtol=0)
        np.testing.assert_allclose(got["lum"][1:], ref["lum"][1:], rtol=2e-5)
        assert_rgba8_close(got["tonemapped"], ref["tonemapped"], 1, what=f"{w}x{h} frame {frame} tonemapped")


@pytest.mark.parametrize("w,h", TINY_SIZES)
def test_chain_tiny_sizes_without_exposure(gr, w, h):
    hdr = synth.make_hdr(w, h)
    ostate, gstate = {}, {}
    for frame in range(2):
        ref = orc.hdr_chain(hdr, ostate, use_lum=False)
        got = run_chain_gpu(gr, hdr, gstate, use_lum=False)
        for name in ("threshold", "d0", "d1", "d2", "d3", "u2", "u1", "u0"):
            assert_rgba16f_close(got[name], ref[name], what=f"{w}x{h} frame {frame} {name}")
        assert_rgba8_close(got["tonemapped"], ref["tonemapped"], 1, what=f"{w}x{h} tonemapped")
    d3 = capi.DeviceImage(gr, 1, 1, F16)
    with pytest.raises(capi.GraniteHipError):
        gr.luminance(d3, capi.DeviceBuffer(gr, 12).ptr, 0.5)


def test_kernels_stagewise_on_identical_inputs(gr):
    """Each kernel against the oracle on the SAME input bits (no error carried between stages)."""
    w, h = 250, 130
    hdr = synth.make_hdr(w, h)
    lum3 = np.array([0.5, 2.0 ** 0.5, 2.0 ** -0.5], np.float32)
    lumbuf = capi.DeviceBuffer(gr, 12).upload(lum3)
    tw, th = orc.level_size(w, h, 0.5)
    ref_t = orc.bloom_threshold(hdr, tw, th, lum3)
    dhdr = capi.DeviceImage(gr, w, h, F16).upload(hdr)
    dt = capi.DeviceImage(gr, tw, th, F16)
    gr.bloom_threshold(dhdr, dt, lumbuf.ptr)
    assert_rgba16f_close(dt.download(), ref_t, what="threshold(dynamic)")
    gr.bloom_threshold(dhdr, dt, None)
    assert_rgba16f_close(dt.download(), orc.bloom_threshold(hdr, tw, th, None), what="threshold(static)")

    dw, dh = orc.level_size(w, h, 0.25)
    dt.upload(ref_t)
    dd = capi.DeviceImage(gr, dw, dh, F16)
    gr.bloom_downsample(dt, dd)
    ref_d = orc.bloom_downsample(ref_t, dw, dh)
    assert_rgba16f_close(dd.download(), ref_d, what="downsample")

    hist_bits = synth.make_hdr(dw, dh, seed=99)
    dhist = capi.DeviceImage(gr, dw, dh, F16).upload(hist_bits)
    gr.bloom_downsample(dt, dd, dhist, 0.0667)
    assert_rgba16f_close(dd.download(), orc.bloom_downsample(ref_t, dw, dh, hist_bits, 0.0667), what="downsample(feedback)")

    du = capi.DeviceImage(gr, tw, th, F16)
    dd.upload(ref_d)
    gr.bloom_upsample(dd, du)
    assert_rgba16f_close(du.download(), orc.bloom_upsample(ref_d, tw, th), what="upsample")

    for fmt, check in ((capi.FORMAT_R8G8B8A8_SRGB, True),):
        dout = capi.DeviceImage(gr, w, h, fmt)
        gr.tonemap(dhdr, dd, dout, lumbuf.ptr, 1.3)
        assert_rgba8_close(dout.download(), orc.tonemap(hdr, ref_d, lum3, 1.3), 1, what="tonemap(dynamic)")
        gr.tonemap(dhdr, dd, dout, None, 0.7)
        assert_rgba8_close(dout.download(), orc.tonemap(hdr, ref_d, None, 0.7), 1, what="tonemap(static)")


def test_constant_image_known_answers(gr):
    """Analytic: constant colour c with avg_lum = 0 => threshold = c/(max+1e-4)*(max+1e-4) ~ c, tent weights sum to 1 so
    every level equals the threshold value; alpha = log2(max + 1e-4)."""
    w, h = 128, 64
    c = np.array([0.5, 1.5, 0.25, 1.0], np.float32)
    hdr = np.broadcast_to(c.astype(np.float16).view(np.uint16), (h, w, 4)).copy()
    got = run_chain_gpu(gr, hdr, {}, use_lum=True)
    expect_alpha = np.log2(1.5 + 1e-4)
    for name in ("threshold", "d0", "d1", "d2", "d3", "u2", "u1", "u0"):
        v = got[name].view(np.float16).astype(np.float32)
        np.testing.assert_allclose(v[..., :3], np.broadcast_to(c[:3], v[..., :3].shape), rtol=2e-3, err_msg=name)
        np.testing.assert_allclose(v[..., 3], expect_alpha, rtol=2e-3, err_msg=name)
    # luminance: mean log-lum clamp [-3,2], lerp from 0 on frame 0
    lum_lerp, _ = orc.frame_lerps(0.01)
    np.testing.assert_allclose(got["lum"][0], lum_lerp * expect_alpha, rtol=2e-3)


def test_argument_validation(gr):
    hdr = capi.DeviceImage(gr, 16, 16, F16)
    bad = capi.DeviceImage(gr, 16, 16, capi.FORMAT_R8G8B8A8_UNORM)
    with pytest.raises(capi.GraniteHipError):
        gr.bloom_threshold(bad, hdr)
    with pytest.raises(capi.GraniteHipError):
        gr.tonemap(hdr, hdr, capi.DeviceImage(gr, 8, 8, capi.FORMAT_R8G8B8A8_SRGB))


@pytest.mark.parametrize("w,h", [(3840, 2160), (7680, 4320), (2048, 2048), (2560, 1440), (1280, 720), (256, 256), (1920, 1080), (1000, 808), (333, 250)])
def test_fused_pyramid_tail_equals_the_five_separate_launches(gr, w, h):
    """gr_bloom_down_tail + gr_bloom_up_tail (downsample-2 -> downsample-3 + feedback -> luminance -> upsample-2 -> upsample-1
    through LDS, two launches) must leave the very bytes of the five separate launches in every level and in the luminance
    buffer: 4K and 1440p (downsample-3 and upsample-2 on the generic tent: 135 -> 68, 90 -> 45 exact but 45 -> 23), sizes where
    both are exact 2:1 / 1:2, small ones with partial tiles; 1080p (135 -> 68: downsample-2 and upsample-1 on the nine generic taps
    as well) and two odd-sized pyramids where no step is exact."""
    sz = [orc.level_size(w, h, s) for s in (0.125, 0.0625, 0.03125)]
    rng = np.random.default_rng(w * 31 + h)
    d1_bits = np.exp2(rng.uniform(-8, 6, (sz[0][1], sz[0][0], 4))).astype(np.float16).view(np.uint16)
    hist_bits = np.exp2(rng.uniform(-8, 4, (sz[2][1], sz[2][0], 4))).astype(np.float16).view(np.uint16)
    d1 = capi.DeviceImage(gr, *sz[0], F16).upload(d1_bits)
    hist = capi.DeviceImage(gr, *sz[2], F16).upload(hist_bits)
    lum0 = np.array([0.25, 2.0 ** 0.25, 2.0 ** -0.25], np.float32)
    lum_lerp, fb_lerp = orc.frame_lerps(0.01)
    got = {}
    for fused in (False, True):
        d2, u2 = capi.DeviceImage(gr, *sz[1], F16), capi.DeviceImage(gr, *sz[1], F16)
        d3, u1 = capi.DeviceImage(gr, *sz[2], F16), capi.DeviceImage(gr, *sz[0], F16)
        lum = capi.DeviceBuffer(gr, 12).upload(lum0)
        if fused:
            assert gr.bloom_tail(d1, d2, d3, hist, u2, u1, fb_lerp, lum.ptr, lum_lerp), "this pyramid must qualify for the fused tail"
        else:
            gr.bloom_downsample(d1, d2)
            gr.bloom_downsample(d2, d3, hist, fb_lerp)
            gr.luminance(d3, lum.ptr, lum_lerp)
            gr.bloom_upsample(d3, u2)
            gr.bloom_upsample(u2, u1)
        gr.sync()
        got[fused] = (d2.download(), d3.download(), u2.download(), u1.download(), lum.download(np.float32))
    for a, b, name in zip(got[True], got[False], ("downsample-2", "downsample-3", "upsample-2", "upsample-1", "luminance")):
        np.testing.assert_array_equal(a, b, err_msg=name)
    assert got[True][4][0] != lum0[0]


@pytest.mark.parametrize("w,h", [(1920, 1080), (2560, 1440), (1000, 808), (333, 250), (256, 256), (70, 38)])
def test_fused_pyramid_middle_equals_the_two_separate_launches(gr, w, h):
    """gr_bloom_down_mid (downsample-0 and downsample-1 through LDS, one launch) must leave the very bytes of two gr_bloom_downsample
    calls in both levels -- exact 2:1 levels (1440p's 640 x 360 -> 320 x 180, 256 x 256), odd ones on the nine generic taps (1080p's 135
    rows, 1000 x 808, 333 x 250), partial tiles -- and, restricted to a band of downsample-1 rows, in those rows and in the rows of
    downsample-0 under their taps.  (Frames above 1440p keep the two launches: gr_bloom_down_mid_supported.)"""
    sz = [orc.level_size(w, h, s) for s in (0.5, 0.25, 0.125)]
    rng = np.random.default_rng(w * 17 + h)
    t_bits = np.exp2(rng.uniform(-8, 6, (sz[0][1], sz[0][0], 4))).astype(np.float16).view(np.uint16)
    t = capi.DeviceImage(gr, *sz[0], F16).upload(t_bits)
    want_d0, want_d1 = capi.DeviceImage(gr, *sz[1], F16), capi.DeviceImage(gr, *sz[2], F16)
    gr.bloom_downsample(t, want_d0)
    gr.bloom_downsample(want_d0, want_d1)
    d0, d1 = capi.DeviceImage(gr, *sz[1], F16), capi.DeviceImage(gr, *sz[2], F16)
    assert gr.bloom_down_mid(t, d0, d1), "a pyramid of InputRelative sizes must qualify (GR_MID_FUSION_ANY_SIZE: also above 1440p)"
    gr.sync()
    ref0, ref1 = want_d0.download(), want_d1.download()
    np.testing.assert_array_equal(d0.download(), ref0, err_msg="downsample-0")
    np.testing.assert_array_equal(d1.download(), ref1, err_msg="downsample-1")
    # a band of downsample-1 rows: those rows, and downsample-0 at least under their taps (1.75 texels either side), nothing of
    # downsample-1 outside the band
    h1, h0 = sz[2][1], sz[1][1]
    first, count = h1 // 3, max(h1 // 4, 1)
    poison = np.full((h1, sz[2][0], 4), 0x7bff, np.uint16)
    b0, b1 = capi.DeviceImage(gr, *sz[1], F16), capi.DeviceImage(gr, *sz[2], F16).upload(poison)
    assert gr.bloom_down_mid(t, b0, b1, rows=(first, count))
    gr.sync()
    got1, got0 = b1.download(), b0.download()
    np.testing.assert_array_equal(got1[first:first + count], ref1[first:first + count])
    assert (got1[:first] == 0x7bff).all() and (got1[first + count:] == 0x7bff).all()
    scale = h0 / h1
    lo = max(int(np.floor((first + 0.5) * scale - 0.5 - 1.75)), 0)
    hi = min(int(np.floor((first + count - 0.5) * scale - 0.5 + 1.75)) + 1, h0 - 1)
    np.testing.assert_array_equal(got0[lo:hi + 1], ref0[lo:hi + 1])


@pytest.mark.parametrize("w,h,packed,dynamic", [(1920, 1080, False, True), (2560, 1440, False, False), (256, 256, False, True), (64, 48, False, True),
                                                (1920, 1080, True, True), (328, 200, True, False)])
def test_fused_pyramid_head_equals_threshold_and_the_two_downsamples(gr, w, h, packed, dynamic):
    """gr_bloom_down_head (threshold, downsample-0, downsample-1 through LDS, one launch) must leave the very bytes of gr_bloom_threshold
    and two gr_bloom_downsample calls in all three levels: RGBA16F and B10G11R11 HDR targets, with and without the exposure buffer,
    partial tiles (1080p's 135 rows of downsample-1, 64 x 48)."""
    sz = [orc.level_size(w, h, s) for s in (0.5, 0.25, 0.125)]
    rng = np.random.default_rng(w * 31 + h)
    hdr_f = np.exp2(rng.uniform(-6, 6, (h, w, 4))).astype(np.float32)
    if packed:
        hdr = capi.DeviceImage(gr, w, h, capi.FORMAT_B10G11R11_UFLOAT_PACK32).upload(orc.pack_b10g11r11(hdr_f[..., :3]))
    else:
        hdr = capi.DeviceImage(gr, w, h, F16).upload(hdr_f.astype(np.float16).view(np.uint16))
    lum = capi.DeviceBuffer(gr, 12).upload(np.array([0.3, 1.7, 1.0 / 1.7], np.float32)) if dynamic else None  # (kept alive: .ptr alone would free it)
    lum_ptr = lum.ptr if lum is not None else None
    want = [capi.DeviceImage(gr, *s, F16) for s in sz]
    gr.bloom_threshold(hdr, want[0], lum_ptr)
    gr.bloom_downsample(want[0], want[1])
    gr.bloom_downsample(want[1], want[2])
    got = [capi.DeviceImage(gr, *s, F16) for s in sz]
    assert gr.bloom_down_head(hdr, got[0], got[1], got[2], lum_ptr), "even sizes up to 1440p must qualify"
    gr.sync()
    for a, b, name in zip(got, want, ("threshold", "downsample-0", "downsample-1")):
        np.testing.assert_array_equal(a.download(), b.download(), err_msg=name)


@pytest.mark.parametrize("w,h,dynamic", [(3840, 2160, True), (1920, 1080, True), (2560, 1440, True), (256, 256, True), (640, 360, False), (200, 136, True),
                                         (72, 40, False)])
def test_fused_upsample_chain_equals_the_separate_launches(gr, w, h, dynamic):
    """gr_bloom_up_all (luminance, upsample-2, upsample-1, upsample-0 through LDS, one launch) must leave the very bytes of gr_luminance
    and three gr_bloom_upsample calls in all three levels and in the luminance buffer: 1080p (upsample-1 on the nine generic taps:
    68 -> 135 rows), 1440p and 256 x 256 (every level on the 1:2 stencil), partial tiles, with and without the exposure buffer."""
    sz = [orc.level_size(w, h, s) for s in (0.25, 0.125, 0.0625, 0.03125)]
    rng = np.random.default_rng(w * 13 + h)
    d3 = capi.DeviceImage(gr, *sz[3], F16).upload(np.exp2(rng.uniform(-8, 4, (sz[3][1], sz[3][0], 4))).astype(np.float16).view(np.uint16))
    lum0 = np.array([0.25, 2.0 ** 0.25, 2.0 ** -0.25], np.float32)
    lum_lerp, _ = orc.frame_lerps(0.01)
    got = {}
    for fused in (False, True, "busy frame"):  # "busy frame": GR_BLOOM_BUSY_FRAME_BIT, 256-thread workgroups (the luminance reduction in its 1024-thread order on 256)
        u2, u1, u0 = (capi.DeviceImage(gr, *sz[i], F16) for i in (2, 1, 0))
        lum = capi.DeviceBuffer(gr, 12).upload(lum0) if dynamic else None
        lum_ptr = lum.ptr if lum is not None else None
        if fused:
            assert gr.bloom_up_all(d3, u2, u1, u0, lum_ptr, lum_lerp, busy_frame=fused == "busy frame"), "a pyramid of InputRelative sizes up to 1440p with an even quarter level must qualify"
        else:
            if dynamic:
                gr.luminance(d3, lum_ptr, lum_lerp)
            gr.bloom_upsample(d3, u2)
            gr.bloom_upsample(u2, u1)
            gr.bloom_upsample(u1, u0)
        gr.sync()
        got[fused] = (u2.download(), u1.download(), u0.download(), lum.download(np.float32) if dynamic else np.zeros(3, np.float32))
    for form in (True, "busy frame"):
        for a, b, name in zip(got[form], got[False], ("upsample-2", "upsample-1", "upsample-0", "luminance")):
            np.testing.assert_array_equal(a, b, err_msg=f"{name} ({form})")
    if dynamic:
        assert got[True][3][0] != lum0[0]


@pytest.mark.parametrize("w,h,packed,dynamic", [(1920, 1080, False, True), (256, 256, False, True), (64, 48, False, True), (640, 360, False, False),
                                                (1920, 1080, True, True), (328, 200, True, False), (1280, 720, False, True), (200, 136, False, True)])
def test_whole_pyramid_in_one_launch_equals_the_separate_launches(gr, w, h, packed, dynamic):
    """gr_bloom_pyramid (threshold, four downsamples + feedback, luminance, three upsamples as three block ranges of ONE grid, a range waiting on
    device-side counters for the one before it) must leave the very bytes of the nine separate launches in all eight levels and in the luminance
    buffer: the sizes of the fused-head / fused-upsample tests (256 x 256 = BASELINE config 1, partial tiles; 1080p with its odd levels on the nine
    generic taps and 720p through the launcher directly -- gr_bloom_pyramid_supported offers the launch up to 640 x 384), both HDR formats, with and without the exposure buffer -- and run three times over the same images, as consecutive frames do
    (the counters of a launch are left at zero by its last workgroup).  No workgroup may have given up waiting."""
    names = ("threshold", "d0", "d1", "d2", "d3")
    scales = dict(threshold=0.5, d0=0.25, d1=0.125, d2=0.0625, d3=0.03125, u2=0.0625, u1=0.125, u0=0.25)
    rng = np.random.default_rng(w * 7 + h)
    hdr_f = np.exp2(rng.uniform(-6, 6, (h, w, 4))).astype(np.float32)
    if packed:
        hdr = capi.DeviceImage(gr, w, h, capi.FORMAT_B10G11R11_UFLOAT_PACK32).upload(orc.pack_b10g11r11(hdr_f[..., :3]))
    else:
        hdr = capi.DeviceImage(gr, w, h, F16).upload(hdr_f.astype(np.float16).view(np.uint16))
    d3_size = orc.level_size(w, h, scales["d3"])
    history = capi.DeviceImage(gr, *d3_size, F16).upload(np.exp2(rng.uniform(-8, 2, (d3_size[1], d3_size[0], 4))).astype(np.float16).view(np.uint16))
    lum0 = np.array([0.3, 1.7, 1.0 / 1.7], np.float32)
    lum_lerp, feedback_lerp = orc.frame_lerps(0.01)
    results = {}
    for fused in (False, True):
        l = {name: capi.DeviceImage(gr, *orc.level_size(w, h, s), F16) for name, s in scales.items()}
        lum = capi.DeviceBuffer(gr, 12).upload(lum0) if dynamic else None
        lum_ptr = lum.ptr if lum is not None else None
        for frame in range(3):
            if fused:
                offered = gr.bloom_pyramid(hdr, l, history, feedback_lerp, lum_ptr, lum_lerp, any_size=w * h > 640 * 384)
                assert offered, "a frame up to 640 x 384 with even half / quarter / eighth levels must qualify"
            else:
                gr.bloom_threshold(hdr, l["threshold"], lum_ptr)
                for src, dst in zip(names[:-1], names[1:]):
                    gr.bloom_downsample(l[src], l[dst], history if dst == "d3" else None, feedback_lerp)
                if dynamic:
                    gr.luminance(l["d3"], lum_ptr, lum_lerp)
                gr.bloom_upsample(l["d3"], l["u2"])
                gr.bloom_upsample(l["u2"], l["u1"])
                gr.bloom_upsample(l["u1"], l["u0"])
        gr.sync()
        results[fused] = {name: img.download() for name, img in l.items()}
        results[fused]["luminance"] = lum.download(np.float32) if dynamic else np.zeros(3, np.float32)
    assert gr.pyramid_giveups() == 0
    for name in results[True]:
        np.testing.assert_array_equal(results[True][name], results[False][name], err_msg=name)
    if dynamic:
        assert results[True]["luminance"][0] != lum0[0]


def test_whole_pyramid_declines_what_it_does_not_cover(gr):
    """Frames above 640 x 384 and pyramids with an odd half / quarter / eighth level keep the separate (fused-by-parts) launches."""
    scales = dict(threshold=0.5, d0=0.25, d1=0.125, d2=0.0625, d3=0.03125, u2=0.0625, u1=0.125, u0=0.25)
    for w, h in ((1920, 1080), (3840, 2160), (330, 202)):
        hdr = capi.DeviceImage(gr, w, h, F16)
        l = {name: capi.DeviceImage(gr, *orc.level_size(w, h, s), F16) for name, s in scales.items()}
        history = capi.DeviceImage(gr, *orc.level_size(w, h, 0.03125), F16)
        assert not gr.bloom_pyramid(hdr, l, history, 0.1)


def test_fused_upsample_chain_declines_what_it_does_not_cover(gr):
    """A quarter level that is not exactly twice the eighth keeps gr_bloom_up_tail + gr_bloom_upsample, and so does a frame above 4K
    (the tiles' recomputed patches cost more than the launch they save there: profiles/r05_up_fusion_by_size.txt)."""
    u0, u1, u2, d3 = (capi.DeviceImage(gr, *orc.level_size(1004, 812, s), F16) for s in (0.25, 0.125, 0.0625, 0.03125))
    assert not gr.bloom_up_all(d3, u2, u1, u0)
    u0, u1, u2, d3 = (capi.DeviceImage(gr, *orc.level_size(7680, 4320, s), F16) for s in (0.25, 0.125, 0.0625, 0.03125))
    assert not gr.bloom_up_all(d3, u2, u1, u0)


def test_fused_pyramid_head_declines_what_it_does_not_cover(gr):
    """Odd level sizes (the nine generic taps) and frames above 1440p keep the separate launches."""
    for w, h in ((1002, 810), (3840, 2160)):
        hdr = capi.DeviceImage(gr, w, h, F16)
        t, d0, d1 = (capi.DeviceImage(gr, *orc.level_size(w, h, s), F16) for s in (0.5, 0.25, 0.125))
        assert not gr.bloom_down_head(hdr, t, d0, d1)


def test_fused_pyramid_middle_is_for_launch_bound_frames_only(gr):
    """At 4K the fused form recomputes more than the saved launch is worth (measured: the frame gets 3 % slower): not offered."""
    t, d0, d1 = (capi.DeviceImage(gr, *orc.level_size(3840, 2160, s), F16) for s in (0.5, 0.25, 0.125))
    assert not gr.bloom_down_mid(t, d0, d1)


def test_fused_pyramid_tail_declines_what_it_does_not_cover(gr):
    """A level that is not ceil(half) of the one above (not a pyramid of render_graph.cpp's InputRelative sizes: the patch of upsample-2
    under a tile of upsample-1 would not fit the kernel's LDS) falls back to the separate launches."""
    d1 = capi.DeviceImage(gr, 101, 57, F16)
    d2, u2 = capi.DeviceImage(gr, 80, 40, F16), capi.DeviceImage(gr, 80, 40, F16)   # 101 -> 80
    d3, hist = capi.DeviceImage(gr, 40, 20, F16), capi.DeviceImage(gr, 40, 20, F16)
    u1 = capi.DeviceImage(gr, 101, 57, F16)
    assert not gr.bloom_tail(d1, d2, d3, hist, u2, u1, 0.1)


def test_a_centre_tap_does_not_read_its_overflowed_neighbour(gr):
    """Under the sampler statement (a coordinate within 2^-8 of a texel centre reads that texel alone) a tap on a pixel centre has
    weight exactly 0 for its neighbours.  An fp16 +inf texel (an overflowed HDR value) one texel further on must not reach the result as
    0 * inf = NaN: the oracle's linear_combine skips it, a hardware sampler returns the texel.  Upsample between equal sizes puts all
    nine taps on texel centres: the 3 x 3 neighbourhood of the inf texel becomes inf, the ring around it must stay finite and equal
    to the oracle's."""
    w = h = 32
    src = synth.make_hdr(w, h)
    src[16, 16, :3] = 0x7c00  # +inf
    dev_in = capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16B16A16_SFLOAT).upload(src)
    dev_out = capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16B16A16_SFLOAT)
    gr.bloom_upsample(dev_in, dev_out)
    gr.sync()
    got = dev_out.download().view(np.float16).astype(np.float32)
    want = orc.bloom_upsample(src, w, h).view(np.float16).astype(np.float32)
    assert not np.isnan(want).any() and np.isinf(want[15:18, 15:18, :3]).all() and np.isfinite(want[13, 13:20]).all()
    assert not np.isnan(got).any(), f"{int(np.isnan(got).any(axis=2).sum())} pixels are NaN beside the overflowed texel"
    np.testing.assert_array_equal(np.isinf(got), np.isinf(want))
    finite = np.isfinite(want)
    np.testing.assert_allclose(got[finite], want[finite], rtol=2e-3, atol=1e-4)
