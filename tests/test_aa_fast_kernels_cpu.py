"""CPU: the fast anti-aliasing kernels (granite_amd/csrc/aa_fast_kernels.hpp, the text the GPU build compiles) run through the
host emulation of tests/cpp/hip_emu.hpp -- a thread per lane, barriers and votes as rendezvous -- and must equal the oracle bit
for bit: what is checked here is the kernels' tiling, halo staging, clamping at the image border, row bands and the per-pixel
arithmetic of aa_core.hpp; what a GPU run adds is the compiler and the hardware."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from granite_amd import synth
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRESET_THRESHOLD = (0.15, 0.1, 0.1, 0.05)  # SMAA.hlsl:304-324


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("aa_fast_host") / "libaa_fast_host.so")
    subprocess.check_call(["g++", "-O1", "-std=c++20", "-pthread", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                           "-I" + os.path.join(ROOT, "tests", "cpp", "emu_include"),
                           os.path.join(ROOT, "tests", "cpp", "aa_fast_host.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.aah_centre_taps_exact.argtypes = [C.c_int, C.c_float]
    lib.aah_centre_taps_exact.restype = C.c_int
    return lib


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def noisy(w, h, seed=7):
    """Blocky random gamma-space bytes with single-pixel speckle: every pixel is an edge candidate (tests/test_gpu_aa.py)."""
    r = np.random.default_rng(seed)
    coarse = r.integers(0, 256, ((h + 2) // 3, (w + 2) // 3, 4), dtype=np.uint8)
    img = np.repeat(np.repeat(coarse, 3, axis=0), 3, axis=1)[:h, :w].copy()
    fine = r.integers(0, 256, (h, w, 4), dtype=np.uint8)
    mask = r.random((h, w)) < 0.3
    img[mask] = fine[mask]
    img[..., 3] = 255
    return img


SIZES = [(70, 40, "pattern"), (67, 35, "noise"), (33, 17, "pattern"), (8, 8, "noise"), (96, 20, "noise")]


def source(w, h, kind):
    return synth.make_ldr_pattern(w, h) if kind == "pattern" else noisy(w, h)


def test_pixel_centre_taps_are_texel_fetches_at_every_size_in_use(host):
    """The launchers' guard (aa_core.hpp: axis_taps_exact) holds for every axis length up to 8K and for the sizes of the
    multi-GPU weak-scaling frames; it fails where the fp32 round trip of a pixel centre leaves the snap radius."""
    for n in list(range(1, 600)) + [1080, 1920, 2160, 3840, 4320, 7680, 8640]:
        assert host.aah_centre_taps_exact(n, np.float32(1.0) / np.float32(n)), n
    assert not host.aah_centre_taps_exact(100000, np.float32(1.0) / np.float32(100000))


def test_diagonal_search_walks_land_on_texels_at_every_size_in_use(host):
    """aa_core.hpp: axis_walk_exact -- 17 fma steps of one texel from any pixel centre (in x also from a quarter texel beside it)
    resolve to the texel the integer walk of the weight kernel reads, for every axis up to 4K; beyond (7680: seventeen roundings
    of a coordinate near 1.0 add up to more than the sampler's snap radius of 1/256 texel) the kernel keeps the float walk."""
    host.aah_diag_walk_exact.argtypes = [C.c_int, C.c_float, C.c_int]
    host.aah_diag_walk_exact.restype = C.c_int
    for n in list(range(1, 300)) + [333, 480, 512, 1080, 1920, 2160, 3840, 4320]:
        inv = np.float32(1.0) / np.float32(n)
        assert host.aah_diag_walk_exact(n, inv, 0) and host.aah_diag_walk_exact(n, inv, 1), n
    assert not host.aah_diag_walk_exact(7680, np.float32(1.0) / np.float32(7680), 0)


# (134, 70) and (140, 64): workgroups whose tile lies inside the image (unclamped neighbours, two texels per load); (101, 60): the
# same geometry with rows that are not 8-byte multiples, i.e. the clamped path everywhere
@pytest.mark.parametrize("w,h,kind", SIZES + [(134, 70, "noise"), (140, 64, "pattern"), (101, 60, "noise")])
def test_fxaa_kernel_equals_oracle(host, w, h, kind):
    src = source(w, h, kind)
    out = np.zeros_like(src)
    host.aah_fxaa(p(src), w, h, p(out), 0, 0)
    np.testing.assert_array_equal(out, orc.fxaa(src, False))


@pytest.mark.parametrize("w,h", [(70, 61), (134, 90)])
def test_fxaa_kernel_row_band(host, w, h):
    src = noisy(w, h, 3)
    ref = orc.fxaa(src, False)
    out = np.full_like(src, 0xAB)
    host.aah_fxaa(p(src), w, h, p(out), 19, 23)
    np.testing.assert_array_equal(out[19:42], ref[19:42])
    assert (out[:19] == 0xAB).all() and (out[42:] == 0xAB).all()


@pytest.mark.parametrize("quality", [0, 3])
@pytest.mark.parametrize("w,h,kind", SIZES)
def test_smaa_edge_kernel_equals_oracle(host, w, h, kind, quality):
    src = source(w, h, kind)
    edges = np.full((h, w, 2), 0xCD, np.uint8)
    host.aah_smaa_edges(p(src), w, h, p(edges), C.c_float(PRESET_THRESHOLD[quality]), 0, 0)
    ref = orc.smaa_edges(src, quality)
    np.testing.assert_array_equal(edges, ref)
    assert ref.any()


@pytest.mark.parametrize("wide", [0, 1])
@pytest.mark.parametrize("w,h,kind", [(72, 40, "pattern"), (68, 36, "noise"), (8, 8, "noise"), (67, 35, "noise")])
def test_smaa_blend_kernel_equals_oracle(host, w, h, kind, wide):
    if wide and w % 4:
        pytest.skip("the 4-pixel form needs width % 4 == 0")
    from granite_amd.data import load_smaa_luts
    area, search = load_smaa_luts()
    src = source(w, h, kind)
    ref = orc.smaa(src, area, search, 3, False)
    out = np.zeros_like(src)
    host.aah_smaa_blend(p(src), p(ref["weights"]), w, h, p(out), wide, 0, 0)
    np.testing.assert_array_equal(out, ref["out"])
    assert (out != src).any()
    # a band
    out2 = np.full_like(src, 0x11)
    host.aah_smaa_blend(p(src), p(ref["weights"]), w, h, p(out2), wide, 5, 13)
    np.testing.assert_array_equal(out2[5:18], ref["out"][5:18])
    assert (out2[:5] == 0x11).all() and (out2[18:] == 0x11).all()


def taa_inputs(w, h, seed=3):
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam, seed)
    cur = synth.make_hdr(w, h, seed)
    mv = synth.make_motion_vectors(w, h)
    V2 = synth.look_at((0.01, 2.0, 8.0), (0.01, 1.0, 0.0))  # VP_prev = the camera translated by 0.01 along x
    T = np.eye(4); T[0, 0] = T[1, 1] = 0.5; T[0, 3] = T[1, 3] = 0.5
    reproj = T @ (cam.P @ V2) @ cam.invVP
    return cur, gbuf["depth"], mv, np.ascontiguousarray(reproj.T, np.float32).reshape(16)


@pytest.mark.parametrize("quality", [0, 1, 2])
@pytest.mark.parametrize("w,h", [(80, 45), (67, 35)])
def test_taa_kernel_within_the_resolve_tolerance(host, w, h, quality):
    """k_taa_fast against the oracle at the tolerance of tests/test_gpu_aa.py (the oracle's history is fed to both, so errors are
    not carried): first frame without history, then two frames with it; a row band writes only its rows."""
    from util import assert_rgba16f_close
    cur, depth, mv, reproj = taa_inputs(w, h)
    col, hist = np.zeros((h, w, 4), np.uint16), np.zeros((h, w, 4), np.uint16)
    host.aah_taa(p(cur), p(depth), p(mv), None, w, h, p(reproj), quality, p(col), p(hist), 0, 0)
    ref_c, ref_h = orc.taa_resolve(cur, depth, mv, None, reproj, quality)
    assert_rgba16f_close(col, ref_c, what="f0 colour")
    assert_rgba16f_close(hist, ref_h, what="f0 history")
    cur2 = synth.make_hdr(w, h, seed=11)
    prev = ref_h
    for f in range(2):
        host.aah_taa(p(cur2), p(depth), p(mv), p(prev), w, h, p(reproj), quality, p(col), p(hist), 0, 0)
        ref_c, ref_h2 = orc.taa_resolve(cur2, depth, mv, prev, reproj, quality)
        assert_rgba16f_close(col, ref_c, ulps=2.0, abs_tol=1e-4, what=f"q{quality} f{f + 1} colour")
        assert_rgba16f_close(hist, ref_h2, ulps=2.0, abs_tol=1e-4, what=f"q{quality} f{f + 1} history")
        same = (hist == ref_h2).mean()
        assert same > 0.9, same
        prev = ref_h2
    band = np.full((h, w, 4), 0x1234, np.uint16)
    band_h = band.copy()
    host.aah_taa(p(cur2), p(depth), p(mv), p(prev), w, h, p(reproj), quality, p(band), p(band_h), 7, 20)
    full_c, full_h = np.zeros_like(band), np.zeros_like(band)
    host.aah_taa(p(cur2), p(depth), p(mv), p(prev), w, h, p(reproj), quality, p(full_c), p(full_h), 0, 0)
    np.testing.assert_array_equal(band[7:27], full_c[7:27])
    np.testing.assert_array_equal(band_h[7:27], full_h[7:27])
    assert (band[:7] == 0x1234).all() and (band[27:] == 0x1234).all()


@pytest.mark.parametrize("quality", [0, 2])
def test_taa_band_says_when_a_pixel_reaches_outside_the_history_rows_it_holds(host, quality):
    """Row bands with a halo of history rows (gr_taa_resolve_band): the band's pixels equal the whole-frame launch as long as every
    fetched history row is among the rows held -- rows outside them are poisoned here -- and the flag is raised, by exactly the
    launches whose motion reaches further than the halo, otherwise."""
    w, h = 64, 96
    cur, depth, _, reproj = taa_inputs(w, h)
    prev = orc.taa_resolve(synth.make_hdr(w, h, seed=11), depth, np.zeros((h, w, 2), np.uint16), None, reproj, quality)[1]
    first, count = 32, 32
    for rows_of_motion, halo, expect_flag in ((3.0, 8, False), (3.0, 2, True), (-6.5, 9, False), (-6.5, 5, True), (0.0, 3, False)):
        mv = np.zeros((h, w, 2), np.float32)
        mv[..., 1] = rows_of_motion / h  # history position = v - mv: rows_of_motion rows up (or down) everywhere
        mv[0, 0, 1] = 1e-3               # (a pixel outside the band may point anywhere)
        mv16 = mv.astype(np.float16).view(np.uint16)
        full_c, full_h = np.zeros((h, w, 4), np.uint16), np.zeros((h, w, 4), np.uint16)
        host.aah_taa(p(cur), p(depth), p(mv16), p(prev), w, h, p(reproj), quality, p(full_c), p(full_h), 0, 0)
        held_first, held_end = max(first - halo, 0), min(first + count + halo, h)
        poisoned = prev.copy()
        poisoned[:held_first] = 0x7bff
        poisoned[held_end:] = 0x7bff
        band_c, band_h = np.zeros_like(full_c), np.zeros_like(full_h)
        flag = np.zeros(1, np.uint32)
        host.aah_taa_band(p(cur), p(depth), p(mv16), p(poisoned), w, h, p(reproj), quality, p(band_c), p(band_h), first, count, 0, 0,
                          held_first, held_end - held_first, p(flag))
        assert bool(flag[0]) == expect_flag, (rows_of_motion, halo, int(flag[0]))
        if not expect_flag:
            np.testing.assert_array_equal(band_c[first:first + count], full_c[first:first + count])
            np.testing.assert_array_equal(band_h[first:first + count], full_h[first:first + count])


def stair_card(w, h):
    """Long straight and stair-stepped edges, so that the orthogonal searches run to their limit (32 steps at Ultra) and the
    diagonal ones find real diagonals; plus the test card's features."""
    img = synth.make_ldr_pattern(w, h).copy()
    y, x = np.mgrid[0:h, 0:w]
    img[(y > h // 3) & (y < h // 3 + 3)] = (250, 250, 250, 255)                      # a bar across the whole image
    img[(x > w // 2) & (x < w // 2 + 2)] = (5, 5, 5, 255)                            # a column down the whole image
    img[(y - x // 7) % 23 == 0] = (255, 40, 40, 255)                                  # shallow stairs: Z patterns with far ends
    img[(x - y // 5) % 31 == 0] = (40, 255, 40, 255)                                  # steep stairs
    return img


@pytest.mark.parametrize("quality", [0, 1, 2, 3])
@pytest.mark.parametrize("w,h,kind", [(200, 120, "stairs"), (70, 40, "pattern"), (67, 35, "noise"), (8, 8, "noise"), (150, 33, "stairs")])
def test_smaa_weight_kernel_over_bit_planes_equals_oracle(host, w, h, kind, quality):
    from granite_amd.data import load_smaa_luts
    area, search = load_smaa_luts()
    src = stair_card(w, h) if kind == "stairs" else source(w, h, kind)
    edges = orc.smaa_edges(src, quality)
    ref = orc.smaa_weights(edges, area, search, quality)
    out = np.full((h, w, 4), 0x77, np.uint8)
    host.aah_smaa_weights(p(edges), w, h, p(area), p(search), quality, p(out), 0, 0)
    np.testing.assert_array_equal(out, ref)
    assert ref.any()


@pytest.mark.parametrize("switch", ["AAH_FLOAT_DIAG_WALKS", "AAH_NO_CENTRE_SNAP"])
def test_smaa_weight_kernel_on_its_float_paths_equals_oracle(host, switch, monkeypatch):
    """What the kernel falls back to where the host cannot prove that coordinates land on texels (axes of 7680 pixels: the diagonal
    walks; beyond ~16K: every pixel-centre tap): the same pass with the diagonal searches sampled in fp32, resp. with no tap treated
    as a texel fetch -- forced here on a small image, results unchanged."""
    from granite_amd.data import load_smaa_luts
    area, search = load_smaa_luts()
    w, h = 200, 120
    edges = orc.smaa_edges(stair_card(w, h), 3)
    ref = orc.smaa_weights(edges, area, search, 3)
    monkeypatch.setenv(switch, "1")
    out = np.full((h, w, 4), 0x77, np.uint8)
    host.aah_smaa_weights(p(edges), w, h, p(area), p(search), 3, p(out), 0, 0)
    np.testing.assert_array_equal(out, ref)


def test_smaa_weight_kernel_row_band(host):
    from granite_amd.data import load_smaa_luts
    area, search = load_smaa_luts()
    w, h = 160, 300
    src = stair_card(w, h)
    edges = orc.smaa_edges(src, 3)
    ref = orc.smaa_weights(edges, area, search, 3)
    out = np.full((h, w, 4), 0x77, np.uint8)
    host.aah_smaa_weights(p(edges), w, h, p(area), p(search), 3, p(out), 130, 41)
    np.testing.assert_array_equal(out[130:171], ref[130:171])
    assert (out[:130] == 0x77).all() and (out[171:] == 0x77).all()


@pytest.mark.parametrize("quality", [0, 2])
def test_taa_kernel_with_b10g11r11_input_and_colour_output(host, quality):
    """renderTargetFp16 = false: the current frame comes in as B10G11R11_UFLOAT_PACK32 words and the resolved colour goes out in the
    same format (rounded once, from fp32: the store conversion is exact integer code -- wherever the fp32 colour agrees the words
    are equal); history RGBA16F."""
    from util import assert_rgba16f_close
    w, h = 80, 45
    cur16, depth, mv, reproj = taa_inputs(w, h)
    cur16 = orc.quantize_b10g11r11(cur16)
    cur = orc.pack_b10g11r11(cur16)
    prev = orc.taa_resolve(synth.make_hdr(w, h, seed=11), depth, mv, None, reproj, quality)[1]
    col, hist = np.zeros((h, w), np.uint32), np.zeros((h, w, 4), np.uint16)
    host.aah_taa_fmt(p(cur), p(depth), p(mv), p(prev), w, h, p(reproj), quality, p(col), p(hist), 0, 0, 1, 1)
    ref_c, ref_h = orc.taa_resolve(cur16, depth, mv, prev, reproj, quality, color_b10g11r11=True)
    assert_rgba16f_close(hist, ref_h, ulps=2.0, abs_tol=1e-4, what="history")
    got_c = orc.unpack_b10g11r11(col)
    assert np.array_equal(orc.pack_b10g11r11(got_c), col)
    # a neighbouring code of a channel only where the fp32 colour (a few ulps apart between kernel and oracle) sits on a tie
    want = orc.pack_b10g11r11(ref_c)
    for shift, bits in ((0, 11), (11, 11), (22, 10)):
        a, b = (col >> shift) & ((1 << bits) - 1), (want >> shift) & ((1 << bits) - 1)
        assert np.abs(a.astype(np.int64) - b.astype(np.int64)).max() <= 1
    assert (col == want).mean() > 0.98
