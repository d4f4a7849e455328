"""CPU: restatements of the host-side math on the path checked against the REFERENCE's own code -- math/muglm compiled from
/root/reference by oracle/ref_build/Makefile into oracle/_ref/libref_muglm.so (the reference is C++, its math library builds
from two files; its shaders cannot be built or run here).  This pins: the half packing of the light records
(muglm::floatToHalf, used by LightClusterer::refresh_bindless_prepare), RenderContext::set_camera's derived parameters as
the C++ host layer computes them, and the synthetic camera the tests feed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from granite_amd import app as gapp, synth
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_muglm.so")


@pytest.fixture(scope="module")
def ref():
    if os.path.isdir("/root/reference/math/muglm"):
        try:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_build")])
        except (subprocess.CalledProcessError, OSError) as e:  # keep going with a prebuilt library if there is one
            print("oracle/ref_build did not build:", e)
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libref_muglm.so not built (needs /root/reference)")
    lib = C.CDLL(REF_LIB)
    lib.ref_float_to_half.restype, lib.ref_float_to_half.argtypes = C.c_uint16, [C.c_float]
    lib.ref_half_to_float.restype, lib.ref_half_to_float.argtypes = C.c_float, [C.c_uint16]
    lib.ref_perspective.argtypes = [C.c_float] * 4 + [C.c_void_p]
    lib.ref_float_to_half_array.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.ref_camera_parameters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    orc.lib().orc_float_to_half_muglm_array.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    return lib


def mat(a):
    return np.ascontiguousarray(a, np.float32).reshape(16)


def test_float_to_half_restatement_equals_muglm_on_every_rounding_case(ref):
    """16.8 M bit patterns: every 256th float plus a dense sweep around every half rounding boundary (where muglm rounds ties
    up instead of to even) and all specials."""
    bits = np.arange(0, 1 << 32, 256, dtype=np.uint64).astype(np.uint32)
    halves = np.arange(0, 0x7C00, dtype=np.uint32)
    as_f32 = ((halves >> 10) + 112) << 23 | ((halves & 0x3FF) << 13)
    around = (as_f32[:, None] + np.arange(0x0FF0, 0x1010, dtype=np.uint32)[None, :]).reshape(-1)  # tie = 0x1000
    special = np.array([0, 0x80000000, 0x7F800000, 0xFF800000, 0x7FC00000, 0x00000001, 0x33000000, 0x33000001, 0x387FFFFF, 0x477FF000,
                        0x477FEFFF, 0x47800000], np.uint32)
    allbits = np.concatenate([bits, around, around | np.uint32(0x80000000), special])
    x = allbits.view(np.float32)
    got = np.empty(x.size, np.uint16)
    want = np.empty(x.size, np.uint16)
    orc.lib().orc_float_to_half_muglm_array(x.ctypes.data, got.ctypes.data, C.c_uint64(x.size))
    ref.ref_float_to_half_array(x.ctypes.data, want.ctypes.data, C.c_uint64(x.size))
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, [(hex(int(allbits[i])), hex(int(got[i])), hex(int(want[i]))) for i in bad[:5]]
    # and back: every half
    for h in list(range(0, 0x10000, 7)) + [0x7C00, 0xFC00, 0x0001, 0x03FF, 0x0400]:
        a, b = orc.lib().orc_half_to_float(h), ref.ref_half_to_float(h)
        assert a == b or (a != a and b != b), hex(h)


def test_host_camera_parameters_equal_the_reference_computation(ref):
    """gra_set_camera -> RenderContext::set_camera restated in the C++ host layer, vs the same quantities computed with the
    reference's own mat4 product / inverse (render_context.cpp:53-86)."""
    a = gapp.Application(640, 360, device=-1)
    rng = np.random.default_rng(8)
    for trial in range(6):
        cam = synth.Camera(640, 360, fovy_deg=40 + 10 * trial, near=0.05 * (trial + 1), far=50.0 + 40 * trial,
                           eye=tuple(rng.uniform(-6, 6, 3)), center=tuple(rng.uniform(-1, 1, 3)))
        P, V = mat(cam.P.T), mat(cam.V.T)
        a.set_camera(P, V)
        got = a.get_render_parameters()
        want = np.zeros(72, np.float32)
        ref.ref_camera_parameters(P.ctypes.data, V.ctypes.data, want.ctypes.data)
        np.testing.assert_array_equal(got[0:16], P)
        np.testing.assert_array_equal(got[16:32], V)
        for name, g, w in (("view_projection", got[32:48], want[32:48]), ("inv_projection", got[48:64], want[0:16]),
                           ("inv_view", got[64:80], want[16:32]), ("inv_view_projection", got[80:96], want[48:64])):
            scale = np.abs(w).max()
            assert np.abs(g - w).max() <= 2e-6 * scale, (name, trial, np.abs(g - w).max(), scale)
        np.testing.assert_allclose(got[96:99], want[64:67], rtol=0, atol=2e-6 * max(1.0, np.abs(want[64:67]).max()))
        np.testing.assert_allclose(got[99:102], want[67:70], rtol=0, atol=2e-6)
        np.testing.assert_allclose(got[102:104], want[70:72], rtol=2e-5)
        # the synthetic camera the parity tests use (float64 numpy) describes the same camera
        sp = cam.render_params()
        np.testing.assert_allclose(sp[48:64], want[0:16], rtol=0, atol=3e-6 * np.abs(want[0:16]).max())
        np.testing.assert_allclose(sp[80:96], want[48:64], rtol=0, atol=3e-6 * np.abs(want[48:64]).max())
    a.close()


def test_default_harness_projection_is_muglm_perspective(ref):
    a = gapp.Application(1280, 720, device=-1)
    want = np.zeros(16, np.float32)
    ref.ref_perspective(1.0471975512, 1280.0 / 720.0, 0.1, 100.0, want.ctypes.data)
    np.testing.assert_allclose(a.get_render_parameters()[0:16], want, rtol=2e-6, atol=1e-7)
    a.close()


def test_fsr_constants_equal_the_vendored_header_in_cpu_mode(ref):
    """FsrEasuCon / FsrRcasCon compiled from assets/shaders/post/ffx-fsr/ffx_fsr1.h (A_CPU): the resolve-position scale and
    offset the EASU restatement uses, and the linear sharpness of RCAS."""
    ref.ref_fsr_easu_constants.argtypes = [C.c_float] * 4 + [C.c_void_p]
    ref.ref_fsr_rcas_constants.argtypes = [C.c_float, C.c_void_p]
    orc.lib().orc_fsr_easu_constants.argtypes = [C.c_int] * 4 + [C.c_void_p]
    for iw, ih, ow, oh in ((1280, 720, 1920, 1080), (2560, 1440, 3840, 2160), (480, 270, 720, 405), (7, 5, 30, 22), (1001, 333, 1920, 1080)):
        want = np.zeros(16, np.uint32)
        ref.ref_fsr_easu_constants(iw, ih, ow, oh, want.ctypes.data)
        got = np.zeros(4, np.float32)
        orc.lib().orc_fsr_easu_constants(iw, ih, ow, oh, got.ctypes.data)
        np.testing.assert_array_equal(got.view(np.uint32), want[0:4])
        # the gather positions are texel corners: con1 = (1/iw, 1/ih, 1/iw, -1/ih), i.e. taps at integer offsets from floor(pp)
        f = want.view(np.float32)
        assert f[4] == np.float32(1) / np.float32(iw) and f[5] == np.float32(1) / np.float32(ih)
        assert f[6] == f[4] and f[7] == -f[5]
    con = np.zeros(4, np.uint32)
    ref.ref_fsr_rcas_constants(0.5, con.ctypes.data)
    assert con.view(np.float32)[0] == np.float32(orc.fsr_rcas_sharpness(0.5))


def test_rec709_to_display_matrix_equals_the_reference(ref):
    """compute_rec709_to_st2020 (hdr.cpp:580-593) in the C++ host layer vs the same three statements over the reference's own
    compute_xyz_matrix (math/transforms.cpp), inverse and mat3 product: ST.2020, DCI-P3 and rec.709 displays."""
    ref.ref_rec709_to_display.argtypes = [C.c_void_p, C.c_void_p]
    lib = gapp.load_library()
    for prim in ((0.708, 0.292, 0.170, 0.797, 0.131, 0.046, 0.3127, 0.3290), (0.680, 0.320, 0.265, 0.690, 0.150, 0.060, 0.3127, 0.3290),
                 (0.640, 0.330, 0.3, 0.6, 0.150, 0.060, 0.3127, 0.3290)):
        p8 = np.array(prim, np.float32)
        want, got = np.zeros(9, np.float32), np.zeros(9, np.float32)
        ref.ref_rec709_to_display(p8.ctypes.data, want.ctypes.data)
        assert lib.gra_compute_rec709_to_display(p8.ctypes.data, got.ctypes.data) == 0
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    np.testing.assert_allclose(want.reshape(3, 3), np.eye(3), atol=2e-6)  # rec.709 display: identity
    np.testing.assert_allclose(orc.rec709_to_display(), np.array([0.6274039, 0.06909728, 0.01639144, 0.3292831, 0.91954035, 0.08801329,
                                                                  0.04331313, 0.01136229, 0.8955956], np.float32), atol=1e-6)
