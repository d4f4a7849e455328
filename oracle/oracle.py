"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/liboracle.so, the CPU restatement of Granite's image-space chain (see oracle_common.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(granite_amd/) never does.  The reference has no tests for this path; the oracle is pinned by executing the
reference's own shader sources on the CPU (oracle/ref_build -> oracle/_ref, tests/test_reference_shaders_cpu.py, bit for bit)
plus analytic known-answer tests and fixtures in tests/golden/.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

LIGHT_DESC_DTYPE = np.dtype([("type", "<i4"), ("color", "<f4", 3), ("inner_cone", "<f4"), ("outer_cone", "<f4"),
                             ("cutoff_range", "<f4"), ("pad", "<f4"), ("transform", "<f4", (3, 4))])
assert LIGHT_DESC_DTYPE.itemsize == 80

LIGHT_INFO_DTYPE = np.dtype([("color", "<f4", 3), ("spot_scale_bias", "<u4"), ("position", "<f4", 3),
                             ("offset_radius", "<u4"), ("direction", "<f4", 3), ("inv_radius", "<f4")])
assert LIGHT_INFO_DTYPE.itemsize == 48

RENDER_PARAMS_DTYPE = np.dtype([("projection", "<f4", 16), ("view", "<f4", 16), ("view_projection", "<f4", 16),
                                ("inv_projection", "<f4", 16), ("inv_view", "<f4", 16), ("inv_view_projection", "<f4", 16),
                                ("camera_position", "<f4", 3), ("camera_front", "<f4", 3), ("z_near", "<f4"),
                                ("z_far", "<f4")])
assert RENDER_PARAMS_DTYPE.itemsize == 104 * 4

CLUSTER_PARAMS_DTYPE = np.dtype([("transform", "<f4", 16), ("clip_scale", "<f4", 4), ("camera_base", "<f4", 3),
                                 ("pad0", "<f4"), ("camera_front", "<f4", 3), ("pad1", "<f4"), ("xy_scale", "<f4", 2),
                                 ("resolution_xy", "<i4", 2), ("inv_resolution_xy", "<f4", 2), ("num_lights", "<i4"),
                                 ("num_lights_32", "<i4"), ("num_decals", "<i4"), ("num_decals_32", "<i4"),
                                 ("decals_texture_offset", "<i4"), ("z_max_index", "<i4"), ("z_scale", "<f4"),
                                 ("pad2", "<f4", 3)])
assert CLUSTER_PARAMS_DTYPE.itemsize == 176


class LightingArgs(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("albedo", C.c_void_p), ("normal", C.c_void_p),
                ("pbr", C.c_void_p), ("depth", C.c_void_p), ("hdr", C.c_void_p), ("rp", C.c_void_p),
                ("cluster", C.c_void_p), ("lights", C.c_void_p), ("type_mask", C.c_void_p), ("bitmask", C.c_void_p),
                ("range", C.c_void_p), ("dir_color", C.c_float * 3), ("dir_direction", C.c_float * 3),
                ("enable_directional", C.c_int32), ("enable_clustered", C.c_int32), ("ambient_fallback", C.c_int32),
                ("wave_tile", C.c_int32), ("ambient_occlusion", C.c_void_p), ("ao_width", C.c_int32), ("ao_height", C.c_int32),
                ("hdr_b10g11r11", C.c_int32), ("fog_color", C.c_float * 3), ("fog_falloff", C.c_float)]


def build(force: bool = False) -> str:
    """Compile the restatement with the committed Makefile (gcc only; no GPU, no reference sources)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_half_to_float.restype = C.c_float
        _lib.orc_half_to_float.argtypes = [C.c_uint16]
        _lib.orc_float_to_half.restype = C.c_uint16
        _lib.orc_float_to_half.argtypes = [C.c_float]
        _lib.orc_float_to_half_muglm.restype = C.c_uint16
        _lib.orc_float_to_half_muglm.argtypes = [C.c_float]
        _lib.orc_float_to_srgb8.restype = C.c_uint8
        _lib.orc_float_to_srgb8.argtypes = [C.c_float]
        _lib.orc_srgb8_to_float.restype = C.c_float
        _lib.orc_srgb8_to_float.argtypes = [C.c_uint8]
        _lib.orc_pack_lights.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _img16(a: np.ndarray):
    assert a.dtype == np.uint16 and a.ndim == 3 and a.shape[2] == 4 and a.flags.c_contiguous
    return a.shape[1], a.shape[0]


# ---- post chain ---------------------------------------------------------------------------------------------------
def level_size(w: int, h: int, scale: float):
    """InputRelative size: ceil(input_dim * scale) (render_graph.cpp:3158-3170)."""
    import math
    return max(int(math.ceil(np.float32(w) * np.float32(scale))), 1), max(int(math.ceil(np.float32(h) * np.float32(scale))), 1)


def bloom_threshold(hdr: np.ndarray, ow: int, oh: int, lum3=None) -> np.ndarray:
    iw, ih = _img16(hdr)
    out = np.zeros((oh, ow, 4), np.uint16)
    l = None if lum3 is None else np.ascontiguousarray(lum3, np.float32)
    lib().orc_bloom_threshold(_p(hdr), iw, ih, _p(out), ow, oh, _p(l))
    return out


def bloom_downsample(src: np.ndarray, ow: int, oh: int, history=None, lerp: float = 0.0) -> np.ndarray:
    iw, ih = _img16(src)
    out = np.zeros((oh, ow, 4), np.uint16)
    lib().orc_bloom_downsample(_p(src), iw, ih, _p(out), ow, oh, _p(history), C.c_float(lerp))
    return out


def bloom_upsample(src: np.ndarray, ow: int, oh: int) -> np.ndarray:
    iw, ih = _img16(src)
    out = np.zeros((oh, ow, 4), np.uint16)
    lib().orc_bloom_upsample(_p(src), iw, ih, _p(out), ow, oh)
    return out


def luminance(d3: np.ndarray, lum3: np.ndarray, lerp: float, lo: float = -3.0, hi: float = 2.0) -> np.ndarray:
    w, h = _img16(d3)
    out = np.array(lum3, np.float32, copy=True)
    lib().orc_luminance(_p(d3), w, h, _p(out), C.c_float(lerp), C.c_float(lo), C.c_float(hi))
    return out


def tonemap(hdr: np.ndarray, bloom: np.ndarray, lum3=None, dynamic_exposure: float = 1.0) -> np.ndarray:
    w, h = _img16(hdr)
    bw, bh = _img16(bloom)
    out = np.zeros((h, w, 4), np.uint8)
    l = None if lum3 is None else np.ascontiguousarray(lum3, np.float32)
    lib().orc_tonemap(_p(hdr), w, h, _p(bloom), bw, bh, _p(l), C.c_float(dynamic_exposure), _p(out))
    return out


def float_to_srgb8(values: np.ndarray) -> np.ndarray:
    """Linear fp32 -> sRGB8 bytes as an *_SRGB attachment store (assets/shaders/inc/srgb.h:12-18 + UNORM8 rounding)."""
    v = np.ascontiguousarray(values, np.float32).reshape(-1)
    out = np.zeros(v.size, np.uint8)
    lib().orc_float_to_srgb8_array(_p(v), C.c_size_t(v.size), _p(out))
    return out.reshape(np.shape(values))


BLIT_FORMATS = {"rgba16f": 0, "rgba8_unorm": 1, "rgba8_srgb": 2}


def blit(src: np.ndarray, in_format: str, ow: int, oh: int, out_format: str, linear: bool) -> np.ndarray:
    """blit.frag over a full-screen quad: `src` (H x W x 4, uint16 half bits or uint8) -> oh x ow x 4 in `out_format`."""
    src = np.ascontiguousarray(src)
    assert src.dtype == (np.uint16 if in_format == "rgba16f" else np.uint8) and src.shape[2] == 4
    out = np.zeros((oh, ow, 4), np.uint16 if out_format == "rgba16f" else np.uint8)
    lib().orc_blit(_p(src), src.shape[1], src.shape[0], BLIT_FORMATS[in_format], _p(out), ow, oh, BLIT_FORMATS[out_format], int(linear))
    return out


def frame_lerps(frame_time: float):
    """(luminance lerp, bloom feedback lerp) as hdr.cpp:93,181 compute them: float(1.0 - pow(0.5|0.001, frame_time))."""
    import math
    return float(np.float32(1.0 - math.pow(0.5, frame_time))), float(np.float32(1.0 - math.pow(0.001, frame_time)))


def hdr_chain(hdr: np.ndarray, state: dict, frame_time: float = 0.01, dynamic_exposure: float = 1.0, use_lum: bool = True):
    """One frame of setup_hdr_postprocess_compute's recorded order (hdr.cpp:354-379) + tonemap (:381-399).

    state carries the cross-frame resources: 'lum' (LuminanceData, zero-initialised) and 'd3_history' (or None).
    Returns dict of every level and the tonemapped RGBA8; updates state in place."""
    w, h = _img16(hdr)
    lum_lerp, fb_lerp = frame_lerps(frame_time)
    lum = state.setdefault("lum", np.zeros(3, np.float32))
    sz = [level_size(w, h, s) for s in (0.5, 0.25, 0.125, 0.0625, 0.03125)]
    t = bloom_threshold(hdr, *sz[0], lum3=lum if use_lum else None)
    d0 = bloom_downsample(t, *sz[1])
    d1 = bloom_downsample(d0, *sz[2])
    d2 = bloom_downsample(d1, *sz[3])
    d3 = bloom_downsample(d2, *sz[4], history=state.get("d3_history"), lerp=fb_lerp)
    if use_lum:
        lum = luminance(d3, lum, lum_lerp)
        state["lum"] = lum
    u2 = bloom_upsample(d3, *sz[3])
    u1 = bloom_upsample(u2, *sz[2])
    u0 = bloom_upsample(u1, *sz[1])
    out = tonemap(hdr, u0, lum if use_lum else None, dynamic_exposure)
    state["d3_history"] = d3
    return {"threshold": t, "d0": d0, "d1": d1, "d2": d2, "d3": d3, "u2": u2, "u1": u1, "u0": u0, "lum": lum.copy(),
            "tonemapped": out}


# ---- the same chain through the REFERENCE's executed shaders (oracle/_ref/libref_shaders.so, built by oracle/ref_build) -----------
_REF = None


def reference_shader_library():
    """oracle/_ref/libref_shaders.so (the reference's own GLSL, compiled for the CPU by oracle/ref_build) or None if it was not
    built (it needs /root/reference at build time; the built library travels).  Test infrastructure, like everything under oracle/."""
    global _REF
    if _REF is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_shaders.so")
        if not os.path.exists(path):
            return None
        ref = C.CDLL(path)
        P = C.c_void_p
        ref.ref_bloom_threshold.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int, P]
        ref.ref_bloom_downsample.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int, P, C.c_float]
        ref.ref_bloom_upsample.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int]
        ref.ref_luminance.argtypes = [P, C.c_int, C.c_int, P, C.c_float, C.c_float, C.c_float]
        ref.ref_tonemap.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int, P, C.c_float, P]
        _REF = ref
    return _REF


def hdr_chain_reference_shaders(hdr: np.ndarray, state: dict, frame_time: float = 0.01, dynamic_exposure: float = 1.0):
    """hdr_chain() with every pass executed from the reference's shader text (bloom_threshold / bloom_downsample / bloom_upsample /
    luminance.comp, tonemap.frag) instead of the restatement: same order, same push constants, same state dictionary."""
    ref = reference_shader_library()
    w, h = _img16(hdr)
    lum_lerp, fb_lerp = frame_lerps(frame_time)
    lum = np.array(state.setdefault("lum", np.zeros(3, np.float32)), np.float32, copy=True)
    sz = [level_size(w, h, s) for s in (0.5, 0.25, 0.125, 0.0625, 0.03125)]

    def level(size):
        return np.zeros((size[1], size[0], 4), np.uint16)

    def ptr(a):
        return None if a is None else a.ctypes.data

    hdr = np.ascontiguousarray(hdr, np.uint16)
    t, d0, d1, d2, d3 = (level(s) for s in sz)
    ref.ref_bloom_threshold(ptr(hdr), w, h, ptr(t), *sz[0], ptr(lum))
    for src, dst, s_src, s_dst in ((t, d0, sz[0], sz[1]), (d0, d1, sz[1], sz[2]), (d1, d2, sz[2], sz[3])):
        ref.ref_bloom_downsample(ptr(src), *s_src, ptr(dst), *s_dst, None, 0.0)
    ref.ref_bloom_downsample(ptr(d2), *sz[3], ptr(d3), *sz[4], ptr(state.get("d3_history")), fb_lerp)
    ref.ref_luminance(ptr(d3), *sz[4], ptr(lum), lum_lerp, -3.0, 2.0)
    state["lum"] = lum
    u2, u1, u0 = level(sz[3]), level(sz[2]), level(sz[1])
    ref.ref_bloom_upsample(ptr(d3), *sz[4], ptr(u2), *sz[3])
    ref.ref_bloom_upsample(ptr(u2), *sz[3], ptr(u1), *sz[2])
    ref.ref_bloom_upsample(ptr(u1), *sz[2], ptr(u0), *sz[1])
    out = np.zeros((h, w, 4), np.uint8)
    ref.ref_tonemap(ptr(hdr), w, h, ptr(u0), *sz[1], ptr(lum), dynamic_exposure, ptr(out))
    state["d3_history"] = d3
    return {"threshold": t, "d0": d0, "d1": d1, "d2": d2, "d3": d3, "u2": u2, "u1": u1, "u0": u0, "lum": lum.copy(), "tonemapped": out}


# ---- lighting -------------------------------------------------------------------------------------------------------
def pack_lights(descs: np.ndarray, camera_front):
    descs = np.ascontiguousarray(descs, LIGHT_DESC_DTYPE)
    n = len(descs)
    lights = np.zeros(4096, LIGHT_INFO_DTYPE)
    model = np.zeros((4096, 3, 4), np.float32)
    type_mask = np.zeros(128, np.uint32)
    order = np.zeros(max(n, 1), np.int32)
    cf = np.ascontiguousarray(camera_front, np.float32)
    count = lib().orc_pack_lights(_p(descs), n, _p(cf), _p(lights), _p(model), _p(type_mask), _p(order))
    return count, lights, model, type_mask, order[:count]


def cluster_params(rp: np.ndarray, res_x: int, res_y: int, res_z: int, num_lights: int) -> np.ndarray:
    out = np.zeros(1, CLUSTER_PARAMS_DTYPE)
    lib().orc_cluster_params(_p(rp), res_x, res_y, res_z, num_lights, _p(out))
    return out


def light_z_ranges(rp, lights, model, type_mask, num_lights: int, res_z: int) -> np.ndarray:
    out = np.zeros((max(num_lights, 1), 2), np.uint32)
    if num_lights == 0:
        out[0] = (0xFFFFFFFF, 0)  # clusterer.cpp:1341-1342
        return out
    lib().orc_light_z_ranges(_p(rp), _p(lights), _p(model), _p(type_mask), num_lights, res_z, _p(out))
    return out


def cluster_build(rp, prm, lights, model, type_mask, num_lights: int, res_z: int, subgroup_tile_h: int = 8):
    """spot_transform -> setup -> binning -> z_range, as LightClusterer::build_cluster_bindless_gpu orders them."""
    res_x, res_y = int(prm["resolution_xy"][0][0]), int(prm["resolution_xy"][0][1])
    n32 = int(prm["num_lights_32"][0])
    spots = np.zeros((4096, 24), np.float32)
    setup = np.zeros((4096, 128), np.float32)
    bitmask = np.zeros(res_x * res_y * max(n32, 1), np.uint32)
    if num_lights > 0:
        lib().orc_cluster_spot_transform(_p(rp), _p(model), num_lights, _p(spots))
        lib().orc_cluster_setup(_p(rp), _p(prm), _p(lights), _p(type_mask), _p(spots), num_lights, _p(setup))
        lib().orc_cluster_binning(_p(prm), _p(type_mask), _p(setup), _p(bitmask), subgroup_tile_h)
    zr = light_z_ranges(rp, lights, model, type_mask, num_lights, res_z)
    ranges = np.zeros((res_z, 2), np.uint32)
    lib().orc_cluster_z_range(_p(zr), len(zr), res_z, _p(ranges))
    return {"spots": spots, "setup": setup, "bitmask": bitmask, "light_ranges": zr, "range": ranges}


def pack_b10g11r11(rgba16f: np.ndarray) -> np.ndarray:
    """RGBA16F bits (h, w, 4) -> B10G11R11_UFLOAT_PACK32 words (h, w): the attachment store conversion (round to the closest finite
    value, ties to even; negatives -> 0)."""
    src = np.ascontiguousarray(rgba16f, np.uint16)
    out = np.zeros(src.shape[:-1], np.uint32)
    lib().orc_pack_b10g11r11_from_rgba16f(_p(src), _p(out), C.c_uint64(out.size))
    return out


def unpack_b10g11r11(words: np.ndarray) -> np.ndarray:
    """B10G11R11 words (h, w) -> the same texels as RGBA16F bits (h, w, 4): exact, alpha 1."""
    src = np.ascontiguousarray(words, np.uint32)
    out = np.zeros(src.shape + (4,), np.uint16)
    lib().orc_unpack_b10g11r11_to_rgba16f(_p(src), _p(out), C.c_uint64(src.size))
    return out


def quantize_b10g11r11(rgba16f: np.ndarray) -> np.ndarray:
    """What an RGBA16F image becomes when it is stored into a B10G11R11 attachment, as RGBA16F bits again."""
    return unpack_b10g11r11(pack_b10g11r11(rgba16f))


def lighting(gbuf: dict, rp, prm, lights, type_mask, bitmask, ranges, dir_color, dir_direction, directional=True,
             clustered=True, ambient_fallback=True, wave_tile=0, bruteforce=False, ambient_occlusion=None, entry=None,
             b10g11r11=False, fog=None) -> np.ndarray:
    """Returns the HDR target (RGBA16F bits) after DeferredLightRenderer::render_light on gbuf['emissive'].
    entry: another implementation taking the same OrcLightingArgs (oracle/_ref's ref_lighting: the reference's own shaders).
    b10g11r11: the target is a B10G11R11_UFLOAT_PACK32 attachment (renderTargetFp16 = false): gbuf['emissive'] must hold packed-
    representable values (quantize_b10g11r11), both blends round to the packed format, the result is returned as RGBA16F bits."""
    h, w = gbuf["depth"].shape
    hdr = np.array(gbuf["emissive"], np.uint16, copy=True)
    a = LightingArgs()
    a.width, a.height = w, h
    keep = [np.ascontiguousarray(gbuf["albedo"], np.uint32), np.ascontiguousarray(gbuf["normal"], np.uint32),
            np.ascontiguousarray(gbuf["pbr"], np.uint16), np.ascontiguousarray(gbuf["depth"], np.float32)]
    a.albedo, a.normal, a.pbr, a.depth = [k.ctypes.data for k in keep]
    a.hdr = hdr.ctypes.data
    a.rp, a.cluster = rp.ctypes.data, prm.ctypes.data
    a.lights, a.type_mask = lights.ctypes.data, type_mask.ctypes.data
    a.bitmask, a.range = bitmask.ctypes.data, ranges.ctypes.data
    a.dir_color = (C.c_float * 3)(*[float(v) for v in dir_color])
    a.dir_direction = (C.c_float * 3)(*[float(v) for v in dir_direction])
    a.enable_directional, a.enable_clustered = int(directional), int(clustered)
    a.ambient_fallback, a.wave_tile = int(ambient_fallback), int(wave_tile)
    a.hdr_b10g11r11 = int(b10g11r11)
    if fog is not None:  # ((r, g, b), falloff): the fog quad of render_light behind the clustered quad (renderer.cpp:1179-1196)
        a.fog_color = (C.c_float * 3)(*[float(v) for v in fog[0]])
        a.fog_falloff = float(fog[1])
    if ambient_occlusion is not None:  # AMBIENT_OCCLUSION variant: R8_UNORM image of any size
        ao = np.ascontiguousarray(ambient_occlusion, np.uint8)
        keep.append(ao)
        a.ambient_occlusion, a.ao_height, a.ao_width = ao.ctypes.data, ao.shape[0], ao.shape[1]
    if entry is not None:
        entry(C.byref(a))
    elif bruteforce:
        lib().orc_lighting_bruteforce_clustered(C.byref(a))
    else:
        lib().orc_lighting(C.byref(a))
    return hdr


# ---- scalar helpers ---------------------------------------------------------------------------------------------------
def half_to_float(bits: np.ndarray) -> np.ndarray:
    return np.asarray(bits, np.uint16).view(np.float16).astype(np.float32)


# ---- anti-aliasing --------------------------------------------------------------------------------------------------------
def fxaa(rgba8: np.ndarray, target_srgb: bool = True) -> np.ndarray:
    h, w = rgba8.shape[:2]
    src = np.ascontiguousarray(rgba8, np.uint8)
    out = np.zeros((h, w, 4), np.uint8)
    lib().orc_fxaa(_p(src), w, h, _p(out), int(target_srgb))
    return out


def smaa_edges(rgba8: np.ndarray, quality: int) -> np.ndarray:
    h, w = rgba8.shape[:2]
    src = np.ascontiguousarray(rgba8, np.uint8)
    out = np.zeros((h, w, 2), np.uint8)
    lib().orc_smaa_edges(_p(src), w, h, _p(out), quality)
    return out


def smaa_weights(edges: np.ndarray, area: np.ndarray, search: np.ndarray, quality: int) -> np.ndarray:
    h, w = edges.shape[:2]
    e = np.ascontiguousarray(edges, np.uint8)
    a, s = np.ascontiguousarray(area, np.uint8), np.ascontiguousarray(search, np.uint8)
    out = np.zeros((h, w, 4), np.uint8)
    lib().orc_smaa_weights(_p(e), w, h, _p(a), _p(s), _p(out), quality)
    return out


def smaa_blend(rgba8: np.ndarray, weights: np.ndarray, target_srgb: bool = True) -> np.ndarray:
    h, w = rgba8.shape[:2]
    src, wt = np.ascontiguousarray(rgba8, np.uint8), np.ascontiguousarray(weights, np.uint8)
    out = np.zeros((h, w, 4), np.uint8)
    lib().orc_smaa_blend(_p(src), _p(wt), w, h, _p(out), int(target_srgb))
    return out


def smaa(rgba8: np.ndarray, area, search, quality: int, target_srgb: bool = True):
    e = smaa_edges(rgba8, quality)
    wt = smaa_weights(e, area, search, quality)
    return {"edges": e, "weights": wt, "out": smaa_blend(rgba8, wt, target_srgb)}


def taa_resolve(current: np.ndarray, depth: np.ndarray, mv: np.ndarray, history, reproj16, quality: int, color_b10g11r11: bool = False):
    """color_b10g11r11: the resolved colour is stored to a B10G11R11 attachment (returned as its exact RGBA16F texels)."""
    w, h = _img16(current)
    d = np.ascontiguousarray(depth, np.float32)
    m = np.ascontiguousarray(mv, np.uint16)
    r = np.ascontiguousarray(reproj16, np.float32)
    out_c = np.zeros((h, w, 4), np.uint16)
    out_h = np.zeros((h, w, 4), np.uint16)
    lib().orc_taa_resolve_fmt(_p(current), _p(d), _p(m), _p(history), w, h, _p(r), quality, _p(out_c), _p(out_h), int(color_b10g11r11))
    return out_c, out_h


# ---- depth hierarchy (renderer/post/spd.cpp:196-232 + assets/shaders/post/hiz.comp) ----------------------------------------
def hiz_layout(iw: int, ih: int, output_downsample: bool = False):
    """setup_depth_hierarchy_pass (spd.cpp:207-218): chain level-0 size, level count; plus push.resolution / push.mips."""
    ds = int(output_downsample)
    levels = max(1, int(math.floor(math.log2(max(iw, ih)))) - ds)
    cw, ch = ((iw + 63) & ~63) >> ds, ((ih + 63) & ~63) >> ds
    return {"chain_w": cw, "chain_h": ch, "levels": levels, "res_w": cw << ds, "res_h": ch << ds, "mips": levels + ds}


def hiz_z_transform(inv_projection16) -> np.ndarray:
    """mat2(inv_projection[2].zw * vec2(-1, 1), inv_projection[3].zw * vec2(-1, 1)), column-major (spd.cpp:164-165)."""
    m = np.asarray(inv_projection16, np.float32).reshape(4, 4)  # m[c] = column c
    return np.array([-m[2][2], m[2][3], -m[3][2], m[3][3]], np.float32)


def hiz(depth: np.ndarray, z_transform, output_downsample: bool = False):
    """Returns the list of chain levels (2-D float32 arrays)."""
    d = np.ascontiguousarray(depth, np.float32)
    ih, iw = d.shape
    lay = hiz_layout(iw, ih, output_downsample)
    fn = lib().orc_mip_chain_offset
    fn.restype = C.c_size_t
    total = fn(lay["chain_w"], lay["chain_h"], lay["levels"])
    out = np.zeros(total, np.float32)
    zt = np.ascontiguousarray(z_transform, np.float32)
    lib().orc_hiz(_p(d), iw, ih, lay["res_w"], lay["res_h"], lay["mips"], _p(zt), int(not output_downsample), _p(out))
    levels = []
    for l in range(lay["levels"]):
        w, h = max(lay["chain_w"] >> l, 1), max(lay["chain_h"] >> l, 1)
        o = fn(lay["chain_w"], lay["chain_h"], l)
        levels.append(out[o:o + w * h].reshape(h, w))
    return levels


# ---- single-pass downsampler (renderer/post/spd.cpp:56-102 + assets/shaders/post/ffx-spd) -------------------------------------
def spd_split(chain: np.ndarray, w0: int, h0: int, mips: int):
    """Views of the levels of a tightly packed RGBA16F chain (uint16 bits)."""
    levels, o = [], 0
    flat = chain.reshape(-1)
    for l in range(mips):
        w, h = max(w0 >> l, 1), max(h0 >> l, 1)
        levels.append(flat[o:o + w * h * 4].reshape(h, w, 4))
        o += w * h * 4
    return levels


def spd_chain_texels(w0: int, h0: int, mips: int) -> int:
    return sum(max(w0 >> l, 1) * max(h0 >> l, 1) for l in range(mips))


def spd(rgba16f_bits: np.ndarray, w0: int, h0: int, mips: int, components: int = 4, depth_mode: bool = False,
        filter_mods=None, entry=None, fill: int = 0):
    """emit_single_pass_downsample: levels of the output chain (level 0 = w0 x h0) as uint16 RGBA16F bits.  `fill` is what
    texels no workgroup reaches keep.  entry: alternative implementation with orc_spd's signature (the executed shader)."""
    src = np.ascontiguousarray(rgba16f_bits, np.uint16)
    ih, iw = src.shape[:2]
    chain = np.full(spd_chain_texels(w0, h0, mips) * 4, fill, np.uint16)
    fm = None if filter_mods is None else np.ascontiguousarray(filter_mods, np.float32).reshape(mips, 4)
    fn = entry if entry is not None else lib().orc_spd
    fn(_p(src), iw, ih, w0, h0, mips, components, int(depth_mode), None if fm is None else _p(fm), _p(chain))
    return spd_split(chain, w0, h0, mips)


# ---- spatial upscaling (renderer/post/aa.cpp:75-174 + assets/shaders/post/ffx-fsr) -----------------------------------------
def fsr_easu(rgba8: np.ndarray, ow: int, oh: int, fp16: bool = True, target_srgb: bool = False) -> np.ndarray:
    ih, iw = rgba8.shape[:2]
    src = np.ascontiguousarray(rgba8, np.uint8)
    out = np.zeros((oh, ow, 4), np.uint8)
    lib().orc_fsr_easu(_p(src), iw, ih, _p(out), ow, oh, int(fp16), int(target_srgb))
    return out


def fsr_rcas_sharpness(stops: float = 0.5) -> float:
    """FsrRcasCon (aa.cpp:64-74): the linear sharpness, exp2(-stops) in fp32."""
    return float(np.exp2(np.float32(-stops), dtype=np.float32))


def fsr_rcas(rgba8: np.ndarray, sharpness: float = None, srgb: bool = True) -> np.ndarray:
    h, w = rgba8.shape[:2]
    src = np.ascontiguousarray(rgba8, np.uint8)
    out = np.zeros((h, w, 4), np.uint8)
    s = fsr_rcas_sharpness() if sharpness is None else sharpness
    lib().orc_fsr_rcas(_p(src), w, h, _p(out), C.c_float(s), int(srgb))
    return out


# ---- HDR10 output (renderer/post/hdr.cpp:562-658 + assets/shaders/post/pq10_encode.frag) -----------------------------------
REC709_PRIMARIES = ((0.640, 0.330), (0.3, 0.6), (0.150, 0.060), (0.3127, 0.3290))  # hdr.cpp:585-589
ST2020_PRIMARIES = ((0.708, 0.292), (0.170, 0.797), (0.131, 0.046), (0.3127, 0.3290))


def xyz_matrix(primaries) -> np.ndarray:
    """compute_xyz_matrix (math/transforms.cpp:353-370) in float64, for known-answer checks (3x3, columns = primaries)."""
    def conv(xy):
        return np.array([xy[0] / xy[1], 1.0, (1.0 - xy[0] - xy[1]) / xy[1]])
    r, g, b, wp = (conv(p) for p in primaries)
    scale = np.linalg.solve(np.stack([r, g, b], axis=1), wp)
    return np.stack([r * scale[0], g * scale[1], b * scale[2]], axis=1)


def rec709_to_display(primaries=ST2020_PRIMARIES) -> np.ndarray:
    """compute_rec709_to_st2020 (hdr.cpp:580-593): column-major 9 floats."""
    m = np.linalg.inv(xyz_matrix(primaries)) @ xyz_matrix(REC709_PRIMARIES)
    return np.ascontiguousarray(m.T, np.float32).reshape(9)


def pq10_encode(hdr: np.ndarray, ui_rgba8: np.ndarray, conversion9, hdr_pre_exposure=500.0, ui_pre_exposure=400.0,
                max_light_level=1000.0) -> np.ndarray:
    w, h = _img16(hdr)
    ui = np.ascontiguousarray(ui_rgba8, np.uint8)
    m = np.ascontiguousarray(conversion9, np.float32)
    out = np.zeros((h, w), np.uint32)
    lib().orc_pq10_encode(_p(hdr), _p(ui), w, h, _p(m), C.c_float(hdr_pre_exposure), C.c_float(ui_pre_exposure),
                          C.c_float(max_light_level), _p(out))
    return out


# ---- screen-space reflections (renderer/post/ssr.cpp + assets/shaders/post/ffx-sssr) -------------------------------------------
class SSRArgs(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("hier", C.c_void_p), ("hier_w", C.c_int32), ("hier_h", C.c_int32),
                ("hier_levels", C.c_int32), ("pbr", C.c_void_p), ("normal", C.c_void_p), ("light", C.c_void_p), ("noise", C.c_void_p),
                ("frame", C.c_int32), ("view_projection", C.c_void_p), ("inv_view_projection", C.c_void_p),
                ("camera_position", C.c_float * 3), ("output", C.c_void_p), ("ray_length", C.c_void_p), ("confidence", C.c_void_p),
                ("ray_list", C.c_void_p), ("ray_counter", C.c_void_p)]


def ssr_trace(hier_levels, pbr, normal, light, noise, frame, view_projection, inv_view_projection, camera_position, entry=None):
    """classify + build_indirect + trace_primary over a depth hierarchy given as a list of float32 levels (level 0 first).
    Returns {"output" RGBA16F bits, "ray_length" R16F bits, "confidence" R8, "ray_list" uint32[count], "ray_counter" uint32[6]}.
    entry: (classify, trace) pair of another implementation over the same struct (oracle/_ref: the reference's own shaders)."""
    h, w = pbr.shape
    chain = np.ascontiguousarray(np.concatenate([np.ascontiguousarray(l, np.float32).reshape(-1) for l in hier_levels]))
    keep = [chain, np.ascontiguousarray(pbr, np.uint16), np.ascontiguousarray(normal, np.uint32), np.ascontiguousarray(light, np.uint16),
            np.ascontiguousarray(noise, np.uint16), np.ascontiguousarray(view_projection, np.float32), np.ascontiguousarray(inv_view_projection, np.float32)]
    out = {"output": np.zeros((h, w, 4), np.uint16), "ray_length": np.zeros((h, w), np.uint16), "confidence": np.zeros((h, w), np.uint8),
           "ray_list": np.zeros(w * h, np.uint32), "ray_counter": np.zeros(6, np.uint32)}
    a = SSRArgs()
    a.width, a.height = w, h
    a.hier, a.hier_w, a.hier_h, a.hier_levels = keep[0].ctypes.data, hier_levels[0].shape[1], hier_levels[0].shape[0], len(hier_levels)
    a.pbr, a.normal, a.light, a.noise = (k.ctypes.data for k in keep[1:5])
    a.frame = int(frame)
    a.view_projection, a.inv_view_projection = keep[5].ctypes.data, keep[6].ctypes.data
    a.camera_position = (C.c_float * 3)(*[float(v) for v in camera_position])
    a.output, a.ray_length, a.confidence = out["output"].ctypes.data, out["ray_length"].ctypes.data, out["confidence"].ctypes.data
    a.ray_list, a.ray_counter = out["ray_list"].ctypes.data, out["ray_counter"].ctypes.data
    classify, trace = entry if entry is not None else (lib().orc_ssr_classify, lib().orc_ssr_trace)
    classify(C.byref(a))
    trace(C.byref(a))
    out["ray_list"] = out["ray_list"][:int(out["ray_counter"][5])].copy()
    return out


def ssr_apply(hdr, reflected, albedo, normal, pbr, depth, brdf_lut, inv_view_projection, camera_position) -> np.ndarray:
    """apply.frag blended ONE / ONE into `hdr` (RGBA16F bits); returns the new target."""
    h, w = depth.shape
    out = np.array(hdr, np.uint16, copy=True)
    lut = np.ascontiguousarray(brdf_lut, np.uint16)
    ivp = np.ascontiguousarray(inv_view_projection, np.float32)
    cam = np.ascontiguousarray(camera_position, np.float32)
    lib().orc_ssr_apply(w, h, _p(np.ascontiguousarray(reflected, np.uint16)), _p(np.ascontiguousarray(albedo, np.uint32)),
                        _p(np.ascontiguousarray(normal, np.uint32)), _p(np.ascontiguousarray(pbr, np.uint16)),
                        _p(np.ascontiguousarray(depth, np.float32)), _p(lut), lut.shape[1], lut.shape[0], _p(ivp), _p(cam), _p(out))
    return out
