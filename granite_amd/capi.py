"""ctypes binding of the gfx950 executor's C ABI (include/granite_hip.h).

This is the Python-side twin of what a Granite maintainer binds from C++: plain pointers and sizes,
no torch types.  The library is built in-tree by ``__graft_entry__.build()`` / ``granite_amd/csrc/Makefile`` and must be
present: there is no CPU fallback, every entry point raises if the HIP library is missing or a call fails.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRANITE_LIB_DIR: an A/B build of the same sources (make OUT=../lib_xyz EXTRA_<unit>=...), for measurements only
LIB_PATH = os.path.join(_HERE, os.environ.get("GRANITE_LIB_DIR", "lib"), "libgranite_hip.so")

# gr_format (VkFormat numeric values)
FORMAT_R8_UNORM = 9
FORMAT_R8G8_UNORM = 16
FORMAT_R8G8B8A8_UNORM = 37
FORMAT_R8G8B8A8_SRGB = 43
FORMAT_A2B10G10R10_UNORM_PACK32 = 64
FORMAT_R16_SFLOAT = 76
FORMAT_R16G16_SFLOAT = 83
FORMAT_R16G16B16A16_SFLOAT = 97
FORMAT_R32_SFLOAT = 100
FORMAT_B10G11R11_UFLOAT_PACK32 = 122
FORMAT_D16_UNORM = 124
FORMAT_D32_SFLOAT = 126

FORMAT_BPP = {
    FORMAT_R8_UNORM: 1,
    FORMAT_R8G8_UNORM: 2,
    FORMAT_R8G8B8A8_UNORM: 4,
    FORMAT_R8G8B8A8_SRGB: 4,
    FORMAT_A2B10G10R10_UNORM_PACK32: 4,
    FORMAT_R16_SFLOAT: 2,
    FORMAT_R16G16_SFLOAT: 4,
    FORMAT_R16G16B16A16_SFLOAT: 8,
    FORMAT_R32_SFLOAT: 4,
    FORMAT_B10G11R11_UFLOAT_PACK32: 4,
    FORMAT_D16_UNORM: 2,
    FORMAT_D32_SFLOAT: 4,
}

LIGHTING_DIRECTIONAL_BIT = 1
LIGHTING_CLUSTERED_BIT = 2
LIGHTING_AMBIENT_FALLBACK_BIT = 4
LIGHTING_AMBIENT_OCCLUSION_BIT = 8
LIGHTING_SHARE_REGISTERS_BIT = 16  # scheduling hint only (include/granite_hip.h)

MAX_LIGHTS_BINDLESS = 4096
CULL_SETUP_BYTES_PER_LIGHT = 512
TRANSFORMED_SPOT_BYTES_PER_LIGHT = 96
TRANSFORMS_OFFSET_LIGHTS = 0
TRANSFORMS_OFFSET_SHADOW = 196608
TRANSFORMS_OFFSET_MODEL = 458752
TRANSFORMS_OFFSET_TYPE_MASK = 655360
TRANSFORMS_OFFSET_DECALS = 655872
TRANSFORMS_SIZE = 852480


class Image(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("pitch_bytes", C.c_uint32),
                ("format", C.c_uint32)]


class TimingEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("count", C.c_uint64), ("total_ms", C.c_double)]


class PushBloomThreshold(C.Structure):
    _fields_ = [("threads", C.c_uint32 * 2), ("inv_output_size", C.c_float * 2)]


class PushBloomDownsample(C.Structure):
    _fields_ = [("threads", C.c_uint32 * 2), ("inv_output_size", C.c_float * 2), ("inv_input_size", C.c_float * 2),
                ("lerp", C.c_float)]


class PushBloomUpsample(C.Structure):
    _fields_ = [("threads", C.c_uint32 * 2), ("inv_output_size", C.c_float * 2), ("inv_input_size", C.c_float * 2)]


class PushLuminance(C.Structure):
    _fields_ = [("size", C.c_uint32 * 2), ("lerp", C.c_float), ("min_loglum", C.c_float), ("max_loglum", C.c_float)]


class BloomPyramidArgs(C.Structure):
    _fields_ = [("hdr", Image), ("threshold", Image), ("d0", Image), ("d1", Image), ("d2", Image), ("d3", Image), ("history", Image),
                ("u2", Image), ("u1", Image), ("u0", Image), ("lum", C.c_void_p), ("push_threshold", PushBloomThreshold),
                ("push_d0", PushBloomDownsample), ("push_d1", PushBloomDownsample), ("push_d2", PushBloomDownsample), ("push_d3", PushBloomDownsample),
                ("push_u2", PushBloomUpsample), ("push_u1", PushBloomUpsample), ("push_u0", PushBloomUpsample), ("push_luminance", PushLuminance)]


class PushTonemap(C.Structure):
    _fields_ = [("dynamic_exposure", C.c_float)]


class ClusterParams(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("clip_scale", C.c_float * 4), ("camera_base", C.c_float * 3),
                ("pad0", C.c_float), ("camera_front", C.c_float * 3), ("pad1", C.c_float), ("xy_scale", C.c_float * 2),
                ("resolution_xy", C.c_int32 * 2), ("inv_resolution_xy", C.c_float * 2), ("num_lights", C.c_int32),
                ("num_lights_32", C.c_int32), ("num_decals", C.c_int32), ("num_decals_32", C.c_int32),
                ("decals_texture_offset", C.c_int32), ("z_max_index", C.c_int32), ("z_scale", C.c_float),
                ("pad2", C.c_float * 3)]


assert C.sizeof(ClusterParams) == 176


class PushSpotTransform(C.Structure):
    _fields_ = [("vp", C.c_float * 16), ("camera_pos", C.c_float * 3), ("num_lights", C.c_uint32),
                ("camera_front", C.c_float * 3), ("z_near", C.c_float), ("z_far", C.c_float)]


assert C.sizeof(PushSpotTransform) == 100


class PushClusterSetup(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("num_lights", C.c_uint32)]


class PushZRange(C.Structure):
    _fields_ = [("num_volumes", C.c_uint32), ("num_volumes_128", C.c_uint32), ("num_ranges", C.c_uint32)]


class ClusterFrontArgs(C.Structure):
    """gr_cluster_front_args (include/granite_hip.h): uploads + spot_transform + setup + z_range as one launch."""
    _fields_ = [("transforms", C.c_void_p), ("src_lights", C.c_void_p), ("src_models", C.c_void_p), ("src_type_mask", C.c_void_p),
                ("transformed_spots", C.c_void_p), ("cull_setup", C.c_void_p), ("params", C.c_void_p), ("spot_push", C.c_void_p),
                ("setup_push", C.c_void_p), ("src_ranges", C.c_void_p), ("light_ranges", C.c_void_p), ("range_out", C.c_void_p),
                ("z_push", C.c_void_p)]


class PushDirectional(C.Structure):
    _fields_ = [("inv_view_proj_col2", C.c_float * 4), ("color", C.c_float * 3), ("environment_intensity", C.c_float),
                ("camera_pos", C.c_float * 3), ("environment_mipscale", C.c_float), ("direction", C.c_float * 3),
                ("cascade_log_bias", C.c_float), ("camera_front", C.c_float * 3), ("pad0", C.c_float),
                ("inv_resolution", C.c_float * 2), ("pad1", C.c_float * 2)]


class PushClustering(C.Structure):
    _fields_ = [("inv_view_proj_col2", C.c_float * 4), ("camera_pos", C.c_float * 3), ("pad0", C.c_float),
                ("inv_resolution", C.c_float * 2), ("pad1", C.c_float * 2)]


class LightingArgs(C.Structure):
    _fields_ = [("albedo", Image), ("normal", Image), ("pbr", Image), ("depth", Image), ("emissive", Image),
                ("hdr", Image),
                ("inv_view_projection", C.c_float * 16), ("directional", PushDirectional), ("clustering", PushClustering),
                ("cluster", ClusterParams), ("transforms", C.c_void_p), ("bitmask", C.c_void_p), ("range", C.c_void_p),
                ("flags", C.c_uint32), ("rows", C.c_uint32 * 2), ("ambient_occlusion", Image),
                ("fog_color", C.c_float * 3), ("fog_falloff", C.c_float)]


class Rows(C.Structure):
    """gr_rows: render area of one launch, output rows [first, first + count).  Passed by pointer: None (NULL) = the whole image,
    count == 0 = NO rows (the launcher returns GR_OK without launching: an empty band must not touch the image).  The one embedded
    use, LightingArgs.rows, keeps {0, 0} = whole target, as a zero-initialised argument struct has it (include/granite_hip.h)."""
    _fields_ = [("first", C.c_uint32), ("count", C.c_uint32)]


class UploadRange(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src_pinned", C.c_void_p), ("bytes", C.c_size_t)]


class PushFxaa(C.Structure):
    _fields_ = [("inv_resolution", C.c_float * 2)]


class PushSmaa(C.Structure):
    _fields_ = [("rt_metrics", C.c_float * 4)]


class PushTaa(C.Structure):
    _fields_ = [("reproj", C.c_float * 16), ("rt_metrics", C.c_float * 4)]


assert C.sizeof(PushTaa) == 80


class HizArgs(C.Structure):
    _fields_ = [("depth", Image), ("chain", C.c_void_p), ("chain_width", C.c_uint32), ("chain_height", C.c_uint32),
                ("chain_levels", C.c_uint32), ("output_downsample", C.c_uint32), ("z_transform", C.c_float * 4),
                ("counter", C.c_void_p)]


class SsrArgs(C.Structure):
    _fields_ = [("depth_chain", C.c_void_p), ("chain_width", C.c_uint32), ("chain_height", C.c_uint32), ("chain_levels", C.c_uint32),
                ("pbr", Image), ("normal", Image), ("light", Image), ("dither_lut", C.c_void_p), ("frame", C.c_uint32),
                ("view_projection", C.c_float * 16), ("inv_view_projection", C.c_float * 16), ("camera_position", C.c_float * 3),
                ("output", Image), ("ray_length", Image), ("ray_confidence", Image), ("ray_list", C.c_void_p), ("ray_counter", C.c_void_p),
                ("scratch", C.c_void_p)]


class SsrApplyArgs(C.Structure):
    _fields_ = [("hdr", Image), ("reflected", Image), ("albedo", Image), ("normal", Image), ("pbr", Image), ("depth", Image),
                ("brdf_lut", Image), ("inv_view_projection", C.c_float * 16), ("camera_position", C.c_float * 3)]


class SpdArgs(C.Structure):
    _fields_ = [("input", Image), ("chain", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("mips", C.c_uint32),
                ("components", C.c_uint32), ("reduction_mode", C.c_uint32), ("filter_mods", C.c_void_p)]


SPD_REDUCTION_COLOR, SPD_REDUCTION_DEPTH = 0, 1


class PushPq10(C.Structure):
    _fields_ = [("primary_conversion", C.c_float * 16), ("hdr_pre_exposure", C.c_float), ("ui_pre_exposure", C.c_float),
                ("max_light_level", C.c_float), ("inv_max_light_level", C.c_float)]


class GraniteHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """dlopen the in-tree libgranite_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GraniteHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp = C.c_void_p
    sigs = {
        "gr_abi_version": (C.c_int, []),
        "gr_create": (vp, [C.c_int]),
        "gr_destroy": (None, [vp]),
        "gr_last_error": (C.c_char_p, [vp]),
        "gr_sync": (C.c_int, [vp, vp]),
        "gr_alloc": (C.c_int, [vp, C.c_size_t, P(vp)]),
        "gr_free": (C.c_int, [vp, vp]),
        "gr_upload": (C.c_int, [vp, vp, vp, vp, C.c_size_t]),
        "gr_download": (C.c_int, [vp, vp, vp, vp, C.c_size_t]),
        "gr_copy": (C.c_int, [vp, vp, vp, vp, C.c_size_t]),
        "gr_fill_zero": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "gr_timing_enable": (C.c_int, [vp, C.c_int]),
        "gr_timing_set_filter": (C.c_int, [vp, C.c_char_p]),
        "gr_timing_set_sampling": (C.c_int, [vp, C.c_uint32]),
        "gr_timing_max_ms": (C.c_int, [vp, C.c_char_p, vp]),
        "gr_timing_span_begin": (C.c_int, [vp, vp, C.c_char_p, vp]),
        "gr_timing_span_end": (C.c_int, [vp, vp, vp]),
        "gr_bandwidth_probe": (C.c_int, [vp, C.c_size_t, C.c_int, P(C.c_double), P(C.c_double)]),
        "gr_timing_reset": (C.c_int, [vp]),
        "gr_timing_query": (C.c_int, [vp, P(TimingEntry), C.c_int]),
        "gr_bloom_threshold": (C.c_int, [vp, vp, P(Image), P(Image), vp, P(PushBloomThreshold)]),
        "gr_bloom_downsample": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), P(PushBloomDownsample)]),
        "gr_bloom_upsample": (C.c_int, [vp, vp, P(Image), P(Image), P(PushBloomUpsample)]),
        "gr_luminance": (C.c_int, [vp, vp, P(Image), vp, P(PushLuminance)]),
        "gr_ssr_scratch_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32]),
        "gr_ssr_trace": (C.c_int, [vp, vp, P(SsrArgs)]),
        "gr_ssr_apply": (C.c_int, [vp, vp, P(SsrApplyArgs)]),
        "gr_bloom_tail_supported": (C.c_int, [P(Image), P(Image), P(Image), P(Image), P(Image), P(PushBloomDownsample), P(PushBloomDownsample),
                                              P(PushBloomUpsample), P(PushBloomUpsample)]),
        "gr_bloom_down_tail": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), P(Image), P(PushBloomDownsample), P(PushBloomDownsample)]),
        "gr_bloom_down_mid_supported": (C.c_int, [P(Image), P(Image), P(Image), P(PushBloomDownsample), P(PushBloomDownsample)]),
        "gr_bloom_down_mid": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), P(PushBloomDownsample), P(PushBloomDownsample), P(Rows)]),
        "gr_bloom_down_head_supported": (C.c_int, [P(Image), P(Image), P(Image), P(Image), P(PushBloomThreshold), P(PushBloomDownsample), P(PushBloomDownsample)]),
        "gr_bloom_down_head": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), P(Image), vp, P(PushBloomThreshold), P(PushBloomDownsample), P(PushBloomDownsample)]),
        "gr_bloom_pyramid_supported": (C.c_int, [P(BloomPyramidArgs)]),
        "gr_bloom_pyramid": (C.c_int, [vp, vp, P(BloomPyramidArgs)]),
        "gr_debug_pyramid_giveups": (C.c_int, [vp, P(C.c_uint32)]),
        "gr_bloom_up_all_supported": (C.c_int, [P(Image), P(Image), P(Image), P(Image), P(PushBloomUpsample), P(PushBloomUpsample), P(PushBloomUpsample)]),
        "gr_bloom_up_all": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), P(Image), vp, P(PushBloomUpsample), P(PushBloomUpsample), P(PushBloomUpsample),
                                      P(PushLuminance), C.c_uint32]),
        "gr_bloom_up_tail": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), vp, P(PushBloomUpsample), P(PushBloomUpsample), P(PushLuminance)]),
        "gr_tonemap": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), vp, P(PushTonemap)]),
        "gr_bloom_threshold_rows": (C.c_int, [vp, vp, P(Image), P(Image), vp, P(PushBloomThreshold), P(Rows)]),
        "gr_bloom_downsample_rows": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), P(PushBloomDownsample), P(Rows)]),
        "gr_bloom_upsample_rows": (C.c_int, [vp, vp, P(Image), P(Image), P(PushBloomUpsample), P(Rows)]),
        "gr_tonemap_rows": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), vp, P(PushTonemap), P(Rows)]),
        "gr_upload_batch": (C.c_int, [vp, vp, P(UploadRange), C.c_uint32]),
        "gr_alloc_host": (C.c_int, [vp, C.c_size_t, P(vp)]),
        "gr_free_host": (C.c_int, [vp, vp]),
        "gr_cluster_spot_transform": (C.c_int, [vp, vp, vp, vp, P(PushSpotTransform)]),
        "gr_cluster_setup": (C.c_int, [vp, vp, vp, vp, vp, P(ClusterParams), P(PushClusterSetup)]),
        "gr_cluster_binning": (C.c_int, [vp, vp, vp, vp, vp, P(ClusterParams)]),
        "gr_cluster_z_range": (C.c_int, [vp, vp, vp, vp, P(PushZRange)]),
        "gr_cluster_front": (C.c_int, [vp, vp, P(ClusterFrontArgs)]),
        "gr_alloc_host": (C.c_int, [vp, C.c_size_t, P(vp)]),
        "gr_free_host": (C.c_int, [vp, vp]),
        "gr_lighting": (C.c_int, [vp, vp, P(LightingArgs)]),
        "gr_smaa_set_luts": (C.c_int, [vp, vp, vp]),
        "gr_fxaa": (C.c_int, [vp, vp, P(Image), P(Image), P(PushFxaa)]),
        "gr_blit": (C.c_int, [vp, vp, P(Image), P(Image), C.c_int]),
        "gr_smaa_edge_detection": (C.c_int, [vp, vp, P(Image), P(Image), P(PushSmaa), C.c_int]),
        "gr_smaa_blend_weight": (C.c_int, [vp, vp, P(Image), P(Image), P(PushSmaa), C.c_int]),
        "gr_smaa_neighbor_blend": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), P(PushSmaa)]),
        "gr_taa_resolve": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), P(Image), P(Image), P(Image), P(PushTaa), C.c_int]),
        "gr_hiz": (C.c_int, [vp, vp, P(HizArgs)]),
        "gr_fill_byte": (C.c_int, [vp, vp, vp, C.c_int, C.c_size_t]),
        "gr_fill_u32": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_size_t]),
        "gr_get_device_info": (C.c_int, [vp, C.c_char_p, C.c_size_t, vp]),
        "gr_spd_downsample": (C.c_int, [vp, vp, P(SpdArgs)]),
        "gr_debug_mix": (C.c_int, [vp, vp, vp, C.c_size_t, vp, vp, C.c_uint32, C.c_uint32]),
        "gr_pack_b10g11r11": (C.c_int, [vp, vp, vp, vp, C.c_uint32]),
        "gr_pq10_encode": (C.c_int, [vp, vp, P(Image), P(Image), P(Image), P(PushPq10)]),
        "gr_fsr_upscale": (C.c_int, [vp, vp, P(Image), P(Image), C.c_int]),
        "gr_fsr_sharpen": (C.c_int, [vp, vp, P(Image), P(Image), C.c_float]),
        "gr_mip_chain_offset": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
        "gr_mip_chain_size": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "gr_abi_version", "gr_create", "gr_destroy", "gr_last_error", "gr_sync", "gr_alloc", "gr_free", "gr_upload",
    "gr_download", "gr_copy", "gr_fill_zero", "gr_upload_batch", "gr_alloc_host", "gr_free_host", "gr_timing_enable", "gr_timing_set_filter", "gr_timing_reset", "gr_timing_query",
    "gr_bloom_threshold", "gr_bloom_downsample", "gr_bloom_upsample", "gr_luminance", "gr_tonemap",
    "gr_bloom_threshold_rows", "gr_bloom_downsample_rows", "gr_bloom_upsample_rows", "gr_tonemap_rows",
    "gr_cluster_spot_transform", "gr_cluster_setup", "gr_cluster_binning", "gr_cluster_z_range", "gr_cluster_front", "gr_lighting",
    "gr_smaa_set_luts", "gr_fxaa", "gr_blit", "gr_smaa_edge_detection", "gr_smaa_blend_weight", "gr_smaa_neighbor_blend", "gr_taa_resolve",
    "gr_hiz", "gr_mip_chain_offset", "gr_mip_chain_size", "gr_fsr_upscale", "gr_fsr_sharpen", "gr_fill_byte", "gr_fill_u32", "gr_pq10_encode", "gr_get_device_info", "gr_spd_downsample", "gr_debug_mix", "gr_pack_b10g11r11",
]


class DeviceBuffer:
    """A zero-initialised HBM allocation owned through gr_alloc/gr_free."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        ctx.check(ctx.lib.gr_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, array: np.ndarray, offset: int = 0) -> "DeviceBuffer":
        a = np.ascontiguousarray(array)
        assert offset + a.nbytes <= self.nbytes, (offset, a.nbytes, self.nbytes)
        self.ctx.check(self.ctx.lib.gr_upload(self.ctx.handle, None, self.ptr + offset, a.ctypes.data, a.nbytes))
        self.ctx.sync()
        return self

    def download(self, dtype=np.uint8, count: Optional[int] = None, offset: int = 0) -> np.ndarray:
        dt = np.dtype(dtype)
        n = (self.nbytes - offset) // dt.itemsize if count is None else count
        out = np.empty(n, dtype=dt)
        self.ctx.check(self.ctx.lib.gr_download(self.ctx.handle, None, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.gr_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceImage:
    """A linear row-major attachment in HBM (tight pitch) + its gr_image descriptor."""

    def __init__(self, ctx: "Context", width: int, height: int, fmt: int, ptr: Optional[int] = None):
        self.ctx = ctx
        self.width, self.height, self.format = int(width), int(height), int(fmt)
        self.bpp = FORMAT_BPP[fmt]
        self.pitch = self.width * self.bpp
        self.buffer = None
        if ptr is None:
            self.buffer = DeviceBuffer(ctx, self.pitch * self.height)
            ptr = self.buffer.ptr
        self.ptr = ptr
        self.desc = Image(ptr, self.width, self.height, self.pitch, self.format)

    def upload(self, array: np.ndarray) -> "DeviceImage":
        a = np.ascontiguousarray(array)
        assert a.nbytes == self.pitch * self.height, (a.shape, a.dtype, self.width, self.height, self.bpp)
        self.ctx.check(self.ctx.lib.gr_upload(self.ctx.handle, None, self.ptr, a.ctypes.data, a.nbytes))
        self.ctx.sync()
        return self

    def download(self) -> np.ndarray:
        out = np.empty(self.pitch * self.height, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.gr_download(self.ctx.handle, None, out.ctypes.data, self.ptr, out.nbytes))
        if self.format == FORMAT_R16G16B16A16_SFLOAT:
            return out.view(np.uint16).reshape(self.height, self.width, 4)
        if self.format in (FORMAT_R8G8B8A8_SRGB, FORMAT_R8G8B8A8_UNORM):
            return out.reshape(self.height, self.width, 4)
        if self.format == FORMAT_R8G8_UNORM:
            return out.reshape(self.height, self.width, 2)
        if self.format in (FORMAT_D32_SFLOAT, FORMAT_R32_SFLOAT):
            return out.view(np.float32).reshape(self.height, self.width)
        if self.format in (FORMAT_A2B10G10R10_UNORM_PACK32, FORMAT_B10G11R11_UFLOAT_PACK32):
            return out.view(np.uint32).reshape(self.height, self.width)
        if self.format == FORMAT_R16G16_SFLOAT:
            return out.view(np.uint16).reshape(self.height, self.width, 2)
        if self.format == FORMAT_R16_SFLOAT:
            return out.view(np.uint16).reshape(self.height, self.width)
        return out.reshape(self.height, self.pitch)


class Context:
    """gr_ctx wrapper. `stream` arguments are raw hipStream_t handles (ints) or None for the default stream."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.handle = self.lib.gr_create(device)
        if not self.handle:
            raise GraniteHipError(f"gr_create({device}) failed: no usable HIP device")

    def check(self, code: int):
        if code < 0:
            raise GraniteHipError(f"[{code}] {self.lib.gr_last_error(self.handle).decode()}")
        return code

    def sync(self, stream=None):
        self.check(self.lib.gr_sync(self.handle, stream))

    def close(self):
        if self.handle:
            self.lib.gr_destroy(self.handle)
            self.handle = None

    # ---- timing -----------------------------------------------------------------------------------------------
    def timing_enable(self, enable: bool):
        self.check(self.lib.gr_timing_enable(self.handle, int(enable)))

    def timing_max_ms(self, name: str) -> float:
        """Longest single bracket recorded under `name` since the last reset."""
        out = C.c_double(0.0)
        self.check(self.lib.gr_timing_max_ms(self.handle, name.encode(), C.byref(out)))
        return float(out.value)

    def timing_reset(self):
        self.check(self.lib.gr_timing_reset(self.handle))

    def timing_query(self):
        arr = (TimingEntry * 64)()
        n = self.check(self.lib.gr_timing_query(self.handle, arr, 64))
        return {arr[i].name.decode(): (int(arr[i].count), float(arr[i].total_ms)) for i in range(n)}

    # ---- post chain -------------------------------------------------------------------------------------------
    @staticmethod
    def _rows(rows):
        return None if rows is None else C.byref(Rows(int(rows[0]), int(rows[1])))

    def bloom_threshold(self, hdr: DeviceImage, out: DeviceImage, lum_ptr=None, stream=None, rows=None):
        push = PushBloomThreshold((out.width, out.height), (1.0 / out.width, 1.0 / out.height))
        self.check(self.lib.gr_bloom_threshold_rows(self.handle, stream, hdr.desc, out.desc, lum_ptr, push, self._rows(rows)))

    def bloom_downsample(self, src: DeviceImage, out: DeviceImage, history: Optional[DeviceImage] = None, lerp: float = 0.0,
                         stream=None, rows=None):
        push = PushBloomDownsample((out.width, out.height), (1.0 / out.width, 1.0 / out.height),
                                   (1.0 / src.width, 1.0 / src.height), lerp)
        self.check(self.lib.gr_bloom_downsample_rows(self.handle, stream, src.desc, out.desc,
                                                     history.desc if history is not None else None, push, self._rows(rows)))

    def bloom_upsample(self, src: DeviceImage, out: DeviceImage, stream=None, rows=None):
        push = PushBloomUpsample((out.width, out.height), (1.0 / out.width, 1.0 / out.height),
                                 (1.0 / src.width, 1.0 / src.height))
        self.check(self.lib.gr_bloom_upsample_rows(self.handle, stream, src.desc, out.desc, push, self._rows(rows)))

    def bloom_tail(self, d1: DeviceImage, d2: DeviceImage, d3: DeviceImage, history: DeviceImage, u2: DeviceImage, u1: DeviceImage,
                   feedback_lerp: float, lum_ptr=None, lum_lerp: float = 0.0, stream=None) -> bool:
        """downsample-2, downsample-3 (+ feedback), luminance, upsample-2, upsample-1 as the two fused launches; False (nothing
        launched) when the pyramid does not qualify."""
        def down(out, src):
            return PushBloomDownsample((out.width, out.height), (1.0 / out.width, 1.0 / out.height), (1.0 / src.width, 1.0 / src.height), feedback_lerp)

        def up(out, src):
            return PushBloomUpsample((out.width, out.height), (1.0 / out.width, 1.0 / out.height), (1.0 / src.width, 1.0 / src.height))
        p_d2, p_d3, p_u2, p_u1 = down(d2, d1), down(d3, d2), up(u2, d3), up(u1, u2)
        if not self.lib.gr_bloom_tail_supported(d1.desc, d2.desc, d3.desc, u2.desc, u1.desc, p_d2, p_d3, p_u2, p_u1):
            return False
        self.check(self.lib.gr_bloom_down_tail(self.handle, stream, d1.desc, d2.desc, d3.desc, history.desc, p_d2, p_d3))
        p_lum = PushLuminance((d3.width // 2, d3.height // 2), lum_lerp, -3.0, 2.0) if lum_ptr is not None else None
        self.check(self.lib.gr_bloom_up_tail(self.handle, stream, d3.desc, u2.desc, u1.desc, lum_ptr, p_u2, p_u1, p_lum))
        return True

    def bloom_down_mid(self, threshold: DeviceImage, d0: DeviceImage, d1: DeviceImage, stream=None, rows=None) -> bool:
        """downsample-0 and downsample-1 as one launch (rows restricts downsample-1); False (nothing launched) when the levels do not qualify."""
        def down(out, src):
            return PushBloomDownsample((out.width, out.height), (1.0 / out.width, 1.0 / out.height), (1.0 / src.width, 1.0 / src.height), 0.0)
        p_d0, p_d1 = down(d0, threshold), down(d1, d0)
        if not self.lib.gr_bloom_down_mid_supported(threshold.desc, d0.desc, d1.desc, p_d0, p_d1):
            return False
        self.check(self.lib.gr_bloom_down_mid(self.handle, stream, threshold.desc, d0.desc, d1.desc, p_d0, p_d1, self._rows(rows)))
        return True

    def bloom_up_all(self, d3: DeviceImage, u2: DeviceImage, u1: DeviceImage, u0: DeviceImage, lum_ptr=None, lum_lerp: float = 0.0, stream=None,
                     busy_frame: bool = False) -> bool:
        """luminance, upsample-2, upsample-1 and upsample-0 as one launch; False (nothing launched) when the frame does not qualify."""
        def up(out, src):
            return PushBloomUpsample((out.width, out.height), (1.0 / out.width, 1.0 / out.height), (1.0 / src.width, 1.0 / src.height))
        p_u2, p_u1, p_u0 = up(u2, d3), up(u1, u2), up(u0, u1)
        if not self.lib.gr_bloom_up_all_supported(d3.desc, u2.desc, u1.desc, u0.desc, p_u2, p_u1, p_u0):
            return False
        p_lum = PushLuminance((d3.width // 2, d3.height // 2), lum_lerp, -3.0, 2.0) if lum_ptr is not None else None
        self.check(self.lib.gr_bloom_up_all(self.handle, stream, d3.desc, u2.desc, u1.desc, u0.desc, lum_ptr, p_u2, p_u1, p_u0, p_lum, 1 if busy_frame else 0))
        return True

    def bloom_pyramid(self, hdr: DeviceImage, levels: dict, history: DeviceImage, feedback_lerp: float, lum_ptr=None, lum_lerp: float = 0.0, stream=None,
                      any_size: bool = False) -> bool:
        """The whole bloom pass as ONE launch (levels: threshold, d0..d3, u2..u0 by name); False (nothing launched) when the frame does not qualify.
        any_size: launch without asking gr_bloom_pyramid_supported (which offers the launch up to 640 x 384 frames; the launcher checks the rest itself)."""
        def down(out, src, lerp=0.0):
            return PushBloomDownsample((out.width, out.height), (1.0 / out.width, 1.0 / out.height), (1.0 / src.width, 1.0 / src.height), lerp)

        def up(out, src):
            return PushBloomUpsample((out.width, out.height), (1.0 / out.width, 1.0 / out.height), (1.0 / src.width, 1.0 / src.height))
        l = levels
        a = BloomPyramidArgs()
        a.hdr, a.history = hdr.desc, history.desc
        for name in ("threshold", "d0", "d1", "d2", "d3", "u2", "u1", "u0"):
            setattr(a, name, l[name].desc)
        a.lum = lum_ptr
        a.push_threshold = PushBloomThreshold((l["threshold"].width, l["threshold"].height), (1.0 / l["threshold"].width, 1.0 / l["threshold"].height))
        a.push_d0, a.push_d1 = down(l["d0"], l["threshold"]), down(l["d1"], l["d0"])
        a.push_d2, a.push_d3 = down(l["d2"], l["d1"], feedback_lerp), down(l["d3"], l["d2"], feedback_lerp)
        a.push_u2, a.push_u1, a.push_u0 = up(l["u2"], l["d3"]), up(l["u1"], l["u2"]), up(l["u0"], l["u1"])
        if lum_ptr is not None:
            a.push_luminance = PushLuminance((l["d3"].width // 2, l["d3"].height // 2), lum_lerp, -3.0, 2.0)
        if not any_size and not self.lib.gr_bloom_pyramid_supported(a):
            return False
        self.check(self.lib.gr_bloom_pyramid(self.handle, stream, a))
        return True

    def pyramid_giveups(self) -> int:
        n = C.c_uint32(0)
        self.check(self.lib.gr_debug_pyramid_giveups(self.handle, C.byref(n)))
        return int(n.value)

    def bloom_down_head(self, hdr: DeviceImage, threshold: DeviceImage, d0: DeviceImage, d1: DeviceImage, lum_ptr=None, stream=None) -> bool:
        """threshold, downsample-0 and downsample-1 as one launch; False (nothing launched) when the frame does not qualify."""
        def down(out, src):
            return PushBloomDownsample((out.width, out.height), (1.0 / out.width, 1.0 / out.height), (1.0 / src.width, 1.0 / src.height), 0.0)
        p_t = PushBloomThreshold((threshold.width, threshold.height), (1.0 / threshold.width, 1.0 / threshold.height))
        p_d0, p_d1 = down(d0, threshold), down(d1, d0)
        if not self.lib.gr_bloom_down_head_supported(hdr.desc, threshold.desc, d0.desc, d1.desc, p_t, p_d0, p_d1):
            return False
        self.check(self.lib.gr_bloom_down_head(self.handle, stream, hdr.desc, threshold.desc, d0.desc, d1.desc, lum_ptr, p_t, p_d0, p_d1))
        return True

    def luminance(self, d3: DeviceImage, lum_ptr, lerp: float, min_loglum: float = -3.0, max_loglum: float = 2.0, stream=None):
        push = PushLuminance((d3.width // 2, d3.height // 2), lerp, min_loglum, max_loglum)
        self.check(self.lib.gr_luminance(self.handle, stream, d3.desc, lum_ptr, push))

    def tonemap(self, hdr: DeviceImage, bloom: DeviceImage, out: DeviceImage, lum_ptr=None, dynamic_exposure: float = 1.0,
                stream=None, rows=None):
        push = PushTonemap(dynamic_exposure)
        self.check(self.lib.gr_tonemap_rows(self.handle, stream, hdr.desc, bloom.desc, out.desc, lum_ptr, push, self._rows(rows)))


    # ---- anti-aliasing --------------------------------------------------------------------------------------------
    def fxaa(self, src: DeviceImage, out: DeviceImage, stream=None):
        push = PushFxaa((1.0 / src.width, 1.0 / src.height))
        self.check(self.lib.gr_fxaa(self.handle, stream, src.desc, out.desc, push))

    def smaa_set_luts(self, area: np.ndarray, search: np.ndarray):
        a, s = np.ascontiguousarray(area, np.uint8), np.ascontiguousarray(search, np.uint8)
        assert a.size == 160 * 560 * 2 and s.size == 64 * 16
        self.check(self.lib.gr_smaa_set_luts(self.handle, a.ctypes.data, s.ctypes.data))

    @staticmethod
    def _smaa_push(img: DeviceImage) -> PushSmaa:
        return PushSmaa((1.0 / img.width, 1.0 / img.height, float(img.width), float(img.height)))

    def smaa_edge_detection(self, color: DeviceImage, edges: DeviceImage, quality: int, stream=None):
        self.check(self.lib.gr_smaa_edge_detection(self.handle, stream, color.desc, edges.desc, self._smaa_push(color), quality))

    def smaa_blend_weight(self, edges: DeviceImage, weights: DeviceImage, quality: int, stream=None):
        self.check(self.lib.gr_smaa_blend_weight(self.handle, stream, edges.desc, weights.desc, self._smaa_push(edges), quality))

    def smaa_neighbor_blend(self, color: DeviceImage, weights: DeviceImage, out: DeviceImage, stream=None):
        self.check(self.lib.gr_smaa_neighbor_blend(self.handle, stream, color.desc, weights.desc, out.desc, self._smaa_push(color)))

    def taa_resolve(self, current: DeviceImage, depth: DeviceImage, mv: DeviceImage, history, out_color: DeviceImage,
                    out_history: DeviceImage, reproj16, quality: int, stream=None):
        push = PushTaa()
        push.reproj[:] = [float(v) for v in reproj16]
        push.rt_metrics[:] = (1.0 / current.width, 1.0 / current.height, float(current.width), float(current.height))
        self.check(self.lib.gr_taa_resolve(self.handle, stream, current.desc, depth.desc, mv.desc,
                                           history.desc if history is not None else None, out_color.desc, out_history.desc, push,
                                           quality))

    def blit(self, src: DeviceImage, out: DeviceImage, linear: bool, stream=None):
        self.check(self.lib.gr_blit(self.handle, stream, src.desc, out.desc, int(linear)))

    def fsr_upscale(self, src: DeviceImage, out: DeviceImage, fp16: bool = True, stream=None):
        self.check(self.lib.gr_fsr_upscale(self.handle, stream, src.desc, out.desc, int(fp16)))

    def fsr_sharpen(self, src: DeviceImage, out: DeviceImage, sharpness: float, stream=None):
        self.check(self.lib.gr_fsr_sharpen(self.handle, stream, src.desc, out.desc, C.c_float(sharpness)))

    def pq10_encode(self, hdr: DeviceImage, ui: DeviceImage, out: DeviceImage, conversion9, hdr_pre_exposure=500.0, ui_pre_exposure=400.0,
                    max_light_level=1000.0, stream=None):
        push = PushPq10()
        m = [float(v) for v in conversion9]
        for col in range(3):
            for row in range(3):
                push.primary_conversion[4 * col + row] = m[3 * col + row]
        push.primary_conversion[15] = 1.0
        push.hdr_pre_exposure, push.ui_pre_exposure = hdr_pre_exposure, ui_pre_exposure
        push.max_light_level, push.inv_max_light_level = max_light_level, float(np.float32(1.0) / np.float32(max_light_level))
        self.check(self.lib.gr_pq10_encode(self.handle, stream, hdr.desc, ui.desc, out.desc, push))

    def hiz(self, depth: DeviceImage, z_transform, output_downsample: bool = False, chain: Optional[DeviceBuffer] = None,
            counter: Optional[DeviceBuffer] = None, stream=None):
        """Depth hierarchy of `depth` sized as setup_depth_hierarchy_pass sizes it (spd.cpp:207-218).  Returns
        (chain buffer, counter buffer, layout dict); chain / counter can be passed back in to reuse them."""
        ds = int(output_downsample)
        levels = max(1, max(depth.width, depth.height).bit_length() - 1 - ds)
        cw, ch = ((depth.width + 63) & ~63) >> ds, ((depth.height + 63) & ~63) >> ds
        if chain is None:
            chain = DeviceBuffer(self, self.lib.gr_mip_chain_size(cw, ch, 4, levels))
        if counter is None:
            counter = DeviceBuffer(self, 4)
        args = HizArgs()
        args.depth = depth.desc
        args.chain, args.chain_width, args.chain_height, args.chain_levels = chain.ptr, cw, ch, levels
        args.output_downsample = ds
        args.z_transform[:] = [float(v) for v in z_transform]
        args.counter = counter.ptr
        self.check(self.lib.gr_hiz(self.handle, stream, args))
        return chain, counter, {"chain_w": cw, "chain_h": ch, "levels": levels}

    def ssr_trace(self, chain: DeviceBuffer, layout: dict, pbr: DeviceImage, normal: DeviceImage, light: DeviceImage, dither: DeviceBuffer,
                  frame: int, view_projection, inv_view_projection, camera_position, stream=None) -> dict:
        """classify + build_indirect + trace_primary.  Returns the output / ray-length / confidence images and the ray buffers."""
        w, h = light.width, light.height
        out = {"output": DeviceImage(self, w, h, FORMAT_R16G16B16A16_SFLOAT), "ray_length": DeviceImage(self, w, h, FORMAT_R16_SFLOAT),
               "confidence": DeviceImage(self, w, h, FORMAT_R8_UNORM), "ray_list": DeviceBuffer(self, w * h * 4),
               "ray_counter": DeviceBuffer(self, 4096), "scratch": DeviceBuffer(self, self.lib.gr_ssr_scratch_bytes(w, h))}
        a = SsrArgs()
        a.depth_chain, a.chain_width, a.chain_height, a.chain_levels = chain.ptr, layout["chain_w"], layout["chain_h"], layout["levels"]
        a.pbr, a.normal, a.light = pbr.desc, normal.desc, light.desc
        a.dither_lut, a.frame = dither.ptr, frame
        a.view_projection[:] = [float(v) for v in view_projection]
        a.inv_view_projection[:] = [float(v) for v in inv_view_projection]
        a.camera_position[:] = [float(v) for v in camera_position]
        a.output, a.ray_length, a.ray_confidence = out["output"].desc, out["ray_length"].desc, out["confidence"].desc
        a.ray_list, a.ray_counter, a.scratch = out["ray_list"].ptr, out["ray_counter"].ptr, out["scratch"].ptr
        self.check(self.lib.gr_ssr_trace(self.handle, stream, a))
        return out

    def ssr_apply(self, hdr: DeviceImage, reflected: DeviceImage, albedo: DeviceImage, normal: DeviceImage, pbr: DeviceImage, depth: DeviceImage,
                  brdf_lut: DeviceImage, inv_view_projection, camera_position, stream=None):
        a = SsrApplyArgs()
        a.hdr, a.reflected, a.albedo, a.normal, a.pbr, a.depth, a.brdf_lut = (i.desc for i in (hdr, reflected, albedo, normal, pbr, depth, brdf_lut))
        a.inv_view_projection[:] = [float(v) for v in inv_view_projection]
        a.camera_position[:] = [float(v) for v in camera_position]
        self.check(self.lib.gr_ssr_apply(self.handle, stream, a))

    def spd_downsample(self, source: DeviceImage, width: int, height: int, mips: int, components: int = 4, depth_mode: bool = False,
                       filter_mods=None, chain: Optional[DeviceBuffer] = None, stream=None) -> DeviceBuffer:
        """emit_single_pass_downsample: fills an RGBA16F chain whose level 0 is width x height from `source`."""
        if chain is None:
            chain = DeviceBuffer(self, self.lib.gr_mip_chain_size(width, height, 8, mips))
        args = SpdArgs()
        args.input = source.desc
        args.chain, args.width, args.height, args.mips, args.components = chain.ptr, width, height, mips, components
        args.reduction_mode = SPD_REDUCTION_DEPTH if depth_mode else SPD_REDUCTION_COLOR
        fm = None if filter_mods is None else np.ascontiguousarray(filter_mods, np.float32).reshape(mips, 4)
        args.filter_mods = None if fm is None else fm.ctypes.data
        self.check(self.lib.gr_spd_downsample(self.handle, stream, args))
        return chain

    def read_mip_chain(self, chain: DeviceBuffer, layout: dict):
        raw = chain.download(np.float32)
        out = []
        for l in range(layout["levels"]):
            w, h = max(layout["chain_w"] >> l, 1), max(layout["chain_h"] >> l, 1)
            o = self.lib.gr_mip_chain_offset(layout["chain_w"], layout["chain_h"], 4, l) // 4
            out.append(raw[o:o + w * h].reshape(h, w))
        return out
