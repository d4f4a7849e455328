"""ctypes binding of the headless harness (include/granite_app.h): composes Granite's image-space render graph on the
HIP executor and runs frames.  One call per N frames keeps Python out of the timed loop."""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Optional

import numpy as np

from . import capi, synth

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRANITE_LIB_DIR: an A/B build of the same sources (make OUT=../lib_xyz EXTRA_<unit>=...), for measurements only
LIB_PATH = os.path.join(_HERE, os.environ.get("GRANITE_LIB_DIR", "lib"), "libgranite_host.so")

POST_AA_NONE, POST_AA_FXAA = 0, 1
POST_AA_SMAA_LOW, POST_AA_SMAA_MEDIUM, POST_AA_SMAA_HIGH, POST_AA_SMAA_ULTRA = 2, 3, 4, 5
POST_AA_TAA_LOW, POST_AA_TAA_MEDIUM, POST_AA_TAA_HIGH = 6, 7, 8


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("enable_lighting", C.c_int32),
                ("hdr_bloom", C.c_int32), ("dynamic_exposure", C.c_int32), ("compute_post", C.c_int32),
                ("post_aa", C.c_int32), ("pre_aa", C.c_int32), ("rmw_emissive", C.c_int32),
                ("cluster_res", C.c_uint32 * 3), ("frame_time", C.c_float), ("directional_color", C.c_float * 3),
                ("directional_direction", C.c_float * 3), ("enable_timestamps", C.c_int32),
                ("strip_index", C.c_uint32), ("strip_count", C.c_uint32),
                ("disable_image_aliasing", C.c_int32), ("depth_hierarchy", C.c_int32),
                ("resolution_scale", C.c_float), ("resolution_scale_sharpen", C.c_int32), ("fsr_fp32", C.c_int32),
                ("ambient_occlusion", C.c_int32), ("hdr10", C.c_int32), ("ssr", C.c_int32), ("aa_bench", C.c_int32), ("output_gather_rgba", C.c_int32),
                ("hdr_packed_float", C.c_int32), ("taa_history_reach_rows", C.c_uint32)]


# void (*gra_exchange_fn)(void *user, const char *tag, void *device_ptr, uint64_t chunk_bytes, uint32_t rank_count, void *stream)
EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p)


def install_ssr_tables():
    """Hands the SSR pass its two constant tables (granite_amd/data: blue-noise values, split-sum BRDF table)."""
    from .data import load_brdf_lut, load_sssr_noise_base
    noise, lut = np.ascontiguousarray(load_sssr_noise_base()), np.ascontiguousarray(load_brdf_lut())
    if load_library().gra_install_ssr_tables(noise.ctypes.data, lut.ctypes.data, lut.shape[1], lut.shape[0]) != 0:
        raise capi.GraniteHipError("gra_install_ssr_tables failed")


class FrameState(C.Structure):
    """gra_frame_state: the host-side state a frame inherits (checkpoint / replay)."""
    _fields_ = [("frames", C.c_uint64), ("elapsed", C.c_double), ("swapchain_index", C.c_uint32), ("jitter_phase", C.c_uint32),
                ("base_view", C.c_float * 16), ("jittered_projection", C.c_float * 16), ("view_proj", (C.c_float * 16) * 16),
                ("inv_view_proj", (C.c_float * 16) * 16), ("jittered_view_proj", (C.c_float * 16) * 16)]


class ResourceInfo(C.Structure):
    _fields_ = [("device_ptr", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32),
                ("size_bytes", C.c_uint64), ("physical_index", C.c_int32), ("levels", C.c_uint32)]


class Timestamp(C.Structure):
    _fields_ = [("tag", C.c_char * 64), ("count", C.c_uint64), ("total_ms", C.c_double)]


EXPORTED_SYMBOLS = [
    "gra_create", "gra_destroy", "gra_last_error", "gra_set_camera", "gra_set_camera_motion", "gra_set_render_parameters",
    "gra_get_render_parameters", "gra_set_lights", "gra_upload_gbuffer", "gra_render_frames", "gra_sync",
    "gra_get_resource", "gra_read_resource", "gra_get_backbuffer", "gra_read_backbuffer", "gra_get_cluster_state",
    "gra_dump_graph", "gra_collect_timestamps", "gra_get_kernel_context", "gra_get_stream", "gra_get_taa_reprojection",
    "gra_set_smaa_luts", "gra_get_host_stats", "gra_get_output_gather_stats", "gra_get_prefetched_refreshes", "gra_get_launch_graph_replays", "gra_get_allocated_bytes", "gra_gtx_probe", "gra_gtx_read", "gra_gtx_write",
    "gra_upload_gbuffer_gtx", "gra_save_resource_gtx", "gra_get_render_size", "gra_upload_ambient_occlusion", "gra_upload_aa_bench_images", "gra_compute_rec709_to_display", "gra_set_exchange_callback", "gra_get_strip_plan", "gra_get_strip_plan_aa", "gra_get_strip_plan_taa_history",
    "gra_comm_create_unique_id", "gra_comm_init", "gra_comm_info", "gra_comm_init_output", "gra_install_ssr_tables", "gra_reset_timestamps", "gra_set_directional_light", "gra_set_fog", "gra_generate_mipmaps", "gra_write_resource", "gra_get_frame_state", "gra_set_frame_state",
]

_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise capi.GraniteHipError(f"{LIB_PATH} not found: run __graft_entry__.build(). There is no CPU fallback.")
    capi.load_library()  # dependency, resolved through $ORIGIN rpath as well
    lib = C.CDLL(LIB_PATH)
    vp, P = C.c_void_p, C.POINTER
    sigs = {
        "gra_create": (vp, [P(Config), C.c_char_p, C.c_size_t]),
        "gra_destroy": (None, [vp]),
        "gra_last_error": (C.c_char_p, [vp]),
        "gra_set_camera": (C.c_int, [vp, vp, vp]),
        "gra_set_camera_motion": (C.c_int, [vp, vp]),
        "gra_set_render_parameters": (C.c_int, [vp, vp]),
        "gra_get_render_parameters": (C.c_int, [vp, vp]),
        "gra_set_lights": (C.c_int, [vp, vp, C.c_uint32]),
        "gra_upload_gbuffer": (C.c_int, [vp, vp, vp, vp, vp, vp, vp]),
        "gra_render_frames": (C.c_int, [vp, C.c_uint32, C.c_int32]),
        "gra_sync": (C.c_int, [vp]),
        "gra_get_resource": (C.c_int, [vp, C.c_char_p, P(ResourceInfo)]),
        "gra_read_resource": (C.c_int, [vp, C.c_char_p, vp, C.c_uint64]),
        "gra_get_backbuffer": (C.c_int, [vp, P(ResourceInfo)]),
        "gra_read_backbuffer": (C.c_int, [vp, vp, C.c_uint64]),
        "gra_get_cluster_state": (C.c_int, [vp, vp, vp, vp, vp, vp]),
        "gra_dump_graph": (C.c_size_t, [vp, C.c_char_p, C.c_size_t]),
        "gra_collect_timestamps": (C.c_int, [vp, P(Timestamp), C.c_int]),
        "gra_get_kernel_context": (vp, [vp]),
        "gra_get_stream": (vp, [vp]),
        "gra_get_taa_reprojection": (C.c_int, [vp, vp]),
        "gra_set_smaa_luts": (C.c_int, [vp, vp, vp]),
        "gra_get_host_stats": (C.c_int, [vp, vp]),
        "gra_get_output_gather_stats": (C.c_int, [vp, vp]),
        "gra_install_ssr_tables": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32]),
        "gra_get_prefetched_refreshes": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "gra_get_launch_graph_replays": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "gra_get_allocated_bytes": (C.c_int, [vp, vp]),
        "gra_get_render_size": (C.c_int, [vp, vp, vp]),
        "gra_upload_ambient_occlusion": (C.c_int, [vp, vp]),
        "gra_upload_aa_bench_images": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32]),
        "gra_compute_rec709_to_display": (C.c_int, [vp, vp]),
        "gra_gtx_probe": (C.c_int, [C.c_char_p, vp, vp, C.c_size_t]),
        "gra_gtx_read": (C.c_int, [C.c_char_p, vp, C.c_uint64, vp, C.c_size_t]),
        "gra_gtx_write": (C.c_int, [C.c_char_p, vp, vp, vp, C.c_size_t]),
        "gra_upload_gbuffer_gtx": (C.c_int, [vp] + [C.c_char_p] * 6),
        "gra_save_resource_gtx": (C.c_int, [vp, C.c_char_p, C.c_char_p]),
        "gra_reset_timestamps": (C.c_int, [vp]),
        "gra_generate_mipmaps": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]),
        "gra_set_directional_light": (C.c_int, [vp, vp, vp]),
        "gra_set_fog": (C.c_int, [vp, vp, C.c_float]),
        "gra_write_resource": (C.c_int, [vp, C.c_char_p, vp, C.c_uint64]),
        "gra_get_frame_state": (C.c_int, [vp, P(FrameState)]),
        "gra_set_frame_state": (C.c_int, [vp, P(FrameState)]),
        "gra_set_exchange_callback": (C.c_int, [vp, EXCHANGE_FN, vp]),
        "gra_get_strip_plan": (C.c_int, [vp, vp]),
        "gra_get_strip_plan_aa": (C.c_int, [vp, vp]),
        "gra_get_strip_plan_taa_history": (C.c_int, [vp, vp]),
        "gra_comm_create_unique_id": (C.c_int, [vp]),
        "gra_comm_init": (C.c_int, [vp, vp, C.c_int32, C.c_int32]),
        "gra_comm_info": (C.c_int, [vp, P(C.c_int32), P(C.c_int32), P(C.c_int32)]),
        "gra_comm_init_output": (C.c_int, [vp, vp, C.c_int32, C.c_int32]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else np.ascontiguousarray(a).ctypes.data


class Application:
    """Headless image-space application.  device=-1 composes/bakes the graph without a GPU (CPU tests)."""

    def __init__(self, width: int, height: int, *, device: int = 0, lighting: bool = True, hdr_bloom: bool = True,
                 dynamic_exposure: bool = True, compute_post: bool = True, post_aa: int = POST_AA_NONE,
                 pre_aa: int = POST_AA_NONE, rmw_emissive: bool = False, cluster_res=synth.CLUSTER_RESOLUTION,
                 frame_time: float = synth.FRAME_TIME, timestamps: bool = False, strip_index: int = 0, strip_count: int = 1,
                 alias_images: bool = True, depth_hierarchy: int = 0,
                 resolution_scale: float = 1.0, resolution_scale_sharpen: bool = True, fsr_fp16: bool = True,
                 ambient_occlusion: bool = False, hdr10: bool = False, ssr: bool = False, aa_bench: bool = False,
                 output_gather_rgba: bool = False, rt_fp16: bool = True, taa_history_reach_rows: int = 0):
        """taa_history_reach_rows > 0 (row bands + TAA): the history bands exchange boundary rows with their neighbours only; a pixel
        that reaches further makes the next render / sync / read raise.  rt_fp16 = False: viewer_config renderTargetFp16 = false (the reference's default): emissive / HDR-main and the TAA colour
        output are B10G11R11_UFLOAT_PACK32; the emissive upload is then (h, w) uint32 words (oracle.pack_b10g11r11)."""
        self.lib = load_library()
        cfg = Config()
        cfg.device, cfg.width, cfg.height = device, width, height
        cfg.enable_lighting, cfg.hdr_bloom = int(lighting), int(hdr_bloom)
        cfg.dynamic_exposure, cfg.compute_post = int(dynamic_exposure), int(compute_post)
        cfg.post_aa, cfg.pre_aa, cfg.rmw_emissive = post_aa, pre_aa, int(rmw_emissive)
        cfg.cluster_res[:] = cluster_res
        cfg.frame_time = frame_time
        cfg.directional_color[:] = synth.DIRECTIONAL_COLOR
        cfg.directional_direction[:] = synth.DIRECTIONAL_DIRECTION
        cfg.enable_timestamps = int(timestamps)
        cfg.strip_index, cfg.strip_count = strip_index, strip_count
        cfg.disable_image_aliasing = int(not alias_images)
        cfg.depth_hierarchy = int(depth_hierarchy)
        cfg.resolution_scale = float(resolution_scale)
        cfg.resolution_scale_sharpen, cfg.fsr_fp32 = int(resolution_scale_sharpen), int(not fsr_fp16)
        cfg.ambient_occlusion = int(ambient_occlusion)
        cfg.hdr10 = int(hdr10)
        cfg.ssr = int(ssr)
        cfg.aa_bench = int(aa_bench)
        cfg.output_gather_rgba = int(output_gather_rgba)
        cfg.hdr_packed_float = int(not rt_fp16)
        cfg.taa_history_reach_rows = int(taa_history_reach_rows)
        if ssr:
            install_ssr_tables()
        self._exchange_ref = None
        self.config = cfg
        self.width, self.height = width, height
        err = C.create_string_buffer(512)
        self.handle = self.lib.gra_create(cfg, err, 512)
        if not self.handle:
            raise capi.GraniteHipError(f"gra_create failed: {err.value.decode()}")
        if device >= 0 and POST_AA_SMAA_LOW <= post_aa <= POST_AA_SMAA_ULTRA:
            from .data import load_smaa_luts
            area, search = load_smaa_luts()
            self._check(self.lib.gra_set_smaa_luts(self.handle, area.ctypes.data, search.ctypes.data))

    def _check(self, code):
        if code < 0:
            raise capi.GraniteHipError(self.lib.gra_last_error(self.handle).decode())
        return code

    def close(self):
        if self.handle:
            self.lib.gra_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene ---------------------------------------------------------------------------------------------------
    def set_render_parameters(self, params104: np.ndarray):
        p = np.ascontiguousarray(params104, np.float32)
        assert p.size == 104
        self._check(self.lib.gra_set_render_parameters(self.handle, p.ctypes.data))

    def set_camera(self, projection16, view16):
        p, v = np.ascontiguousarray(projection16, np.float32), np.ascontiguousarray(view16, np.float32)
        self._check(self.lib.gra_set_camera(self.handle, p.ctypes.data, v.ctypes.data))

    def set_camera_motion(self, translation):
        """The eye moves by `translation` (world units) every frame from now on; (0, 0, 0) stops it."""
        t = np.ascontiguousarray(translation, np.float32)
        assert t.size == 3
        self._check(self.lib.gra_set_camera_motion(self.handle, t.ctypes.data))

    def get_render_parameters(self) -> np.ndarray:
        out = np.zeros(104, np.float32)
        self._check(self.lib.gra_get_render_parameters(self.handle, out.ctypes.data))
        return out

    def set_lights(self, descs: np.ndarray):
        d = np.ascontiguousarray(descs, synth.LIGHT_DESC_DTYPE)
        self._check(self.lib.gra_set_lights(self.handle, d.ctypes.data if len(d) else None, len(d)))

    def upload_gbuffer(self, gbuf: dict, motion_vectors=None):
        keep = [np.ascontiguousarray(gbuf[k]) if k in gbuf and gbuf[k] is not None else None
                for k in ("emissive", "albedo", "normal", "pbr", "depth")]
        mv = None if motion_vectors is None else np.ascontiguousarray(motion_vectors)
        self._check(self.lib.gra_upload_gbuffer(self.handle, *[_ptr(k) for k in keep], _ptr(mv)))

    def upload_aa_bench_images(self, first: np.ndarray, second: np.ndarray):
        """The two RGBA8 (sRGB) input images of the aa_bench graph, both H x W x 4 uint8 of one size."""
        a, b = np.ascontiguousarray(first, np.uint8), np.ascontiguousarray(second, np.uint8)
        assert a.shape == b.shape and a.ndim == 3 and a.shape[2] == 4
        self._check(self.lib.gra_upload_aa_bench_images(self.handle, a.ctypes.data, b.ctypes.data, a.shape[1], a.shape[0]))

    def upload_ambient_occlusion(self, ao: np.ndarray):
        """Render-sized uint8 image for "ssao-output-main" (needs ambient_occlusion=True)."""
        a = np.ascontiguousarray(ao, np.uint8)
        self._check(self.lib.gra_upload_ambient_occlusion(self.handle, a.ctypes.data))

    def upload_gbuffer_gtx(self, emissive=None, albedo=None, normal=None, pbr=None, depth=None, motion_vectors=None):
        """G-buffer attachments from .gtx files (paths; None = unchanged)."""
        paths = [None if p is None else str(p).encode() for p in (emissive, albedo, normal, pbr, depth, motion_vectors)]
        self._check(self.lib.gra_upload_gbuffer_gtx(self.handle, *paths))

    def save_gtx(self, path: str, name: Optional[str] = None):
        """Write graph texture `name` (None = the last backbuffer) as .gtx."""
        self._check(self.lib.gra_save_resource_gtx(self.handle, None if name is None else name.encode(), str(path).encode()))

    def upload_hdr(self, hdr_bits: np.ndarray):
        self.upload_gbuffer({"emissive": hdr_bits})

    # ---- frames --------------------------------------------------------------------------------------------------
    def render_frames(self, count: int = 1, sync: bool = True):
        self._check(self.lib.gra_render_frames(self.handle, count, int(sync)))

    def sync(self):
        self._check(self.lib.gra_sync(self.handle))

    # ---- introspection ----------------------------------------------------------------------------------------------
    def resource(self, name: str) -> ResourceInfo:
        info = ResourceInfo()
        self._check(self.lib.gra_get_resource(self.handle, name.encode(), info))
        return info

    def _shape(self, info: ResourceInfo, raw: np.ndarray):
        f, w, h = info.format, info.width, info.height
        if f == capi.FORMAT_R16G16B16A16_SFLOAT:
            return raw.view(np.uint16).reshape(h, w, 4)
        if f in (capi.FORMAT_R8G8B8A8_SRGB, capi.FORMAT_R8G8B8A8_UNORM):
            return raw.reshape(h, w, 4)
        if f == capi.FORMAT_R8G8_UNORM:
            return raw.reshape(h, w, 2)
        if f in (capi.FORMAT_D32_SFLOAT, capi.FORMAT_R32_SFLOAT):
            return raw.view(np.float32).reshape(h, w)
        if f in (capi.FORMAT_A2B10G10R10_UNORM_PACK32, capi.FORMAT_B10G11R11_UFLOAT_PACK32):
            return raw.view(np.uint32).reshape(h, w)
        if f == capi.FORMAT_R16G16_SFLOAT:
            return raw.view(np.uint16).reshape(h, w, 2)
        return raw

    def read(self, name: str) -> np.ndarray:
        info = self.resource(name)
        raw = np.empty(info.size_bytes, np.uint8)
        self._check(self.lib.gra_read_resource(self.handle, name.encode(), raw.ctypes.data, raw.nbytes))
        if info.levels > 1:
            return raw
        return self._shape(info, raw) if info.width else raw

    def write(self, name: str, data: np.ndarray):
        """The write-side twin of read(): what a later frame reads as the previous frame's value of `name`."""
        a = np.ascontiguousarray(data)
        self._check(self.lib.gra_write_resource(self.handle, name.encode(), a.ctypes.data, a.nbytes))

    def frame_state(self) -> "FrameState":
        st = FrameState()
        self._check(self.lib.gra_get_frame_state(self.handle, st))
        return st

    def set_frame_state(self, st: "FrameState"):
        self._check(self.lib.gra_set_frame_state(self.handle, st))

    def read_mip_chain(self, name: str):
        """Levels of a mip-chain attachment (R32_SFLOAT), finest first."""
        info = self.resource(name)
        raw = self.read(name).view(np.float32)
        lib = capi.load_library()
        out = []
        for l in range(info.levels):
            w, h = max(info.width >> l, 1), max(info.height >> l, 1)
            o = lib.gr_mip_chain_offset(info.width, info.height, 4, l) // 4
            out.append(raw[o:o + w * h].reshape(h, w))
        return out

    def read_backbuffer(self) -> np.ndarray:
        info = ResourceInfo()
        self._check(self.lib.gra_get_backbuffer(self.handle, info))
        raw = np.empty(info.size_bytes, np.uint8)
        self._check(self.lib.gra_read_backbuffer(self.handle, raw.ctypes.data, raw.nbytes))
        return self._shape(info, raw)

    def backbuffer_info(self) -> ResourceInfo:
        info = ResourceInfo()
        self._check(self.lib.gra_get_backbuffer(self.handle, info))
        return info

    def cluster_state(self):
        lights = np.zeros(4096 * 48, np.uint8)
        models = np.zeros((4096, 3, 4), np.float32)
        type_mask = np.zeros(128, np.uint32)
        params = np.zeros(176, np.uint8)
        ranges = np.zeros((4096, 2), np.uint32)
        n = self._check(self.lib.gra_get_cluster_state(self.handle, lights.ctypes.data, models.ctypes.data, type_mask.ctypes.data,
                                                       params.ctypes.data, ranges.ctypes.data))
        return {"count": n, "lights": lights, "models": models, "type_mask": type_mask, "params": params,
                "light_ranges": ranges[:max(n, 1)]}

    def graph(self) -> dict:
        need = self.lib.gra_dump_graph(self.handle, None, 0)
        if need == 0:
            raise capi.GraniteHipError(self.lib.gra_last_error(self.handle).decode())
        buf = C.create_string_buffer(need)
        self.lib.gra_dump_graph(self.handle, buf, need)
        return json.loads(buf.value.decode())

    def taa_reprojection(self) -> np.ndarray:
        out = np.zeros(16, np.float32)
        self._check(self.lib.gra_get_taa_reprojection(self.handle, out.ctypes.data))
        return out

    def timestamps(self) -> dict:
        arr = (Timestamp * 64)()
        n = self._check(self.lib.gra_collect_timestamps(self.handle, arr, 64))
        return {arr[i].tag.decode(): (int(arr[i].count), float(arr[i].total_ms)) for i in range(n)}

    def generate_mipmaps(self, level0_bits: np.ndarray, levels: int, components: int = 4, filter_mods=None) -> np.ndarray:
        """RGBA16F image with `levels` mip levels: level 0 = level0_bits, levels 1.. from the single-pass downsampler.
        Returns the tightly packed chain (uint16 bits, all levels)."""
        src = np.ascontiguousarray(level0_bits, np.uint16)
        h, w = src.shape[:2]
        texels = sum(max(w >> l, 1) * max(h >> l, 1) for l in range(levels))
        chain = np.zeros(texels * 4, np.uint16)
        fm = None if filter_mods is None else np.ascontiguousarray(filter_mods, np.float32).reshape(levels - 1, 4)
        self._check(self.lib.gra_generate_mipmaps(self.handle, src.ctypes.data, w, h, levels, components,
                                                  None if fm is None else fm.ctypes.data, chain.ctypes.data))
        return chain

    def reset_timestamps(self):
        self._check(self.lib.gra_reset_timestamps(self.handle))

    def set_directional(self, direction, color):
        d, c = np.asarray(direction, np.float32), np.asarray(color, np.float32)
        self._check(self.lib.gra_set_directional_light(self.handle, d.ctypes.data, c.ctypes.data))

    def set_fog(self, color, falloff: float):
        """LightingParameters::fog: render_light's fog quad (renderer.cpp:1179-1196) when falloff > 0."""
        c = np.ascontiguousarray(color, np.float32)
        self._check(self.lib.gra_set_fog(self.handle, c.ctypes.data, float(falloff)))

    # ---- row-band tiling ---------------------------------------------------------------------------------------------
    def set_exchange_callback(self, fn):
        """fn(tag: str, device_ptr: int, chunk_bytes: int, rank_count: int, stream: int) is called by the executor where
        the bands of all ranks meet; None removes it."""
        if fn is None:
            self._exchange_ref = None
            self._check(self.lib.gra_set_exchange_callback(self.handle, C.cast(None, EXCHANGE_FN), None))
            return

        def trampoline(_user, tag, ptr, chunk_bytes, ranks, stream):
            fn(tag.decode(), int(ptr or 0), int(chunk_bytes), int(ranks), int(stream or 0))

        self._exchange_ref = EXCHANGE_FN(trampoline)  # keep alive
        self._check(self.lib.gra_set_exchange_callback(self.handle, self._exchange_ref, None))

    @staticmethod
    def comm_create_unique_id() -> bytes:
        """Rank 0: the 128-byte RCCL id every rank passes to comm_init."""
        buf = (C.c_uint8 * 128)()
        if load_library().gra_comm_create_unique_id(buf) != 0:
            raise capi.GraniteHipError("gra_comm_create_unique_id failed (librccl.so.1 not loadable?)")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, ranks: int):
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.gra_comm_init(self.handle, buf, rank, ranks))

    def comm_info(self) -> dict:
        """What the in-frame communicator says about itself (for run records): ncclCommCount, ncclGetVersion, stand-in or RCCL."""
        n, v, s_ = C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
        self._check(self.lib.gra_comm_info(self.handle, C.byref(n), C.byref(v), C.byref(s_)))
        return {"nranks": n.value, "version": v.value, "library": "test stand-in (GRANITE_RCCL_LIBRARY)" if s_.value else "librccl.so.1"}

    def comm_init_output(self, unique_id: bytes, rank: int, ranks: int):
        """Second communicator: the tonemapped bands are gathered beside the frame, on a stream of their own."""
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.gra_comm_init_output(self.handle, buf, rank, ranks))

    def strip_plan(self) -> dict:
        out = np.zeros(24, np.uint32)
        self._check(self.lib.gra_get_strip_plan(self.handle, out.ctypes.data))
        plan = {"index": int(out[0]), "count": int(out[1]), "width": int(out[2]), "height": int(out[3])}
        for i, name in enumerate(("lighting", "threshold", "d0", "d1", "u0", "tonemap")):
            whole, first, count = (int(v) for v in out[4 + 3 * i:7 + 3 * i])
            plan[name] = None if whole else (first, count)
        plan["d1_chunk_rows"], plan["out_chunk_rows"] = int(out[22]), int(out[23])
        aa = np.zeros(12, np.uint32)
        self._check(self.lib.gra_get_strip_plan_aa(self.handle, aa.ctypes.data))
        for i, name in enumerate(("taa", "smaa_edges", "smaa_weights", "aa_out")):
            whole, first, count = (int(v) for v in aa[3 * i:3 * i + 3])
            plan[name] = None if whole else (first, count)
        th = np.zeros(5, np.uint32)
        self._check(self.lib.gra_get_strip_plan_taa_history(self.handle, th.ctypes.data))
        plan["taa_history_reach_rows"], plan["taa_exchange_rows"] = int(th[0]), int(th[1])
        plan["taa_history_held"] = None if th[2] else (int(th[3]), int(th[4]))
        return plan

    def host_stats(self) -> dict:
        out = np.zeros(3, np.float64)
        self._check(self.lib.gra_get_host_stats(self.handle, out.ctypes.data))
        return {"frames": int(out[0]), "seconds": float(out[1]), "blocked_seconds": float(out[2])}

    def output_gather_stats(self) -> dict:
        """Output gather beside the frame: how often the next writer of an output image found its gather still in flight."""
        out = np.zeros(2, np.uint64)
        self._check(self.lib.gra_get_output_gather_stats(self.handle, out.ctypes.data))
        return {"acquires": int(out[0]), "waits": int(out[1])}

    def launch_graph_replays(self) -> int:
        """Pre-recorded launch sequences replayed so far (one hipGraph launch instead of one API call per kernel)."""
        out = C.c_uint64(0)
        self._check(self.lib.gra_get_launch_graph_replays(self.handle, C.byref(out)))
        return int(out.value)

    def prefetched_refreshes(self) -> int:
        """Frames whose light refresh was done ahead of time by the clusterer's helper thread."""
        out = C.c_uint64(0)
        self._check(self.lib.gra_get_prefetched_refreshes(self.handle, C.byref(out)))
        return int(out.value)

    def render_size(self):
        """(width, height) the G-buffer is rendered and uploaded at (backbuffer size x resolution_scale)."""
        w, h = C.c_uint32(0), C.c_uint32(0)
        self._check(self.lib.gra_get_render_size(self.handle, C.byref(w), C.byref(h)))
        return int(w.value), int(h.value)

    def allocated_bytes(self) -> int:
        out = C.c_uint64(0)
        self._check(self.lib.gra_get_allocated_bytes(self.handle, C.byref(out)))
        return int(out.value)

    def kernel_context(self) -> "KernelContextView":
        return KernelContextView(self.lib.gra_get_kernel_context(self.handle))


class KernelContextView:
    """Borrowed gr_ctx of a running application: per-kernel hipEvent timing (gr_timing_*)."""

    def __init__(self, handle):
        self.lib = capi.load_library()
        self.handle = handle

    def timing_enable(self, enable: bool):
        self.lib.gr_timing_enable(self.handle, int(enable))

    def timing_reset(self):
        self.lib.gr_timing_reset(self.handle)

    def bandwidth_probe(self, nbytes: int = 1 << 30, repeats: int = 5):
        """Measured HBM copy / triad rates in GB/s (float4 kernels over arrays beyond the Infinity Cache)."""
        copy, triad = C.c_double(), C.c_double()
        if self.lib.gr_bandwidth_probe(self.handle, nbytes, repeats, C.byref(copy), C.byref(triad)) < 0:
            raise capi.GraniteHipError(self.lib.gr_last_error(self.handle).decode())
        return copy.value, triad.value

    def timing_set_sampling(self, every_nth: int):
        self.lib.gr_timing_set_sampling(self.handle, int(every_nth))

    def timing_set_filter(self, name=None):
        self.lib.gr_timing_set_filter(self.handle, None if name is None else name.encode())

    def timing_query(self) -> dict:
        arr = (capi.TimingEntry * 64)()
        n = self.lib.gr_timing_query(self.handle, arr, 64)
        if n < 0:
            raise capi.GraniteHipError(self.lib.gr_last_error(self.handle).decode())
        return {arr[i].name.decode(): (int(arr[i].count), float(arr[i].total_ms)) for i in range(n)}

    def timing_max_ms(self, name: str) -> float:
        """Longest single bracket recorded under `name` since the last reset (a collective that stalled once shows here, not in the mean)."""
        out = C.c_double(0.0)
        if self.lib.gr_timing_max_ms(self.handle, name.encode(), C.byref(out)) < 0:
            raise capi.GraniteHipError(self.lib.gr_last_error(self.handle).decode())
        return float(out.value)
