"""GTX ("GRANITE TEXFMT1") files through the host library's C ABI (gra_gtx_*): the container Granite keeps textures and
image dumps in (vulkan/texture/memory_mapped_texture.cpp:29-44; payload layout vulkan/texture/texture_format.cpp:349-387)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List

import numpy as np

from . import app as gapp
from . import capi

HEADER_SIZE = 64


class GtxInfo(C.Structure):
    _fields_ = [("type", C.c_uint32), ("format", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("depth", C.c_uint32), ("layers", C.c_uint32), ("levels", C.c_uint32), ("flags", C.c_uint32),
                ("payload_size", C.c_uint64)]


class GtxError(RuntimeError):
    pass


@dataclass
class GtxFile:
    info: GtxInfo
    payload: np.ndarray  # uint8

    def level_offset(self, level: int) -> int:
        return level_offset(self.info, level)

    def level(self, level: int = 0) -> np.ndarray:
        """Raw bytes of one mip level (all layers), shaped (layers, height, width, bytes per texel)."""
        w, h = max(self.info.width >> level, 1), max(self.info.height >> level, 1)
        bpp = capi.FORMAT_BPP[self.info.format]
        o = self.level_offset(level)
        return self.payload[o:o + self.info.layers * w * h * bpp].reshape(self.info.layers, h, w, bpp)


def level_offset(info: GtxInfo, level: int) -> int:
    bpp = capi.FORMAT_BPP[info.format]
    offset = 0
    for l in range(level + 1):
        offset = (offset + 15) & ~15
        if l == level:
            return offset
        offset += max(info.width >> l, 1) * max(info.height >> l, 1) * max(info.depth >> l, 1) * info.layers * bpp
    return offset


def payload_size(info: GtxInfo) -> int:
    last = info.levels - 1
    bpp = capi.FORMAT_BPP[info.format]
    return level_offset(info, last) + max(info.width >> last, 1) * max(info.height >> last, 1) * max(info.depth >> last, 1) * info.layers * bpp


def _lib():
    lib = gapp.load_library()
    return lib


def probe(path: str) -> GtxInfo:
    info = GtxInfo()
    err = C.create_string_buffer(512)
    if _lib().gra_gtx_probe(path.encode(), C.byref(info), err, len(err)) < 0:
        raise GtxError(err.value.decode())
    return info


def read(path: str) -> GtxFile:
    info = probe(path)
    payload = np.empty(info.payload_size, np.uint8)
    err = C.create_string_buffer(512)
    if _lib().gra_gtx_read(path.encode(), payload.ctypes.data, payload.nbytes, err, len(err)) < 0:
        raise GtxError(err.value.decode())
    return GtxFile(info, payload)


def write(path: str, fmt: int, levels: List[np.ndarray], flags: int = 0, layers: int = 1):
    """levels[l]: array whose bytes are level l (all layers), level 0 first; shape[-3:-1] or [0:2] of level 0 gives h, w."""
    first = np.ascontiguousarray(levels[0])
    bpp = capi.FORMAT_BPP[fmt]
    h, w = (first.shape[1], first.shape[2]) if layers > 1 else (first.shape[0], first.shape[1])
    info = GtxInfo(1, fmt, w, h, 1, layers, len(levels), flags, 0)
    info.payload_size = payload_size(info)
    payload = np.zeros(info.payload_size, np.uint8)
    for l, a in enumerate(levels):
        raw = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        want = max(w >> l, 1) * max(h >> l, 1) * layers * bpp
        if raw.size != want:
            raise GtxError(f"level {l}: {raw.size} bytes, layout wants {want}")
        o = level_offset(info, l)
        payload[o:o + raw.size] = raw
    err = C.create_string_buffer(512)
    if _lib().gra_gtx_write(path.encode(), C.byref(info), payload.ctypes.data, err, len(err)) < 0:
        raise GtxError(err.value.decode())
