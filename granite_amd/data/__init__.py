"""SMAA lookup tables (see README.md)."""
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
AREA_SHAPE = (560, 160, 2)
SEARCH_SHAPE = (16, 64)


def _payload(path: str) -> bytes:
    raw = open(path, "rb").read()
    if raw[:15] == b"GRANITE TEXFMT1":  # MemoryMappedHeader: magic[16], 8 x u32, payload_size u64, reserved u64
        payload_size = struct.unpack_from("<Q", raw, 48)[0]
        return raw[64:64 + payload_size]
    return raw


def load_smaa_luts(area_path: str = None, search_path: str = None):
    area = np.frombuffer(_payload(area_path or os.path.join(_HERE, "smaa_area_rg8_160x560.bin")), np.uint8)
    search = np.frombuffer(_payload(search_path or os.path.join(_HERE, "smaa_search_r8_64x16.bin")), np.uint8)
    assert area.size == 560 * 160 * 2 and search.size == 16 * 64
    return area.reshape(AREA_SHAPE).copy(), search.reshape(SEARCH_SHAPE).copy()


# ---- screen-space reflection tables (see README.md) -----------------------------------------------------------------------
SSSR_NOISE_BASE_SHAPE = (128, 128, 2)
SSSR_DITHER_LAYERS = 64
BRDF_LUT_SHAPE = (256, 256, 2)


def load_sssr_noise_base(path: str = None) -> np.ndarray:
    """uint8[128, 128, 2]: the blue-noise sampler's integer values for sample 0, dimensions 0 and 1."""
    raw = np.fromfile(path or os.path.join(_HERE, "sssr_blue_noise_128x128_rg8.bin"), np.uint8)
    assert raw.size == 128 * 128 * 2
    return raw.reshape(SSSR_NOISE_BASE_SHAPE).copy()


def expand_sssr_dither(base: np.ndarray) -> np.ndarray:
    """The 64-layer R8G8 dither texture of renderer/post/ssr.cpp:178-199 as uint16[64, 128, 128] (r | g << 8), fp32 arithmetic:
    sample = (0.5 + value) / 256, + GOLDEN_RATIO * layer, fract, * 255 + 0.5, truncate."""
    golden = np.float32(1.61803398875)
    sample = (np.float32(0.5) + base.astype(np.float32)) / np.float32(256.0)
    out = np.zeros((SSSR_DITHER_LAYERS, 128, 128), np.uint16)
    for z in range(SSSR_DITHER_LAYERS):
        offset = golden * np.float32(z)
        v = (sample + offset).astype(np.float32)
        v = (v - np.floor(v)).astype(np.float32)
        q = (v * np.float32(255.0) + np.float32(0.5)).astype(np.uint32)
        out[z] = (q[..., 0] | (q[..., 1] << 8)).astype(np.uint16)
    return out


def load_brdf_lut(path: str = None) -> np.ndarray:
    """uint16[256, 256, 2]: R16G16_SFLOAT bits of the split-sum BRDF table (assets/textures/ibl_brdf_lut.gtx payload)."""
    raw = np.frombuffer(_payload(path or os.path.join(_HERE, "ibl_brdf_lut_rg16f_256x256.bin")), np.uint16)
    assert raw.size == 256 * 256 * 2
    return raw.reshape(BRDF_LUT_SHAPE).copy()
