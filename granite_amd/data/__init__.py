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
