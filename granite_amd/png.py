"""Minimal PNG codec (8-bit RGBA / RGB, non-interlaced) on zlib: the headless runner's --png-path / --png-reference-path
outputs (application_headless.cpp:398-417 writes the swapchain with stbi_write_png) and the inputs of image_compare."""
from __future__ import annotations

import struct
import zlib

import numpy as np

SIGNATURE = b"\x89PNG\r\n\x1a\n"


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path: str, rgba: np.ndarray, level: int = 3) -> None:
    img = np.ascontiguousarray(rgba, np.uint8)
    if img.ndim != 3 or img.shape[2] not in (3, 4):
        raise ValueError("write_png wants (H, W, 3|4) uint8")
    h, w, c = img.shape
    rows = np.zeros((h, 1 + w * c), np.uint8)  # filter type 0 on every scanline
    rows[:, 1:] = img.reshape(h, w * c)
    with open(path, "wb") as f:
        f.write(SIGNATURE)
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 0)))
        f.write(_chunk(b"IDAT", zlib.compress(rows.tobytes(), level)))
        f.write(_chunk(b"IEND", b""))


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def read_png(path: str) -> np.ndarray:
    """Returns (H, W, 4) uint8; RGB files get alpha 255.  All five scanline filters are handled.  Anything malformed is a
    ValueError."""
    try:
        return _read_png(path)
    except (zlib.error, struct.error, IndexError) as e:
        raise ValueError(f"{path}: corrupt PNG ({e})") from e


def _read_png(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != SIGNATURE:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, header = 8, [], None
    while pos + 8 <= len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            header = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    if header is None:
        raise ValueError(f"{path}: no IHDR")
    w, h, depth, colour, _, _, interlace = header
    if depth != 8 or colour not in (2, 6) or interlace:
        raise ValueError(f"{path}: only 8-bit RGB / RGBA non-interlaced PNG is supported")
    c = 4 if colour == 6 else 3
    pixels = zlib.decompress(b"".join(idat))
    if len(pixels) != h * (1 + w * c):
        raise ValueError(f"{path}: {len(pixels)} bytes of image data, IHDR announces {h * (1 + w * c)}")
    data = np.frombuffer(pixels, np.uint8).reshape(h, 1 + w * c)
    out = np.zeros((h, w * c), np.uint8)
    prev = np.zeros(w * c, np.uint8)
    for y in range(h):
        kind, line = int(data[y, 0]), data[y, 1:]
        if kind == 0:
            cur = line.copy()
        elif kind == 2:
            cur = line + prev
        else:  # 1, 3, 4 depend on the pixel to the left: walk pixel by pixel, channels vectorised
            cur = np.zeros(w * c, np.uint8)
            left, upleft = np.zeros(c, np.uint8), np.zeros(c, np.uint8)
            for x in range(w):
                s = slice(x * c, x * c + c)
                up = prev[s]
                if kind == 1:
                    pred = left
                elif kind == 3:
                    pred = ((left.astype(np.int32) + up) >> 1).astype(np.uint8)
                elif kind == 4:
                    pred = _paeth(left, up, upleft)
                else:
                    raise ValueError(f"{path}: bad filter type {kind}")
                left = line[s] + pred
                cur[s] = left
                upleft = up
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, c)
    if c == 3:
        img = np.concatenate([img, np.full((h, w, 1), 255, np.uint8)], axis=2)
    return img
