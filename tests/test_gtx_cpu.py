"""CPU: the GTX ("GRANITE TEXFMT1") reader / writer of the host library, pinned against the .gtx files the reference ships
(tests/golden/gtx_reference_files.json, made by tests/golden/make_gtx_golden.py from /root/reference/assets/textures)."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from granite_amd import capi, data, gtx

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "gtx_reference_files.json")))


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def test_writer_reproduces_the_reference_files_bit_for_bit(tmp_path):
    """The SMAA lookup tables are committed as raw payloads; wrapping them with our writer must give exactly the files
    Granite ships (same header bytes, same layout, same size)."""
    area, search = data.load_smaa_luts()
    gtx.write(str(tmp_path / "area.gtx"), capi.FORMAT_R8G8_UNORM, [area])
    gtx.write(str(tmp_path / "search.gtx"), capi.FORMAT_R8_UNORM, [search[..., None]])
    for ours, name in (("area.gtx", "smaa/area.gtx"), ("search.gtx", "smaa/search.gtx")):
        raw = open(tmp_path / ours, "rb").read()
        assert len(raw) == GOLDEN[name]["size"]
        assert raw[:64].hex() == GOLDEN[name]["header_hex"]
        assert hashlib.sha256(raw).hexdigest() == GOLDEN[name]["sha256"]
    # and the reader gets the payload back
    f = gtx.read(str(tmp_path / "area.gtx"))
    assert (f.info.type, f.info.format, f.info.width, f.info.height, f.info.layers, f.info.levels) == (1, 16, 160, 560, 1, 1)
    np.testing.assert_array_equal(f.level(0)[0], area)
    assert hashlib.sha256(f.payload.tobytes()).hexdigest() == GOLDEN["smaa/area.gtx"]["payload_sha256"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/assets/textures"), reason="reference tree not present")
def test_reader_accepts_every_gtx_file_of_the_reference():
    for name, want in GOLDEN.items():
        f = gtx.read(os.path.join("/root/reference/assets/textures", name))
        assert f.payload.size + gtx.HEADER_SIZE == want["size"]
        assert hashlib.sha256(f.payload.tobytes()).hexdigest() == want["payload_sha256"]
    lut = gtx.read("/root/reference/assets/textures/ibl_brdf_lut.gtx")
    assert (lut.info.format, lut.info.width, lut.info.height, lut.info.levels) == (capi.FORMAT_R16G16_SFLOAT, 256, 256, 1)


def test_mip_levels_start_on_16_byte_boundaries(tmp_path):
    """texture_format.cpp:359-361: offset = (offset + 15) & ~15 before every level."""
    rng = np.random.default_rng(5)
    w, h = 13, 7  # R8: 91, 18, 3, 1 bytes per level
    levels = [rng.integers(0, 256, (max(h >> l, 1), max(w >> l, 1), 1), dtype=np.uint8) for l in range(4)]
    path = str(tmp_path / "mips.gtx")
    gtx.write(path, capi.FORMAT_R8_UNORM, levels)
    raw = open(path, "rb").read()
    assert [gtx.level_offset(gtx.probe(path), l) for l in range(4)] == [0, 96, 128, 144]
    assert len(raw) == 64 + 144 + 1
    assert struct.unpack_from("<8IQQ", raw, 16) == (1, 9, 13, 7, 1, 1, 4, 0, 145, 0)
    f = gtx.read(path)
    for l in range(4):
        np.testing.assert_array_equal(f.level(l)[0], levels[l])
    # padding between levels is zero
    assert raw[64 + 91:64 + 96] == bytes(5)


def test_malformed_files_are_rejected_with_a_reason(tmp_path):
    good = str(tmp_path / "good.gtx")
    gtx.write(good, capi.FORMAT_R8G8B8A8_UNORM, [np.zeros((4, 4, 4), np.uint8)])
    raw = bytearray(open(good, "rb").read())

    def expect(mutated: bytes, what: str):
        p = str(tmp_path / "bad.gtx")
        open(p, "wb").write(mutated)
        with pytest.raises(gtx.GtxError, match=what):
            gtx.read(p)

    expect(b"NOT A TEXTURE..." + bytes(raw[16:]), "magic")
    expect(bytes(raw[:-8]), "truncated")
    bad = bytearray(raw)
    struct.pack_into("<Q", bad, 48, 63)
    expect(bytes(bad), "payload size")
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 20, 147)  # VK_FORMAT_BC7_UNORM_BLOCK: not handled by this executor
    expect(bytes(bad), "format")
    # extents whose byte count wraps 64 bits (65536 x 65536 x 32768 x 32768 layers x 4 B = 2^66 = 0 mod 2^64) with an empty payload
    bad = bytearray(raw[:64])
    struct.pack_into("<IIII", bad, 24, 65536, 65536, 32768, 32768)
    struct.pack_into("<Q", bad, 48, 0)
    expect(bytes(bad), "implausible")
    with pytest.raises(gtx.GtxError, match="cannot open"):
        gtx.read(str(tmp_path / "missing.gtx"))


def test_gtx_reader_survives_corrupted_files(tmp_path):
    """Header fields and length mutated at random: the reader either parses a self-consistent file or reports an error; it never
    reads past the file (the probe / read entry points validate payload_size against the file and the layout)."""
    r = np.random.default_rng(7)
    img = r.integers(0, 256, (16, 24, 4), dtype=np.uint8)
    good = str(tmp_path / "good.gtx")
    gtx.write(good, 37, [img, img[:8, :12], img[:4, :6]])
    blob = bytearray(open(good, "rb").read())
    ok = bad = 0
    for trial in range(400):
        b = bytearray(blob)
        for _ in range(int(r.integers(1, 4))):
            kind = int(r.integers(0, 3)) if len(b) >= 64 else 2
            if kind == 0:                                  # flip a byte somewhere in the 64-byte header
                b[int(r.integers(0, 64))] = int(r.integers(0, 256))
            elif kind == 1:                                # overwrite one of the eight u32 header fields with an extreme value
                field = 16 + 4 * int(r.integers(0, 8))
                b[field:field + 4] = int(r.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFF, 65536, 3])).to_bytes(4, "little")
            else:                                          # truncate or pad the file
                n = int(r.integers(0, len(b) + 64))
                b = b[:n] if n <= len(b) else b + bytearray(n - len(b))
        path = str(tmp_path / f"m{trial}.gtx")
        with open(path, "wb") as f:
            f.write(b)
        try:
            info = gtx.probe(path)
            data = gtx.read(path)
            assert data.payload.nbytes == info.payload_size
            assert info.payload_size <= max(len(b) - gtx.HEADER_SIZE, 0)
            ok += 1
        except gtx.GtxError:
            bad += 1
        os.remove(path)
    assert bad > 100 and ok + bad == 400
