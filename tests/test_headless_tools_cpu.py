"""Host-side tooling around the path: PNG codec, the image_compare PSNR gate (tools/image_compare.cpp), the
viewer_config / lights.json readers and the --stat document of the headless runner (application_headless.cpp:627-652)."""
import json
import math
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

from granite_amd import app as gapp, gtx, headless, image_compare, png, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_png_round_trip(tmp_path):
    r = np.random.default_rng(0)
    for shape in ((7, 13, 4), (16, 16, 3), (1, 1, 4)):
        img = r.integers(0, 256, shape, dtype=np.uint8)
        p = str(tmp_path / "a.png")
        png.write_png(p, img)
        got = png.read_png(p)
        assert got.shape == (shape[0], shape[1], 4)
        np.testing.assert_array_equal(got[..., :shape[2]], img)
        if shape[2] == 3:
            assert (got[..., 3] == 255).all()
    with pytest.raises(ValueError):
        png.write_png(str(tmp_path / "b.png"), np.zeros((4, 4), np.uint8))
    (tmp_path / "c.png").write_bytes(b"not a png at all")
    with pytest.raises(ValueError):
        png.read_png(str(tmp_path / "c.png"))


def _filtered_png(path, img):
    """Encode with scanline filters 0..4 in rotation (what stb / libpng writers emit), to exercise the decoder."""
    h, w, c = img.shape
    raw = bytearray()
    prev = np.zeros(w * c, np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        upleft = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
        kind = y % 5
        if kind == 0:
            pred = np.zeros_like(cur)
        elif kind == 1:
            pred = left
        elif kind == 2:
            pred = prev
        elif kind == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        raw.append(kind)
        raw += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(png.SIGNATURE + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 0)))
        half = len(raw) // 2
        comp = zlib.compress(bytes(raw))
        f.write(chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"tEXt", b"k\0v") + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b""))


def test_png_reader_handles_all_filters_and_split_idat(tmp_path):
    img = np.random.default_rng(1).integers(0, 256, (23, 9, 4), dtype=np.uint8)
    p = str(tmp_path / "f.png")
    _filtered_png(p, img)
    np.testing.assert_array_equal(png.read_png(p), img)


def _psnr(a, b):
    d = a[..., :3].astype(np.float64) - b[..., :3]
    return 10.0 * math.log10(255.0 ** 2 * d.size / (d * d).sum())


def test_image_compare_psnr_gate(tmp_path, capsys):
    r = np.random.default_rng(2)
    a = r.integers(0, 256, (32, 48, 4), dtype=np.uint8)
    b = a.copy()
    b[..., :3] = np.clip(a[..., :3].astype(int) + r.integers(-3, 4, (32, 48, 3)), 0, 255)
    b[..., 3] = 7   # alpha never counts
    pa, pb = str(tmp_path / "a.gtx"), str(tmp_path / "b.gtx")
    gtx.write(pa, 43, [a])
    gtx.write(pb, 43, [b])
    want = _psnr(a, b)
    assert image_compare.compare_images(image_compare.load_image(pa), image_compare.load_image(pb)) == pytest.approx(want, abs=1e-9)
    assert image_compare.main([pa, pb, "--threshold", str(want - 1)]) == 0
    assert f"PSNR: {want:.0f} dB" in capsys.readouterr().out
    assert image_compare.main([pa, pb, "--threshold", str(want + 1)]) == 1
    assert image_compare.main([pa, pa, "--threshold", "200"]) == 0          # identical: infinite PSNR passes any gate
    assert image_compare.main([pa]) == 1                                     # "Need two inputs."
    assert image_compare.main([pa, str(tmp_path / "missing.gtx")]) == 1
    # mismatching formats / sizes report 0 dB (compare_images :89-107) and so fail any non-negative threshold
    pu = str(tmp_path / "u.gtx")
    gtx.write(pu, 37, [a])
    assert image_compare.main([pa, pu]) == 0 and image_compare.main([pa, pu, "--threshold", "0.5"]) == 1
    ps = str(tmp_path / "s.gtx")
    gtx.write(ps, 43, [a[:16]])
    assert image_compare.main([pa, ps, "--threshold", "0.5"]) == 1
    # the diff image: 16 x the signed byte difference, clamped above, alpha 255
    pd = str(tmp_path / "d.png")
    assert image_compare.main([pa, pb, "--diff", pd]) == 0
    d = png.read_png(pd)
    diff = (a[..., :3].astype(int) - b[..., :3].astype(int)) * 16
    np.testing.assert_array_equal(d[..., :3], (np.minimum(diff, 255) & 255).astype(np.uint8))
    assert (d[..., 3] == 255).all()
    # PNG inputs compare like GTX ones
    ppa, ppb = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    png.write_png(ppa, a)
    png.write_png(ppb, b)
    assert image_compare.compare_images(image_compare.load_image(ppa), image_compare.load_image(ppb)) == pytest.approx(want, abs=1e-9)


def test_image_compare_directories(tmp_path):
    r = np.random.default_rng(3)
    da, db = tmp_path / "A", tmp_path / "B"
    da.mkdir(), db.mkdir()
    for i in range(3):
        img = r.integers(0, 256, (8, 8, 4), dtype=np.uint8)
        other = img.copy()
        if i == 1:
            other[..., 0] ^= 0x40
        gtx.write(str(da / f"f{i}.gtx"), 43, [img])
        gtx.write(str(db / f"f{i}.gtx"), 43, [other])
    assert image_compare.main([str(da), str(db)]) == 0
    assert image_compare.main([str(da), str(db), "--threshold", "40"]) == 1
    os.remove(str(db / "f2.gtx"))
    assert image_compare.main([str(da), str(db)]) == 1   # "Folder size is not identical."
    exe = os.path.join(ROOT, "tools", "image_compare")
    assert subprocess.run([exe, str(da / "f0.gtx"), str(db / "f0.gtx"), "--threshold", "60"]).returncode == 0
    assert subprocess.run([exe, str(da / "f1.gtx"), str(db / "f1.gtx"), "--threshold", "60"]).returncode == 1


def test_viewer_config_mapping():
    kw = headless.viewer_config_to_kwargs({})
    assert kw == dict(lighting=True, hdr_bloom=True, dynamic_exposure=True, post_aa=gapp.POST_AA_NONE, resolution_scale=1.0,
                      resolution_scale_sharpen=True, hdr10=False, ambient_occlusion=False, ssr=False)
    kw = headless.viewer_config_to_kwargs({"renderer": "deferred", "msaa": 1, "hdrBloom": True, "hdrBloomDynamicExposure": False,
                                           "postAA": "smaaHigh", "resolutionScale": 0.75, "resolutionScaleSharpen": False,
                                           "clusteredLights": True, "directionalLightShadows": True, "ssao": True,
                                           "shadowMapResolution": 2048.0})
    assert kw["post_aa"] == gapp.POST_AA_SMAA_HIGH and kw["resolution_scale"] == 0.75 and not kw["resolution_scale_sharpen"]
    assert not kw["dynamic_exposure"] and kw["ambient_occlusion"]
    assert headless.viewer_config_to_kwargs({"hdr10": True})["hdr_bloom"] is False
    assert headless.viewer_config_to_kwargs({"ssr": True})["ssr"] is True
    for bad in ({"renderer": "forward"}, {"msaa": 4}, {"postAA": "taaFSR2"}, {"postAA": "bogus"}):
        with pytest.raises(headless.ConfigError):
            headless.viewer_config_to_kwargs(bad)
    # every AA name the reference's string_to_post_antialiasing_type knows is either mapped or refused by name
    assert set(headless.POST_AA) | set(headless.UNSUPPORTED_AA) == {"none", "fxaa", "fxaa2phase", "smaaLow", "smaaMedium", "smaaHigh",
                                                                     "smaaUltra", "smaaUltraT2X", "taaLow", "taaMedium", "taaHigh",
                                                                     "taaFSR2"}
    # the mapped configs bake (no GPU needed)
    for doc in ({"postAA": "fxaa"}, {"postAA": "taaMedium"}, {"resolutionScale": 0.5}, {"hdr10": True}, {"hdrBloom": False}, {"ssr": True}):
        a = gapp.Application(640, 360, device=-1, **headless.viewer_config_to_kwargs(doc))
        assert a.graph()["passes"]
        a.close()


def test_lights_json():
    doc = {"directional": {"direction": [0.0, -1.0, 0.0], "color": [1.0, 2.0, 3.0]},
           "spot": [{"innerCone": 0.9, "outerCone": 0.7, "color": [4, 5, 6], "range": 9.0, "position": [1, 2, 3], "direction": [0, 0, -1]},
                    {"innerCone": 0.8, "outerCone": 0.6, "color": [1, 1, 1], "position": [0, 0, 0], "direction": [0, -2, 0]}],
           "point": [{"color": [7, 8, 9], "range": 2.5, "position": [-1, -2, -3]}]}
    descs, directional = headless.lights_from_json(doc)
    assert directional == {"direction": [0.0, 1.0, 0.0], "color": [1.0, 2.0, 3.0]}   # read_lights negates the direction
    assert list(descs["type"]) == [0, 0, 1] and list(descs["cutoff_range"]) == [9.0, 0.0, 2.5]
    np.testing.assert_allclose(descs["transform"][:, :, 3], [[1, 2, 3], [0, 0, 0], [-1, -2, -3]])
    for i, want in ((0, (0, 0, -1)), (1, (0, -1, 0))):
        rot = descs["transform"][i][:, :3].astype(np.float64)
        np.testing.assert_allclose(rot @ rot.T, np.eye(3), atol=1e-6)           # orthonormal, right-handed
        assert np.linalg.det(rot) == pytest.approx(1.0, abs=1e-6)
        np.testing.assert_allclose(-rot[:, 2], want, atol=1e-6)                 # -Z of the node = light direction
    np.testing.assert_allclose(descs["transform"][2][:, :3], np.eye(3))
    assert headless.lights_from_json({})[0].shape == (0,)


def test_stat_document_is_what_the_sweep_scripts_read():
    stamps = {"lighting-main": (10, 2.5), "bloom-downsample-0": (20, 1.0), "never-ran": (0, 0.0)}
    doc = headless.stat_document(291.5, "AMD Instinct MI355X", 70253, stamps, 10)
    parsed = json.loads(json.dumps(doc))
    # sweep_scene.py run_test(): averageFrameTimeUs, gpu, driverVersion, optional performance
    assert parsed["averageFrameTimeUs"] == 291.5 and parsed["gpu"] == "AMD Instinct MI355X" and parsed["driverVersion"] == 70253
    assert set(parsed["performance"]) == {"lighting-main", "bloom-downsample-0"}
    rep = parsed["performance"]["bloom-downsample-0"]
    assert rep == {"timePerAccumulationUs": 50.0, "timePerFrameContextUs": 100.0, "accumulationsPerFrameContext": 2.0}
    assert "performance" not in headless.stat_document(1.0, "x", 1, {}, 1)


def test_headless_cli_refuses_what_the_reference_refuses(tmp_path):
    exe = os.path.join(ROOT, "tools", "granite-hip-headless")
    r = subprocess.run([exe, "synthetic", "--width", "64", "--height", "64"], capture_output=True, text=True)
    assert r.returncode == 1 and "--frames" in r.stderr
    cfg = tmp_path / "fwd.json"
    cfg.write_text(json.dumps({"renderer": "forward"}))
    r = subprocess.run([exe, "synthetic", "--frames", "1", "--config", str(cfg)], capture_output=True, text=True)
    assert r.returncode == 1 and "deferred" in r.stderr


def test_corrupt_png_is_a_load_failure_not_a_crash(tmp_path):
    r = np.random.default_rng(9)
    img = r.integers(0, 256, (12, 10, 4), dtype=np.uint8)
    good = str(tmp_path / "good.png")
    png.write_png(good, img)
    blob = open(good, "rb").read()
    failures = 0
    for trial in range(200):
        b = bytearray(blob)
        for _ in range(int(r.integers(1, 4))):
            if r.random() < 0.7:
                b[int(r.integers(8, len(b)))] ^= 1 << int(r.integers(0, 8))
            else:
                b = b[:int(r.integers(8, len(b)))]
        path = str(tmp_path / "m.png")
        with open(path, "wb") as f:
            f.write(b)
        try:
            out = png.read_png(path)
            assert out.shape[2] == 4 and out.dtype == np.uint8   # CRCs are not verified: a flipped pixel bit still decodes
        except ValueError:
            failures += 1
            assert image_compare.main([good, path]) == 1          # "Failed to load texture", exit code 1
    assert failures > 50


@pytest.mark.skipif(not os.path.exists("/root/reference/tools/sweep_scene.py"), reason="needs the Granite checkout")
def test_the_references_sweep_script_consumes_our_stat_file(tmp_path):
    """Granite's own tools/sweep_scene.py, executed, with a stand-in viewer binary that takes the command line the script builds
    (sweep_scene.py:94-117) through the headless runner's argument parser and writes the --stat document with stat_document():
    the script must get through run_test / map_result_to_json and write its results file."""
    stub = tmp_path / "viewer-stub"
    stub.write_text(f"""#!{sys.executable}
import json, sys
sys.path.insert(0, {ROOT!r})
from granite_amd import headless
args = headless.parse_args(sys.argv[1:])
assert args.frames == 7 and args.width == 640 and args.height == 360 and args.timestamp and args.config.endswith(".json")
kw = headless.viewer_config_to_kwargs(json.load(open(args.config)))
stamps = {{"gbuffer-main + lighting-main": (7, 1.4), "tonemap": (7, 0.21)}}
json.dump(headless.stat_document(250.0 if kw["post_aa"] == 0 else 300.0, "stub gpu", 42, stamps, args.frames), open(args.stat, "w"))
""")
    stub.chmod(0o755)
    cfgs = []
    for name, doc in (("plain", {"renderer": "deferred", "hdrBloom": True}), ("fxaa", {"postAA": "fxaa"})):
        p = tmp_path / f"{name}.json"
        p.write_text(json.dumps(doc))
        cfgs.append(str(p))
    results = tmp_path / "results.json"
    r = subprocess.run([sys.executable, "/root/reference/tools/sweep_scene.py", "--viewer-binary", str(stub), "--scene", "synthetic",
                        "--configs", *cfgs, "--width", "640", "--height", "360", "--frames", "7", "--iterations", "2", "--timestamp",
                        "--results", str(results)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    runs = json.loads(results.read_text())["runs"]
    assert [x["config"] for x in runs] == ["plain", "fxaa"] and [x["avg"] for x in runs] == [250.0, 300.0]
    assert runs[0]["gpu"] == "stub gpu" and runs[0]["version"] == 42 and runs[0]["width"] == 640
    assert runs[0]["performance"]["tonemap"] == {"timePerAccumulationUs": 30.0, "timePerFrameContextUs": 30.0,
                                                   "accumulationsPerFrameContext": 1.0}


@pytest.mark.skipif(not os.path.exists("/root/reference/tools/bench_aa.py"), reason="needs the Granite checkout")
def test_the_references_aa_bench_script_drives_our_stand_in(tmp_path):
    """Granite's own tools/bench_aa.py, executed: the command line it builds for aa-bench-headless (bench_aa.py:144-148 +
    '--aa-method' per method, :161-166) must parse with the stand-in's argument parser and its run_test / map_result_to_json must
    consume the --stat document.  The stub stands for the GPU run; tests/test_gpu_headless.py runs the real one."""
    stub = tmp_path / "aa-bench-stub"
    stub.write_text(f"""#!{sys.executable}
import json, sys
sys.path.insert(0, {ROOT!r})
from granite_amd import aa_bench, headless
args = aa_bench.parse_args(sys.argv[1:])
assert args.frames == 9 and args.width == 320 and args.height == 200 and args.input_images == ["a.png", "b.png"]
kw = aa_bench.method_to_kwargs(args.aa_method)
cost = 100.0 + 10.0 * kw["post_aa"] + 20.0 * kw["pre_aa"]
json.dump(headless.stat_document(cost, "stub gpu", 7, {{"tonemap": (9, 0.9)}}, args.frames), open(args.stat, "w"))
""")
    stub.chmod(0o755)
    results = tmp_path / "aa.json"
    r = subprocess.run([sys.executable, "/root/reference/tools/bench_aa.py", "--binary", str(stub), "--images", "a.png", "b.png", "--width", "320",
                        "--height", "200", "--frames", "9", "--iterations", "2", "--results", str(results)], capture_output=True, text=True,
                       timeout=180)
    assert r.returncode == 0, r.stderr[-1500:]
    runs = {x["method"]: x for x in json.loads(results.read_text())["runs"]}
    # bench_aa.py:158-159: every method it knows, live in the reference or not
    assert set(runs) == {"none", "fxaa", "fxaa2phase", "smaaLow", "smaaMedium", "smaaHigh", "smaaUltra", "smaaUltraT2X", "taaLow", "taaMedium",
                         "taaHigh", "taaUltra", "taaExtreme", "taaNightmare"}
    from granite_amd import app as gapp
    assert runs["none"]["avg"] == 100.0 and runs["fxaa"]["avg"] == 100.0 + 10.0 * gapp.POST_AA_FXAA
    assert runs["taaHigh"]["avg"] == 100.0 + 20.0 * gapp.POST_AA_TAA_HIGH and runs["taaNightmare"]["avg"] == 100.0
    assert runs["smaaUltra"]["gpu"] == "stub gpu" and runs["smaaUltra"]["version"] == 7 and runs["smaaUltra"]["width"] == 320
