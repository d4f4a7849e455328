"""The headless runner driven the way Granite's tools/sweep_scene.py drives gltf-viewer-headless (command line built as in
sweep_scene.py:94-117, stat file parsed as in run_test :17-37), plus the image_compare gate over what it wrote."""
import json
import os
import subprocess

import numpy as np
import pytest

from granite_amd import app as gapp, gtx, image_compare, png, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "granite-hip-headless")


def sweep(tmp_path, name, config, scene="synthetic", extra=(), width=640, height=360, frames=4):
    cfg = tmp_path / f"{name}.json"
    cfg.write_text(json.dumps(config))
    stat = tmp_path / f"{name}.stat"
    out_png = tmp_path / f"{name}.png"
    cmd = [EXE, scene, "--frames", str(frames), "--width", str(width), "--height", str(height), "--stat", str(stat), "--timestamp",
           "--config", str(cfg), "--png-reference-path", str(out_png), *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    parsed = json.loads(stat.read_text())
    return parsed, str(out_png), r.stdout


def test_sweep_protocol_and_stat_file(tmp_path):
    parsed, out_png, stdout = sweep(tmp_path, "deferred_hdr", {"renderer": "deferred", "hdrBloom": True, "clusteredLights": True},
                                    extra=("--lights", "300"))
    assert "=== Begin run ===" in stdout and "Average frame time:" in stdout
    assert parsed["averageFrameTimeUs"] > 0 and ("MI3" in parsed["gpu"] or "gfx9" in parsed["gpu"]) and parsed["driverVersion"] > 0
    perf = parsed["performance"]
    # physical-pass tags, as the reference reports them: the G-buffer and lighting passes share one VkRenderPass there
    for tag in ("gbuffer-main + lighting-main", "clustering-bindless", "bloom-compute", "tonemap"):
        assert tag in perf, sorted(perf)
        rep = perf[tag]
        assert rep["timePerAccumulationUs"] > 0 and rep["accumulationsPerFrameContext"] == 1.0   # warm-up frame was reset away
        assert rep["timePerFrameContextUs"] == pytest.approx(rep["timePerAccumulationUs"])
    # the frame written by --png-reference-path is the frame the library renders for the same inputs
    cam = synth.Camera(640, 360)
    a = gapp.Application(640, 360)
    a.set_render_parameters(cam.render_params())
    a.set_lights(synth.make_lights(cam, 300))
    a.upload_gbuffer(synth.make_gbuffer(cam), motion_vectors=synth.make_motion_vectors(640, 360))
    a.render_frames(5)   # warm-up + 4
    np.testing.assert_array_equal(png.read_png(out_png), a.read_backbuffer())
    a.close()


def test_psnr_gate_between_configs(tmp_path):
    _, plain, _ = sweep(tmp_path, "plain", {"postAA": "none"}, extra=("--lights", "100"))
    _, again, _ = sweep(tmp_path, "again", {"postAA": "none"}, extra=("--lights", "100"))
    _, fxaa, _ = sweep(tmp_path, "fxaa", {"postAA": "fxaa"}, extra=("--lights", "100"))
    assert image_compare.main([plain, again, "--threshold", "100"]) == 0           # deterministic: identical frames
    psnr = image_compare.compare_images(image_compare.load_image(plain), image_compare.load_image(fxaa))
    assert 20.0 < psnr < 80.0
    assert image_compare.main([plain, fxaa, "--threshold", str(psnr + 1.0)]) == 1


def test_graphics_variant_tags_merged_passes_and_gtx_scene(tmp_path):
    """--quirks useAsyncComputePost=false selects the graphics HDR chain; a scene directory of .gtx attachments + lights.json
    renders the same frame as the arrays it was saved from."""
    w, h = 320, 180
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    scene = tmp_path / "scene"
    scene.mkdir()
    fmts = {"emissive": 97, "albedo": 43, "normal": 64, "pbr": 16, "depth": 126}
    for name, fmt in fmts.items():
        arr = np.ascontiguousarray(gbuf[name])
        gtx.write(str(scene / f"{name}.gtx"), fmt, [arr.view(np.uint8).reshape(h, w, -1)])
    lights = {"directional": {"direction": [-0.3, -1.0, -0.2], "color": [3.0, 3.0, 2.5]},
              "point": [{"color": [20, 5, 5], "range": 6.0, "position": [0.0, 1.5, 2.0]}],
              "spot": [{"innerCone": 0.95, "outerCone": 0.8, "color": [5, 30, 5], "range": 9.0, "position": [1.0, 3.0, 3.0],
                        "direction": [-0.2, -1.0, -0.5]}]}
    (scene / "lights.json").write_text(json.dumps(lights))
    quirks = tmp_path / "quirks.json"
    quirks.write_text(json.dumps({"useAsyncComputePost": False}))
    parsed, out_png, _ = sweep(tmp_path, "gfx", {"hdrBloom": True}, scene=str(scene), width=w, height=h,
                               extra=("--quirks", str(quirks)))
    tags = set(parsed["performance"])
    assert "gbuffer-main + lighting-main" in tags and "bloom-threshold" in tags and "bloom-compute" not in tags, sorted(tags)

    from granite_amd import headless
    descs, directional = headless.lights_from_json(lights)
    b = gapp.Application(w, h, compute_post=False)
    b.set_directional(directional["direction"], directional["color"])
    b.set_render_parameters(cam.render_params())
    b.set_lights(descs)
    b.upload_gbuffer(gbuf)
    b.render_frames(5)
    np.testing.assert_array_equal(png.read_png(out_png), b.read_backbuffer())
    assert len(np.unique(b.read_backbuffer()[..., 1])) > 20
    b.close()


AA_EXE = os.path.join(ROOT, "tools", "aa-bench-headless")


def aa_bench(tmp_path, images, method, frames, width, height, scale=None):
    stat, out_png = tmp_path / f"{method}.stat", tmp_path / f"{method}.png"
    cmd = [AA_EXE, "--frames", str(frames), "--width", str(width), "--height", str(height), "--input-images", *images, "--stat", str(stat),
           "--aa-method", method, "--png-reference-path", str(out_png)]
    if scale is not None:
        cmd += ["--scale", str(scale)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(stat.read_text()), png.read_png(str(out_png)), r


def test_aa_bench_stand_in_runs_the_references_graph(tmp_path):
    """tools/aa-bench-headless with the command line tools/bench_aa.py builds (bench_aa.py:144-148): the graph of
    tools/aa_bench.cpp -- blit of the alternating input images into the HDR target, AA, blit into the swapchain -- checked
    against the oracle's blit / FXAA / SMAA for the frame the run ends on, and the --stat document bench_aa.py reads."""
    from oracle import oracle as orc
    from granite_amd.data import load_smaa_luts
    w, h, frames = 320, 180, 5
    imgs = [synth.make_ldr_pattern(200, 120, seed) for seed in (1, 2)]   # another size than the frame: the blit filters
    paths = []
    for i, img in enumerate(imgs):
        paths.append(str(tmp_path / f"in{i}.png"))
        png.write_png(paths[-1], img)
    last = imgs[frames & 1]   # warm-up frame = image 0, then `frames` more: the input index has advanced `frames` times
    hdr = orc.blit(last, "rgba8_srgb", w, h, "rgba16f", True)
    plain = orc.blit(hdr, "rgba16f", w, h, "rgba8_srgb", False)

    stat, got, _ = aa_bench(tmp_path, paths, "none", frames, w, h)
    np.testing.assert_array_equal(got, plain)
    assert stat["averageFrameTimeUs"] > 0 and stat["driverVersion"] > 0 and {"main", "tonemap"} <= set(stat["performance"])
    assert stat["performance"]["tonemap"]["accumulationsPerFrameContext"] == 1.0

    stat, got, _ = aa_bench(tmp_path, paths, "fxaa", frames, w, h)
    assert np.abs(got.astype(np.int16) - orc.fxaa(plain).astype(np.int16)).max() <= 1
    assert "fxaa" in stat["performance"]

    stat, got, _ = aa_bench(tmp_path, paths, "smaaUltra", frames, w, h)
    assert np.abs(got.astype(np.int16) - orc.smaa(plain, *load_smaa_luts(), 3)["out"].astype(np.int16)).max() <= 1
    assert {"smaa-edge", "smaa-weights", "smaa-blend"} <= set(stat["performance"])

    # temporal methods resolve in front of the blit; the frame differs from the unresolved one and carries the pass's timestamp
    stat, got, _ = aa_bench(tmp_path, paths, "taaHigh", frames, w, h)
    assert "taa-resolve" in stat["performance"] and (got != plain).any() and np.abs(got.astype(np.int16) - plain.astype(np.int16)).mean() < 40

    # methods bench_aa.py sweeps that are not live in the reference either run as "none"
    stat, got, r = aa_bench(tmp_path, paths, "taaNightmare", frames, w, h)
    np.testing.assert_array_equal(got, plain)
    assert "no live implementation" in r.stderr

    # --scale: HDR target at half size, FSR 1.0 + sharpen behind the blit
    stat, got, _ = aa_bench(tmp_path, paths, "none", frames, w, h, scale=0.5)
    assert got.shape == (h, w, 4) and {"post-scale-output-scale", "post-scale-output-sharpen"} <= set(stat["performance"])
