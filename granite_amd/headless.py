"""Headless runner for the image-space chain with the command line and --stat file of Granite's `gltf-viewer-headless`
(application/platforms/application_headless.cpp:497-700, viewer/viewer.cpp:44-70), so the reference's own measurement
tooling drives the HIP executor unchanged:

    python tools/sweep_scene.py --viewer-binary tools/granite-hip-headless --scene SCENE --configs cfg/*.json \\
        --width 3840 --height 2160 --frames 200 --timestamp --results sweep.json        # Granite's script, our binary
    python tools/sweep_stat_diff.py --stats vulkan.json sweep.json

    granite-hip-headless SCENE --frames N --width W --height H [--stat out.json] [--timestamp] [--config viewer.json]
        [--png-path prefix] [--png-reference-path out.png] [--gtx-reference-path out.gtx] [--camera-index i] [--lights N]

SCENE is what stands in for the glTF file: a directory holding the G-buffer attachments a Granite build dumped as
`emissive.gtx albedo.gtx normal.gtx pbr.gtx depth.gtx` (+ optional `lights.json`, read_lights()'s format,
scene_viewer_application.cpp:48-138, and `camera.json` {"fovy","aspect","znear","zfar","eye","center"}), or the word
`synthetic` (or any path that is not a directory) for the seeded synthetic G-buffer and lights of the benchmarks.

Protocol = the reference's (application_headless.cpp:581-654): one warm-up frame, wait idle, reset timestamps, N timed
frames, wait idle; `averageFrameTimeUs` = wall time / frames.  --stat keys: averageFrameTimeUs, gpu, driverVersion and,
with --timestamp, performance{pass: timePerAccumulationUs, timePerFrameContextUs, accumulationsPerFrameContext}.
--config understands the viewer_config keys that select image-space work (read_config, scene_viewer_application.cpp:
163-260): renderer, hdrBloom, hdrBloomDynamicExposure, clusteredLights, postAA, resolutionScale, resolutionScaleSharpen,
hdr10, ssao (as the lighting pass's ambient-occlusion input), ssr; keys that concern geometry or shadow passes are accepted and
ignored, a forward renderer or MSAA is refused (no G-buffer for this executor to consume)."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

from . import app as gapp
from . import capi, png, synth

POST_AA = {"none": gapp.POST_AA_NONE, "fxaa": gapp.POST_AA_FXAA, "smaaLow": gapp.POST_AA_SMAA_LOW,
           "smaaMedium": gapp.POST_AA_SMAA_MEDIUM, "smaaHigh": gapp.POST_AA_SMAA_HIGH, "smaaUltra": gapp.POST_AA_SMAA_ULTRA,
           "taaLow": gapp.POST_AA_TAA_LOW, "taaMedium": gapp.POST_AA_TAA_MEDIUM, "taaHigh": gapp.POST_AA_TAA_HIGH}
# Not parity targets: aa_sharpen_resolve.frag (fxaa2phase) and smaa_t2x_resolve.frag (smaaUltraT2X) call convert_input /
# convert_to_output / luminance, which no header of the reference defines any more (they do not compile there either);
# taaFSR2 is the external FidelityFX FSR2 library.
UNSUPPORTED_AA = ("fxaa2phase", "smaaUltraT2X", "taaFSR2")
IGNORED_KEYS = ("directionalLightShadows", "directionalLightShadowsCascaded", "directionalLightShadowsVSM", "PCFKernelWide",
                "clusteredLightsShadows", "clusteredLightsShadowsResolution", "clusteredLightsShadowsVSM", "showUi",
                "forwardDepthPrepass", "shadowMapResolution", "rescaleScene", "lodBias", "debugProbes",
                "cameraIndex", "clusteredLightsBindless", "maxSpotLights", "maxPointLights", "volumetricFog",
                "volumetricDiffuse", "deferredClusteredStencilCulling")


class ConfigError(ValueError):
    pass


def viewer_config_to_kwargs(doc: dict) -> dict:
    """viewer_config JSON -> Application keyword arguments.  Defaults are the viewer's (scene_viewer_application.hpp
    Config: deferred, hdr_bloom on, dynamic exposure on, clustered lights off, post AA none, resolution scale 1)."""
    kw = dict(lighting=True, hdr_bloom=True, dynamic_exposure=True, post_aa=gapp.POST_AA_NONE, resolution_scale=1.0,
              resolution_scale_sharpen=True, hdr10=False, ambient_occlusion=False, ssr=False)
    for key, value in doc.items():
        if key == "renderer":
            if value != "deferred":
                raise ConfigError(f"renderer '{value}': only the deferred renderer has an image-space lighting pass")
        elif key == "msaa":
            if int(value) > 1:
                raise ConfigError("msaa > 1 is a forward-renderer option")
        elif key == "hdrBloom":
            kw["hdr_bloom"] = bool(value)
        elif key == "hdrBloomDynamicExposure":
            kw["dynamic_exposure"] = bool(value)
        elif key == "clusteredLights":
            pass  # positional lights are always clustered here (the path's lighting pass)
        elif key == "postAA":
            if value in UNSUPPORTED_AA:
                raise ConfigError(f"postAA '{value}' is outside the built path")
            if value not in POST_AA:
                raise ConfigError(f"Unrecognized AA type: {value}")
            kw["post_aa"] = POST_AA[value]
        elif key == "resolutionScale":
            kw["resolution_scale"] = float(value)
        elif key == "resolutionScaleSharpen":
            kw["resolution_scale_sharpen"] = bool(value)
        elif key == "hdr10":
            kw["hdr10"] = bool(value)
        elif key == "ssao":
            kw["ambient_occlusion"] = bool(value)
        elif key == "ssr":
            kw["ssr"] = bool(value)  # setup_ssr_pass on the deferred path (scene_viewer_application.cpp:1206-1212)
        elif key == "renderTargetFp16":
            # false = the viewer's default: emissive / HDR-main as B10G11R11_UFLOAT_PACK32 (scene_viewer_application.cpp:881-883).
            # Without the key the runner follows the scene's emissive.gtx (a dump carries its format); the synthetic scene is RGBA16F.
            kw["rt_fp16"] = bool(value)
        elif key in IGNORED_KEYS:
            continue
        else:
            print(f"[WARN]: viewer config key '{key}' is not understood, ignored.", file=sys.stderr)
    if kw["hdr10"]:
        kw["hdr_bloom"] = False  # the viewer's HDR10 branch replaces bloom + tonemap (scene_viewer_application.cpp:1262-1290)
    return kw


def lights_from_json(doc: dict):
    """read_lights() (scene_viewer_application.cpp:48-138) -> (light descriptors, directional dict or None)."""
    spots, points = doc.get("spot", []), doc.get("point", [])
    descs = np.zeros(len(spots) + len(points), synth.LIGHT_DESC_DTYPE)
    for i, l in enumerate(list(spots) + list(points)):
        is_spot = i < len(spots)
        descs[i]["type"] = 0 if is_spot else 1
        descs[i]["color"] = l["color"]
        descs[i]["cutoff_range"] = l.get("range", 0.0)
        tr = np.zeros((3, 4))
        tr[:, :3] = np.eye(3)
        if is_spot:
            descs[i]["inner_cone"], descs[i]["outer_cone"] = l["innerCone"], l["outerCone"]
            fwd = np.asarray(l["direction"], np.float64)
            fwd /= np.linalg.norm(fwd)   # the node's -Z axis looks along "direction" (look_at_arbitrary_up, conjugated)
            helper = np.array([0.0, 1.0, 0.0]) if abs(fwd[1]) < 0.999 else np.array([1.0, 0.0, 0.0])
            z = -fwd
            x = np.cross(helper, z)
            x /= np.linalg.norm(x)
            tr[:, 0], tr[:, 1], tr[:, 2] = x, np.cross(z, x), z
        tr[:, 3] = l["position"]
        descs[i]["transform"] = tr.astype(np.float32)
    directional = None
    if "directional" in doc:
        d = doc["directional"]
        directional = {"direction": [-float(v) for v in d["direction"]], "color": [float(v) for v in d["color"]]}
    return descs, directional


def stat_document(average_frame_time_us: float, gpu: str, driver_version: int, timestamps: dict, frames: int) -> dict:
    """The --stat JSON (application_headless.cpp:627-652).  timestamps: {pass tag: (accumulations, total ms)}."""
    doc = {"averageFrameTimeUs": average_frame_time_us, "gpu": gpu, "driverVersion": int(driver_version)}
    if timestamps:
        perf = {}
        for tag, (count, total_ms) in timestamps.items():
            if count == 0:
                continue
            perf[tag] = {"timePerAccumulationUs": 1e3 * total_ms / count,
                         "timePerFrameContextUs": 1e3 * total_ms / max(frames, 1),
                         "accumulationsPerFrameContext": count / max(frames, 1)}
        doc["performance"] = perf
    return doc


def device_info(app: gapp.Application):
    ctx = app.kernel_context()
    name = C.create_string_buffer(256)
    version = C.c_uint32(0)
    if ctx.lib.gr_get_device_info(ctx.handle, name, len(name), C.byref(version)) < 0:
        raise capi.GraniteHipError(ctx.lib.gr_last_error(ctx.handle).decode())
    return name.value.decode(), version.value


def parse_args(argv):
    ap = argparse.ArgumentParser(prog="granite-hip-headless", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("scene", nargs="?", default="synthetic")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--width", type=int, default=1280)    # application_headless.cpp:489-490
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--time-step", type=float, default=0.01)
    ap.add_argument("--stat", default="")
    ap.add_argument("--png-path", default="")
    ap.add_argument("--png-reference-path", default="")
    ap.add_argument("--gtx-reference-path", default="")
    ap.add_argument("--config", default="")
    ap.add_argument("--quirks", default="")
    ap.add_argument("--timestamp", action="store_true")
    ap.add_argument("--camera-index", type=int, default=-1)
    ap.add_argument("--lights", type=int, default=-1, help="synthetic scene: number of positional lights (default 4096)")
    ap.add_argument("--device", type=int, default=0)
    for ignored in ("--fs-assets", "--fs-builtin", "--fs-cache", "--video-encode-path"):
        ap.add_argument(ignored, default="")
    return ap.parse_args(argv)


def backbuffer_rgba8(app: gapp.Application) -> np.ndarray:
    img = app.read_backbuffer()
    if img.dtype == np.uint32:  # A2B10G10R10: keep the top 8 bits of each channel for the preview file
        return np.stack([((img >> s) & 1023) >> 2 for s in (0, 10, 20)] + [np.full(img.shape, 255)], axis=-1).astype(np.uint8)
    return img


def main(argv=None) -> int:
    args = parse_args(sys.argv[1:] if argv is None else argv)
    if args.frames == 0:
        print("[ERROR]: Need to specify --frames for a headless run.", file=sys.stderr)   # application_headless.cpp:531
        return 1
    doc = {}
    if args.config:
        try:
            with open(args.config) as f:
                doc = json.load(f)
        except OSError:
            print("[ERROR]: Failed to read config file. Assuming defaults.", file=sys.stderr)   # read_config :168-171
    try:
        kw = viewer_config_to_kwargs(doc)
    except ConfigError as e:
        print(f"[ERROR]: {e}", file=sys.stderr)
        return 1

    if args.quirks:
        # read_quirks (scene_viewer_application.cpp:140-165): the one quirk that selects image-space work is
        # useAsyncComputePost -- compute-queue HDR chain (hdr.cpp:312-400) or the 10-pass graphics form (:402-561).
        try:
            with open(args.quirks) as f:
                quirks = json.load(f)
            if "useAsyncComputePost" in quirks:
                kw["compute_post"] = bool(quirks["useAsyncComputePost"])
        except OSError:
            print("[ERROR]: Failed to read quirks file. Assuming defaults.", file=sys.stderr)

    scene_dir = args.scene if os.path.isdir(args.scene) else None
    descs, directional, cam_doc = None, None, {}
    if scene_dir:
        if os.path.exists(os.path.join(scene_dir, "lights.json")):
            with open(os.path.join(scene_dir, "lights.json")) as f:
                descs, directional = lights_from_json(json.load(f))
        if os.path.exists(os.path.join(scene_dir, "camera.json")):
            with open(os.path.join(scene_dir, "camera.json")) as f:
                cam_doc = json.load(f)

    if "rt_fp16" not in kw and scene_dir and os.path.exists(os.path.join(scene_dir, "emissive.gtx")):
        from . import gtx
        kw["rt_fp16"] = gtx.probe(os.path.join(scene_dir, "emissive.gtx")).format != capi.FORMAT_B10G11R11_UFLOAT_PACK32
    try:
        app = gapp.Application(args.width, args.height, device=args.device, timestamps=args.timestamp,
                               frame_time=args.time_step, **kw)
    except capi.GraniteHipError as e:
        print(f"[ERROR]: {e}", file=sys.stderr)
        return 1
    if directional:
        app.set_directional(directional["direction"], directional["color"])
    rw, rh = app.render_size()
    cam = synth.Camera(rw, rh, fovy_deg=math.degrees(cam_doc["fovy"]) if "fovy" in cam_doc else 60.0,
                       near=cam_doc.get("znear", 0.1), far=cam_doc.get("zfar", 100.0),
                       eye=tuple(cam_doc.get("eye", (0.0, 2.0, 8.0))), center=tuple(cam_doc.get("center", (0.0, 1.0, 0.0))))
    app.set_render_parameters(cam.render_params())
    if scene_dir:
        paths = {k: os.path.join(scene_dir, k + ".gtx") for k in ("emissive", "albedo", "normal", "pbr", "depth")}
        mv = os.path.join(scene_dir, "mv.gtx")
        app.upload_gbuffer_gtx(**paths, motion_vectors=mv if os.path.exists(mv) else None)
        app.set_lights(descs if descs is not None else np.zeros(0, synth.LIGHT_DESC_DTYPE))
    else:
        gbuf = synth.make_gbuffer(cam)
        if not kw.get("rt_fp16", True):
            gbuf["emissive"] = synth.pack_b10g11r11(gbuf["emissive"])
        app.upload_gbuffer(gbuf, motion_vectors=synth.make_motion_vectors(rw, rh))
        app.set_lights(synth.make_lights(cam, 4096 if args.lights < 0 else args.lights))

    gpu, driver_version = device_info(app)

    app.render_frames(1)                  # warm-up frame, then wait idle and reset the timestamp log
    app.timestamps()
    app.reset_timestamps()
    print("[INFO]: === Begin run ===")
    start = time.perf_counter_ns()
    rendered = 0
    for frame in range(args.frames):
        app.render_frames(1, sync=False)
        rendered += 1
        if args.png_path:
            app.sync()
            png.write_png(f"{args.png_path}_{frame:05d}.png", backbuffer_rgba8(app))
            print(f"[INFO]:    Queued frame {frame} (Total time = {1e-6 * (time.perf_counter_ns() - start):.3f} ms).")
    app.sync()
    end = time.perf_counter_ns()
    print("[INFO]: === End run ===")

    usec = 1e-3 * (end - start) / rendered
    print(f"[INFO]: Average frame time: {usec:.3f} usec")
    stamps = app.timestamps() if args.timestamp else {}
    for tag, (count, total_ms) in stamps.items():
        if count:
            print(f"[INFO]: Timestamp tag report: {tag}\n[INFO]:   {total_ms / count:.3f} ms / iteration")
    if args.stat:
        with open(args.stat, "w") as f:
            json.dump(stat_document(usec, gpu, driver_version, stamps, rendered), f, indent=4)
    if args.png_reference_path:
        png.write_png(args.png_reference_path, backbuffer_rgba8(app))
    if args.gtx_reference_path:
        app.save_gtx(args.gtx_reference_path)
    app.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
