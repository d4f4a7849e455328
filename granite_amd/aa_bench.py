"""Stand-in for Granite's `aa-bench-headless` (tools/aa_bench.cpp) on the HIP executor, with the command line and --stat
file that Granite's own tools/bench_aa.py drives:

    python /path/to/Granite/tools/bench_aa.py --binary tools/aa-bench-headless --images a.png b.png \\
        --width 1920 --height 1080 --frames 100 --results aa.json

    aa-bench-headless --frames N --width W --height H --input-images A B --aa-method M [--scale S] [--stat out.json]

The frame graph is aa_bench.cpp:65-161 (gra_config.aa_bench): "main" blits one of the two input images (alternating per
frame) into the HDR target, the temporal method resolves in front of a blit into the swapchain-sized "tonemap" target, FXAA /
SMAA follow it, and with --scale < 1 FSR 1.0 + sharpen ends the frame.  Run protocol and --stat keys are the headless
application's (application_headless.cpp:581-652): one warm-up frame, N frames, wait idle; averageFrameTimeUs = wall time / N;
per-pass timestamps always on (aa_bench.cpp:155).

--aa-method takes the names of string_to_post_antialiasing_type (aa.cpp:255-290).  bench_aa.py also asks for methods that
have no live implementation in the reference checkout either: fxaa2phase and smaaUltraT2X (their shaders call functions no
header defines any more), taaUltra / taaExtreme / taaNightmare (not in the table: the reference falls through to "none").
The stand-in runs those as "none" and says so on stderr, so a sweep gets through all of its methods.
Inputs: .png (8-bit RGB / RGBA) or .gtx (R8G8B8A8); images of another size than W x H are what the blit is for."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

from . import app as gapp
from . import capi, gtx, headless, png

METHODS = dict(headless.POST_AA)
TEMPORAL = {gapp.POST_AA_TAA_LOW, gapp.POST_AA_TAA_MEDIUM, gapp.POST_AA_TAA_HIGH}
NOT_LIVE = ("fxaa2phase", "smaaUltraT2X", "taaFSR2")


def parse_args(argv):
    ap = argparse.ArgumentParser(prog="aa-bench-headless", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--input-images", nargs=2, default=None, metavar=("IMAGE0", "IMAGE1"))
    ap.add_argument("--aa-method", default="none")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--stat", default="")
    ap.add_argument("--png-reference-path", default="")
    ap.add_argument("--time-step", type=float, default=0.01)
    ap.add_argument("--device", type=int, default=0)
    for ignored in ("--fs-assets", "--fs-builtin", "--fs-cache", "--hw-counter-lib"):
        ap.add_argument(ignored, default="")
    return ap.parse_args(argv)


def load_image(path: str) -> np.ndarray:
    if path.lower().endswith(".gtx"):
        tex = gtx.read(path)
        img = np.ascontiguousarray(tex.level(0)[0])
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 4:
            raise ValueError(f"{path}: an R8G8B8A8 texture is needed")
        return img
    return png.read_png(path)


def method_to_kwargs(name: str) -> dict:
    """--aa-method -> (pre_aa, post_aa): the temporal methods sit in front of the tonemap blit, the others behind it."""
    if name in NOT_LIVE or name not in METHODS:
        print(f"[WARN]: AA method '{name}' has no live implementation in the reference; running 'none'.", file=sys.stderr)
        name = "none"
    value = METHODS[name]
    return {"pre_aa": value, "post_aa": gapp.POST_AA_NONE} if value in TEMPORAL else {"pre_aa": gapp.POST_AA_NONE, "post_aa": value}


def main(argv=None) -> int:
    args = parse_args(sys.argv[1:] if argv is None else argv)
    if args.frames == 0:
        print("[ERROR]: Need to specify --frames for a headless run.", file=sys.stderr)
        return 1
    if not (0.0 < args.scale <= 1.0):
        print("[ERROR]: --scale must be in (0, 1].", file=sys.stderr)
        return 1
    images = None
    if args.input_images:
        try:
            images = [load_image(p) for p in args.input_images]
        except (OSError, ValueError) as e:
            print(f"[ERROR]: Failed to load texture: {e}", file=sys.stderr)
            return 1
        if images[0].shape != images[1].shape:
            print("[ERROR]: The two input images must have one size.", file=sys.stderr)
            return 1
    try:
        app = gapp.Application(args.width, args.height, device=args.device, lighting=False, hdr_bloom=False, dynamic_exposure=False,
                               aa_bench=True, timestamps=True, frame_time=args.time_step, resolution_scale=args.scale,
                               **method_to_kwargs(args.aa_method))
    except capi.GraniteHipError as e:
        print(f"[ERROR]: {e}", file=sys.stderr)
        return 1
    if images:
        app.upload_aa_bench_images(images[0], images[1])
    gpu, driver_version = headless.device_info(app)

    app.render_frames(1)
    app.timestamps()
    app.reset_timestamps()
    print("[INFO]: === Begin run ===")
    start = time.perf_counter_ns()
    app.render_frames(args.frames, sync=False)
    app.sync()
    end = time.perf_counter_ns()
    print("[INFO]: === End run ===")
    usec = 1e-3 * (end - start) / args.frames
    print(f"[INFO]: Average frame time: {usec:.3f} usec")
    stamps = app.timestamps()
    for tag, (count, total_ms) in stamps.items():
        if count:
            print(f"[INFO]: Timestamp tag report: {tag}\n[INFO]:   {total_ms / count:.3f} ms / iteration")
    if args.stat:
        with open(args.stat, "w") as f:
            json.dump(headless.stat_document(usec, gpu, driver_version, stamps, args.frames), f, indent=4)
    if args.png_reference_path:
        png.write_png(args.png_reference_path, headless.backbuffer_rgba8(app))
    app.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
