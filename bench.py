#!/usr/bin/env python3
"""bench.py — Mpixels/s of the clustered-light + bloom + tonemap chain at 4K on MI355X (BASELINE.json metric).

A "step" is one frame of the hot path over one synthetic G-buffer (config 3: 3840x2160, 4096 clustered point+spot
lights, 5 reduced bloom levels + luminance + tonemap), run through the RenderGraph executor exactly as the reference's
headless platform runs frames (application_headless.cpp:581-612: warm-up, then timed frames).  Inputs are resident in HBM
before the timed region.  Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (width, height, lights, description)
    "config3_4k_4096lights": (3840, 2160, 4096, "3840x2160, 4096 clustered point+spot lights, bloom pyramid + luminance + tonemap"),
    "config2_1080p_256lights": (1920, 1080, 256, "1920x1080, 256 point lights, full light+post chain"),
    # BASELINE config 1 (the reference's own CPU-runnable case): no lighting pass, HDR input -> bloom-compute -> tonemap
    "config1_256_post_only": (256, 256, 0, "256x256 HDR input, bloom pyramid + luminance + tonemap (no lighting pass)"),
    # BASELINE config 4: config 3 + TAA High in front of the post chain (previous-frame history) + SMAA Ultra behind the tonemap
    "config4_4k_smaa_taa": (3840, 2160, 4096, "3840x2160, 4096 clustered point+spot lights, TAA High (history feedback edge) + bloom pyramid + "
                                              "luminance + tonemap + SMAA Ultra; the camera translates 0.01 units per frame under the 16-phase TAA jitter, motion vectors 0 with a constant-motion region"),
    # config 3 on the reference's default HDR format (viewer_config renderTargetFp16 = false): emissive / HDR-main are B10G11R11_UFLOAT_PACK32
    "config3_4k_4096lights_b10g11r11": (3840, 2160, 4096, "3840x2160, 4096 clustered point+spot lights, bloom pyramid + luminance + tonemap; HDR targets "
                                                           "B10G11R11_UFLOAT_PACK32 (renderTargetFp16 = false, the reference's shipped default)"),
    # BASELINE config 5 as stated: ONE 7680x4320 frame tiled into --gpus row bands (strong scaling; 1 GPU renders it whole)
    "config5_8k": (7680, 4320, 4096, "7680x4320 screen-tiled across the GPUs, 4096 clustered point+spot lights, bloom pyramid + luminance + tonemap"),
}
FIXED_FRAME_WORKLOADS = {"config5_8k"}  # the frame does not grow with the number of ranks

# algorithmic bytes per full-resolution pixel, each pass reading its declared inputs once and writing
# its outputs once (RGBA16F HDR; G-buffer RGBA8 + A2B10G10R10 + RG8 + D32F).
ALGO_BYTES_PER_PX = {
    "lighting": 22.0 + 8.0,
    "bloom_threshold": 8.0 + 2.0,
    "tonemap": 8.5 + 4.0,
    "chain": 56.66,
}
# SURVEY 8d table rows for the other configurations: post only 26.7 B/px; + SMAA (3 passes) 14 + 12; + TAA (RGBA16F everywhere) 24 + 16
# packed HDR targets: lighting 18 + 4, threshold 4 + 2, pyramid and luminance as before (3.33 + 0.008 + 0.82), tonemap 4.5 + 4
CHAIN_BYTES_PER_PX = {"config1_256_post_only": 26.7, "config4_4k_smaa_taa": 56.66 + 26.0 + 40.0, "config3_4k_4096lights_b10g11r11": 40.66}
PACKED_HDR_WORKLOADS = {"config3_4k_4096lights_b10g11r11"}
ALGO_BYTES_PER_PX_PACKED = {"lighting": 18.0 + 4.0, "bloom_threshold": 4.0 + 2.0, "tonemap": 4.5 + 4.0}
SINGLE_GPU_WORKLOADS = {"config1_256_post_only", "config4_4k_smaa_taa", "config3_4k_4096lights_b10g11r11"}  # not tiled by bench.py: N ranks run N replicas
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy); the run measures its own too
VALU_SIMDS, VALU_CLOCK_HZ, VALU_CYCLES_PER_INST = 1024, 2.4e9, 2.0  # 256 CUs x 4 SIMD-32, max clock, wave64 fp32 op
BRACKET_EVERY = 8  # the dominant kernel keeps its hipEvent bracket on every 8th launch of the timed region (short runs: >= 16 brackets, or all)
MIN_BRACKETS = 16
SETTLE_MS = float(os.environ.get("GRANITE_BENCH_SETTLE_MS", "40"))  # of untimed, un-bracketed load immediately in front of the timed region


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config3_4k_4096lights", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="length of the unbracketed run after the timed region (0 = off); 6 s by default so that a 5 s utilisation sampler sees the GPU busy")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the frame the CPU baseline renders (0 = auto)")
    ap.add_argument("--scene", default="default", choices=["default", "depth_split", "hot_spot"],
                    help="synthetic scene (granite_amd/synth.py): default = the scene every headline number is quoted on; depth_split = a third of the "
                         "lighting tiles straddles a 1-7 / 30 unit depth discontinuity; hot_spot = 128 lights on 5 %% of the screen (64-94 in range per pixel)")
    ap.add_argument("--allow-replicas", action="store_true",
                    help="N > 1: when the row-band transport (RCCL) cannot be set up, fall back to N independent replicas instead of failing "
                         "(the line then says so in config.parallelism; it is not a scaling measurement)")
    return ap.parse_args()


def vulkan_probe():
    """The reference's own path as a CPU baseline needs a Vulkan loader + a software ICD (lavapipe) on this host: say whether one
    is there, so that the day it is, the record shows it (none in this image)."""
    import ctypes.util
    import glob
    icds = os.environ.get("VK_ICD_FILENAMES") or os.environ.get("VK_DRIVER_FILES")
    found = glob.glob("/usr/share/vulkan/icd.d/*.json") + glob.glob("/etc/vulkan/icd.d/*.json")
    return {"libvulkan": ctypes.util.find_library("vulkan"), "VK_ICD_FILENAMES": icds, "icd_manifests": found[:4]}


def cpu_baseline(cam, gbuf, descs, width, height, sample_rows):
    """The oracle (CPU restatement of the reference GLSL, OpenMP over rows) timed on this host's cores on a bounded
    sample of the same workload: the full cluster build + the lighting/bloom/tonemap chain on a centred horizontal band
    of `sample_rows` rows of the same G-buffer (same lights, same camera)."""
    from oracle import oracle as orc
    from granite_amd import synth

    cores = os.cpu_count() or 1
    rp = cam.render_params()
    y0 = (height - sample_rows) // 2
    band = {k: np.ascontiguousarray(v[y0:y0 + sample_rows]) for k, v in gbuf.items()}

    # Repeat the frame until ~10 s of wall time have been spent (at least once, at most 16 times) and average.
    t_cluster = t_light = t_post = 0.0
    frames = 0
    state = {}
    reference = None  # the oracle's frame 2 (lit HDR target + backbuffer): what parity_check() holds the GPU frame to
    started = time.perf_counter()
    full = dict(gbuf)
    depth = np.zeros_like(gbuf["depth"])
    depth[y0:y0 + sample_rows] = gbuf["depth"][y0:y0 + sample_rows]
    full = dict(gbuf, depth=depth)
    while frames < 16 and (frames < 2 or time.perf_counter() - started < 10.0):
        t0 = time.perf_counter()
        n, lights, model, tmask, _ = orc.pack_lights(descs, rp[99:102])
        prm = orc.cluster_params(rp, *synth.CLUSTER_RESOLUTION, n)
        cb = orc.cluster_build(rp, prm, lights, model, tmask, n, synth.CLUSTER_RESOLUTION[2])
        t_cluster += time.perf_counter() - t0
        # The band keeps its true screen position: a full-height depth buffer that is sky outside the band.
        t0 = time.perf_counter()
        hdr = orc.lighting(full, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
        t_light += time.perf_counter() - t0
        hdr_band = np.ascontiguousarray(hdr[y0:y0 + sample_rows])
        t0 = time.perf_counter()
        chain = orc.hdr_chain(hdr_band, state)
        t_post += time.perf_counter() - t0
        frames += 1
        if frames == 2 and sample_rows == height:
            reference = {"frames": 2, "hdr": hdr, "tonemapped": chain["tonemapped"].copy()}
    t_cluster, t_light, t_post = t_cluster / frames, t_light / frames, t_post / frames
    del band
    # Extrapolate to a whole frame: the cluster build is paid once per frame, the per-pixel passes scale with rows.
    frame_s = t_cluster + (t_light + t_post) * (height / sample_rows)
    return reference, {
        "value": width * height / frame_s / 1e6,
        "unit": "Mpixels/s",
        "cores": cores,
        "kind": "port",  # the CPU restatement (oracle/): BASELINE.md 3 calls it "cpu-oracle", as opposed to the reference's Vulkan path on lavapipe
        "name": "cpu-oracle",
        "vulkan_icd": vulkan_probe(),
        "sample": f"{frames} frame(s) averaged, {width}x{sample_rows} band of the same G-buffer (all {n} lights, full cluster build): "
                  f"cluster {t_cluster:.2f}s + lighting {t_light:.2f}s + bloom/tonemap {t_post:.2f}s on {cores} OpenMP threads; "
                  f"value = full-frame rate extrapolated as cluster + per-pixel passes x {height}/{sample_rows}",
    }


def cpu_baseline_reference_shaders(cam, gbuf, descs, width, height, band_rows=270):
    """The nearest attainable stand-in for the reference's own CPU path (application_headless.cpp:581-654 on a software Vulkan device, which
    this image cannot run): the reference's GLSL itself -- directional.frag, clustering.frag, the bloom shaders, luminance.comp, tonemap.frag
    -- executed on the host's cores by oracle/_ref/libref_shaders.so (oracle/ref_build: the shader text compiled as C++ against a small
    GLSL environment, rows of invocations spread over the cores by the runner).  Timed on a centred band of the workload's frame (lighting
    + post chain; the cluster build, whose shaders run as teams of real threads there, is taken from the port's timing) and on the whole of
    BASELINE config 1 (256 x 256 bloom + tonemap).  None when the library was not built (it needs /root/reference at build time)."""
    from oracle import oracle as orc
    from granite_amd import synth
    if orc.reference_shader_library() is None:
        return None
    import ctypes as C
    ref = orc.reference_shader_library()
    cores = os.cpu_count() or 1
    rp = cam.render_params()
    band_rows = min(band_rows, height)
    y0 = (height - band_rows) // 2
    depth = np.zeros_like(gbuf["depth"])
    depth[y0:y0 + band_rows] = gbuf["depth"][y0:y0 + band_rows]
    full = dict(gbuf, depth=depth)
    n, lights, model, tmask, _ = orc.pack_lights(descs, rp[99:102])
    prm = orc.cluster_params(rp, *synth.CLUSTER_RESOLUTION, n)
    t0 = time.perf_counter()
    cb = orc.cluster_build(rp, prm, lights, model, tmask, n, synth.CLUSTER_RESOLUTION[2])
    t_cluster = time.perf_counter() - t0
    t_light = t_post = 0.0
    frames, state, started = 0, {}, time.perf_counter()
    while frames < 8 and (frames < 1 or time.perf_counter() - started < 8.0):
        t0 = time.perf_counter()
        hdr = orc.lighting(full, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION,
                           entry=ref.ref_lighting)
        t_light += time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.hdr_chain_reference_shaders(np.ascontiguousarray(hdr[y0:y0 + band_rows]), state)
        t_post += time.perf_counter() - t0
        frames += 1
    t_light, t_post = t_light / frames, t_post / frames
    frame_s = t_cluster + (t_light + t_post) * (height / band_rows)
    # BASELINE config 1 as the reference's headless runner renders it: 256 x 256, bloom + tonemap, whole frame
    small, small_state, small_frames, t_small = synth.make_hdr(256, 256), {}, 0, 0.0
    while small_frames < 20 and (small_frames < 2 or t_small < 1.0):
        t0 = time.perf_counter()
        orc.hdr_chain_reference_shaders(small, small_state)
        t_small += time.perf_counter() - t0
        small_frames += 1
    return {
        "value": width * height / frame_s / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "reference-shaders",
        "name": "the reference's GLSL (assets/shaders/{lights,post}) executed on the host by oracle/_ref/libref_shaders.so",
        "sample": f"{frames} frame(s): lighting {t_light:.2f}s + bloom/tonemap {t_post:.2f}s on a {width}x{band_rows} band (all {n} lights), x {height}/{band_rows}, "
                  f"+ cluster build {t_cluster:.2f}s (the port's; the reference's cluster shaders run as thread teams there) on {cores} OpenMP threads",
        "config1_256x256_post_chain": {"value": 256 * 256 / (t_small / small_frames) / 1e6, "unit": "Mpixels/s", "ms_per_frame": 1e3 * t_small / small_frames,
                                        "frames": small_frames},
    }


def parity_check(cam, gbuf, descs, device, reference):
    """Outside the timed region: a fresh executor renders the frames the oracle rendered in cpu_baseline() and is held to
    the tolerances the tests state (lit HDR target 2 ulp fp16 + 1e-4, backbuffer +-1 LSB).  Returns (ok, detail)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import rgba16f_mismatch
    from granite_amd import app as gapp
    a = gapp.Application(cam.width, cam.height, device=device)
    try:
        a.set_render_parameters(cam.render_params())
        a.set_lights(descs)
        a.upload_gbuffer(gbuf)
        a.render_frames(reference["frames"], sync=True)
        hdr_bad = int(rgba16f_mismatch(a.read("HDR-main"), reference["hdr"], 2.0, 1e-4).sum())
        diff = np.abs(a.read_backbuffer().astype(np.int16) - reference["tonemapped"].astype(np.int16))
        tm_bad = int((diff > 1).sum())
    finally:
        a.close()
    return hdr_bad == 0 and tm_bad == 0, {"hdr_channels_beyond_2ulp": hdr_bad, "backbuffer_bytes_beyond_1lsb": tm_bad,
                                          "frames": reference["frames"]}


def main():
    args = parse_args()
    if os.environ.get("GRANITE_BENCH_WATCHDOG_S"):
        # diagnosis of a run that does not come back: after so many seconds every thread's Python stack goes to stderr and the process exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["GRANITE_BENCH_WATCHDOG_S"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "GRANITE_BENCH_DEVICE" in os.environ:  # test hook: several ranks on one GPU (exercises the replicas fallback)
        local_rank = int(os.environ["GRANITE_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP executor has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        # Control plane only (barriers, the max-over-ranks reduction, shipping the RCCL id): gloo on the host.  The data
        # path's collectives are RCCL all-gathers issued by the executor itself on its own stream (gra_comm_init).
        import torch.distributed as dist
        # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."): the one line this script owes its caller
        # on stdout is the JSON line, so stdout points at stderr while the group comes up and meets for the first time
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from granite_amd import app as gapp, multigpu, synth

    base_width, base_height, num_lights, base_desc = WORKLOADS[args.workload]
    fixed_frame = args.workload in FIXED_FRAME_WORKLOADS

    def build(bands: bool, workload: str = None):
        """One frame tiled into `world` row bands (bands=True) or this rank's own base frame (single GPU / fallback)."""
        workload = workload or args.workload
        base_width, base_height, num_lights, base_desc = WORKLOADS[workload]
        fixed_frame = workload in FIXED_FRAME_WORKLOADS
        width, height, desc, name = base_width, base_height, base_desc, workload
        if bands and fixed_frame:
            # BASELINE config 5 as stated: the same 7680x4320 frame whatever the number of ranks (strong scaling).
            name = f"{workload}_{world}_rowbands"
            desc = f"{width}x{height} tiled into {world} row bands (one per GPU), {num_lights} clustered lights, bloom pyramid + luminance + tonemap"
        elif bands:
            # Weak scaling: one base frame's worth of pixels per rank, the frame tiled into `world` row bands
            # (7680x4320 at 4 ranks is BASELINE config 5's frame); same camera, same 4096 lights, same cluster grid.
            width, height = multigpu.weak_scaled_frame(world, (width, height))
            name = f"{workload}_x{world}_rowbands_{width}x{height}"
            desc = f"{width}x{height} tiled into {world} row bands (one per GPU), {num_lights} clustered lights, bloom pyramid + luminance + tonemap"
        cam = synth.Camera(width, height)
        gbuf = synth.make_gbuffer(cam, scene=args.scene)
        descs = synth.make_lights(cam, num_lights, spot_fraction=0.25 if num_lights > 256 else 0.0, scene=args.scene)
        if args.scene != "default":
            name, desc = f"{name}_scene_{args.scene}", f"{desc}; scene {args.scene} (granite_amd/synth.py)"
        strips = dict(strip_index=rank if bands else 0, strip_count=world if bands else 1,
                      output_gather_rgba=os.environ.get("GRANITE_BENCH_GATHER_RGBA", "0") == "1")
        if workload == "config1_256_post_only":
            app = gapp.Application(width, height, device=local_rank, lighting=False, hdr_bloom=True, dynamic_exposure=True, compute_post=True)
            app.upload_hdr(gbuf["emissive"])
        elif workload == "config4_4k_smaa_taa":
            app = gapp.Application(width, height, device=local_rank, lighting=True, hdr_bloom=True, dynamic_exposure=True, compute_post=True,
                                   pre_aa=gapp.POST_AA_TAA_HIGH, post_aa=gapp.POST_AA_SMAA_ULTRA)
            app.set_camera(np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16))
            app.set_lights(descs)
            app.upload_gbuffer(gbuf, synth.make_motion_vectors(width, height))
            app.set_camera_motion((0.01, 0.0, 0.0))  # SURVEY 8d, config 4 extras: the camera translates 0.01 units per frame
        elif workload in PACKED_HDR_WORKLOADS:
            app = gapp.Application(width, height, device=local_rank, lighting=True, hdr_bloom=True, dynamic_exposure=True, compute_post=True,
                                   rt_fp16=False)
            app.set_render_parameters(cam.render_params())
            app.set_lights(descs)
            app.upload_gbuffer(dict(gbuf, emissive=synth.pack_b10g11r11(gbuf["emissive"])))
        else:
            app = gapp.Application(width, height, device=local_rank, lighting=True, hdr_bloom=True, dynamic_exposure=True,
                                   compute_post=True, **strips)
            app.set_render_parameters(cam.render_params())
            app.set_lights(descs)
            app.upload_gbuffer(gbuf)
        if bands:
            # two RCCL communicators: the 1/8 bloom level meets inside the frame (on the executor's stream), the tonemapped
            # bands beside it (their own stream; GRANITE_BENCH_GATHER=inframe keeps them in the frame for an A/B)
            ids = [gapp.Application.comm_create_unique_id() if rank == 0 else None, gapp.Application.comm_create_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            app.comm_init(ids[0], rank, world)
            if os.environ.get("GRANITE_BENCH_GATHER", "beside") != "inframe":
                app.comm_init_output(ids[1], rank, world)
        # Set-up, not a step: bake the graph and let the executor allocate what it creates lazily (physical images, the
        # spare copies of hand-over resources, its event rings) so that no hipMalloc lands in a counted frame.  With bands
        # this also runs the first all-gathers.
        app.render_frames(4, sync=True)
        return app, cam, gbuf, descs, width, height, desc, name

    bands = world > 1 and os.environ.get("GRANITE_BENCH_MULTI", "bands") != "replicas" and args.workload not in SINGLE_GPU_WORKLOADS
    fallback_reason = None
    if bands:
        # Every rank must end up in the same mode: agree on success through the control plane.
        try:
            built = build(True)
            ok = 1
        except Exception as e:  # noqa: BLE001 - any failure of the RCCL transport or the band set-up
            built, ok, fallback_reason = None, 0, f"{type(e).__name__}: {e}"
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if built is not None:
                built[0].close()
            bands = False
            fallback_reason = fallback_reason or "another rank failed to set up the row-band transport"
            if not args.allow_replicas:
                # N ranks that do not exchange anything are N single-GPU runs: a line from them must not pass for a scaling point
                print(f"[bench] rank {rank}: row-band set-up failed ({fallback_reason}); --allow-replicas runs {world} independent replicas instead",
                      file=sys.stderr)
                dist.barrier()
                dist.destroy_process_group()
                raise SystemExit(3)
            built = build(False)
    else:
        built = build(False)
    application, cam, gbuf, descs, width, height, desc, workload_name = built
    plan = application.strip_plan()
    kctx = application.kernel_context()

    def check_bands(app, cam_, gbuf_, descs_, w_, h_):
        """Outside the timed region: the frame the bands assembled (4 set-up frames so far) against the same frames of ONE executor rendering
        the whole target on rank 0's GPU -- every byte of the backbuffer must agree (GRANITE_BENCH_CHECK_BANDS=0 skips it)."""
        if os.environ.get("GRANITE_BENCH_CHECK_BANDS", "1") != "1":
            return None
        verdict = torch.tensor([1], dtype=torch.int32)
        if rank == 0:
            try:
                whole = gapp.Application(w_, h_, device=local_rank, lighting=True, hdr_bloom=True, dynamic_exposure=True, compute_post=True)
                whole.set_render_parameters(cam_.render_params())
                whole.set_lights(descs_)
                whole.upload_gbuffer(gbuf_)
                whole.render_frames(4, sync=True)
                verdict[0] = int(np.array_equal(whole.read_backbuffer(), app.read_backbuffer()))
                whole.close()
            except Exception as e:  # noqa: BLE001 - e.g. not enough HBM for the whole frame beside the band
                print(f"[bench] band check not possible: {type(e).__name__}: {e}", file=sys.stderr)
                verdict[0] = -1
        dist.broadcast(verdict, src=0)
        return None if int(verdict.item()) < 0 else bool(int(verdict.item()))

    def gather_brackets(per_kernel_, kctx_):
        """Row bands: device time of the two collectives on their streams (hipEvent brackets like every launcher's; "inframe_gather" = the
        1/8 bloom level inside the frame, "output_gather" = the finished bands beside it), from fully bracketed warm-up frames."""
        found = {}
        for name in ("inframe_gather", "output_gather"):
            if name in per_kernel_ and per_kernel_[name][0]:
                found[name] = {"mean": 1000.0 * per_kernel_[name][1] / per_kernel_[name][0], "max": 1000.0 * kctx_.timing_max_ms(name),
                               "brackets": int(per_kernel_[name][0])}
        return found

    def rccl_record(app, plan_, w_, h_, gather_us_, og0_, og1_, rank_ms_):
        """What the communicator itself says (did RCCL see N ranks, which library version, how many bytes each rank sends per frame), the
        per-collective device time -- mean on this rank and the max over the ranks: one slow link shows up as the max -- and whether the
        output gather stayed hidden: `waits` = times a pass of a later frame found the gather of the image it was about to overwrite
        still in flight during the timed steps (0 = every gather had finished before its image was needed again)."""
        info = app.comm_info()
        out_rows = plan_["out_chunk_rows"] if "out_chunk_rows" in plan_ else -(-h_ // world)
        packed_rgb = os.environ.get("GRANITE_BENCH_GATHER_RGBA", "0") != "1"
        info.update({"output_bytes_per_rank": int(out_rows) * w_ * (3 if packed_rgb else 4),
                     "bloom_level_bytes_per_rank": int(plan_["d1_chunk_rows"]) * (w_ // 8) * 8 if "d1_chunk_rows" in plan_ else None,
                     "ms_per_step_over_ranks": rank_ms_})
        for name, key in (("inframe_gather", "inframe_gather_us"), ("output_gather", "output_gather_us")):
            mine = gather_us_.get(name)
            t = torch.tensor([mine["mean"] if mine else -1.0, mine["max"] if mine else -1.0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            info[key] = None if mine is None and float(t[0]) < 0 else {"mean_rank0": mine["mean"] if mine else None, "mean_max_over_ranks": float(t[0]),
                                                                        "max_over_ranks": float(t[1]), "source": "hipEvent brackets on the collective's stream, warm-up frames"}
        waits = torch.tensor([og1_["waits"] - og0_["waits"], og1_["acquires"] - og0_["acquires"]], dtype=torch.int64)
        dist.all_reduce(waits, op=dist.ReduceOp.MAX)
        info["output_gather_waits_in_timed_steps"] = int(waits[0])
        info["overlapped"] = bool(int(waits[1]) > 0 and int(waits[0]) == 0) if os.environ.get("GRANITE_BENCH_GATHER", "beside") != "inframe" else False
        return info

    def over_ranks(seconds, steps):
        """every rank's own clock over the same K steps: the spread says whether one rank (or one link) holds the others up"""
        mine = torch.tensor([seconds], dtype=torch.float64)
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return {"min": 1000.0 * float(lo.item()) / steps, "max": 1000.0 * float(hi.item()) / steps}, float(hi.item())

    bands_checked = check_bands(application, cam, gbuf, descs, width, height) if bands else None

    def barrier():
        torch.cuda.synchronize()
        application.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def hbm_probe():
        # measured HBM ceiling of THIS device in THIS run (1 GiB arrays: beyond the 256 MiB Infinity Cache)
        try:
            return kctx.bandwidth_probe(1 << 30, 5)
        except Exception:  # noqa: BLE001 - e.g. not enough free HBM beside an 8K frame: the spec peak stands alone
            return None, None

    copy_gbs, triad_gbs = hbm_probe()

    # ---- warm-up (also finds the dominant kernel with every launcher bracketed) ----
    kctx.timing_set_sampling(1)
    kctx.timing_enable(True)
    kctx.timing_set_filter(None)
    kctx.timing_reset()
    t_warm = time.perf_counter()
    application.render_frames(max(args.warmup, 1), sync=True)
    warm_ms_per_frame = 1000.0 * (time.perf_counter() - t_warm) / max(args.warmup, 1)
    per_kernel = kctx.timing_query()
    # the dominant kernel = the longest single-launch kernel among those whose algorithmic bytes SURVEY 8d states
    known = {k: v for k, v in per_kernel.items() if k in ALGO_BYTES_PER_PX and k != "chain" and v[0]}
    dominant = max(known.items(), key=lambda kv: kv[1][1] / kv[1][0])[0] if known else "lighting"
    warm_breakdown = {k: {"launches": c, "avg_us": 1000.0 * ms / max(c, 1)} for k, (c, ms) in per_kernel.items()}
    gather_us = gather_brackets(per_kernel, kctx)

    # ---- timed region: only the dominant kernel keeps its hipEvent bracket, on every BRACKET_EVERY-th launch.  (An event
    # pair around a kernel stops the command processor from overlapping it with its neighbours on the stream: bracketing
    # every launch costs the frame ~8 %; the launch duration is still measured live, inside the timed frames.) ----
    kctx.timing_set_filter(dominant)
    kctx.timing_set_sampling(max(1, min(BRACKET_EVERY, args.steps // MIN_BRACKETS)))
    kctx.timing_reset()
    # ---- clock settle: the device reaches its sustained clocks only after tens of milliseconds of uninterrupted load, and the host-side
    # pause above (the warm-up's per-kernel read-out) is enough to lose them again: a timed region that starts cold runs its first ~30
    # frames 7 % slower (profiles/r03_launch_gap_experiments.txt: 20 frames at 0.260 ms cold, 0.248 after 40 frames of load, 0.242 after
    # 160).  So the SAME frames run, un-bracketed and untimed, for about SETTLE_MS right up to the barrier that opens the timed region.
    # They are warm-up beyond the W the command line asks for and are reported as such ("clock_settle"); GRANITE_BENCH_SETTLE_MS=0 = off.
    settle_frames = 0
    if SETTLE_MS > 0:
        kctx.timing_enable(False)
        if dist is None:
            # the host runs at most two frames ahead of the device (Device::next_frame_context), so its clock follows the device's load
            t_settle = time.perf_counter()
            while 1000.0 * (time.perf_counter() - t_settle) < SETTLE_MS and settle_frames < 8000:
                application.render_frames(16, sync=False)
                settle_frames += 16
        else:
            # the ranks render the same number of frames (a frame is a collective): a count agreed on beforehand
            settle_frames = int(min(4000, max(100, -(-SETTLE_MS // max(warm_ms_per_frame, 1e-3)))))
            t = torch.tensor([settle_frames], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            settle_frames = int(t.item())
            application.render_frames(settle_frames, sync=False)
    barrier()
    kctx.timing_enable(True)
    og0 = application.output_gather_stats()
    t0 = time.perf_counter()
    hs0 = application.host_stats()
    application.render_frames(args.steps, sync=False)
    hs1 = application.host_stats()
    og1 = application.output_gather_stats()
    # host side of the frame loop (light sort/pack + launches), excluding time blocked on GPU back-pressure
    host_busy = (hs1["seconds"] - hs0["seconds"]) - (hs1["blocked_seconds"] - hs0["blocked_seconds"])
    barrier()
    elapsed = time.perf_counter() - t0
    timed = kctx.timing_query()
    kctx.timing_enable(False)
    kctx.timing_set_sampling(1)

    # ---- after the timed region: the same loop, unbracketed, kept running for about a second.  Reported beside `value`
    # (never instead of it): a K-step run at ~0.25 ms per step is over before a 1 Hz utilisation sampler sees the GPU busy,
    # and it pays the executor's three-frame pipeline fill once per K frames. ----
    sustained = None
    if args.sustain_seconds > 0:
        # every rank must render the SAME number of frames (each frame is a set of collectives): the count comes from the slowest rank's clock
        agreed = elapsed
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            agreed = float(t.item())
        n_sus = max(args.steps, int(args.sustain_seconds / max(agreed / args.steps, 1e-6)))
        barrier()
        ts0 = time.perf_counter()
        application.render_frames(n_sus, sync=False)
        barrier()
        sus = time.perf_counter() - ts0
        if dist is not None:
            t = torch.tensor([sus], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sus = float(t.item())
        sustained = {"steps": n_sus, "seconds": sus, "ms_per_step": 1000.0 * sus / n_sus}

    rank_ms = None
    if dist is not None:
        rank_ms, elapsed = over_ranks(elapsed, args.steps)

    # one frame per step: at N > 1 its row bands are spread over the ranks; in the replicas fallback every rank renders its own
    pixels_per_step = width * height * (1 if bands or world == 1 else world)
    value = pixels_per_step * args.steps / elapsed / 1e6

    dom_count, dom_ms = timed.get(dominant, (0, 0.0))
    dom_avg_s = (dom_ms / 1000.0) / max(dom_count, 1)
    bpp = (ALGO_BYTES_PER_PX_PACKED if args.workload in PACKED_HDR_WORKLOADS else ALGO_BYTES_PER_PX).get(dominant)
    roofline = None
    if bpp and dom_avg_s > 0:
        # rank 0's launch of the dominant kernel covers its own band only
        band = {"lighting": plan["lighting"], "tonemap": plan["tonemap"], "bloom_threshold": plan["threshold"]}.get(dominant)
        rows = height if band is None else band[1] * (2 if dominant == "bloom_threshold" else 1)
        algo_bytes = bpp * width * rows
        achieved = algo_bytes / dom_avg_s / 1e9
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": algo_bytes,
                    "avg_launch_us": dom_avg_s * 1e6, "launches": dom_count,
                    "launches_bracketed": f"{dom_count} of {args.steps} launches of the timed region carry the hipEvent bracket"}
        if copy_gbs:
            # the practical ceiling beside the spec peak: float4 copy / triad kernels of this run
            roofline["peak_measured"] = max(copy_gbs, triad_gbs)
            roofline["frac_of_measured"] = achieved / max(copy_gbs, triad_gbs)
            roofline["peak_measured_detail"] = {"copy_GBps": copy_gbs, "triad_GBps": triad_gbs, "array_bytes": 1 << 30}
        # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes of this very
        # command, corrected as MI355X_MICROARCH.md prescribes; tools/pmc_passes.sh -> profiles/pmc_traffic.json).  Only
        # quoted for the workload it was collected on.
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            entry = pmc["kernels"].get(dominant)
            if entry and args.workload == "config3_4k_4096lights" and world == 1:
                # Counters of another build of the kernel: the HBM traffic figure is not quoted for it; the instruction counts are, marked
                # stale (they move by a few per cent between builds, the binding resource does not).
                import hashlib
                src = entry.get("source_file")
                current = hashlib.sha256(open(os.path.join(ROOT, src), "rb").read()).hexdigest() if src else None
                stale = current is None or current != entry.get("source_sha256")
                if stale:
                    roofline["traffic_source"] = (f"profiles/pmc_traffic.json is stale: {src or 'the kernel source'} changed since the counters were "
                                                  "collected (tools/pmc_passes.sh regenerates it); traffic not quoted")
                else:
                    roofline["traffic"] = entry["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = "profiles/pmc_traffic.json (" + pmc["source"] + "; " + entry["correction"] + ")"
                valu = entry.get("SQ_INSTS_VALU")
                roofline["valu_instructions_per_launch"] = valu
                if valu:
                    # wave64 VALU instructions issued per launch x 2 cycles against 1024 SIMDs at the 2.4 GHz maximum clock
                    valu_peak_us = 1e6 * valu * VALU_CYCLES_PER_INST / (VALU_SIMDS * VALU_CLOCK_HZ)
                    roofline["valu_peak_us"] = valu_peak_us
                    roofline["valu_issue"] = {"frac": valu_peak_us / (1e6 * dom_avg_s),
                                              "peak": "1024 SIMD-32 x 2.4 GHz, 2 cycles per wave64 fp32 instruction",
                                              "class_histogram": entry.get("valu_class_histogram"), "counters_stale": stale}
                    # What the launch measures (profiles/r06_lighting_residency.txt): ONE vector instruction per SIMD and quad-cycle.  Round 5 printed a
                    # "class-weighted model" here (every opcode class at its stand-alone cost: 118 us); the classes do not add up that way in a mixed
                    # stream (profiles/r06_valu_mix_bench.txt), so the line now carries the rate that the occupancy sweep and the counters confirm.
                    roofline["valu_issue"]["one_per_quad_cycle_us"] = 1e6 * valu * 4.0 / (VALU_SIMDS * 2.04e9)
                    roofline["valu_issue"]["one_per_quad_cycle"] = ("SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.04 GHz, the shader clock measured under this load): "
                                                                   "the launch alone on the machine takes this long; inside the frame the other streams' kernels share the same issue slots")
        except (OSError, KeyError, ValueError):
            pass
        # Config 4: what the anti-aliasing kernels fetch and write against their algorithmic bytes (tools/pmc_aa.sh -> profiles/aa_traffic.json;
        # FETCH_SIZE / WRITE_SIZE passes over the kernels alone at 3840x2160), quoted while the kernels' sources hash to what was measured.
        if args.workload == "config4_4k_smaa_taa" and world == 1:
            try:
                import hashlib
                with open(os.path.join(ROOT, "profiles", "aa_traffic.json")) as f:
                    aa = json.load(f)
                fresh = all(hashlib.sha256(open(os.path.join(ROOT, src), "rb").read()).hexdigest() == digest for src, digest in aa["sources_sha256"].items())
                if fresh:
                    roofline["aa_kernels"] = {k: {"kernel": v["kernel"], "hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                                                  "algorithmic_bytes_per_launch": v["algorithmic_read_bytes"] + v["algorithmic_write_bytes"],
                                                  "fetch_over_algorithmic_reads": v["fetch_over_algorithmic_reads"],
                                                  "write_over_algorithmic_writes": v["write_over_algorithmic_writes"]} for k, v in aa["kernels"].items()}
                    roofline["aa_kernels_source"] = "profiles/aa_traffic.json (" + aa["source"] + "; " + aa["correction"] + ")"
                else:
                    roofline["aa_kernels_source"] = "profiles/aa_traffic.json is stale: an AA kernel source changed since the counters were collected (tools/pmc_aa.sh regenerates it)"
            except (OSError, KeyError, ValueError):
                pass
        # The binding resource of the dominant kernel: whichever ceiling it sits closer to.  `frac` stays the HBM fraction the metric asks for.
        vf = (roofline.get("valu_issue") or {}).get("frac")
        roofline["hbm_frac"] = roofline["frac"]
        if vf is not None and vf > roofline["frac"]:
            roofline["bound"] = "valu"
            roofline["bound_detail"] = ("VALU issue: %.2f of the 2-cycle issue peak against %.2f of the HBM peak; traffic == algorithmic bytes" % (vf, roofline["frac"]))
    chain_bytes = CHAIN_BYTES_PER_PX.get(args.workload, ALGO_BYTES_PER_PX["chain"]) * width * height
    chain_gbs = chain_bytes * args.steps / elapsed / 1e9

    result = {
        "metric": "Mpixels/s for clustered-light+post chain @4K; achieved HBM GB/s vs roofline",
        "value": value,
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if (fixed_frame and world > 1 and bands) else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload_name, "description": desc, "width": width, "height": height, "lights": num_lights,
                   "cluster_grid": list(synth.CLUSTER_RESOLUTION),
                   # which BASELINE.json configuration this line is, in words a reader of a SCALE record needs: at N > 1 the default line
                   # is WEAK scaling (BASELINE config 3's 3840x2160 worth of pixels per rank, one frame of N times that tiled into N row
                   # bands), so that value(N) / (N x value(1)) is an efficiency; BASELINE config 5 (ONE 7680x4320 frame over the same N
                   # ranks, strong scaling) is the `config5_8k` record of the same line (or `--workload config5_8k` at the top level)
                   "baseline_config": ("config 5 (7680x4320 over %d ranks, strong scaling)" % world if (fixed_frame and bands) else
                                       "config 3 x %d (weak scaling: %dx%d = %d frames of 3840x2160; BASELINE config 5 is in config5_8k)" % (world, width, height, world)
                                       if bands and args.workload == "config3_4k_4096lights" else args.workload.split("_")[0].replace("config", "config ")),
                   "parallelism": ("single" if world == 1 else
                                   f"{world} row bands, RCCL all-gather of the 1/8 bloom level (in frame) and of the tonemapped bands as RGB888 (alpha is constant: 3/4 of the bytes per xGMI link; GRANITE_BENCH_GATHER_RGBA=1 sends RGBA8) "
                                   f"({'in frame' if os.environ.get('GRANITE_BENCH_GATHER', 'beside') == 'inframe' else 'beside the frame, own stream + communicator'})" if bands else
                                   f"{world} independent replicas ({'this workload is not tiled by bench.py' if args.workload in SINGLE_GPU_WORKLOADS else f'row-band set-up failed: {fallback_reason}'})"),
                   "hdr_format": ("B10G11R11_UFLOAT_PACK32 (packed-float storage, fp32 arithmetic)" if args.workload in PACKED_HDR_WORKLOADS
                                  else "R16G16B16A16_SFLOAT (fp16 storage, fp32 arithmetic)"), "seed": synth.SEED, "scene": args.scene,
                   "timed_region": {"config1_256_post_only": "bloom pyramid + luminance + tonemap, every frame; the HDR input is resident in HBM",
                                    "config4_4k_smaa_taa": "cluster build + per-frame light refresh + lighting + TAA resolve + bloom pyramid + luminance + "
                                                           "tonemap + SMAA (edges, weights, blend), every frame; the synthetic G-buffer is resident in HBM"}.get(
                       args.workload, "cluster build + per-frame light refresh + lighting + bloom pyramid + luminance + tonemap, every frame; "
                                      "the synthetic G-buffer is resident in HBM (its production is outside the path, as in the reference)")},
        "roofline": roofline,
        "chain": {"algorithmic_GBps": chain_gbs, "frac_of_hbm_peak": chain_gbs / HBM_PEAK_GBS,
                  "algorithmic_bytes_per_frame": chain_bytes},
        "kernels_warmup": warm_breakdown,
        "host_busy_ms_per_step": 1000.0 * host_busy / args.steps,
        # untimed frames in front of the timed region beyond --warmup (see "clock settle" above); 0 frames = switched off
        "clock_settle": {"frames": settle_frames, "target_ms_of_load": SETTLE_MS,
                         "why": "the device needs tens of ms of uninterrupted load to reach its sustained clocks; the warm-up's host-side read-out loses them again"},
    }
    if bands:
        # the assembled frame == the same frames of one executor rendering the whole target (null = check skipped / not possible)
        result["bands_checked"] = bands_checked
        info = rccl_record(application, plan, width, height, gather_us, og0, og1, rank_ms)
        result["rccl"] = info
    if sustained:
        sustained["value"] = pixels_per_step * sustained["steps"] / sustained["seconds"] / 1e6
        sustained["unit"] = "Mpixels/s"
        result["sustained"] = sustained

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload not in SINGLE_GPU_WORKLOADS:
        rows = args.cpu_sample_rows or height  # whole frame: ~4 s on a 256-thread host, ~20 s on 8 cores
        reference, result["cpu_baseline"] = cpu_baseline(cam, gbuf, descs, width, height, rows)
        # beside it, never instead of it: the reference's own shader text executed on the same cores
        result["cpu_baseline_reference_shaders"] = cpu_baseline_reference_shaders(cam, gbuf, descs, width, height)
        # the frame the baseline just rendered is what the GPU frame is compared with (outside the timed region)
        if reference is not None:
            ok, detail = parity_check(cam, gbuf, descs, local_rank, reference)
            result["parity_checked"] = bool(ok)
            result["parity_detail"] = detail
        else:
            result["parity_checked"] = False
            result["parity_detail"] = {"reason": "the CPU baseline rendered a band of the frame only (--cpu-sample-rows)"}
    else:
        result["cpu_baseline"] = None
        result["cpu_baseline_reference_shaders"] = None
        result["parity_checked"] = False
        result["parity_detail"] = {"reason": "no CPU baseline in this run (multi-GPU rank, --no-cpu-baseline, or a workload whose oracle comparison "
                                             "lives in tests/test_gpu_fullsize.py: config 1 / config 4)"}

    application.close()
    # ---- BASELINE config 5 beside the default workload (N > 1): the default line is weak scaling (one 4K frame's worth of pixels per
    # rank); the configuration BASELINE.json names for the multi-GPU case is ONE 7680x4320 frame over the ranks -- strong scaling.  Same
    # contract on a second executor: W warm-up frames, then exactly K frames between two barriers, max over the ranks.
    if bands and args.workload == "config3_4k_4096lights" and os.environ.get("GRANITE_BENCH_CONFIG5", "1") == "1":
        sub = None
        try:
            app5, cam5, gbuf5, descs5, w5, h5, desc5, name5 = build(True, "config5_8k")
            ok = 1
        except Exception as e:  # noqa: BLE001
            app5, ok, sub = None, 0, {"error": f"{type(e).__name__}: {e}"}
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            application = app5  # barrier() syncs the executor it finds under this name
            checked5 = check_bands(app5, cam5, gbuf5, descs5, w5, h5)
            k5 = app5.kernel_context()
            # warm-up with every launcher and both collectives bracketed (their device times), then the clocks (see "clock settle")
            k5.timing_set_sampling(1); k5.timing_enable(True); k5.timing_set_filter(None); k5.timing_reset()
            application.render_frames(max(args.warmup, 1), sync=True)
            gather5 = gather_brackets(k5.timing_query(), k5)
            k5.timing_enable(False)
            application.render_frames(max(args.steps, 40), sync=False)
            barrier()
            og5_0 = app5.output_gather_stats()
            t0 = time.perf_counter()
            application.render_frames(args.steps, sync=False)
            og5_1 = app5.output_gather_stats()
            barrier()
            rank_ms5, slowest = over_ranks(time.perf_counter() - t0, args.steps)
            sub = {"workload": name5, "description": desc5, "width": w5, "height": h5, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": 1000.0 * slowest / args.steps, "value": w5 * h5 * args.steps / slowest / 1e6, "unit": "Mpixels/s",
                   "scaling": "strong", "bands_checked": checked5,
                   "rccl": rccl_record(app5, app5.strip_plan(), w5, h5, gather5, og5_0, og5_1, rank_ms5)}
            app5.close()
        elif app5 is not None:
            app5.close()
        result["config5_8k"] = sub or {"error": "another rank could not set up the 8K frame"}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
