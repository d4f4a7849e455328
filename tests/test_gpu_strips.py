"""GPU parity for row-band tiling: N executor instances on the one available device, one thread per rank,
bands meeting through multigpu.LocalExchange, must reproduce the single-instance frame bit for bit -- every kernel keeps
full-image coordinates, so a band holds the very values of the same rows of a whole-frame launch."""
import ctypes as C
import threading

import numpy as np
import pytest

from granite_amd import app as gapp, capi, multigpu, synth

pytestmark = pytest.mark.gpu


def make_app(w, h, cam, gbuf, descs, mv=None, **kw):
    a = gapp.Application(w, h, **kw)
    if mv is None:
        a.set_render_parameters(cam.render_params())
    else:  # temporal AA jitters the projection itself: it needs the camera, not verbatim render parameters
        a.set_camera(np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16))
    a.set_lights(descs)
    a.upload_gbuffer(gbuf, mv)
    return a


def run_emulated_ranks(world, frames, make, reads, expect_errors=False):
    """`world` instances on the one device, one thread each, bands meeting through LocalExchange; returns got[rank][frame] =
    tuple of the `reads` resources (backbuffer first) and the plans (expect_errors: the errors the ranks raised instead)."""
    lib = capi.load_library()
    lib.gr_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.gr_sync.argtypes = [C.c_void_p, C.c_void_p]
    apps = [make(strip_index=r, strip_count=world) for r in range(world)]
    ctx = apps[0].lib.gra_get_kernel_context(apps[0].handle)

    def copy(dst, src, nbytes, stream):
        assert lib.gr_copy(ctx, stream, dst, src, nbytes) == 0

    def sync(stream):
        assert lib.gr_sync(ctx, stream) == 0

    local = multigpu.LocalExchange(world, copy, sync)
    got = [[] for _ in range(world)]
    errors = []

    def run(rank):
        try:
            a = apps[rank]
            a.set_exchange_callback(local.for_rank(rank))
            for _ in range(frames):
                a.render_frames(1)
                got[rank].append(tuple([a.read_backbuffer().copy()] + [a.read(name).copy() for name in reads]))
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            local.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    if expect_errors:
        for a in apps:
            a.close()
        return errors
    assert not errors, errors
    plans = [a.strip_plan() for a in apps]
    for a in apps:
        a.close()
    return got, plans


@pytest.mark.parametrize("world,w,h,post,pre", [(2, 480, 272, "POST_AA_FXAA", 0), (3, 333, 250, "POST_AA_SMAA_ULTRA", "POST_AA_TAA_HIGH"),
                                                (4, 512, 512, "POST_AA_SMAA_LOW", "POST_AA_TAA_LOW"), (2, 960, 540, "POST_AA_SMAA_HIGH", 0)])
def test_emulated_ranks_with_anti_aliasing_reproduce_the_single_device_frame(world, w, h, post, pre):
    """FXAA / SMAA behind the tonemap and the TAA resolve in front of the post chain under row bands (config
    4's chain tiles).  The halo rows are recomputed, the TAA history bands meet in an all-gather of their own; backbuffer,
    the resolved HDR history and the exposure must be those of the single-instance frames, bit for bit, over 4 frames (the
    history and the jitter sequence carry state from frame to frame)."""
    frames = 4
    kw = dict(post_aa=getattr(gapp, post) if post else 0, pre_aa=getattr(gapp, pre) if pre else 0)
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, 200)
    mv = synth.make_motion_vectors(w, h) if pre else None
    reads = ["average-luminance"] + (["HDR-resolved-history"] if pre else [])

    ref = make_app(w, h, cam, gbuf, descs, mv, **kw)
    want = []
    for _ in range(frames):
        ref.render_frames(1)
        want.append(tuple([ref.read_backbuffer().copy()] + [ref.read(name).copy() for name in reads]))
    ref.close()
    assert any((want[f][0] != want[0][0]).any() for f in range(1, frames)), "frames must differ (exposure, jitter, history)"

    got, plans = run_emulated_ranks(world, frames, lambda **strip: make_app(w, h, cam, gbuf, descs, mv, **kw, **strip), reads)
    for rank in range(world):
        assert plans[rank]["aa_out"] is not None and plans[rank]["tonemap"][1] > plans[rank]["aa_out"][1]
        assert (plans[rank]["taa"] is not None) == bool(pre)
        for f in range(frames):
            for i, name in enumerate(["backbuffer"] + reads):
                np.testing.assert_array_equal(got[rank][f][i], want[f][i], err_msg=f"rank {rank} frame {f}: {name}")


@pytest.mark.parametrize("world,w,h,post,quality", [(2, 480, 544, "POST_AA_SMAA_LOW", "POST_AA_TAA_HIGH"), (3, 320, 600, "POST_AA_FXAA", "POST_AA_TAA_LOW"),
                                                    (4, 256, 1024, 0, "POST_AA_TAA_MEDIUM")])
def test_taa_history_meets_its_neighbours_rows_only_under_a_bounded_reach(world, w, h, post, quality):
    """gra_config.taa_history_reach_rows: the history bands exchange their boundary rows with the neighbouring
    ranks in one small all-gather instead of meeting whole in every rank; frames (backbuffer, exposure) are still the
    single-instance frames bit for bit over 5 frames, and each rank's history is the single instance's on the rows it holds."""
    frames, reach = 5, 6  # make_motion_vectors moves by one row; + the Catmull-Rom footprint, + the jitter
    kw = dict(post_aa=getattr(gapp, post) if post else 0, pre_aa=getattr(gapp, quality))
    cam = synth.Camera(w, h)
    gbuf, descs, mv = synth.make_gbuffer(cam), synth.make_lights(cam, 150), synth.make_motion_vectors(w, h)
    reads = ["average-luminance", "HDR-resolved-history"]
    ref = make_app(w, h, cam, gbuf, descs, mv, **kw)
    want = []
    for _ in range(frames):
        ref.render_frames(1)
        want.append(tuple([ref.read_backbuffer().copy()] + [ref.read(name).copy() for name in reads]))
    ref.close()
    got, plans = run_emulated_ranks(world, frames, lambda **strip: make_app(w, h, cam, gbuf, descs, mv, taa_history_reach_rows=reach, **kw, **strip), reads)
    for rank in range(world):
        depth, held = plans[rank]["taa_exchange_rows"], plans[rank]["taa_history_held"]
        assert 0 < depth < plans[rank]["out_chunk_rows"] and held is not None
        rows = slice(held[0], held[0] + held[1])
        for f in range(frames):
            np.testing.assert_array_equal(got[rank][f][0], want[f][0], err_msg=f"rank {rank} frame {f}: backbuffer")
            np.testing.assert_array_equal(got[rank][f][1], want[f][1], err_msg=f"rank {rank} frame {f}: exposure")
            np.testing.assert_array_equal(got[rank][f][2][rows], want[f][2][rows], err_msg=f"rank {rank} frame {f}: history rows held")


@pytest.mark.filterwarnings("ignore::pytest.PytestUnraisableExceptionWarning")  # the other rank's barrier breaks inside its callback
def test_motion_beyond_the_history_reach_is_reported_not_rendered_silently():
    """A pixel whose motion vector points further than taa_history_reach_rows fetches history rows its rank does not hold: the
    resolve notices, and the next frame / sync / read-back of that rank fails with an error that names the setting."""
    world, w, h, reach = 2, 256, 512, 6
    cam = synth.Camera(w, h)
    gbuf, descs = synth.make_gbuffer(cam), synth.make_lights(cam, 50)
    mv = np.zeros((h, w, 2), np.float32)
    mv[:, :, 1] = -40.0 / h  # history 40 rows further down: the upper band's last rows reach into the lower band
    mv16 = mv.astype(np.float16).view(np.uint16)
    errors = run_emulated_ranks(world, 4, lambda **strip: make_app(w, h, cam, gbuf, descs, mv16, pre_aa=gapp.POST_AA_TAA_HIGH, taa_history_reach_rows=reach, **strip),
                                [], expect_errors=True)
    assert errors and any("taa_history_reach_rows" in message for _, message in errors), errors


@pytest.mark.parametrize("world,w,h,lights", [(2, 480, 272, 300), (3, 333, 250, 200), (4, 512, 512, 64)])
def test_emulated_ranks_reproduce_the_single_device_frame(world, w, h, lights):
    frames = 3
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, lights)

    ref = make_app(w, h, cam, gbuf, descs)
    ref_frames = []
    for _ in range(frames):
        ref.render_frames(1)
        ref_frames.append((ref.read_backbuffer().copy(), ref.read("downsample-1").copy(), ref.read("average-luminance").copy()))
    ref.close()

    lib = capi.load_library()
    lib.gr_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.gr_sync.argtypes = [C.c_void_p, C.c_void_p]
    apps = [make_app(w, h, cam, gbuf, descs, strip_index=r, strip_count=world) for r in range(world)]
    ctx = apps[0].lib.gra_get_kernel_context(apps[0].handle)

    def copy(dst, src, nbytes, stream):
        assert lib.gr_copy(ctx, stream, dst, src, nbytes) == 0

    def sync(stream):
        assert lib.gr_sync(ctx, stream) == 0

    local = multigpu.LocalExchange(world, copy, sync)
    got = [[] for _ in range(world)]
    errors = []

    def run(rank):
        try:
            a = apps[rank]
            a.set_exchange_callback(local.for_rank(rank))
            for _ in range(frames):
                a.render_frames(1)
                got[rank].append((a.read_backbuffer().copy(), a.read("downsample-1").copy(), a.read("average-luminance").copy()))
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            local.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for rank in range(world):
        plan = apps[rank].strip_plan()
        assert plan["count"] == world and plan["tonemap"] is not None
        for f in range(frames):
            np.testing.assert_array_equal(got[rank][f][1], ref_frames[f][1], err_msg=f"rank {rank} frame {f}: downsample-1")
            np.testing.assert_array_equal(got[rank][f][2], ref_frames[f][2], err_msg=f"rank {rank} frame {f}: average-luminance")
            np.testing.assert_array_equal(got[rank][f][0], ref_frames[f][0], err_msg=f"rank {rank} frame {f}: backbuffer")
    for a in apps:
        a.close()


def test_single_rank_rccl_exchange_is_identity():
    """ranks = 1 through the real transport: dlopen(librccl), ncclCommInitRank, the in-place ncclAllGather on the
    executor's stream at both exchange points of a (degenerate) band plan.  The frame must equal the plain frame."""
    w, h = 256, 144
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, 64)
    ref = make_app(w, h, cam, gbuf, descs)
    ref.render_frames(2)
    want = ref.read_backbuffer().copy()
    ref.close()

    a = make_app(w, h, cam, gbuf, descs)
    uid = gapp.Application.comm_create_unique_id()
    assert len(uid) == 128 and any(uid)
    a.comm_init(uid, 0, 1)
    a.render_frames(2)
    np.testing.assert_array_equal(a.read_backbuffer(), want)
    with pytest.raises(capi.GraniteHipError):
        a.comm_init(uid, 0, 1)  # already initialised
    a.close()


def test_single_rank_rccl_exchange_through_the_library_a_torch_process_resolves():
    """bench.py imports torch before it creates the executor, and torch maps the librccl bundled with it (SONAME librccl.so.1): the
    executor's dlopen("librccl.so.1") then finds THAT library already loaded, where this pytest process gets /opt/rocm's.  The same
    one-rank exchange as above in a child process that imports torch first, so that the library the first multi-GPU bench run
    talks to has been through a test; the child says which file it mapped."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import os, sys
import torch  # first, as bench.py does
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from granite_amd import app as gapp, synth
from test_gpu_strips import make_app
w, h = 256, 144
cam = synth.Camera(w, h); gbuf = synth.make_gbuffer(cam); descs = synth.make_lights(cam, 64)
ref = make_app(w, h, cam, gbuf, descs); ref.render_frames(2); want = ref.read_backbuffer().copy(); ref.close()
a = make_app(w, h, cam, gbuf, descs)
a.comm_init(gapp.Application.comm_create_unique_id(), 0, 1)
a.render_frames(2)
assert np.array_equal(a.read_backbuffer(), want)
print("RCCL_VERSION", a.comm_info().get("version"))
a.close()
print("RCCL_MAPPED", sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l}))
print("TORCH_LIB", os.path.join(os.path.dirname(torch.__file__), "lib"))
"""
    r = subprocess.run([sys.executable, "-c", child, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    mapped = [l for l in r.stdout.splitlines() if l.startswith("RCCL_MAPPED")]
    assert mapped and "librccl" in mapped[0], r.stdout
    print(r.stdout)  # -s shows which library a torch process resolves (profiles/ keeps the line of the round's run)


def test_output_gather_beside_the_frame_gives_the_same_frames():
    """The tonemapped bands gathered on the device's collective stream through a second communicator (gra_comm_init_output),
    overlapping the following frames: un-synchronised runs of more frames than there are swapchain images (so that the
    write-after-gather wait is exercised) must leave the same backbuffer, exposure and history as the in-frame form."""
    w, h = 480, 270
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, 200)
    results = []
    for beside in (False, True):
        a = make_app(w, h, cam, gbuf, descs)
        a.comm_init(gapp.Application.comm_create_unique_id(), 0, 1)
        if beside:
            with pytest.raises(capi.GraniteHipError):
                a.comm_init_output(gapp.Application.comm_create_unique_id(), 1, 2)  # does not match this instance's plan
            a.comm_init_output(gapp.Application.comm_create_unique_id(), 0, 1)
        a.render_frames(11, sync=False)
        a.sync()
        results.append((a.read_backbuffer().copy(), a.read("average-luminance").copy(), a.read("downsample-3").copy()))
        a.close()
    for got, want in zip(results[1], results[0]):
        np.testing.assert_array_equal(got, want)
    # without the in-frame communicator there is nothing to move beside the frame
    b = make_app(w, h, cam, gbuf, descs)
    with pytest.raises(capi.GraniteHipError):
        b.comm_init_output(gapp.Application.comm_create_unique_id(), 0, 1)
    b.close()


@pytest.mark.parametrize("w,h", [(64, 16), (333, 21), (3840, 9), (6, 5)])
def test_rgb888_transport_form_round_trips(w, h):
    """gr_pack_rgb8_rows / gr_unpack_rgb8_rows: the 24-bit form the output bands travel in.  Packing rows [a, b) and unpacking them
    into another image restores exactly those rows with alpha 255 and touches nothing else."""
    gr = capi.Context(0)
    lib = gr.lib
    lib.gr_pack_rgb8_rows.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(capi.Image), C.POINTER(capi.Rows), C.c_void_p]
    lib.gr_unpack_rgb8_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(capi.Image), C.POINTER(capi.Rows)]
    rng = np.random.default_rng(w * 131 + h)
    src = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    src[..., 3] = 255
    image = capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8B8A8_SRGB).upload(src)
    other = capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8B8A8_SRGB).upload(np.full((h, w, 4), 7, np.uint8))
    packed = capi.DeviceBuffer(gr, w * h * 3)
    first, count = h // 3, max(h // 2, 1)
    rows = capi.Rows(first, count)
    gr.check(lib.gr_pack_rgb8_rows(gr.handle, None, image.desc, rows, packed.ptr))
    gr.check(lib.gr_unpack_rgb8_rows(gr.handle, None, packed.ptr, other.desc, rows))
    gr.sync()
    got = other.download()
    np.testing.assert_array_equal(got[first:first + count], src[first:first + count])
    assert (got[:first] == 7).all() and (got[first + count:] == 7).all()
    raw = packed.download(np.uint8).reshape(h, w, 3)
    np.testing.assert_array_equal(raw[first:first + count], src[first:first + count, :, :3])


def test_packed_and_rgba_output_gathers_give_the_same_frames():
    """The finished bands travel as RGB888 by default (gra_config.output_gather_rgba = 0); sending the RGBA8 rows instead must not
    change a byte of any rank's frame."""
    world, w, h, frames = 3, 332, 250, 3
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, 128)
    results = []
    for rgba in (False, True):
        got, plans = run_emulated_ranks(world, frames, lambda **strip: make_app(w, h, cam, gbuf, descs, output_gather_rgba=rgba, **strip), [])
        results.append(got)
    for rank in range(world):
        for f in range(frames):
            np.testing.assert_array_equal(results[0][rank][f][0], results[1][rank][f][0])
            assert (results[0][rank][f][0][..., 3] == 255).all()
    for rank in range(1, world):
        np.testing.assert_array_equal(results[0][rank][-1][0], results[0][0][-1][0])
