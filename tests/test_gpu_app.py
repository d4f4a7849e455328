"""GPU parity, end to end: frames rendered by the RenderGraph executor (C++ host layer + HIP kernels) vs the CPU oracle
driven in the reference's recorded order."""
import os

import numpy as np
import pytest

from granite_amd import app as gapp, capi, synth
from oracle import oracle as orc
from util import assert_rgba16f_close, assert_rgba8_close

pytestmark = pytest.mark.gpu


def oracle_frames(cam, gbuf, descs, frames, res=synth.CLUSTER_RESOLUTION):
    rp = cam.render_params()
    n, lights, model, tmask, _ = orc.pack_lights(descs, rp[99:102])
    prm = orc.cluster_params(rp, *res, n)
    cb = orc.cluster_build(rp, prm, lights, model, tmask, n, res[2])
    hdr = orc.lighting(gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
    state, out = {}, None
    for _ in range(frames):
        out = orc.hdr_chain(hdr, state)
    return {"n": n, "lights": lights, "model": model, "type_mask": tmask, "prm": prm, "cluster": cb, "hdr": hdr, "chain": out}


# The one-frame-ahead light refresh only uses the helper threads from ~1000 lights up (LightClusterer::prefetch); the tests of that
# path keep their small scenes and lower the threshold.  Read once per process, so it is set before the first application exists.
os.environ.setdefault("GRANITE_LIGHT_PREFETCH_MIN", "0")


@pytest.fixture(scope="module")
def scene():
    cam = synth.Camera(480, 270)
    return cam, synth.make_gbuffer(cam), synth.make_lights(cam, 700)


def make_app(cam, gbuf, descs, **kw):
    a = gapp.Application(cam.width, cam.height, **kw)
    a.set_render_parameters(cam.render_params())
    a.set_lights(descs)
    a.upload_gbuffer(gbuf)
    return a


def test_host_light_packing_and_cluster_build_bit_exact(scene):
    cam, gbuf, descs = scene
    ref = oracle_frames(cam, gbuf, descs, 1)
    a = make_app(cam, gbuf, descs)
    a.render_frames(1)
    st = a.cluster_state()
    n = ref["n"]
    assert st["count"] == n
    np.testing.assert_array_equal(st["lights"][:n * 48], ref["lights"].view(np.uint8)[:n * 48])
    np.testing.assert_array_equal(st["models"][:n].view(np.uint32), ref["model"][:n].view(np.uint32))
    np.testing.assert_array_equal(st["type_mask"], ref["type_mask"])
    np.testing.assert_array_equal(st["params"], ref["prm"].view(np.uint8).reshape(-1))
    np.testing.assert_array_equal(st["light_ranges"], ref["cluster"]["light_ranges"])
    n32 = (n + 31) // 32
    bitmask = a.read("cluster-bitmask").view(np.uint32)[:128 * 64 * n32]
    np.testing.assert_array_equal(bitmask, ref["cluster"]["bitmask"])
    np.testing.assert_array_equal(a.read("cluster-range").view(np.uint32).reshape(-1, 2), ref["cluster"]["range"])
    a.close()


@pytest.mark.parametrize("compute_post", [True, False])
def test_frames_match_oracle(scene, compute_post):
    cam, gbuf, descs = scene
    frames = 3
    ref = oracle_frames(cam, gbuf, descs, frames)
    a = make_app(cam, gbuf, descs, compute_post=compute_post)
    a.render_frames(frames)
    assert_rgba16f_close(a.read("HDR-main"), ref["hdr"], ulps=2.0, what="HDR-main")
    names = ({"threshold": "threshold", "downsample-3": "d3", "upsample-0": "u0"} if compute_post else
             {"threshold": "threshold", "bloom-downsample-3": "d3", "bloom-upsample-2": "u0"})
    # the post chain stage by stage on the DEVICE's lit HDR target (static scene: the same target every frame), so that no lighting
    # difference is carried into it: SURVEY 8a's 2 ulp + 1e-4 on every level (profiles/r05_pyramid_ulp_histogram_4k.json)
    state, chain = {}, None
    for _ in range(frames):
        chain = orc.hdr_chain(a.read("HDR-main"), state)
    for res, key in names.items():
        assert_rgba16f_close(a.read(res), chain[key], ulps=2.0, abs_tol=1e-4, what=res)
    lum_name = "average-luminance" if compute_post else "average-luminance-updated"
    lum = a.read(lum_name).view(np.float32)
    np.testing.assert_allclose(lum[0], ref["chain"]["lum"][0], atol=2e-5)
    assert_rgba8_close(a.read_backbuffer(), ref["chain"]["tonemapped"], 1, what="backbuffer")
    a.close()


def test_reference_rmw_declaration_equals_attachment_input_form(scene):
    """lighting declared add_color_output("HDR", info, "emissive") (reference form, G-buffer restored per frame) and the
    5-attachment-input form give bit-identical frames."""
    cam, gbuf, descs = scene
    outs = []
    for rmw in (True, False):
        a = make_app(cam, gbuf, descs, rmw_emissive=rmw)
        a.render_frames(4)
        outs.append((a.read("HDR-main").copy(), a.read_backbuffer().copy(), a.graph()))
        a.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    phys = {w["name"]: w["phys"] for p in outs[0][2]["passes"] for w in p["writes"]}
    assert phys["HDR-main"] == phys["emissive-main"], "RMW output must alias its input's physical image"


def test_config1_bloom_tonemap_only():
    """Config 1: 256x256 bloom + tonemap, no lighting pass."""
    hdr = synth.make_hdr(256, 256)
    a = gapp.Application(256, 256, lighting=False)
    a.upload_hdr(hdr)
    state, ref = {}, None
    for _ in range(5):
        ref = orc.hdr_chain(hdr, state)
    a.render_frames(5)
    assert_rgba16f_close(a.read("upsample-0"), ref["u0"], what="upsample-0")
    assert_rgba8_close(a.read_backbuffer(), ref["tonemapped"], 1, what="config1 backbuffer")
    np.testing.assert_allclose(a.read("average-luminance").view(np.float32)[0], ref["lum"][0], atol=1e-5)
    a.close()


def test_frames_are_deterministic_and_timestamps_reported(scene):
    cam, gbuf, descs = scene
    a = make_app(cam, gbuf, descs, timestamps=True)
    a.render_frames(6)
    first = a.read_backbuffer().copy()
    b = make_app(cam, gbuf, descs)
    b.render_frames(6)
    np.testing.assert_array_equal(first, b.read_backbuffer())
    ts = a.timestamps()
    # tags are the reference's physical passes (render_graph.cpp:2274-2289): merged subpasses are reported as one
    assert {"clustering-bindless", "gbuffer-main + lighting-main", "bloom-compute", "tonemap"} <= set(ts)
    assert all(c == 6 and ms > 0 for c, ms in ts.values())
    a.close()
    b.close()


@pytest.mark.parametrize("kw", [dict(), dict(pre_aa=gapp.POST_AA_TAA_HIGH, post_aa=gapp.POST_AA_FXAA)])
def test_pipelined_frames_equal_synchronised_frames(kw):
    """Frame pipelining (front of frame N+1 on the second stream while the back of frame N is in flight, double-buffered
    hand-over) must not change a single byte: 16 frames enqueued back to back vs the same frames with a full device sync
    after each one."""
    w, h = 960, 540
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, 1024)
    mv = synth.make_motion_vectors(w, h)
    P, V = np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16)
    results = []
    for sync_every_frame in (True, False):
        a = gapp.Application(w, h, **kw)
        a.set_camera(P, V)
        a.set_lights(descs)
        a.upload_gbuffer(gbuf, mv)
        if sync_every_frame:
            for _ in range(16):
                a.render_frames(1, sync=True)
        else:
            a.render_frames(16, sync=False)
            a.sync()
        results.append((a.read_backbuffer().copy(), a.read("average-luminance").copy(), a.read("HDR-main").copy(),
                        a.read("downsample-3").copy()))
        a.close()
    for got, want in zip(results[1], results[0]):
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("lighting", [False, True])
def test_aliased_attachments_do_not_change_a_byte(scene, lighting):
    """Graphics form of the chain: the upsample levels live in the allocations of the dead downsample levels
    (RenderGraph::build_aliases, render_graph.cpp:1548-1746).  Frames must equal those of a graph where every image has
    its own allocation -- across several frames, pipelined -- while the executor holds less HBM."""
    cam, gbuf, descs = scene
    results, held = [], []
    for alias in (False, True):
        if lighting:
            a = make_app(cam, gbuf, descs, compute_post=False, alias_images=alias)
        else:
            a = gapp.Application(480, 270, lighting=False, compute_post=False, alias_images=alias)
            a.upload_hdr(synth.make_hdr(480, 270))
        g = {r["name"]: r for r in a.graph()["resources"]}
        assert (g["bloom-upsample-2"]["alias_of"] == g["bloom-downsample-0"]["phys"]) == alias
        a.render_frames(8)
        a.sync()
        if alias:
            up, down = a.resource("bloom-upsample-2"), a.resource("bloom-downsample-0")
            assert up.device_ptr == down.device_ptr
        results.append((a.read_backbuffer().copy(), a.read("average-luminance").copy(), a.read("bloom-upsample-2").copy(),
                        a.read("bloom-downsample-3").copy()))
        held.append(a.allocated_bytes())
        a.close()
    for got, want in zip(results[1], results[0]):
        np.testing.assert_array_equal(got, want)
    saved = held[0] - held[1]
    w, h = 480, 270
    levels = [((w + 3) // 4) * ((h + 3) // 4), ((w + 7) // 8) * ((h + 7) // 8), ((w + 15) // 16) * ((h + 15) // 16)]
    assert saved == 8 * sum(levels), (held, levels)


def test_ambient_occlusion_input_of_the_lighting_pass(scene):
    """viewer_config "ssao" on the deferred path: "ssao-output-main" (R8_UNORM, filled from an upload in place of FFX CACAO)
    reaches the lighting kernel as LightingParameters::ambient_occlusion.  White until an image is uploaded."""
    cam, gbuf, descs = scene
    plain = make_app(cam, gbuf, descs)
    plain.render_frames(3)
    a = make_app(cam, gbuf, descs, ambient_occlusion=True)
    a.render_frames(3)
    np.testing.assert_array_equal(a.read("HDR-main"), plain.read("HDR-main"))
    np.testing.assert_array_equal(a.read_backbuffer(), plain.read_backbuffer())
    plain.close()
    rng = np.random.default_rng(4)
    ao = rng.integers(0, 256, (cam.height, cam.width), dtype=np.uint8)
    a.upload_ambient_occlusion(ao)
    a.render_frames(3)
    np.testing.assert_array_equal(a.read("ssao-output-main").reshape(cam.height, cam.width), ao)
    o = oracle_frames(cam, gbuf, descs, 0)
    want = orc.lighting(gbuf, cam.render_params(), o["prm"], o["lights"], o["type_mask"], o["cluster"]["bitmask"], o["cluster"]["range"],
                        synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION, ambient_occlusion=ao)
    assert_rgba16f_close(a.read("HDR-main"), want, ulps=2.0, abs_tol=1e-4, what="HDR with ambient occlusion")
    a.close()


def test_light_refresh_done_ahead_by_the_helper_thread_is_the_same_refresh(scene):
    """LightClusterer::refresh of frame N+1 runs on the clusterer's helper thread while frame N is enqueued.  The packed
    lights, parameters and slice intervals a later frame uploads must be exactly what a synchronous refresh produces (= the
    oracle's packing), the prefetch must actually be used, and a light or camera change must fall back to packing in place."""
    cam, gbuf, descs = scene
    ref = oracle_frames(cam, gbuf, descs, 0)
    a = make_app(cam, gbuf, descs)
    a.render_frames(6)
    assert a.prefetched_refreshes() >= 4
    st = a.cluster_state()
    n = ref["n"]
    np.testing.assert_array_equal(st["lights"][:n * 48], ref["lights"].view(np.uint8)[:n * 48])
    np.testing.assert_array_equal(st["models"][:n].view(np.uint32), ref["model"][:n].view(np.uint32))
    np.testing.assert_array_equal(st["type_mask"], ref["type_mask"])
    np.testing.assert_array_equal(st["params"], ref["prm"].view(np.uint8).reshape(-1))
    np.testing.assert_array_equal(st["light_ranges"], ref["cluster"]["light_ranges"])
    first = a.read("HDR-main").copy()
    # change the lights between frames: the stale prefetch must not be used
    fewer = descs[:300].copy()
    a.set_lights(fewer)
    a.render_frames(3)
    ref2 = oracle_frames(cam, gbuf, fewer, 0)
    st2 = a.cluster_state()
    assert st2["count"] == ref2["n"]
    np.testing.assert_array_equal(st2["lights"][:ref2["n"] * 48], ref2["lights"].view(np.uint8)[:ref2["n"] * 48])
    assert_rgba16f_close(a.read("HDR-main"), ref2["hdr"], ulps=2.0, what="HDR-main after a light change")
    assert (a.read("HDR-main") != first).any()
    # move the camera between frames
    cam2 = synth.Camera(cam.width, cam.height, eye=(0.5, 2.2, 8.0))
    a.set_render_parameters(cam2.render_params())
    a.render_frames(2)
    ref3 = oracle_frames(cam2, gbuf, fewer, 0)
    st3 = a.cluster_state()
    np.testing.assert_array_equal(st3["params"], ref3["prm"].view(np.uint8).reshape(-1))
    np.testing.assert_array_equal(st3["lights"][:ref3["n"] * 48], ref3["lights"].view(np.uint8)[:ref3["n"] * 48])
    a.close()


def test_camera_motion_equals_explicit_parameters_and_keeps_the_light_prefetch():
    """gra_set_camera_motion (BASELINE config 4: the camera translates every frame): each frame's render parameters, read back and
    installed verbatim in a second executor, give the same frame bit for bit; the parameters really move; and the clusterer's
    one-frame-ahead light refresh keeps predicting the right camera (no frame falls back to packing on the submitting thread)."""
    w, h, frames = 480, 270, 6
    cam = synth.Camera(w, h)
    gbuf = synth.make_gbuffer(cam)
    descs = synth.make_lights(cam, 600)
    P, V = np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16)
    a = gapp.Application(w, h)
    a.set_camera(P, V)
    a.set_lights(descs)
    a.upload_gbuffer(gbuf)
    a.set_camera_motion((0.01, 0.0, 0.004))
    b = gapp.Application(w, h)
    b.set_lights(descs)
    b.upload_gbuffer(gbuf)
    params = []
    for f in range(frames):
        a.render_frames(1)
        rp = a.get_render_parameters()
        params.append(rp.copy())
        b.set_render_parameters(rp)
        b.render_frames(1)
        np.testing.assert_array_equal(a.read("HDR-main"), b.read("HDR-main"), err_msg=f"frame {f}: lit target")
        np.testing.assert_array_equal(a.read_backbuffer(), b.read_backbuffer(), err_msg=f"frame {f}: backbuffer")
    eye = np.array([p[96:99] for p in params])   # camera_position
    np.testing.assert_allclose(np.diff(eye, axis=0), np.tile([0.01, 0.0, 0.004], (frames - 1, 1)), atol=2e-6)
    assert a.prefetched_refreshes() >= frames - 1
    a.close()
    b.close()

    # with a temporal resolve: the jittered, moving camera still runs with every refresh prefetched
    c = gapp.Application(w, h, pre_aa=gapp.POST_AA_TAA_HIGH)
    c.set_camera(P, V)
    c.set_lights(descs)
    c.upload_gbuffer(gbuf, synth.make_motion_vectors(w, h))
    c.set_camera_motion((0.01, 0.0, 0.0))
    c.render_frames(8)
    assert c.prefetched_refreshes() >= 7
    c.close()


def test_prerecorded_launch_sequences_do_not_change_a_byte():
    """HIP::CommandBuffer::replayable (opt-in, GRANITE_LAUNCH_GRAPHS=1): once their arguments repeat, the bloom pass's launches (ONE at
    this size, gr_bloom_pyramid; three above 640 x 384) go out as one pre-instantiated hipGraph (the cluster build is two launches that read the frame's slot of the pinned staging
    ring since round 3: nothing to pre-record).  Twenty-four pipelined frames with them (a separate
    process with the variable set) equal the same frames with every kernel launched directly, byte for byte; the sequences really
    are replayed; per-kernel timing brackets switch the affected sequence back to direct launches; a moving camera never captures
    the cluster build."""
    import subprocess, sys, os, tempfile
    w, h, frames = 640, 360, 24
    cam = synth.Camera(w, h)
    a = gapp.Application(w, h)
    a.set_render_parameters(cam.render_params())
    a.set_lights(synth.make_lights(cam, 700))
    a.upload_gbuffer(synth.make_gbuffer(cam))
    a.render_frames(frames, sync=False)
    a.sync()
    assert a.launch_graph_replays() == 0  # the default launches every kernel directly
    direct = dict(bb=a.read_backbuffer().copy(), hdr=a.read("HDR-main").copy(), d3=a.read("downsample-3").copy(), lum=a.read("average-luminance").copy())
    a.close()

    out = os.path.join(tempfile.mkdtemp(), "graphs.npz")
    code = f"""
import sys, numpy as np
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from granite_amd import app as gapp, synth
w, h, frames = {w}, {h}, {frames}
cam = synth.Camera(w, h)
gbuf, descs = synth.make_gbuffer(cam), synth.make_lights(cam, 700)
a = gapp.Application(w, h)
a.set_render_parameters(cam.render_params()); a.set_lights(descs); a.upload_gbuffer(gbuf)
a.render_frames(frames, sync=False); a.sync()
# one bloom replay per frame from the second occurrence of its key on: the bloom pass's attachments rotate (feedback history, the
# executor's spare copies of hand-over resources) with a period of a few frames
assert a.launch_graph_replays() >= frames - 12, a.launch_graph_replays()
np.savez({out!r}, bb=a.read_backbuffer(), hdr=a.read("HDR-main"), d3=a.read("downsample-3"), lum=a.read("average-luminance"))
# brackets on a kernel of the bloom sequence: that sequence is launched directly again
k = a.kernel_context()
before = a.launch_graph_replays()
k.timing_set_filter("bloom_pyramid"); k.timing_enable(True); k.timing_reset()
a.render_frames(4)
assert k.timing_query()["bloom_pyramid"][0] == 4
k.timing_enable(False); k.timing_set_filter(None)
assert a.launch_graph_replays() - before == 0, a.launch_graph_replays() - before
a.close()
# a camera that moves every frame: the bloom sequence still replays
m = gapp.Application(w, h)
m.set_camera(np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16))
m.set_lights(descs); m.upload_gbuffer(gbuf); m.set_camera_motion((0.01, 0.0, 0.0))
m.render_frames(frames)
assert frames - 12 <= m.launch_graph_replays() <= frames, m.launch_graph_replays()
m.close()
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GRANITE_LAUNCH_GRAPHS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2500:]
    graphs = np.load(out)
    for key, want in direct.items():
        np.testing.assert_array_equal(graphs[key], want, err_msg=key)


@pytest.mark.parametrize("temporal", [False, True])
def test_saved_state_restores_into_a_fresh_application(temporal):
    """Checkpoint / replay: after 8 frames the state a frame inherits -- exposure adaptation, bloom feedback level, TAA
    history (device side: gra_read_resource / gra_write_resource) and the frame state (elapsed time, swapchain position, moving
    camera, jitter ring: gra_get / set_frame_state) -- goes into a FRESH application, whose next four frames equal frames 9-12 of
    the original run byte for byte."""
    w, h, n = 320, 180, 200
    cam = synth.Camera(w, h)
    gbuf, descs, mv = synth.make_gbuffer(cam), synth.make_lights(cam, n), synth.make_motion_vectors(w, h)
    P, V = np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16)

    def make():
        a = gapp.Application(w, h, pre_aa=gapp.POST_AA_TAA_HIGH if temporal else gapp.POST_AA_NONE)
        a.set_camera(P, V)
        a.set_lights(descs)
        a.upload_gbuffer(gbuf, mv if temporal else None)
        a.set_camera_motion((0.01, 0.0, 0.0))
        return a

    inherited = ["average-luminance", "downsample-3"] + (["HDR-resolved-history"] if temporal else [])
    a = make()
    a.render_frames(8)
    saved = {name: a.read(name).copy() for name in inherited}
    state = a.frame_state()
    want = []
    for _ in range(4):
        a.render_frames(1)
        want.append((a.read_backbuffer().copy(), a.read("average-luminance").copy(), a.read("HDR-main").copy()))
    a.close()

    b = make()
    b.set_frame_state(state)
    for name, data in saved.items():
        b.write(name, data)
    for frame in range(4):
        b.render_frames(1)
        np.testing.assert_array_equal(b.read_backbuffer(), want[frame][0], err_msg=f"backbuffer of resumed frame {9 + frame}")
        np.testing.assert_array_equal(b.read("average-luminance"), want[frame][1])
        np.testing.assert_array_equal(b.read("HDR-main"), want[frame][2])
    # and a fresh application WITHOUT the inherited state does differ (the comparison sees the state)
    c = make()
    c.set_frame_state(state)
    c.render_frames(1)
    assert (c.read_backbuffer() != want[0][0]).any()
    b.close()
    c.close()


def test_cpu_timeline_trace_of_a_few_frames(tmp_path):
    """GRANITE_TIMELINE_TRACE=<file> (the reference's switch, threading/thread_group.cpp:174): the host layer writes a chrome://tracing
    timeline -- frames, graph passes by name, the clusterer's refresh, the helper threads' prefetch, waits -- in the event shape of
    util/timeline_trace_file.cpp.  Read by the environment once per process, hence the subprocess."""
    import json, os, subprocess, sys
    trace = str(tmp_path / "timeline.json")
    code = ("import numpy as np\n"
            "from granite_amd import app as gapp, synth\n"
            "cam = synth.Camera(320, 180)\n"
            "a = gapp.Application(320, 180)\n"
            "a.set_render_parameters(cam.render_params())\n"
            "a.set_lights(synth.make_lights(cam, 100))\n"
            "a.upload_gbuffer(synth.make_gbuffer(cam))\n"
            "a.render_frames(6, sync=True)\n"
            "a.close()\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GRANITE_TIMELINE_TRACE=trace), cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    events = [e for e in json.load(open(trace)) if e["ph"] in "BE"]
    begun = {}
    for e in events:
        if e["ph"] == "B":
            begun.setdefault((e["tid"], e["name"]), 0)
            begun[(e["tid"], e["name"])] += 1
    count = lambda name: sum(n for (tid, nm), n in begun.items() if nm == name)  # noqa: E731
    assert count("render-frame") == 6 and count("enqueue-render-passes") == 6 and count("clusterer-refresh") == 6
    for pass_name in ("clustering-bindless", "lighting-main", "bloom-compute", "tonemap"):
        assert count(pass_name) == 6, (pass_name, sorted({nm for _, nm in begun}))
    assert count("bake-render-graph") == 1 and count("wait-idle") >= 1
    assert any(tid.startswith("light-worker-") for tid, _ in begun)
    assert sum(1 for e in events if e["ph"] == "B") == sum(1 for e in events if e["ph"] == "E")
