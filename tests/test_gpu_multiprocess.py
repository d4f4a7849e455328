"""The multi-PROCESS path on the one-GPU box: N ranks = N processes launched exactly as the driver launches bench.py
(torch.distributed.run, 127.0.0.1), all on GPU 0, with the executor's collectives going through tests/rccl_shim (RCCL itself
refuses two ranks on one device).  What runs for the first time on real multi-GPU hardware otherwise -- rendezvous, id broadcast,
two communicators, the in-frame gather, the output gather beside the frame in its RGB888 transport form, max-over-ranks timing,
the JSON line -- runs here, and the assembled frames are compared with the single-process ones."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from granite_amd import app as gapp, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "rccl_shim", "libgranite_rccl_shim.so")


def launch(world, script_args, port, extra_env=None, timeout=420):
    if not os.path.exists(SHIM):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(SHIM)])
    hooks_lib = os.path.join(ROOT, "granite_amd", "lib_testhooks", "libgranite_host.so")
    if not os.path.exists(hooks_lib):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "granite_amd", "csrc"), "testhooks"])
    # the stand-in loader only exists in the -DGRANITE_TEST_HOOKS build of the host layer (granite_amd/lib_testhooks)
    env = dict(os.environ, GRANITE_LIB_DIR="lib_testhooks", GRANITE_RCCL_LIBRARY=SHIM, GRANITE_RCCL_LIBRARY_IS_A_TEST_STAND_IN="1", GRANITE_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), *script_args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("world,w,h,post_aa,pre_aa,reach", [(2, 480, 272, 0, 0, 0), (3, 512, 250, 0, 0, 0), (2, 480, 272, gapp.POST_AA_FXAA, 0, 0),
                                                             (2, 480, 272, gapp.POST_AA_SMAA_ULTRA, gapp.POST_AA_TAA_HIGH, 0),
                                                             (3, 320, 600, gapp.POST_AA_FXAA, gapp.POST_AA_TAA_HIGH, 6)])
def test_ranks_in_separate_processes_assemble_the_single_process_frame(tmp_path, world, w, h, post_aa, pre_aa, reach):
    """reach > 0: the TAA history bands exchange boundary rows only (gra_config.taa_history_reach_rows)."""
    frames, lights = 6, 200
    out = str(tmp_path / "rank{rank}.npz")
    r = launch(world, [os.path.join(ROOT, "tests", "band_worker_gpu.py"), str(w), str(h), str(lights), str(frames), out, str(post_aa), str(pre_aa), str(reach)],
               29600 + (os.getpid() % 300))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    cam = synth.Camera(w, h)
    a = gapp.Application(w, h, post_aa=post_aa, pre_aa=pre_aa)
    if pre_aa:
        a.set_camera(np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16))
    else:
        a.set_render_parameters(cam.render_params())
    a.set_lights(synth.make_lights(cam, lights))
    a.upload_gbuffer(synth.make_gbuffer(cam), synth.make_motion_vectors(w, h) if pre_aa else None)
    a.render_frames(frames)
    want = (a.read_backbuffer().copy(), a.read("downsample-1").copy(), a.read("average-luminance").copy())
    a.close()
    for rank in range(world):
        got = np.load(out.format(rank=rank))
        np.testing.assert_array_equal(got["d1"], want[1], err_msg=f"rank {rank}: 1/8 level")
        np.testing.assert_array_equal(got["lum"], want[2], err_msg=f"rank {rank}: exposure")
        np.testing.assert_array_equal(got["backbuffer"], want[0], err_msg=f"rank {rank}: backbuffer")


def test_bench_runs_as_the_driver_launches_it_with_two_ranks():
    """python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 --steps K --warmup W: one JSON line from rank 0, the
    frame tiled into two row bands (not the replicas fallback), the assembled frame equal to the whole-frame executor's."""
    r = launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--workload", "config2_1080p_256lights",
                   "--no-cpu-baseline", "--sustain-seconds", "0"], 29900 + (os.getpid() % 90))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    # ... and nothing else on stdout (gloo's connection banner goes to stderr)
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines, r.stdout[-2000:]
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["steps"] == 12 and doc["warmup"] == 3 and doc["scaling"] == "weak"
    assert "2 row bands" in doc["config"]["parallelism"] and "RGB888" in doc["config"]["parallelism"], doc["config"]["parallelism"]
    assert doc["bands_checked"] is True
    assert doc["value"] > 0 and doc["ms_per_step"] > 0 and doc["config"]["height"] == 2 * 1080
