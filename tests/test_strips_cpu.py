"""CPU: row-band tiling -- the StripPlan arithmetic through the C ABI and the N>1 band exchange on
torch.distributed/gloo with world_size 2 (each rank = one process, like the GPU launch)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from granite_amd import multigpu, synth
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,width,height", [(2, 3840, 4320), (4, 7680, 4320), (8, 7680, 8640), (3, 333, 250), (8, 64, 40)])
def test_plan_partitions_the_frame(world, width, height):
    h_d1 = orc.level_size(width, height, 0.125)[1]
    out_rows, d1_rows = np.zeros(height, int), np.zeros(h_d1, int)
    for rank in range(world):
        p = multigpu.plan_numpy(rank, world, width, height)
        assert (p["index"], p["count"], p["width"], p["height"]) == (rank, world, width, height)
        for name, limit in (("lighting", height), ("threshold", orc.level_size(width, height, 0.5)[1]),
                            ("d0", orc.level_size(width, height, 0.25)[1]), ("d1", h_d1), ("u0", orc.level_size(width, height, 0.25)[1]),
                            ("tonemap", height)):
            first, count = p[name]
            assert first + count <= limit
        f, c = p["tonemap"]
        out_rows[f:f + c] += 1
        assert (f, c) == (min(rank * p["out_chunk_rows"], height), max(0, min((rank + 1) * p["out_chunk_rows"], height) - rank * p["out_chunk_rows"]))
        f, c = p["d1"]
        d1_rows[f:f + c] += 1
        # the lit band contains the tonemapped band (tonemap reads HDR at the same rows)
        lf, lc = p["lighting"]
        tf, tc = p["tonemap"]
        assert tc == 0 or (lf <= tf and lf + lc >= tf + tc)
        assert p["out_chunk_rows"] * world >= height and p["d1_chunk_rows"] * world >= h_d1
    assert (out_rows == 1).all() and (d1_rows == 1).all()


def test_single_rank_plan_is_unrestricted():
    p = multigpu.plan_numpy(0, 1, 1920, 1080)
    assert all(p[k] is None for k in ("lighting", "threshold", "d0", "d1", "u0", "tonemap"))
    assert p["out_chunk_rows"] == 1080 and p["d1_chunk_rows"] == 135


def test_weak_scaled_frames():
    assert [multigpu.weak_scaled_frame(n) for n in (1, 2, 4, 8)] == [(3840, 2160), (3840, 4320), (7680, 4320), (7680, 8640)]


@pytest.mark.parametrize("width,height", [(256, 192), (200, 150)])
def test_two_rank_gloo_bands_equal_single_process_frame(tmp_path, width, height):
    frames = 2
    out = str(tmp_path / "rank{rank}.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + (os.getpid() % 2000)), WORLD_SIZE="2",
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "strip_worker.py"), str(width), str(height), str(frames), out],
                              env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0

    hdr = synth.make_hdr(width, height)
    state = {}
    refs = [orc.hdr_chain(hdr, state) for _ in range(frames)]
    for rank in range(2):
        got = np.load(out.format(rank=rank))
        for f in range(frames):
            np.testing.assert_array_equal(got["d1"][f], refs[f]["d1"], err_msg=f"rank {rank} frame {f}: 1/8 level after all-gather")
            np.testing.assert_array_equal(got["tm"][f], refs[f]["tonemapped"], err_msg=f"rank {rank} frame {f}: tonemapped frame")
            np.testing.assert_array_equal(got["lum"][f], refs[f]["lum"])
