"""One rank of the CPU row-band test (launched by test_strips_cpu.py, world_size 2, gloo).

Each rank runs the CPU oracle's post chain but only trusts the rows its StripPlan says it computes: everything outside is
overwritten with a poison value before the next stage reads it, so a halo that is too small, or an all-gather that lands a
chunk in the wrong place, changes the rank's output.  Bands meet through granite_amd.multigpu.all_gather_chunks_inplace on
torch.distributed -- the same call the GPU path makes, on the gloo backend."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

POISON16 = np.float16(60000.0).view(np.uint16)


def keep_rows(img: np.ndarray, rng, poison):
    """Rows outside the plan's range were never computed on this rank."""
    out = np.full_like(img, poison)
    if rng is None:
        return img
    first, count = rng
    out[first:first + count] = img[first:first + count]
    return out


def gather_rows(img: np.ndarray, rank: int, world: int, chunk_rows: int) -> np.ndarray:
    import torch
    from granite_amd import multigpu
    h = img.shape[0]
    row_bytes = int(np.prod(img.shape[1:])) * img.dtype.itemsize
    padded = np.zeros((world * chunk_rows,) + img.shape[1:], img.dtype)
    padded[:h] = img
    flat = torch.from_numpy(padded.reshape(-1).view(np.uint8))  # bytes, exactly what the GPU path gathers
    multigpu.all_gather_chunks_inplace(flat, rank, chunk_rows * row_bytes)
    return padded[:h]


def gather_halo(img: np.ndarray, rank: int, world: int, chunk_rows: int, depth: int) -> np.ndarray:
    """The TAA history under a bounded reach, as host/post/aa.cpp does it: every rank's first and last `depth` rows of its chunk
    meet in one all-gather of 2 * depth rows per rank; a rank takes its upper neighbour's last block and its lower neighbour's
    first block.  Everything else outside the own chunk is poison afterwards."""
    h = img.shape[0]
    own_first, own_end = min(rank * chunk_rows, h), min((rank + 1) * chunk_rows, h)
    staging = np.zeros((world * 2 * depth,) + img.shape[1:], img.dtype)
    staging[2 * depth * rank:2 * depth * rank + depth] = img[own_first:own_first + depth]
    staging[2 * depth * rank + depth:2 * depth * (rank + 1)] = img[own_end - depth:own_end]
    staging = gather_rows(staging, rank, world, 2 * depth)
    out = np.full_like(img, POISON16)
    out[own_first:own_end] = img[own_first:own_end]
    if rank > 0:
        out[own_first - depth:own_first] = staging[2 * depth * (rank - 1) + depth:2 * depth * rank]
    if rank + 1 < world:
        out[own_end:own_end + depth] = staging[2 * depth * (rank + 1):2 * depth * (rank + 1) + depth]
    return out


def edgy_hdr(width: int, height: int) -> np.ndarray:
    """The synthetic HDR frame with long edges laid over it, in three column regions: flat background with nearly vertical
    bars that step sideways by one texel every 66 rows (Z patterns whose far end is 2 x 32 texels away: SMAA's vertical
    searches go to their limit, and the distance they find decides the weights), diagonal stripes of both slopes (the diagonal searches; FXAA's taps lean as far
    as they go), and horizontal bars over the noise."""
    from granite_amd import synth
    hdr = synth.make_hdr(width, height).view(np.float16).astype(np.float32)
    ys, xs = np.mgrid[0:height, 0:width]
    third = width // 3
    rgb = hdr[..., :3] * 0.05
    flat = xs < third
    rgb[flat] = 0.3
    rgb[flat & (((xs - (ys + 10) // 66) // 11) % 3 == 0)] = 6.0
    diag = (xs >= third) & (xs < 2 * third)
    rgb[diag & ((((xs + 2 * ys) // 31) % 4 == 0) | (((3 * xs - ys) // 41) % 5 == 0))] += 6.0
    rgb[(xs >= 2 * third) & ((ys // 37) % 4 == 1)] += 6.0
    hdr[..., :3] = rgb
    return hdr.astype(np.float16).view(np.uint16)


def aa_inputs(width: int, height: int):
    """Depth, motion vectors and the reprojection matrix for the TAA resolve of the `smaa+taa` mode."""
    from granite_amd import synth
    cam = synth.Camera(width, height)
    depth = synth.make_gbuffer(cam)["depth"]
    mv = synth.make_motion_vectors(width, height)
    reproj = np.eye(4, dtype=np.float32)
    reproj[0, 0] = reproj[1, 1] = 0.5
    reproj[0, 3] = reproj[1, 3] = 0.5  # clip -> uv of an unmoved camera
    return depth, mv, np.ascontiguousarray(reproj.T).reshape(-1)  # column-major, as the push constant


MODES = {"none": (0, 0), "fxaa": ("POST_AA_FXAA", 0), "smaa+taa": ("POST_AA_SMAA_ULTRA", "POST_AA_TAA_HIGH"),
         "smaa+taa-halo": ("POST_AA_SMAA_ULTRA", "POST_AA_TAA_HIGH")}
TAA_REACH_ROWS = 5  # make_motion_vectors moves by one row at most, + 3 for the 4 x 4 Catmull-Rom footprint and its rounding


def main():
    import torch.distributed as dist
    from granite_amd import app as gapp, multigpu
    from granite_amd.data import load_smaa_luts
    from oracle import oracle as orc

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    width, height, frames, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "none"
    post_aa, pre_aa = (getattr(gapp, v) if isinstance(v, str) else v for v in MODES[mode])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    halo = mode.endswith("-halo")
    mode = mode.replace("-halo", "")
    plan = multigpu.plan_numpy(rank, world, width, height, post_aa=post_aa, pre_aa=pre_aa, taa_history_reach_rows=TAA_REACH_ROWS if halo else 0)
    assert plan["count"] == world and plan["index"] == rank
    assert not halo or plan["taa_exchange_rows"] > 0

    hdr = edgy_hdr(width, height)
    if mode == "smaa+taa":
        depth, mv, reproj = aa_inputs(width, height)
        area, search = load_smaa_luts()
    taa_history = None
    sz = [orc.level_size(width, height, s) for s in (0.5, 0.25, 0.125, 0.0625, 0.03125)]
    lum_lerp, fb_lerp = orc.frame_lerps(0.01)
    lum, d3_history = np.zeros(3, np.float32), None
    results = []
    for _ in range(frames):
        lit = keep_rows(hdr, plan["lighting"], POISON16)
        if mode == "smaa+taa":
            color, hist = orc.taa_resolve(lit, depth, mv, taa_history, reproj, 2)
            lit = keep_rows(color, plan["taa"], POISON16)
            if halo:
                taa_history = gather_halo(keep_rows(hist, plan["taa"], POISON16), rank, world, plan["out_chunk_rows"], plan["taa_exchange_rows"])
                held = plan["taa_history_held"]
                assert (taa_history[held[0]:held[0] + held[1]] != POISON16).any(axis=(1, 2)).all()
            else:
                taa_history = gather_rows(keep_rows(hist, plan["taa"], POISON16), rank, world, plan["out_chunk_rows"])
        t = keep_rows(orc.bloom_threshold(lit, *sz[0], lum3=lum), plan["threshold"], POISON16)
        d0 = keep_rows(orc.bloom_downsample(t, *sz[1]), plan["d0"], POISON16)
        d1 = keep_rows(orc.bloom_downsample(d0, *sz[2]), plan["d1"], POISON16)
        d1 = gather_rows(d1, rank, world, plan["d1_chunk_rows"])
        d2 = orc.bloom_downsample(d1, *sz[3])
        d3 = orc.bloom_downsample(d2, *sz[4], history=d3_history, lerp=fb_lerp)
        lum = orc.luminance(d3, lum, lum_lerp)
        u2 = orc.bloom_upsample(d3, *sz[3])
        u1 = orc.bloom_upsample(u2, *sz[2])
        u0 = keep_rows(orc.bloom_upsample(u1, *sz[1]), plan["u0"], POISON16)
        tm = keep_rows(orc.tonemap(lit, u0, lum), plan["tonemap"], 0x5A)
        if mode == "fxaa":
            tm = keep_rows(orc.fxaa(tm), plan["aa_out"], 0x5A)
        elif mode == "smaa+taa":
            edges = keep_rows(orc.smaa_edges(tm, 3), plan["smaa_edges"], 0xFF)
            weights = keep_rows(orc.smaa_weights(edges, area, search, 3), plan["smaa_weights"], 0xA5)
            tm = keep_rows(orc.smaa_blend(tm, weights), plan["aa_out"], 0x5A)
        tm = gather_rows(tm, rank, world, plan["out_chunk_rows"])
        d3_history = d3
        results.append((d1.copy(), tm.copy(), lum.copy()))
    np.savez(out_path.format(rank=rank), d1=np.stack([r[0] for r in results]), tm=np.stack([r[1] for r in results]),
             lum=np.stack([r[2] for r in results]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
