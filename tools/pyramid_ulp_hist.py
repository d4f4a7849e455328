"""fp16-ulp histograms of every level of the bloom pyramid against the oracle at full size: which tolerance the levels hold (a) when
the oracle's chain starts from the device's lit HDR target and every rounding difference of a level is CARRIED down and up the
pyramid (what tests/test_gpu_fullsize.py and test_gpu_app.py compare: their 4 ulp + 2e-4 on downsample-3 / upsample-0), and (b) stage by
stage, the oracle fed with the device's own input of each stage (what SURVEY 8a's 2 ulp + 1e-4 is stated for).
Usage (GPU box): python tools/pyramid_ulp_hist.py [W H] -> JSON."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from granite_amd import app as gapp, synth
from oracle import oracle as orc
from util import half_bits_to_f32, ulp_fp16, rgba16f_mismatch

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
cam = synth.Camera(w, h)
a = gapp.Application(w, h)
a.set_render_parameters(cam.render_params()); a.set_lights(synth.make_lights(cam, 4096)); a.upload_gbuffer(synth.make_gbuffer(cam))
frames = 2
a.render_frames(frames - 1)
device_history = a.read("downsample-3").copy()  # what the last frame's feedback tap reads
a.render_frames(1)
hdr = a.read("HDR-main").copy()
state, chain = {}, None
for _ in range(frames):
    chain = orc.hdr_chain(hdr, state)


def hist(got, want):
    x = half_bits_to_f32(got).astype(np.float64); y = half_bits_to_f32(want).astype(np.float64)
    fin = np.isfinite(x) & np.isfinite(y)
    d = np.where(fin, np.abs(x - y) / ulp_fp16(np.maximum(np.abs(x), np.abs(y))), 0.0)
    counts, _ = np.histogram(d, [0, 0.5, 1.5, 2.5, 3.5, 4.5, 1e9])
    return {"channels": int(d.size), "ulp_0": int(counts[0]), "ulp_1": int(counts[1]), "ulp_2": int(counts[2]), "ulp_3": int(counts[3]), "ulp_4": int(counts[4]),
            "ulp_gt4": int(counts[5]), "max_ulp": float(d.max()), "beyond_2ulp_plus_1e-4": int(rgba16f_mismatch(got, want, 2.0, 1e-4).sum()),
            "beyond_4ulp_plus_2e-4": int(rgba16f_mismatch(got, want, 4.0, 2e-4).sum())}


names = {"threshold": "threshold", "d0": "downsample-0", "d1": "downsample-1", "d2": "downsample-2", "d3": "downsample-3", "u2": "upsample-2", "u1": "upsample-1",
         "u0": "upsample-0"}
dev = {k: a.read(r).copy() for k, r in names.items()}
result = {"size": [w, h], "frames": frames, "carried": {k: hist(dev[k], chain[k]) for k in names}}
# stage by stage: each oracle stage on the device's own input
sz = {k: (dev[k].shape[1], dev[k].shape[0]) for k in names}
lum_lerp, fb_lerp = orc.frame_lerps(0.01)
stage = {"d0": orc.bloom_downsample(dev["threshold"], *sz["d0"]), "d1": orc.bloom_downsample(dev["d0"], *sz["d1"]), "d2": orc.bloom_downsample(dev["d1"], *sz["d2"]),
         "d3": orc.bloom_downsample(dev["d2"], *sz["d3"], history=device_history, lerp=fb_lerp),
         "u2": orc.bloom_upsample(dev["d3"], *sz["u2"]), "u1": orc.bloom_upsample(dev["u2"], *sz["u1"]), "u0": orc.bloom_upsample(dev["u1"], *sz["u0"])}
result["stage_by_stage"] = {k: hist(dev[k], v) for k, v in stage.items()}
print(json.dumps(result))
a.close()
