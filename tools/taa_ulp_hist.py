"""fp16-ulp histogram of the TAA resolve against the oracle at 4K: which tolerance does the kernel hold, per quality,
for the resolved colour and for the history target, on the first frame (no history) and with a history -- and where the channels that
leave SURVEY 8a's 2 ulp + 1e-4 sit.  Usage (GPU box): python tools/taa_ulp_hist.py [W H] [packed] -> JSON.
packed: the current colour is a B10G11R11 image and the colour target a B10G11R11 attachment (rt_fp16 = false); the histogram is that
of the RGBA16F history target, the colour target is compared as packed codes (equal / one code apart / more)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from granite_amd import capi, synth
from oracle import oracle as orc
from util import half_bits_to_f32, ulp_fp16, rgba16f_mismatch
from test_gpu_aa import taa_inputs

PACKED = "packed" in sys.argv[1:]
args = [a for a in sys.argv[1:] if a != "packed"]
w, h = (int(args[0]), int(args[1])) if len(args) > 1 else (3840, 2160)
F16 = capi.FORMAT_R16G16B16A16_SFLOAT
B10 = capi.FORMAT_B10G11R11_UFLOAT_PACK32
gr = capi.Context(0)
cur, depth, mv, reproj = taa_inputs(w, h)
if PACKED:
    cur = orc.quantize_b10g11r11(cur)
dcur = capi.DeviceImage(gr, w, h, B10 if PACKED else F16)
upload_cur = (lambda img: dcur.upload(orc.pack_b10g11r11(img))) if PACKED else dcur.upload
upload_cur(cur)
ddepth = capi.DeviceImage(gr, w, h, capi.FORMAT_D32_SFLOAT).upload(depth)
dmv = capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16_SFLOAT).upload(mv)
dcol = capi.DeviceImage(gr, w, h, B10 if PACKED else F16)
dh = [capi.DeviceImage(gr, w, h, F16), capi.DeviceImage(gr, w, h, F16)]


def hist(got, want):
    a = half_bits_to_f32(got).astype(np.float64); b = half_bits_to_f32(want).astype(np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    d = np.where(fin, np.abs(a - b) / ulp_fp16(np.maximum(np.abs(a), np.abs(b))), 0.0)
    edges = [0, 0.5, 1.5, 2.5, 3.5, 1e9]
    counts, _ = np.histogram(d, edges)
    out = {"channels": int(d.size), "ulp_0": int(counts[0]), "ulp_1": int(counts[1]), "ulp_2": int(counts[2]), "ulp_3": int(counts[3]), "ulp_gt3": int(counts[4]),
           "max_ulp": float(d.max()), "beyond_2ulp_plus_1e-4": int(rgba16f_mismatch(got, want, 2.0, 1e-4).sum()),
           "beyond_3ulp_plus_2e-4": int(rgba16f_mismatch(got, want, 3.0, 2e-4).sum())}
    bad = rgba16f_mismatch(got, want, 2.0, 1e-4)
    if bad.any():
        # what the offending channels look like: magnitude of the values and of the absolute difference
        va, vb = a[bad], b[bad]
        out["offenders"] = {"value_min": float(np.minimum(np.abs(va), np.abs(vb)).min()), "value_median": float(np.median(np.abs(vb))),
                            "abs_diff_max": float(np.abs(va - vb).max()), "abs_diff_median": float(np.median(np.abs(va - vb))),
                            "by_channel": [int(bad[..., c].sum()) for c in range(4)],
                            "in_motion_vector_region": float(np.mean(np.any(bad, axis=-1)[:, int(0.45 * w):int(0.55 * w)].sum() / max(np.any(bad, axis=-1).sum(), 1)))}
    return out


def codes(got_words, want16):
    """packed colour target against the oracle's RGBA16F result stored into the same attachment: per-field code distance"""
    want = orc.pack_b10g11r11(want16)
    worst = np.zeros(got_words.shape, np.int64)
    for shift, bits in ((0, 11), (11, 11), (22, 10)):
        a, b = (got_words >> shift) & ((1 << bits) - 1), (want >> shift) & ((1 << bits) - 1)
        worst = np.maximum(worst, np.abs(a.astype(np.int64) - b.astype(np.int64)))
    return {"texels": int(worst.size), "equal": int((worst == 0).sum()), "one_code": int((worst == 1).sum()), "more": int((worst > 1).sum())}


colour = (lambda got, want: codes(got, want)) if PACKED else hist
result = {"size": [w, h], "packed": PACKED}
for q, name in enumerate(("low", "medium", "high")):
    gr.taa_resolve(dcur, ddepth, dmv, None, dcol, dh[0], reproj, q); gr.sync()
    ref_c, ref_h = orc.taa_resolve(cur, depth, mv, None, reproj, q, color_b10g11r11=PACKED)
    r = {"frame0_colour": colour(dcol.download(), ref_c), "frame0_history": hist(dh[0].download(), ref_h)}
    cur2 = synth.make_hdr(w, h, seed=11)
    if PACKED:
        cur2 = orc.quantize_b10g11r11(cur2)
    upload_cur(cur2); dh[0].upload(ref_h)
    gr.taa_resolve(dcur, ddepth, dmv, dh[0], dcol, dh[1], reproj, q); gr.sync()
    ref_c2, ref_h2 = orc.taa_resolve(cur2, depth, mv, ref_h, reproj, q, color_b10g11r11=PACKED)
    r["frame1_colour"] = colour(dcol.download(), ref_c2); r["frame1_history"] = hist(dh[1].download(), ref_h2)
    upload_cur(cur)
    result[name] = r
print(json.dumps(result))
gr.close()
