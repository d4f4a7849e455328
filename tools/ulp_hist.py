"""Histogram of the fp16-ulp distance between the HIP lighting pass and the oracle (justifies the tolerance the tests state).
Usage (GPU box): python tools/ulp_hist.py [W H LIGHTS [scene]]  -> one JSON line; with a scene name also the channels beyond 2 ulp + 1e-4, one per line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from granite_amd import capi, synth
from oracle import oracle as orc
from gpu_scene import Scene
from util import half_bits_to_f32, ulp_fp16

w, h, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (3840, 2160, 4096)
scene = sys.argv[4] if len(sys.argv) > 4 else "default"
gr = capi.Context(0)
sc = Scene(w, h, n, scene=scene)
ref_c = orc.cluster_build(sc.rp, sc.prm, sc.lights, sc.model, sc.type_mask, sc.n, sc.res[2])
dev = sc.build_clusters_gpu(gr)
flags = capi.LIGHTING_DIRECTIONAL_BIT | capi.LIGHTING_CLUSTERED_BIT | capi.LIGHTING_AMBIENT_FALLBACK_BIT
ref = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, ref_c["bitmask"], ref_c["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
args, imgs = sc.lighting_args(gr, dev, flags)
gr.check(gr.lib.gr_lighting(gr.handle, None, args)); gr.sync()
got = imgs["hdr"].download()
a = half_bits_to_f32(got)[..., :3].astype(np.float64); b = half_bits_to_f32(ref)[..., :3].astype(np.float64)
d = np.abs(a - b) / ulp_fp16(np.maximum(np.abs(a), np.abs(b)))
edges = [0, 0.5, 1.5, 2.5, 3.5, 1e9]
hist, _ = np.histogram(d, edges)
print(json.dumps({"size": [w, h], "lights": n, "scene": scene, "channels": int(d.size), "ulp_0": int(hist[0]), "ulp_1": int(hist[1]), "ulp_2": int(hist[2]),
                  "ulp_3": int(hist[3]), "ulp_gt3": int(hist[4]), "max_ulp": float(d.max()),
                  "frac_exact": float(hist[0] / d.size), "frac_gt1": float(hist[2:].sum() / d.size), "frac_gt2": float(hist[3:].sum() / d.size)}))
if len(sys.argv) > 4:
    from util import rgba16f_mismatch
    bad = np.argwhere(rgba16f_mismatch(got, ref, 2.0, 1e-4)[..., :3])
    for y, x, c in bad[:64]:
        print("beyond 2 ulp + 1e-4: pixel (%d, %d) channel %d: kernel %.7g oracle %.7g = %.2f ulp; depth %.6g normal 0x%08x pbr 0x%04x albedo 0x%08x"
              % (x, y, c, a[y, x, c], b[y, x, c], d[y, x, c], sc.gbuf["depth"][y, x], sc.gbuf["normal"][y, x], sc.gbuf["pbr"][y, x], sc.gbuf["albedo"][y, x]))
gr.close()
