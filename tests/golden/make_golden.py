"""Generates tests/golden/oracle_golden_v1.npz.

The reference cannot be built or run in this environment (no Vulkan / GLSL toolchain, empty third_party/; SURVEY.md
§8c) and holds no golden vectors for this path, so these fixtures are outputs of the ORACLE on small seeded inputs.
They pin the oracle against regressions and give the GPU tests a second, frozen comparison target; they do not
add independent evidence about the reference ("parity unpinned" stands).

    python -m tests.golden.make_golden        # rewrites the .npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from granite_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

W, H, LIGHTS, FRAMES = 96, 54, 120, 3


def inputs():
    cam = synth.Camera(W, H)
    return cam, synth.make_gbuffer(cam, seed=77), synth.make_lights(cam, LIGHTS, seed=77)


def compute():
    cam, gbuf, descs = inputs()
    rp = cam.render_params()
    n, lights, model, tmask, order = orc.pack_lights(descs, rp[99:102])
    prm = orc.cluster_params(rp, *synth.CLUSTER_RESOLUTION, n)
    cb = orc.cluster_build(rp, prm, lights, model, tmask, n, synth.CLUSTER_RESOLUTION[2])
    hdr = orc.lighting(gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
    state, chain = {}, None
    for _ in range(FRAMES):
        chain = orc.hdr_chain(hdr, state)
    nz = np.flatnonzero(cb["bitmask"])
    return {
        "light_order": order.astype(np.int32),
        "lights_bytes": lights.view(np.uint8)[:n * 48].copy(),
        "type_mask": tmask.copy(),
        "cluster_params_bytes": prm.view(np.uint8).reshape(-1).copy(),
        "light_ranges": cb["light_ranges"].copy(),
        "bitmask_nonzero_index": nz.astype(np.int64),
        "bitmask_nonzero_value": cb["bitmask"][nz].copy(),
        "range": cb["range"].copy(),
        "hdr": hdr,
        "threshold": chain["threshold"],
        "d3": chain["d3"],
        "u0": chain["u0"],
        "lum": chain["lum"].astype(np.float32),
        "tonemapped": chain["tonemapped"],
    }


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden_v1.npz")
    np.savez_compressed(out, **compute())
    print("wrote", out, os.path.getsize(out), "bytes")
