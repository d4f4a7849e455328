"""GPU vs the frozen fixtures under tests/golden/ (the oracle is called once, for the post chain of the executor's own lit target)."""
import os

import numpy as np
import pytest

from granite_amd import app as gapp
from golden.make_golden import FRAMES, inputs
from util import assert_rgba16f_close, assert_rgba8_close

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "oracle_golden_v1.npz")


def test_executor_matches_committed_fixtures():
    ref = np.load(GOLDEN)
    cam, gbuf, descs = inputs()
    a = gapp.Application(cam.width, cam.height)
    a.set_render_parameters(cam.render_params())
    a.set_lights(descs)
    a.upload_gbuffer(gbuf)
    a.render_frames(FRAMES)
    st = a.cluster_state()
    n = st["count"]
    np.testing.assert_array_equal(st["lights"][:n * 48], ref["lights_bytes"])
    np.testing.assert_array_equal(st["type_mask"], ref["type_mask"])
    np.testing.assert_array_equal(st["params"], ref["cluster_params_bytes"])
    np.testing.assert_array_equal(st["light_ranges"], ref["light_ranges"])
    n32 = (n + 31) // 32
    bitmask = a.read("cluster-bitmask").view(np.uint32)[:128 * 64 * n32]
    nz = np.flatnonzero(bitmask)
    np.testing.assert_array_equal(nz, ref["bitmask_nonzero_index"])
    np.testing.assert_array_equal(bitmask[nz], ref["bitmask_nonzero_value"])
    np.testing.assert_array_equal(a.read("cluster-range").view(np.uint32).reshape(-1, 2), ref["range"])
    assert_rgba16f_close(a.read("HDR-main"), ref["hdr"], ulps=2.0, what="HDR-main")
    assert_rgba8_close(a.read_backbuffer(), ref["tonemapped"], 1, what="tonemapped")
    # the hand-over from lighting to the post chain INSIDE this graph: this executor's pyramid levels against the oracle's chain
    # evaluated on this executor's own lit target -- a wrong alias, a missing wait between lighting and threshold or the wrong copy of the
    # rotating HDR-main would show here at 2 ulp + 1e-4, not only in the +-1 LSB of the backbuffer
    from oracle import oracle as orc
    state, chain = {}, None
    own_hdr = a.read("HDR-main").copy()
    for _ in range(FRAMES):
        chain = orc.hdr_chain(own_hdr, state)
    assert_rgba16f_close(a.read("threshold"), chain["threshold"], ulps=2.0, abs_tol=1e-4, what="threshold of the executor's own lit target")
    assert_rgba16f_close(a.read("downsample-3"), chain["d3"], ulps=2.0, abs_tol=1e-4, what="downsample-3 of the executor's own lit target")
    assert_rgba16f_close(a.read("upsample-0"), chain["u0"], ulps=2.0, abs_tol=1e-4, what="upsample-0 of the executor's own lit target")
    np.testing.assert_allclose(a.read("average-luminance").view(np.float32)[0], chain["lum"][0], atol=2e-5)
    a.close()
    # The post chain on the fixture's own HDR target (a second executor without the lighting pass): SURVEY 8a's 2 ulp + 1e-4 on every
    # level -- nothing of the lighting tolerance is carried into it (profiles/r05_pyramid_ulp_histogram_4k.json: no channel of any level
    # beyond 2 ulp + 1e-4 with every rounding difference carried down and up the pyramid).
    b = gapp.Application(cam.width, cam.height, lighting=False)
    b.upload_hdr(ref["hdr"])
    b.render_frames(FRAMES)
    assert_rgba16f_close(b.read("threshold"), ref["threshold"], ulps=2.0, abs_tol=1e-4, what="threshold")
    assert_rgba16f_close(b.read("downsample-3"), ref["d3"], ulps=2.0, abs_tol=1e-4, what="downsample-3")
    assert_rgba16f_close(b.read("upsample-0"), ref["u0"], ulps=2.0, abs_tol=1e-4, what="upsample-0")
    np.testing.assert_allclose(b.read("average-luminance").view(np.float32)[0], ref["lum"][0], atol=2e-5)
    assert_rgba8_close(b.read_backbuffer(), ref["tonemapped"], 1, what="tonemapped (post chain on the fixture's HDR)")
    b.close()
