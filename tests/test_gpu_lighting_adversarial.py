"""GPU parity of the lighting kernel on inputs built to hit the clamps and slice decisions of the reference's BRDF that the
synthetic benchmark scene never reaches:

  * a light (almost) on the -V axis behind a surface whose normal faces away from the camera: HoV = |V + L| / 2 drops below
    the reference's lower clamp 0.001 (point.h:128, spot.h:131) while NoL ~ 1, so the light dominates the pixel and
    f = (1 - HoV)^5 decides the result (0.995 with the clamp, 1.0 without: five fp16 ulps);
  * roughness byte 0 with N perpendicular to both V and L: NoV and NoL at their 0.001 clamps, the smallest Gv * Gl the
    material model can produce (the reference's max(Gv Gl, 0.001) must not be what decides the value);
  * un-normalised normals (clustering.frag:35 does not renormalise): |N| up to sqrt(3), so NoL and NoH exceed 1 and their
    upper clamps engage;
  * a surface lying exactly on cluster Z-slice boundaries, lit by lights whose radius ends exactly there and by lights that
    straddle it: whichever neighbouring slice a pixel is put in, the same lights reach it.

Every case is a whole small frame compared with the oracle at the lighting tolerance (2 ulp fp16 + 1e-4)."""
import numpy as np
import pytest

from granite_amd import capi, synth
from oracle import oracle as orc
from gpu_scene import Scene
from util import assert_rgba16f_close

pytestmark = pytest.mark.gpu

W, H = 256, 128


def world_positions(cam, view_z):
    h, w = view_z.shape
    depth = cam.depth_from_view_distance(view_z)
    ys, xs = np.mgrid[0:h, 0:w]
    ndc = np.stack([2.0 * (xs + 0.5) / w - 1.0, 2.0 * (ys + 0.5) / h - 1.0, depth.astype(np.float64), np.ones((h, w))], axis=0).reshape(4, -1)
    clip = cam.invVP @ ndc
    return depth, (clip[:3] / clip[3]).T.reshape(h, w, 3)


def encode_normals(n):
    q = np.clip(np.rint((0.5 * n + 0.5) * 1023.0), 0, 1023).astype(np.uint32)
    return (q[..., 0] | (q[..., 1] << 10) | (q[..., 2] << 20) | np.uint32(3 << 30)).astype(np.uint32)


def point_lights(positions, colour, radius):
    d = np.zeros(len(positions), synth.LIGHT_DESC_DTYPE)
    d["type"] = 1
    d["color"] = colour
    d["inner_cone"], d["outer_cone"] = np.cos(np.radians(20.0)), np.cos(np.radians(30.0))
    d["cutoff_range"] = radius
    tr = np.zeros((len(positions), 3, 4))
    tr[:, :, :3] = np.eye(3)
    tr[:, :, 3] = positions
    d["transform"] = tr.astype(np.float32)
    return d


def run_case(gr, sc, descs, what, flags=capi.LIGHTING_CLUSTERED_BIT, **oracle_kw):
    sc.descs = descs
    sc.n, sc.lights, sc.model, sc.type_mask, sc.order = orc.pack_lights(descs, sc.rp[99:102])
    sc.prm = orc.cluster_params(sc.rp, sc.res[0], sc.res[1], sc.res[2], sc.n)
    ref_c = orc.cluster_build(sc.rp, sc.prm, sc.lights, sc.model, sc.type_mask, sc.n, sc.res[2])
    dev = sc.build_clusters_gpu(gr)
    ref = orc.lighting(sc.gbuf, sc.rp, sc.prm, sc.lights, sc.type_mask, ref_c["bitmask"], ref_c["range"], synth.DIRECTIONAL_COLOR,
                       synth.DIRECTIONAL_DIRECTION, directional=False, **oracle_kw)
    args, imgs = sc.lighting_args(gr, dev, flags)
    gr.check(gr.lib.gr_lighting(gr.handle, None, args))
    gr.sync()
    got = imgs["hdr"].download()
    assert_rgba16f_close(got, ref, ulps=2.0, abs_tol=1e-4, what=what)
    return got, ref


def flat_scene(view_z_value=5.0):
    sc = Scene(W, H, 0)
    view_z = np.full((H, W), view_z_value)
    depth, pos = world_positions(sc.cam, view_z)
    sc.gbuf["depth"] = depth.astype(np.float32)
    sc.gbuf["emissive"] = np.zeros((H, W, 4), np.uint16)
    sc.gbuf["emissive"][..., 3] = np.float16(1.0).view(np.uint16)
    sc.gbuf["albedo"] = np.full((H, W), 0xff808080, np.uint32)
    return sc, pos


def test_light_on_the_minus_v_axis_behind_a_back_facing_surface(gr):
    sc, pos = flat_scene()
    cam_pos = sc.cam.position
    V = cam_pos[None, None, :] - pos
    V /= np.linalg.norm(V, axis=2, keepdims=True)
    sc.gbuf["normal"] = encode_normals(-V)                             # faces away from the camera: NoV -> 0.001
    sc.gbuf["pbr"] = np.full((H, W), 255 << 8, np.uint16)             # roughness 1 (D flat: the ill-conditioned H direction cannot matter), metallic 0
    rng = np.random.default_rng(3)
    ys, xs = np.mgrid[8:H:16, 8:W:16]
    ys, xs = ys.ravel(), xs.ravel()
    p, v = pos[ys, xs], V[ys, xs]
    # angle between L and -V: from well inside the clamp (0.03 deg: HoV = 2.6e-4) across its edge (0.115 deg: HoV = 0.001) outwards
    angles = np.radians(np.resize([0.03, 0.06, 0.1, 0.115, 0.13, 0.3, 1.0, 5.0], len(ys)))
    side = np.cross(v, np.array([0.0, 1.0, 0.0]))
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    direction = -v * np.cos(angles)[:, None] + side * np.sin(angles)[:, None]
    dist = rng.uniform(0.5, 1.5, len(ys))
    descs = point_lights(p + direction * dist[:, None], (40.0, 30.0, 20.0), 4.0)
    got, ref = run_case(gr, sc, descs, "lights on the -V axis")
    lit = ref.view(np.float16).astype(np.float32)[ys, xs, :3]
    assert (lit.max(axis=1) > 1.0).all(), "each probed pixel must be dominated by its light"


def test_grazing_normals_at_the_smallest_gv_gl(gr):
    sc, pos = flat_scene()
    cam_pos = sc.cam.position
    V = cam_pos[None, None, :] - pos
    V /= np.linalg.norm(V, axis=2, keepdims=True)
    light = np.array([0.0, 9.0, -1.0])                                  # one strong light above the plane
    L = light[None, None, :] - pos
    L /= np.linalg.norm(L, axis=2, keepdims=True)
    N = np.cross(V, L)                                                  # perpendicular to both: NoV = NoL = 0 -> clamps 0.001
    N /= np.linalg.norm(N, axis=2, keepdims=True)
    sc.gbuf["normal"] = encode_normals(N)
    pbr = np.zeros((H, W), np.uint16)                                   # roughness byte 0 -> 0.25, metallic 0 ...
    pbr[:, W // 2:] = 255                                               # ... and metallic 1 on the right half
    sc.gbuf["pbr"] = pbr
    descs = point_lights(light[None, :], (4.0e5, 3.0e5, 2.0e5), 12.0)   # bright: the 0.001 * 0.001-weighted term is visible
    descs["cutoff_range"] = 12.0
    got, ref = run_case(gr, sc, descs, "grazing normals")
    assert (ref.view(np.float16).astype(np.float32)[..., :3].max(axis=2) > 1e-3).mean() > 0.5


def test_unnormalised_normals_engage_the_upper_clamps(gr):
    sc = Scene(W, H, 300)
    rng = np.random.default_rng(9)
    n = rng.uniform(-1.0, 1.0, (H, W, 3))
    n[::2] = np.sign(n[::2])                                            # every other row: (+-1, +-1, +-1), |N| = sqrt(3)
    sc.gbuf["normal"] = encode_normals(n)
    run_case(gr, sc, sc.descs, "un-normalised normals")


def test_surface_on_slice_boundaries_with_lights_ending_there(gr):
    sc0 = Scene(W, H, 0)
    extent = min(0.5, float(sc0.rp[103]) / sc0.res[2])                  # Z slice thickness (clusterer.cpp:700-703)
    # rows alternate between exact slice boundaries k * extent, k around 5 / extent
    k0 = int(round(5.0 / extent))
    view_z = ((k0 + (np.arange(H) % 7))[:, None] * extent) * np.ones((1, W))
    sc = Scene(W, H, 0)
    depth, pos = world_positions(sc.cam, view_z)
    sc.gbuf["depth"] = depth.astype(np.float32)
    front, cam_pos = sc.cam.front, sc.cam.position
    rng = np.random.default_rng(21)
    ys, xs = rng.integers(0, H, 160), rng.integers(0, W, 160)
    p = pos[ys, xs]
    radius = rng.uniform(0.3, 1.2, 160)
    # first half: the light's far extent along the view direction ends exactly on the pixel's slice boundary (the pixel is on
    # the rim: distance == radius up to rounding); second half: lights straddling the boundary
    along = np.where(np.arange(160) < 80, -radius, rng.uniform(-0.5, 0.5, 160) * radius)
    centres = p + front[None, :] * along[:, None]
    descs = point_lights(centres, (30.0, 30.0, 30.0), 1.0)
    descs["cutoff_range"] = radius.astype(np.float32)
    descs["color"] = (rng.uniform(5.0, 40.0, (160, 3))).astype(np.float32)
    run_case(gr, sc, descs, "surface on slice boundaries")
