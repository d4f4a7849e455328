"""Deterministic synthetic inputs for the image-space chain.

The G-buffer producer (Granite's scene/mesh renderer) is out of scope, so the harness, the parity tests and bench.py all
draw their inputs from here: camera, G-buffer attachments in the reference's storage formats, and a light list.
Generator: numpy PCG64 seeded with 1234 (+ a per-stream offset); everything is a pure function of (seed, size).

Matrices are column-major float32[16] like muglm (m[4*col + row]).
"""
from __future__ import annotations

import math

import numpy as np

SEED = 1234

LIGHT_DESC_DTYPE = np.dtype([("type", "<i4"), ("color", "<f4", 3), ("inner_cone", "<f4"), ("outer_cone", "<f4"),
                             ("cutoff_range", "<f4"), ("pad", "<f4"), ("transform", "<f4", (3, 4))])

# SceneViewerApplication defaults (application/scene_viewer_application.cpp:43-46,380,407)
DIRECTIONAL_COLOR = (6.0, 5.5, 4.5)
DIRECTIONAL_DIRECTION = tuple((np.array([0.5, 1.2, 0.8]) / np.linalg.norm([0.5, 1.2, 0.8])).astype(np.float32))
CLUSTER_RESOLUTION = (128, 64, 4096)
FRAME_TIME = 0.01  # application_headless.cpp:420


def perspective(fovy: float, aspect: float, near: float, far: float) -> np.ndarray:
    """muglm::perspective (math/muglm/muglm.cpp:319-337): reverse-Z, Vulkan Y-flip. Returns 4x4 (row, col) float64."""
    t = math.tan(fovy / 2.0)
    m = np.zeros((4, 4), np.float64)
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = -1.0 - far / (near - far)
    m[2, 3] = -(far * near) / (near - far)
    m[3, 2] = -1.0
    m[1, :] *= -1.0
    return m


def look_at(eye, center, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    eye, center, up = (np.asarray(v, np.float64) for v in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s @ eye, -u @ eye, f @ eye
    return m


def _cm(m: np.ndarray) -> np.ndarray:
    """(row, col) float64 -> column-major float32[16]."""
    return np.ascontiguousarray(m.T, np.float32).reshape(16)


class Camera:
    """RenderContext::set_camera (renderer/render_context.cpp:53-86) for the survey's fixed view."""

    def __init__(self, width: int, height: int, fovy_deg: float = 60.0, near: float = 0.1, far: float = 100.0,
                 eye=(0.0, 2.0, 8.0), center=(0.0, 1.0, 0.0)):
        self.width, self.height = width, height
        self.near, self.far = near, far
        self.fovy = math.radians(fovy_deg)
        self.aspect = width / height
        self.P = perspective(self.fovy, self.aspect, near, far)
        self.V = look_at(eye, center)
        self.VP = self.P @ self.V
        self.invP = np.linalg.inv(self.P)
        self.invV = np.linalg.inv(self.V)
        self.invVP = np.linalg.inv(self.VP)
        self.position = self.invV[:3, 3].copy()
        self.front = -self.invV[:3, 2].copy()

    def render_params(self) -> np.ndarray:
        """Packed like oracle.RENDER_PARAMS_DTYPE / the harness C struct: 6 mat4 + pos + front + near + far."""
        out = np.zeros(104, np.float32)
        for i, m in enumerate((self.P, self.V, self.VP, self.invP, self.invV, self.invVP)):
            out[16 * i:16 * i + 16] = _cm(m)
        out[96:99] = self.position
        out[99:102] = self.front
        # RenderContext derives these from inv_projection; for this projection they equal near/far.
        out[102], out[103] = self.near, self.far
        return out

    def depth_from_view_distance(self, d: np.ndarray) -> np.ndarray:
        """Reverse-Z depth written by a surface at distance d along the camera front."""
        z_clip = self.P[2, 2] * (-d) + self.P[2, 3]
        return (z_clip / d).astype(np.float32)


def _rng(stream: int, seed: int = SEED) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, stream]))


def _f32_to_f16_bits(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float16).view(np.uint16)


def make_hdr(width: int, height: int, seed: int = SEED) -> np.ndarray:
    """Emissive / HDR RGBA16F bits: log-uniform luminance 2^-6..2^6, 0.5 % hot pixels at 2^8."""
    r = _rng(1, seed)
    lum = np.exp2(r.uniform(-6.0, 6.0, (height, width))).astype(np.float32)
    hot = r.random((height, width)) < 0.005
    lum[hot] = 256.0
    hue = r.uniform(0.2, 1.0, (height, width, 3)).astype(np.float32)
    hue /= hue.max(axis=2, keepdims=True)
    rgba = np.empty((height, width, 4), np.float32)
    rgba[..., :3] = hue * lum[..., None]
    rgba[..., 3] = 1.0
    return _f32_to_f16_bits(rgba)


SCENES = ("default", "depth_split", "hot_spot")
DEPTH_SPLIT_BAR_PX, DEPTH_SPLIT_FAR_Z = 24, 30.0  # "depth_split": bars of this many pixels alternate between the default surface and a far wall
HOT_SPOT_REGION, HOT_SPOT_LIGHTS, HOT_SPOT_RANGE = (0.39, 0.61), 128, 1.5  # "hot_spot": u and v interval of the region, lights moved into it, their range


def surface_view_z(u, v):
    """View-space distance of the default synthetic surface under normalised screen coordinates (u, v)."""
    return 4.0 + 3.0 * np.sin(8.0 * np.pi * u) * np.cos(6.0 * np.pi * v)


def make_gbuffer(cam: Camera, seed: int = SEED, scene: str = "default") -> dict:
    """G-buffer attachments in the reference's formats (application/scene_viewer_application.cpp:880-931):
    emissive RGBA16F, albedo RGBA8_SRGB (a = ambient), normal A2B10G10R10, pbr RG8, depth D32F.

    scene = "depth_split": every other vertical bar of DEPTH_SPLIT_BAR_PX pixels is a wall at view distance DEPTH_SPLIT_FAR_Z: a third of
    the 16-pixel-wide lighting tiles straddles a depth discontinuity and their light-index windows span every light between the two depths
    (the worst case of clusterer_bindless.h:49-81's subgroup min / max window).  "hot_spot" changes the lights only (make_lights)."""
    assert scene in SCENES
    w, h = cam.width, cam.height
    r = _rng(2, seed)
    u = (np.arange(w, dtype=np.float64) + 0.5) / w
    v = (np.arange(h, dtype=np.float64) + 0.5) / h
    view_z = surface_view_z(u[None, :], v[:, None])
    if scene == "depth_split":
        far = ((np.arange(w) // DEPTH_SPLIT_BAR_PX) % 2) == 1
        view_z = np.where(far[None, :], DEPTH_SPLIT_FAR_Z + 0.25 * np.cos(6.0 * np.pi * v)[:, None], view_z)
    depth = cam.depth_from_view_distance(view_z)
    sky = r.random((h, w)) < 0.02
    depth[sky] = 0.0

    albedo_rgb = r.integers(10, 231, (h, w, 3), dtype=np.uint32)
    albedo = albedo_rgb[..., 0] | (albedo_rgb[..., 1] << 8) | (albedo_rgb[..., 2] << 16) | np.uint32(255 << 24)

    n = r.normal(size=(h, w, 3))
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    facing = n @ (-cam.front)
    n[facing < 0] *= -1.0
    q = np.clip(np.rint((0.5 * n + 0.5) * 1023.0), 0, 1023).astype(np.uint32)
    normal = q[..., 0] | (q[..., 1] << 10) | (q[..., 2] << 20) | np.uint32(3 << 30)

    metallic = (r.random((h, w)) < 0.2).astype(np.uint16) * 255
    roughness = np.clip(np.rint(r.uniform(0.05, 1.0, (h, w)) * 255.0), 0, 255).astype(np.uint16)
    pbr = (metallic | (roughness << 8)).astype(np.uint16)

    return {"emissive": make_hdr(w, h, seed), "albedo": albedo.astype(np.uint32), "normal": normal.astype(np.uint32),
            "pbr": pbr, "depth": depth.astype(np.float32)}


def make_lights(cam: Camera, count: int, spot_fraction: float = 0.25, z_lo: float = 1.0, z_hi: float = 40.0,
                max_range: float = 4.0, seed: int = SEED, scene: str = "default") -> np.ndarray:
    """Point + spot lights uniform (by volume) in the view-frustum slab z in [z_lo, z_hi]; colour = hue * intensity with
    intensity log-uniform 1..50; spots: inner = cos 20 deg, outer = cos 30 deg, random orientation;
    PositionalLight::set_maximum_range(max_range).

    scene = "hot_spot": the first HOT_SPOT_LIGHTS lights sit 0.3 units in front of the default surface inside the screen region
    HOT_SPOT_REGION x HOT_SPOT_REGION (4.8 % of the screen) with range HOT_SPOT_RANGE: well over 64 lights in range per pixel there."""
    assert scene in SCENES
    r = _rng(3, seed)
    descs = np.zeros(count, LIGHT_DESC_DTYPE)
    if count == 0:
        return descs
    d = np.cbrt(r.random(count) * (z_hi ** 3 - z_lo ** 3) + z_lo ** 3)
    t = math.tan(cam.fovy / 2.0)
    xs, ys = r.uniform(-1.0, 1.0, count), r.uniform(-1.0, 1.0, count)
    hot = min(HOT_SPOT_LIGHTS, count) if scene == "hot_spot" else 0
    if hot:
        hr = _rng(7, seed)
        hu, hv = hr.uniform(*HOT_SPOT_REGION, hot), hr.uniform(*HOT_SPOT_REGION, hot)
        d[:hot] = surface_view_z(hu, hv) - 0.3
        xs[:hot], ys[:hot] = 2.0 * hu - 1.0, 2.0 * hv - 1.0  # (the surface is symmetric under v -> 1 - v: the sign of the y axis does not matter)
    xv = xs * d * t * cam.aspect
    yv = ys * d * t
    view_pos = np.stack([xv, yv, -d, np.ones(count)], axis=1)
    world = (cam.invV @ view_pos.T).T[:, :3]

    hue = r.uniform(0.1, 1.0, (count, 3))
    hue /= hue.max(axis=1, keepdims=True)
    intensity = np.exp(r.uniform(math.log(1.0), math.log(50.0), count))
    descs["color"] = (hue * intensity[:, None]).astype(np.float32)
    is_spot = r.random(count) < spot_fraction
    descs["type"] = np.where(is_spot, 0, 1)
    descs["inner_cone"] = math.cos(math.radians(20.0))
    descs["outer_cone"] = math.cos(math.radians(30.0))
    descs["cutoff_range"] = max_range
    if hot:
        descs["cutoff_range"][:hot] = HOT_SPOT_RANGE

    # orientation: -Z axis of the node = light direction
    fwd = r.normal(size=(count, 3))
    fwd /= np.linalg.norm(fwd, axis=1, keepdims=True)
    helper = np.where(np.abs(fwd[:, 1:2]) < 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    zaxis = -fwd
    xaxis = np.cross(helper, zaxis)
    xaxis /= np.linalg.norm(xaxis, axis=1, keepdims=True)
    yaxis = np.cross(zaxis, xaxis)
    tr = np.zeros((count, 3, 4))
    tr[:, :, 0], tr[:, :, 1], tr[:, :, 2], tr[:, :, 3] = xaxis, yaxis, zaxis, world
    descs["transform"] = tr.astype(np.float32)
    return descs


def make_motion_vectors(width: int, height: int) -> np.ndarray:
    """Config 4: MV RG16F = 0 (camera reprojection branch) with a 10 % region of constant (2 px, 1 px)/resolution."""
    mv = np.zeros((height, width, 2), np.float32)
    x0, x1 = int(0.45 * width), int(0.55 * width)
    mv[:, x0:x1, 0] = 2.0 / width
    mv[:, x0:x1, 1] = 1.0 / height
    return _f32_to_f16_bits(mv)


def make_ldr_pattern(width: int, height: int, seed: int = SEED) -> np.ndarray:
    """Gamma-space RGBA8 test card for the AA passes: flat regions, axis-aligned steps, 45-degree and shallow diagonal
    edges, circles and a noisy band, so FXAA spans, SMAA orthogonal + diagonal searches, corner rounding and LUT fetches
    are all exercised."""
    r = _rng(5, seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    img = np.zeros((height, width, 3), np.float64)
    img[...] = (0.18, 0.22, 0.30)
    img[(x > 0.15 * width) & (x < 0.45 * width) & (y > 0.1 * height) & (y < 0.5 * height)] = (0.85, 0.80, 0.20)  # box
    img[(x + y) % max(width // 6, 8) < max(width // 12, 4)] *= 0.55  # 45-degree stripes
    img[(y - 0.23 * x) > 0.62 * height] = (0.05, 0.55, 0.65)  # shallow diagonal
    img[(y + 3.1 * x) < 0.35 * height] = (0.9, 0.9, 0.95)  # steep diagonal
    for cx, cy, rad, col in ((0.7, 0.3, 0.12, (0.95, 0.15, 0.1)), (0.62, 0.72, 0.2, (0.1, 0.1, 0.1)), (0.3, 0.8, 0.07, (1.0, 1.0, 1.0))):
        img[(x - cx * width) ** 2 + (y - cy * height) ** 2 < (rad * min(width, height)) ** 2] = col
    band = (y > 0.88 * height)
    img[band] = r.uniform(0.0, 1.0, (int(band.sum()), 3))
    rgba = np.empty((height, width, 4), np.uint8)
    rgba[..., :3] = np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)
    rgba[..., 3] = 255
    return rgba


def pack_b10g11r11(rgba16f_bits: np.ndarray) -> np.ndarray:
    """RGBA16F bits (h, w, 4) -> B10G11R11_UFLOAT_PACK32 words (h, w): the G-buffer's emissive attachment as the reference declares
    it with renderTargetFp16 = false (scene_viewer_application.cpp:881-883).  A half float and the packed floats share exponent
    width and bias, so the conversion is a mantissa rounding of the half's bits (to nearest, ties to even; the closest FINITE value:
    above the largest finite packed value -> it); negative values -> 0; +inf and NaN stay."""
    hb = np.asarray(rgba16f_bits, np.uint16).astype(np.uint32)

    def channel(h, mant_bits):
        drop = 10 - mant_bits
        v = h & 0x7fff
        q = (v + (1 << (drop - 1)) - 1 + ((v >> drop) & 1)) >> drop
        inf = 31 << mant_bits
        q = np.minimum(q, inf - 1)
        q = np.where(v == 0x7c00, inf, q)
        q = np.where(v > 0x7c00, inf | 1, q)
        return np.where((h & 0x8000) != 0, np.where(v > 0x7c00, inf | 1, 0), q).astype(np.uint32)

    return channel(hb[..., 0], 6) | (channel(hb[..., 1], 6) << 11) | (channel(hb[..., 2], 5) << 22)
