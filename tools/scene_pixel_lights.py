"""Pixel-light counts of a synthetic scene at 4K (CPU, numpy; every 4th pixel in x and y): lights in range per pixel, i.e. the BRDF evaluations
a lighting launch cannot avoid -- the denominator for "time per pixel-light" in profiles/r06_scenes_depth_split_hot_spot.txt.
    python tools/scene_pixel_lights.py [scene ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from granite_amd import synth

W, H, N, STEP = 3840, 2160, 4096, 4
for scene in (sys.argv[1:] or synth.SCENES):
    cam = synth.Camera(W, H)
    g = synth.make_gbuffer(cam, scene=scene)
    l = synth.make_lights(cam, N, scene=scene)
    depth = g["depth"][::STEP, ::STEP].astype(np.float64)
    h, w = depth.shape
    x = (2 * (np.arange(0, W, STEP) + 0.5) / W - 1)[None, :] * np.ones((h, 1))
    y = (2 * (np.arange(0, H, STEP) + 0.5) / H - 1)[:, None] * np.ones((1, w))
    clip = np.stack([x, y, depth, np.ones_like(x)], -1) @ cam.invVP.T
    P = (clip[..., :3] / clip[..., 3:4]).astype(np.float32)
    pos, rng = l["transform"][:, :, 3].astype(np.float32), l["cutoff_range"].astype(np.float32)
    count = np.zeros((h, w), np.int32)
    for i in range(0, N, 128):
        d2 = ((P[:, :, None, :] - pos[None, None, i:i + 128, :]) ** 2).sum(-1)
        count += (d2 < (rng[i:i + 128] ** 2)[None, None, :]).sum(-1)
    lit = depth != 0
    c = count[lit]
    print("%-12s lit pixels %.3f of the frame; lights in range per lit pixel: mean %.2f  median %d  p95 %d  max %d; share with >= 64: %.4f; pixel-lights per 4K frame %.1f M"
          % (scene, lit.mean(), c.mean(), np.median(c), np.percentile(c, 95), c.max(), (c >= 64).mean(), c.mean() * lit.mean() * W * H / 1e6))
