#!/usr/bin/env python3
"""Derives the two constant tables the screen-space reflection pass needs from a Granite checkout (run in the build container,
where /root/reference exists; the outputs are committed under granite_amd/data/ like the SMAA tables):

  sssr_blue_noise_128x128_rg8.bin   the 128 x 128 x 2 bytes that samplerBlueNoiseErrorDistribution_128x128_OptimizedFor_2d2d2d2d_1spp
                                    (Eric Heitz's sampler, renderer/utils/blue/) returns for sample 0, dimensions 0 and 1, as its
                                    integer `value` (the float is (0.5 + value) / 256).  renderer/post/ssr.cpp:178-199 builds its
                                    64-layer dither texture from exactly these; the executor repeats that arithmetic at set-up.
  ibl_brdf_lut_rg16f_256x256.bin    payload of assets/textures/ibl_brdf_lut.gtx (R16G16_SFLOAT, the split-sum BRDF table apply.frag samples).
"""
import os, re, sys
import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from granite_amd import gtx  # noqa: E402

src = open(os.path.join(REF, "renderer/utils/blue/samplerBlueNoiseErrorDistribution_128x128_OptimizedFor_2d2d2d2d_1spp.hpp")).read()


def table(name):
    m = re.search(r"static const int %s\[[^\]]*\]\s*=\s*\{([^}]*)\}" % name, src)
    return np.array([int(v) for v in m.group(1).split(",") if v.strip()], np.int64)


sobol, scrambling, ranking = table("sobol_256spp_256d"), table("scramblingTile"), table("rankingTile")
out = np.zeros((128, 128, 2), np.uint8)
for dim in range(2):
    for y in range(128):          # the header's pixel_j
        for x in range(128):      # pixel_i
            ranked = 0 ^ int(ranking[dim + (x + y * 128) * 8])
            value = int(sobol[dim + ranked * 256]) ^ int(scrambling[(dim % 8) + (x + y * 128) * 8])
            out[y, x, dim] = value
data = os.path.join(ROOT, "granite_amd", "data")
out.tofile(os.path.join(data, "sssr_blue_noise_128x128_rg8.bin"))
lut = gtx.read(os.path.join(REF, "assets/textures/ibl_brdf_lut.gtx"))
assert lut.info.format == 83 and lut.info.width == 256 and lut.info.height == 256  # VK_FORMAT_R16G16_SFLOAT
np.ascontiguousarray(lut.level(0)[0]).tofile(os.path.join(data, "ibl_brdf_lut_rg16f_256x256.bin"))
print("wrote", out.shape, lut.level(0)[0].shape)
