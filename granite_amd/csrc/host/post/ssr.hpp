// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// renderer/post/ssr.{hpp,cpp} restated on the HIP executor: the screen-space reflection passes (FidelityFX SSSR as Granite
// vendors it).  Same pass / resource names, formats and sizes; the classify, build_indirect and trace_primary dispatches are one
// gr_ssr_trace call, the blended apply quad one gr_ssr_apply call.
#pragma once
#include <string>
#include "../render_context.hpp"
#include "../render_graph.hpp"

namespace Granite
{
// ssr.hpp:30-35.  Adds (ssr.cpp:238-323):
//   * the depth hierarchy pass `input_depth + "-hier"` (setup_depth_hierarchy_pass, output_downsample = false),
//   * compute pass `output + "-trace"`: storage textures `output + "-sssr"` (format of the light input), `output + "-length"`
//     (R16_SFLOAT), `output + "-confidence"` (R8_UNORM), storage buffers "ssr-ray-list" (4 bytes per pixel) and
//     "ssr-ray-counter" (4096 bytes), plus the executor's "ssr-tile-scan" scratch (the ray list is built by a scan, not by
//     atomics: csrc/ssr.hip),
//   * graphics pass `output`: colour output `output` = read-modify-write of `input_light`, blend ONE / ONE of
//     reflected * (F * brdf.x + brdf.y), depth test NOT_EQUAL against the quad at z = 1.
// trace_fallback.comp (ssr.cpp:138-169) needs a volumetric-diffuse probe set, which this path does not have.
void setup_ssr_pass(RenderGraph &graph, const RenderContext &context, const std::string &input_depth, const std::string &input_base_color,
                    const std::string &input_normal, const std::string &input_pbr, const std::string &input_light, const std::string &output);

// The two constant tables of the pass, which Granite gets from its own tree: the 128 x 128 x 2 integer values of the blue-noise
// sampler (renderer/utils/blue, sample 0, dimensions 0 / 1) from which SSRState::setup builds the 64-layer dither texture
// (ssr.cpp:178-205; the same arithmetic runs here), and the R16G16_SFLOAT split-sum BRDF table
// (builtin://textures/ibl_brdf_lut.gtx, 256 x 256).  Process-wide; install before the first frame.
void ssr_install_tables(const uint8_t *blue_noise_128x128_rg8, const uint16_t *brdf_lut_rg16f, unsigned brdf_width, unsigned brdf_height);
bool ssr_tables_installed();
} // namespace Granite
