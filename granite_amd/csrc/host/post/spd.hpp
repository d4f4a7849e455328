// Depth hierarchy pass: renderer/post/spd.hpp:58-60 / spd.cpp:196-232 restated on the HIP executor.
// (emit_single_pass_downsample, the FFX SPD colour path, is not part of this build: SURVEY.md §8f.)
#pragma once
#include <string>
#include "../render_context.hpp"
#include "../render_graph.hpp"

namespace Granite
{
// Adds compute pass `output`: reads texture `input` (the depth attachment), writes the R32_SFLOAT storage image `output`
// -- a mip chain of max-reduced linear depth, sized to the input rounded up to multiples of 64 (halved when
// output_downsample, which also drops the full-resolution level) with floor_log2(max(w, h)) - output_downsample levels --
// and the 4-byte storage buffer `output + "-counter"`.  `context` supplies inv_projection at execution time.
void setup_depth_hierarchy_pass(RenderGraph &graph, const std::string &input, const std::string &output,
                                const RenderContext *context, bool output_downsample);
} // namespace Granite
