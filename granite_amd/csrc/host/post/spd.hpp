// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// renderer/post/spd.{hpp,cpp} restated on the HIP executor: the single-pass downsampler (emit_single_pass_downsample, FFX SPD)
// and the depth hierarchy pass built next to it.
#pragma once
#include <string>
#include "../render_context.hpp"
#include "../render_graph.hpp"

namespace Granite
{
// spd.hpp:35-58.  A Vulkan::ImageView of one mip level is a gr_image here (HIP::Image::get_level_view).
bool supports_single_pass_downsample(HIP::Device &device, VkFormat format);

enum ReductionMode : int
{
	Color = 0,
	Depth
};

struct SPDInfo
{
	const gr_image *input;
	const gr_image *const *output_mips; // num_mips consecutive levels of one mip chain
	unsigned num_mips;
	const HIP::Buffer *counter_buffer;  // the shader's atomic ticket; accepted for call-site parity, not touched
	VkDeviceSize counter_buffer_offset;
	unsigned num_components;
	const vec4 *filter_mod;
	ReductionMode mode;
};

static constexpr unsigned MaxSPDMips = 12; // the shader binds uImages[12] (spd.comp:38); spd.hpp says 13
// spd.cpp:56-102: one gr_spd_downsample call where the reference binds the source, the counter and the storage views and
// dispatches spd.comp.  Throws std::logic_error when output_mips are not consecutive levels of one tightly packed chain.
void emit_single_pass_downsample(HIP::CommandBuffer &cmd, const SPDInfo &info);

// Adds compute pass `output`: reads texture `input` (the depth attachment), writes the R32_SFLOAT storage image `output`
// -- a mip chain of max-reduced linear depth, sized to the input rounded up to multiples of 64 (halved when
// output_downsample, which also drops the full-resolution level) with floor_log2(max(w, h)) - output_downsample levels --
// and the 4-byte storage buffer `output + "-counter"`.  `context` supplies inv_projection at execution time.
void setup_depth_hierarchy_pass(RenderGraph &graph, const std::string &input, const std::string &output,
                                const RenderContext *context, bool output_downsample);
} // namespace Granite
