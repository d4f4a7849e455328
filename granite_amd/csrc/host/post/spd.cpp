// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
#include "spd.hpp"
#include <algorithm>

namespace Granite
{
namespace
{
unsigned floor_log2(unsigned v)
{
	unsigned l = 0;
	while (v >>= 1)
		l++;
	return l;
}

// HiZPassState (spd.cpp:141-194): one gr_hiz launch where the reference binds 13 storage views and dispatches hiz.comp.
struct DepthHierarchyPass : RenderPassInterface
{
	RenderGraph *graph = nullptr;
	RenderTextureResource *input = nullptr;
	RenderTextureResource *chain = nullptr;
	RenderBufferResource *counter = nullptr;
	const RenderContext *context = nullptr;
	bool output_downsample = false;

	void build_render_pass(HIP::CommandBuffer &cmd) override
	{
		auto &depth = graph->get_physical_texture_resource(*input);
		auto &out = graph->get_physical_texture_resource(*chain);
		auto &count = graph->get_physical_buffer_resource(*counter);

		gr_hiz_args args = {};
		args.depth = depth.get_view();
		args.chain = out.get_device_pointer();
		args.chain_width = out.get_width();
		args.chain_height = out.get_height();
		args.chain_levels = out.get_levels();
		args.output_downsample = output_downsample ? 1u : 0u;
		// mat2(inv_projection[2].zw * vec2(-1, 1), inv_projection[3].zw * vec2(-1, 1)), spd.cpp:164-165
		auto &inv_projection = context->get_render_parameters().inv_projection;
		args.z_transform[0] = -inv_projection[2].z;
		args.z_transform[1] = inv_projection[2].w;
		args.z_transform[2] = -inv_projection[3].z;
		args.z_transform[3] = inv_projection[3].w;
		args.counter = static_cast<uint32_t *>(count.get_device_pointer());
		cmd.check(gr_hiz(cmd.get_context(), cmd.get_stream(), &args), "depth hierarchy");
	}
};
} // namespace

bool supports_single_pass_downsample(HIP::Device &, VkFormat format)
{
	// spd.cpp:31-54 asks for subgroup quad operations, 256-wide workgroups and format-less storage access; on this device the
	// only question left is the storage format the kernel is written for.
	return format == VK_FORMAT_R16G16B16A16_SFLOAT;
}

void emit_single_pass_downsample(HIP::CommandBuffer &cmd, const SPDInfo &info)
{
	if (!info.input || !info.output_mips || info.num_mips == 0 || info.num_mips > MaxSPDMips)
		throw std::logic_error("emit_single_pass_downsample: bad SPDInfo.");
	const gr_image &top = *info.output_mips[0];
	for (unsigned i = 0; i < info.num_mips; i++)
	{
		const gr_image &mip = *info.output_mips[i];
		const auto *expected = static_cast<const uint8_t *>(top.ptr) + gr_mip_chain_offset(top.width, top.height, 8, i);
		if (mip.format != VK_FORMAT_R16G16B16A16_SFLOAT || mip.ptr != expected || mip.width != std::max(top.width >> i, 1u) ||
		    mip.height != std::max(top.height >> i, 1u))
			throw std::logic_error("emit_single_pass_downsample: output_mips must be consecutive RGBA16F levels of one mip chain.");
	}

	gr_spd_args args = {};
	args.input = *info.input;
	args.chain = top.ptr;
	args.width = top.width;   // push.base_image_resolution, spd.cpp:85-86
	args.height = top.height;
	args.mips = info.num_mips;
	args.components = info.num_components;
	args.reduction_mode = info.mode == ReductionMode::Depth ? GR_SPD_REDUCTION_DEPTH : GR_SPD_REDUCTION_COLOR;
	args.filter_mods = info.filter_mod ? &info.filter_mod[0].x : nullptr;
	cmd.check(gr_spd_downsample(cmd.get_context(), cmd.get_stream(), &args), "single pass downsample");
}

void setup_depth_hierarchy_pass(RenderGraph &graph, const std::string &input, const std::string &output,
                                const RenderContext *context, bool output_downsample)
{
	auto &pass = graph.add_pass(output, RENDER_GRAPH_QUEUE_COMPUTE_BIT);
	auto state = std::make_shared<DepthHierarchyPass>();
	state->graph = &graph;
	state->context = context;
	state->output_downsample = output_downsample;
	state->input = &pass.add_texture_input(input);

	// Whole 64 x 64 tiles, so that six halvings are exact and culling never has to fold below mip 7; stop at 2x1 / 1x2.
	const auto dim = graph.get_resource_dimensions(*state->input);
	const unsigned shift = output_downsample ? 1u : 0u;
	AttachmentInfo att;
	att.size_class = SizeClass::Absolute;
	att.format = VK_FORMAT_R32_SFLOAT;
	att.size_x = float(((dim.width + 63u) & ~63u) >> shift);
	att.size_y = float(((dim.height + 63u) & ~63u) >> shift);
	att.levels = unsigned(std::max(1, int(floor_log2(std::max(dim.width, dim.height))) - int(shift)));
	att.layers = dim.layers;
	state->chain = &pass.add_storage_texture_output(output, att);

	BufferInfo count_info;
	count_info.size = 4 * dim.layers;
	count_info.usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT;
	state->counter = &pass.add_storage_output(output + "-counter", count_info);
	pass.set_render_pass_interface(std::move(state));
}
} // namespace Granite
