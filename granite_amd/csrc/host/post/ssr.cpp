// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
#include "ssr.hpp"
#include "spd.hpp"
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace Granite
{
namespace
{
struct Tables
{
	std::mutex lock;
	std::vector<uint16_t> dither; // R8G8, 128 x 128 x 64 layers
	std::vector<uint16_t> brdf;   // R16G16_SFLOAT
	unsigned brdf_width = 0, brdf_height = 0;
} tables;
constexpr unsigned NumDitherIterations = 64; // ssr.cpp:236

void fill_matrices(const RenderContext &context, float view_projection[16], float inv_view_projection[16], float camera[3])
{
	auto &rp = context.get_render_parameters();
	memcpy(view_projection, rp.view_projection.data(), 16 * sizeof(float));
	memcpy(inv_view_projection, rp.inv_view_projection.data(), 16 * sizeof(float));
	for (int i = 0; i < 3; i++)
		camera[i] = rp.camera_position[i];
}

// SSRState (ssr.cpp:84-236)
struct SSRState : RenderPassInterface
{
	RenderGraph *graph = nullptr;
	const RenderContext *context = nullptr;
	RenderTextureResource *output = nullptr, *depth = nullptr, *normal = nullptr, *base_color = nullptr, *pbr = nullptr, *light = nullptr;
	RenderTextureResource *ray_length = nullptr, *ray_confidence = nullptr;
	RenderBufferResource *ray_counter = nullptr, *ray_list = nullptr, *scan = nullptr;
	HIP::BufferHandle dither_lut;
	unsigned frame = 0;

	void setup(HIP::Device &device) override
	{
		std::lock_guard<std::mutex> holder{tables.lock};
		if (tables.dither.empty())
			throw std::logic_error("SSR: the blue-noise table has not been installed (ssr_install_tables).");
		dither_lut = device.create_buffer(tables.dither.size() * sizeof(uint16_t), VK_BUFFER_USAGE_STORAGE_BUFFER_BIT, "blue-noise-lut");
		if (gr_upload(device.get_context(), nullptr, dither_lut->get_device_pointer(), tables.dither.data(), tables.dither.size() * sizeof(uint16_t)) < 0 ||
		    gr_sync(device.get_context(), nullptr) < 0)
			throw std::runtime_error(gr_last_error(device.get_context()));
	}

	void enqueue_prepare_render_pass(RenderGraph &, TaskComposer &) override
	{
		frame = (frame + 1) % NumDitherIterations; // ssr.cpp:161
	}

	void build_render_pass(HIP::CommandBuffer &cmd) override
	{
		if (!dither_lut)
			setup(cmd.get_device());
		auto &out = graph->get_physical_texture_resource(*output);
		auto &hier = graph->get_physical_texture_resource(*depth);
		gr_ssr_args args = {};
		args.depth_chain = hier.get_device_pointer();
		args.chain_width = hier.get_width();
		args.chain_height = hier.get_height();
		args.chain_levels = hier.get_levels();
		args.pbr = graph->get_physical_texture_resource(*pbr).get_view();
		args.normal = graph->get_physical_texture_resource(*normal).get_view();
		args.light = graph->get_physical_texture_resource(*light).get_view();
		args.dither_lut = dither_lut->get_device_pointer();
		args.frame = frame;
		fill_matrices(*context, args.view_projection, args.inv_view_projection, args.camera_position);
		args.output = out.get_view();
		args.ray_length = graph->get_physical_texture_resource(*ray_length).get_view();
		args.ray_confidence = graph->get_physical_texture_resource(*ray_confidence).get_view();
		args.ray_list = static_cast<uint32_t *>(graph->get_physical_buffer_resource(*ray_list).get_device_pointer());
		args.ray_counter = static_cast<uint32_t *>(graph->get_physical_buffer_resource(*ray_counter).get_device_pointer());
		args.scratch = graph->get_physical_buffer_resource(*scan).get_device_pointer();
		cmd.check(gr_ssr_trace(cmd.get_context(), cmd.get_stream(), &args), "ssr trace");
	}
};

struct SSRApply
{
	RenderGraph *graph = nullptr;
	const RenderContext *context = nullptr;
	RenderTextureResource *result = nullptr, *target = nullptr, *depth = nullptr, *base_color = nullptr, *normal = nullptr, *pbr = nullptr;
	HIP::ImageHandle brdf;

	void record(HIP::CommandBuffer &cmd)
	{
		if (!brdf)
		{
			std::lock_guard<std::mutex> holder{tables.lock};
			if (tables.brdf.empty())
				throw std::logic_error("SSR: the BRDF table has not been installed (ssr_install_tables).");
			auto &device = cmd.get_device();
			brdf = device.create_image(tables.brdf_width, tables.brdf_height, VK_FORMAT_R16G16_SFLOAT, "ibl-brdf-lut");
			if (gr_upload(device.get_context(), nullptr, brdf->get_device_pointer(), tables.brdf.data(), tables.brdf.size() * sizeof(uint16_t)) < 0 ||
			    gr_sync(device.get_context(), nullptr) < 0)
				throw std::runtime_error(gr_last_error(device.get_context()));
		}
		gr_ssr_apply_args args = {};
		args.hdr = graph->get_physical_texture_resource(*target).get_view();
		args.reflected = graph->get_physical_texture_resource(*result).get_view();
		args.albedo = graph->get_physical_texture_resource(*base_color).get_view();
		args.normal = graph->get_physical_texture_resource(*normal).get_view();
		args.pbr = graph->get_physical_texture_resource(*pbr).get_view();
		args.depth = graph->get_physical_texture_resource(*depth).get_view();
		args.brdf_lut = brdf->get_view();
		float unused[16];
		fill_matrices(*context, unused, args.inv_view_projection, args.camera_position);
		cmd.check(gr_ssr_apply(cmd.get_context(), cmd.get_stream(), &args), "ssr apply");
	}
};
} // namespace

void ssr_install_tables(const uint8_t *blue_noise, const uint16_t *brdf_lut, unsigned brdf_width, unsigned brdf_height)
{
	if (!blue_noise || !brdf_lut || !brdf_width || !brdf_height)
		throw std::logic_error("ssr_install_tables: null table");
	std::lock_guard<std::mutex> holder{tables.lock};
	// SSRState::setup (ssr.cpp:178-199): sample = (0.5 + value) / 256 is what the blue-noise sampler returns; each layer adds
	// GOLDEN_RATIO * layer, takes the fraction and quantises to 8 bits.
	constexpr int W = 128, H = 128;
	constexpr float GOLDEN_RATIO = 1.61803398875f;
	const auto encode = [](float x, float y, float offset) -> uint16_t {
		x += offset;
		y += offset;
		x = x - std::floor(x);
		y = y - std::floor(y);
		const auto ix = uint32_t(x * 255.0f + 0.5f), iy = uint32_t(y * 255.0f + 0.5f);
		return uint16_t(ix | (iy << 8));
	};
	tables.dither.resize(size_t(W) * H * NumDitherIterations);
	for (int z = 0; z < int(NumDitherIterations); z++)
		for (int y = 0; y < H; y++)
			for (int x = 0; x < W; x++)
			{
				const uint8_t *v = blue_noise + (size_t(y) * W + x) * 2;
				tables.dither[(size_t(z) * H + y) * W + x] = encode((0.5f + float(v[0])) / 256.0f, (0.5f + float(v[1])) / 256.0f, GOLDEN_RATIO * float(z));
			}
	tables.brdf.assign(brdf_lut, brdf_lut + size_t(brdf_width) * brdf_height * 2);
	tables.brdf_width = brdf_width;
	tables.brdf_height = brdf_height;
}

bool ssr_tables_installed()
{
	std::lock_guard<std::mutex> holder{tables.lock};
	return !tables.dither.empty() && !tables.brdf.empty();
}

void setup_ssr_pass(RenderGraph &graph, const RenderContext &context, const std::string &input_depth, const std::string &input_base_color,
                    const std::string &input_normal, const std::string &input_pbr, const std::string &input_light, const std::string &output)
{
	// "TODO: Fixme." in the reference (ssr.cpp:245-246): the pass builds its own hierarchy of the depth attachment.
	setup_depth_hierarchy_pass(graph, input_depth, input_depth + "-hier", &context, false);

	auto &pass = graph.add_pass(output + "-trace", RENDER_GRAPH_QUEUE_COMPUTE_BIT);
	auto state = std::make_shared<SSRState>();
	state->graph = &graph;
	state->context = &context;
	state->normal = &pass.add_texture_input(input_normal);
	state->pbr = &pass.add_texture_input(input_pbr);
	state->depth = &pass.add_texture_input(input_depth + "-hier");
	state->light = &pass.add_texture_input(input_light);
	state->base_color = &pass.add_texture_input(input_base_color);

	const auto light_dim = graph.get_resource_dimensions(*state->light);

	AttachmentInfo att;
	att.size_class = SizeClass::InputRelative;
	att.size_relative_name = input_depth;
	att.format = light_dim.format;
	state->output = &pass.add_storage_texture_output(output + "-sssr", att);
	att.format = VK_FORMAT_R16_SFLOAT;
	state->ray_length = &pass.add_storage_texture_output(output + "-length", att);
	att.format = VK_FORMAT_R8_UNORM;
	state->ray_confidence = &pass.add_storage_texture_output(output + "-confidence", att);

	BufferInfo buf;
	buf.size = size_t(light_dim.width) * light_dim.height * sizeof(uint32_t);
	state->ray_list = &pass.add_storage_output("ssr-ray-list", buf);
	buf.size = 4096;
	buf.usage = VK_BUFFER_USAGE_INDIRECT_BUFFER_BIT | VK_BUFFER_USAGE_STORAGE_BUFFER_BIT;
	state->ray_counter = &pass.add_storage_output("ssr-ray-counter", buf);
	// executor extension: per-tile counts and offsets of the scan that orders the ray list
	buf.size = gr_ssr_scratch_bytes(light_dim.width, light_dim.height);
	buf.usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT;
	state->scan = &pass.add_storage_output("ssr-tile-scan", buf);
	pass.set_render_pass_interface(std::move(state));

	// "Apply results with plain blending." (ssr.cpp:286-322)
	auto &apply_pass = graph.add_pass(output, RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
	auto apply = std::make_shared<SSRApply>();
	apply->graph = &graph;
	apply->context = &context;
	apply->result = &apply_pass.add_texture_input(output + "-sssr");
	AttachmentInfo output_attr;
	output_attr.size_class = SizeClass::InputRelative;
	output_attr.size_relative_name = input_light;
	output_attr.format = graph.get_resource_dimensions(*apply->result).format;
	apply->target = &apply_pass.add_color_output(output, output_attr, input_light);
	apply_pass.set_depth_stencil_input(input_depth);
	apply->base_color = &apply_pass.add_attachment_input(input_base_color);
	apply->normal = &apply_pass.add_attachment_input(input_normal);
	apply->pbr = &apply_pass.add_attachment_input(input_pbr);
	apply->depth = &apply_pass.add_attachment_input(input_depth);
	apply_pass.set_build_render_pass([apply](HIP::CommandBuffer &cmd) { apply->record(cmd); });
}
} // namespace Granite
