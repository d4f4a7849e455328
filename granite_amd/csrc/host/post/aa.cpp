// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
#include "aa.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

namespace Granite
{
// ---- TemporalJitter (temporal.cpp:40-197) ----------------------------------------------------------------------------------
TemporalJitter::TemporalJitter()
{
	init(Type::None, vec2(0.0f));
}

void TemporalJitter::init(Type type_, vec2 res)
{
	type = type_;
	phase = 0;
	auto ndc_shift = [&](float scale, float px, float py) { return translate(vec3(scale * px / res.x, scale * py / res.y, 0.0f)); };
	jitter_table.clear();

	switch (type)
	{
	case Type::FXAA_2Phase:
		jitter_table = {ndc_shift(2.0f, 0.5f, 0.0f), ndc_shift(2.0f, 0.0f, 0.5f)};
		break;
	case Type::SMAA_T2X:
		jitter_table = {ndc_shift(2.0f, -0.25f, -0.25f), ndc_shift(2.0f, 0.25f, 0.25f)};
		break;
	case Type::TAA_8Phase:
	{
		static const float offsets[8][2] = {{-7, 1}, {-5, -5}, {-1, -3}, {3, -7}, {-5, -1}, {7, 7}, {1, 3}, {-3, 5}};
		for (auto &o : offsets)
			jitter_table.push_back(ndc_shift(0.125f, o[0], o[1]));
		break;
	}
	case Type::TAA_16Phase:
	{
		// Sub-pixel offsets in 1/8-pixel NDC units (x2 for the [-1,1] range is folded into the 0.125 factor).
		static const float offsets[16][2] = {{-8, 0}, {-6, -4}, {-3, -2}, {-2, -6}, {1, -1}, {2, -5}, {6, -7}, {5, -3},
		                                     {4, 1},  {7, 4},   {3, 5},   {0, 7},   {-1, 3}, {-4, 6}, {-7, 8}, {-5, 2}};
		for (auto &o : offsets)
			jitter_table.push_back(ndc_shift(0.125f, o[0], o[1]));
		break;
	}
	default:
		jitter_table = {mat4(1.0f)};
		break;
	}
	jitter_count = unsigned(jitter_table.size());
	saved_jittered_view_proj.assign(jitter_count, mat4(1.0f));
	saved_view_proj.assign(jitter_count, mat4(1.0f));
	saved_inv_view_proj.assign(jitter_count, mat4(1.0f));
}

void TemporalJitter::step(const mat4 &proj, const mat4 &view)
{
	phase = (phase + 1 >= jitter_count) ? 0 : phase + 1;
	saved_view_proj[phase] = proj * view;
	saved_jittered_projection = jitter_table[phase] * proj;
	saved_jittered_view_proj[phase] = jitter_table[phase] * saved_view_proj[phase];
	saved_inv_view_proj[phase] = inverse(saved_view_proj[phase]);
}

unsigned TemporalJitter::get_offset_phase(int frames) const
{
	// Note: like the reference, stepping back from phase 0 lands on jitter_count - frames.
	return phase >= unsigned(frames) ? phase - unsigned(frames) : jitter_count - unsigned(frames);
}

namespace
{
// The plan when there is anything to restrict (or a transport to exercise), else nullptr.
const StripPlan *live(const StripPlan *plan)
{
	return plan && (plan->active() || plan->exchange) ? plan : nullptr;
}

// The pass that writes the frame's output image under row bands: waits for the gather that last used the image before
// writing its band (acquire), and sends the bands to meet in every rank's image afterwards.
void acquire_output(const StripPlan *strip, HIP::CommandBuffer &cmd, HIP::Image &output)
{
	if (strip && strip->acquire_output)
		strip->acquire_output(cmd, output);
}

void gather_output(const StripPlan *strip, HIP::CommandBuffer &cmd, HIP::Image &output, const char *tag)
{
	if (strip && strip->exchange_output)
		strip->exchange_output(cmd, output, strip->out_chunk_rows, tag);
	else if (strip && strip->exchange)
		strip->exchange(cmd, output, strip->out_chunk_rows, tag);
}
} // namespace

// ---- FXAA (fxaa.cpp:28-55) ------------------------------------------------------------------------------------------------
void setup_fxaa_postprocess(RenderGraph &graph, const std::string &input, const std::string &output, VkFormat output_format, const StripPlan *plan)
{
	graph.get_texture_resource(input).get_attachment_info().flags |= ATTACHMENT_INFO_UNORM_SRGB_ALIAS_BIT;

	auto &fxaa = graph.add_pass("fxaa", RenderGraph::get_default_post_graphics_queue());
	AttachmentInfo fxaa_output;
	fxaa_output.flags |= ATTACHMENT_INFO_SUPPORTS_PREROTATE_BIT;
	fxaa_output.size_class = SizeClass::InputRelative;
	fxaa_output.size_relative_name = input;
	fxaa_output.format = output_format;
	fxaa.add_color_output(output, fxaa_output);
	auto &fxaa_input = fxaa.add_texture_input(input);

	fxaa.set_build_render_pass([&graph, &fxaa, &fxaa_input, plan](HIP::CommandBuffer &cmd) {
		const StripPlan *strip = live(plan);
		auto &input_image = graph.get_physical_texture_resource(fxaa_input);
		auto &output_image = graph.get_physical_texture_resource(*fxaa.get_color_outputs()[0]);
		gr_push_fxaa push = {{1.0f / float(input_image.get_width()), 1.0f / float(input_image.get_height())}};
		gr_rows rows;
		acquire_output(strip, cmd, output_image);
		if (to_rows(strip ? &strip->aa_out : nullptr, rows))
			cmd.check(gr_fxaa_rows(cmd.get_context(), cmd.get_stream(), &input_image.get_view(), &output_image.get_view(), &push, &rows), "fxaa");
		gather_output(strip, cmd, output_image, "fxaa");
	});
}

// ---- SMAA (smaa.cpp:32-208) ------------------------------------------------------------------------------------------------
void setup_smaa_postprocess(RenderGraph &graph, TemporalJitter &jitter, float, const std::string &input, const std::string &,
                            const std::string &output, SMAAPreset preset, const StripPlan *plan)
{
	if (preset == SMAAPreset::Ultra_T2X)
		throw std::logic_error("SMAA T2X is not live in the reference (smaa_t2x_resolve.frag does not compile) and is not provided.");
	int smaa_quality = preset == SMAAPreset::Low ? 0 : preset == SMAAPreset::Medium ? 1 : preset == SMAAPreset::High ? 2 : 3;
	jitter.init(TemporalJitter::Type::None, vec2(1.0f));

	graph.get_texture_resource(input).get_attachment_info().flags |= ATTACHMENT_INFO_UNORM_SRGB_ALIAS_BIT;

	AttachmentInfo smaa_edge_output;
	smaa_edge_output.size_class = SizeClass::InputRelative;
	smaa_edge_output.size_relative_name = input;
	smaa_edge_output.format = VK_FORMAT_R8G8_UNORM;

	AttachmentInfo smaa_weight_output = smaa_edge_output;
	smaa_weight_output.format = VK_FORMAT_R8G8B8A8_UNORM;

	AttachmentInfo smaa_output_final;
	smaa_output_final.size_class = SizeClass::InputRelative;
	smaa_output_final.size_relative_name = input;

	AttachmentInfo smaa_depth = smaa_edge_output;
	smaa_depth.format = VK_FORMAT_D16_UNORM;

	auto &smaa_edge = graph.add_pass("smaa-edge", RenderGraph::get_default_post_graphics_queue());
	auto &smaa_weight = graph.add_pass("smaa-weights", RenderGraph::get_default_post_graphics_queue());
	auto &smaa_blend = graph.add_pass("smaa-blend", RenderGraph::get_default_post_graphics_queue());

	auto &edge_output_res = smaa_edge.add_color_output("smaa-edge", smaa_edge_output);
	auto &edge_input_res = smaa_edge.add_texture_input(input);
	// The reference masks the weight pass with a D16 attachment written where the edge shader does not discard.  The HIP
	// weight kernel reads that predicate off the edge texel, so "smaa-mask" stays declared (graph-compatible) but is not
	// touched by a kernel.
	smaa_edge.set_depth_stencil_output("smaa-mask", smaa_depth);

	auto &weight_output_res = smaa_weight.add_color_output("smaa-weights", smaa_weight_output);
	auto &weight_input_res = smaa_weight.add_texture_input("smaa-edge");
	smaa_weight.set_depth_stencil_input("smaa-mask");

	smaa_blend.add_color_output(output, smaa_output_final);
	auto &blend_input_res = smaa_blend.add_texture_input(input);
	auto &blend_weight_res = smaa_blend.add_texture_input("smaa-weights");

	auto metrics = [](const HIP::ImageView &img) {
		gr_push_smaa push = {{1.0f / float(img.get_width()), 1.0f / float(img.get_height()), float(img.get_width()), float(img.get_height())}};
		return push;
	};

	// Both kernels write every pixel (0 where the shader would discard / be masked), which subsumes the reference's
	// LOAD_OP_CLEAR to 0 (smaa.cpp:139-143,180-184): no get_clear_color callbacks are installed.
	// Row bands (plan): each pass covers the rows the next one reads around this rank's output chunk (StripPlan::build).
	smaa_edge.set_build_render_pass([&graph, &edge_input_res, &edge_output_res, metrics, q = smaa_quality, plan](HIP::CommandBuffer &cmd) {
		const StripPlan *strip = live(plan);
		auto &input_image = graph.get_physical_texture_resource(edge_input_res);
		auto &edges = graph.get_physical_texture_resource(edge_output_res);
		auto push = metrics(input_image);
		gr_rows rows;
		if (to_rows(strip ? &strip->smaa_edges : nullptr, rows))
			cmd.check(gr_smaa_edge_detection_rows(cmd.get_context(), cmd.get_stream(), &input_image.get_view(), &edges.get_view(), &push, q, &rows),
			          "smaa-edge");
	});

	smaa_weight.set_build_render_pass([&graph, &weight_input_res, &weight_output_res, metrics, q = smaa_quality, plan](HIP::CommandBuffer &cmd) {
		const StripPlan *strip = live(plan);
		auto &edges = graph.get_physical_texture_resource(weight_input_res);
		auto &weights = graph.get_physical_texture_resource(weight_output_res);
		auto push = metrics(edges);
		gr_rows rows;
		if (to_rows(strip ? &strip->smaa_weights : nullptr, rows))
			cmd.check(gr_smaa_blend_weight_rows(cmd.get_context(), cmd.get_stream(), &edges.get_view(), &weights.get_view(), &push, q, &rows),
			          "smaa-weights");
	});

	smaa_blend.set_build_render_pass([&graph, &smaa_blend, &blend_input_res, &blend_weight_res, metrics, plan](HIP::CommandBuffer &cmd) {
		const StripPlan *strip = live(plan);
		auto &input_image = graph.get_physical_texture_resource(blend_input_res);
		auto &blend_image = graph.get_physical_texture_resource(blend_weight_res);
		auto &output_image = graph.get_physical_texture_resource(*smaa_blend.get_color_outputs()[0]);
		auto push = metrics(input_image);
		gr_rows rows;
		acquire_output(strip, cmd, output_image);
		if (to_rows(strip ? &strip->aa_out : nullptr, rows))
			cmd.check(gr_smaa_neighbor_blend_rows(cmd.get_context(), cmd.get_stream(), &input_image.get_view(), &blend_image.get_view(),
			                                      &output_image.get_view(), &push, &rows),
			          "smaa-blend");
		gather_output(strip, cmd, output_image, "smaa-blend");
	});
}

// ---- TAA (temporal.cpp:199-266) ---------------------------------------------------------------------------------------------
void setup_taa_resolve(RenderGraph &graph, TemporalJitter &jitter, float scaling_factor, const std::string &input, const std::string &input_depth,
                       const std::string &input_mv, const std::string &output, TAAQuality quality, const StripPlan *plan)
{
	jitter.init(TemporalJitter::Type::TAA_16Phase,
	            vec2(float(graph.get_backbuffer_dimensions().width) * scaling_factor, float(graph.get_backbuffer_dimensions().height) * scaling_factor));

	AttachmentInfo taa_output;
	taa_output.size_class = SizeClass::InputRelative;
	taa_output.size_relative_name = input;
	// temporal.cpp:211-216: B10G11R11_UFLOAT_PACK32 where renderable, history RGBA16F.  Here the colour output takes the format of
	// its input: packed when the HDR targets are (renderTargetFp16 = false), RGBA16F with the RGBA16F targets SURVEY 8d measures on.
	const VkFormat input_format = graph.get_texture_resource(input).get_attachment_info().format;
	taa_output.format = input_format == VK_FORMAT_B10G11R11_UFLOAT_PACK32 ? VK_FORMAT_B10G11R11_UFLOAT_PACK32 : VK_FORMAT_R16G16B16A16_SFLOAT;
	AttachmentInfo taa_history = taa_output;
	taa_history.format = VK_FORMAT_R16G16B16A16_SFLOAT;

	auto &resolve = graph.add_pass("taa-resolve", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
	auto &out_color = resolve.add_color_output(output, taa_output);
	auto &out_history = resolve.add_color_output(output + "-history", taa_history);
	auto &input_res = resolve.add_texture_input(input);
	auto &input_res_mv = resolve.add_texture_input(input_mv);
	auto &input_depth_res = resolve.add_texture_input(input_depth);
	auto &history = resolve.add_history_input(output + "-history");

	// Row bands with a bounded history reach: the boundary blocks of every rank's history chunk travel through this image (rank g's
	// rows [2 X g, 2 X g + X) = the first X rows of its chunk, the next X rows = the last X), one all-gather per frame.
	auto halo_staging = std::make_shared<HIP::ImageHandle>();

	resolve.set_build_render_pass(
	    [&graph, &jitter, &out_color, &out_history, &input_res, &input_res_mv, &input_depth_res, &history, q = int(quality), plan, halo_staging](HIP::CommandBuffer &cmd) {
		    const StripPlan *strip = live(plan);
		    auto &image = graph.get_physical_texture_resource(input_res);
		    auto &image_mv = graph.get_physical_texture_resource(input_res_mv);
		    auto &depth = graph.get_physical_texture_resource(input_depth_res);
		    auto *prev = graph.get_physical_history_texture_resource(history);
		    auto &color = graph.get_physical_texture_resource(out_color);
		    auto &hist = graph.get_physical_texture_resource(out_history);

		    gr_push_taa push = {};
		    mat4 reproj = translate(vec3(0.5f, 0.5f, 0.0f)) * scale(vec3(0.5f, 0.5f, 1.0f)) * jitter.get_history_view_proj(1) *
		                  jitter.get_history_inv_view_proj(0);
		    memcpy(push.reproj, reproj.data(), sizeof(push.reproj));
		    push.rt_metrics[0] = 1.0f / float(image.get_width());
		    push.rt_metrics[1] = 1.0f / float(image.get_height());
		    push.rt_metrics[2] = float(image.get_width());
		    push.rt_metrics[3] = float(image.get_height());
		    gr_rows rows, held;
		    const bool bounded = strip && strip->exchange && strip->taa_exchange_rows != 0;
		    if (to_rows(strip ? &strip->taa : nullptr, rows))
		    {
			    if (bounded && prev && to_rows(&strip->taa_history_held, held))
				    cmd.check(gr_taa_resolve_band(cmd.get_context(), cmd.get_stream(), &image.get_view(), &depth.get_view(), &image_mv.get_view(),
				                                  &prev->get_view(), &color.get_view(), &hist.get_view(), &push, q, &rows, &held, strip->taa_reach_flag),
				              "taa-resolve");
			    else
				    cmd.check(gr_taa_resolve_rows(cmd.get_context(), cmd.get_stream(), &image.get_view(), &depth.get_view(), &image_mv.get_view(),
				                                  prev ? &prev->get_view() : nullptr, &color.get_view(), &hist.get_view(), &push, q, &rows),
				              "taa-resolve");
		    }
		    if (!strip || !strip->exchange)
			    return;
		    if (!bounded)
		    {
			    // Next frame's reprojection may read the history anywhere: the bands meet in every rank's history image (same
			    // chunking as the output image; rows a rank resolved beyond its chunk are overwritten with the owner's
			    // identical values).
			    strip->exchange(cmd, hist, strip->out_chunk_rows, "taa-history");
			    return;
		    }
		    // Bounded reach: only the boundary blocks travel.  X rows from either end of every chunk into the staging image, one
		    // all-gather, the upper neighbour's last block above the own chunk and the lower neighbour's first block below it.
		    const uint32_t X = strip->taa_exchange_rows, C = strip->out_chunk_rows, H = hist.get_height(), g = strip->index;
		    const size_t pitch = hist.get_view().pitch_bytes;
		    auto &staging = *halo_staging;
		    if (!staging || staging->get_width() != hist.get_width() || staging->get_height() != strip->count * 2 * X)
			    staging = cmd.get_device().create_image(hist.get_width(), strip->count * 2 * X, hist.get_format(), "taa-history-halo");
		    auto *base = static_cast<uint8_t *>(hist.get_device_pointer());
		    auto *stage = static_cast<uint8_t *>(staging->get_device_pointer());
		    auto copy_rows = [&](uint8_t *dst, const uint8_t *src, uint32_t count) {
			    cmd.check(gr_copy(cmd.get_context(), cmd.get_stream(), dst, src, size_t(count) * pitch), "taa-history-halo");
		    };
		    const uint32_t own_first = std::min(g * C, H), own_end = std::min(own_first + C, H);
		    if (own_end - own_first >= X)
		    {
			    copy_rows(stage + size_t(2 * X * g) * pitch, base + size_t(own_first) * pitch, X);
			    copy_rows(stage + size_t(2 * X * g + X) * pitch, base + size_t(own_end - X) * pitch, X);
		    }
		    strip->exchange(cmd, *staging, 2 * X, "taa-history-halo");
		    if (g > 0 && own_first >= X)
			    copy_rows(base + size_t(own_first - X) * pitch, stage + size_t(2 * X * (g - 1) + X) * pitch, X);
		    if (g + 1 < strip->count && own_end + X <= H)
			    copy_rows(base + size_t(own_end) * pitch, stage + size_t(2 * X * (g + 1)) * pitch, X);
	    });
}

// ---- dispatcher (aa.cpp:176-290) ----------------------------------------------------------------------------------------------
bool setup_before_post_chain_antialiasing(PostAAType type, RenderGraph &graph, TemporalJitter &jitter, const RenderContext &, float scaling_factor,
                                          const std::string &input, const std::string &input_depth, const std::string &input_mv,
                                          const std::string &output, const StripPlan *strip)
{
	TAAQuality taa_quality;
	switch (type)
	{
	case PostAAType::TAA_Low: taa_quality = TAAQuality::Low; break;
	case PostAAType::TAA_Medium: taa_quality = TAAQuality::Medium; break;
	case PostAAType::TAA_High: taa_quality = TAAQuality::High; break;
	case PostAAType::TAA_FSR2: throw std::logic_error("FSR2 needs third_party/fsr2, which the reference checkout does not contain.");
	default: return false;
	}
	setup_taa_resolve(graph, jitter, scaling_factor, input, input_depth, input_mv, output, taa_quality, strip);
	return true;
}

// ---- spatial upscaling (aa.cpp:64-174) ------------------------------------------------------------------------------------
bool setup_after_post_chain_upscaling(RenderGraph &graph, const std::string &input, const std::string &output, bool use_sharpen, bool fp16)
{
	auto &upscale = graph.add_pass(output + "-scale", RenderGraph::get_default_post_graphics_queue());
	AttachmentInfo upscale_info; // swapchain sized
	upscale_info.format = VK_FORMAT_R8G8B8A8_UNORM;
	if (use_sharpen)
		upscale_info.flags |= ATTACHMENT_INFO_UNORM_SRGB_ALIAS_BIT; // the sharpen pass reads it through an sRGB view
	else
		upscale_info.flags |= ATTACHMENT_INFO_SUPPORTS_PREROTATE_BIT;
	auto &upscaled = upscale.add_color_output(use_sharpen ? output + "-scale" : output, upscale_info);
	auto &source = upscale.add_texture_input(input);
	graph.get_texture_resource(input).get_attachment_info().flags |= ATTACHMENT_INFO_UNORM_SRGB_ALIAS_BIT; // read as stored bytes

	upscale.set_build_render_pass([&graph, &upscaled, &source, fp16](HIP::CommandBuffer &cmd) {
		auto &in = graph.get_physical_texture_resource(source);
		auto &out = graph.get_physical_texture_resource(upscaled);
		cmd.check(gr_fsr_upscale(cmd.get_context(), cmd.get_stream(), &in.get_view(), &out.get_view(), fp16 ? 1 : 0), "fsr upscale");
	});

	if (use_sharpen)
	{
		auto &sharpen = graph.add_pass(output + "-sharpen", RenderGraph::get_default_post_graphics_queue());
		AttachmentInfo sharpen_info; // swapchain size and format
		sharpen_info.flags |= ATTACHMENT_INFO_SUPPORTS_PREROTATE_BIT;
		auto &sharpened = sharpen.add_color_output(output, sharpen_info);
		auto &scaled = sharpen.add_texture_input(output + "-scale");
		sharpen.set_build_render_pass([&graph, &sharpened, &scaled](HIP::CommandBuffer &cmd) {
			auto &in = graph.get_physical_texture_resource(scaled);
			auto &out = graph.get_physical_texture_resource(sharpened);
			const float sharpness = exp2f(-0.5f); // FsrRcasCon(constants.params, 0.5f), aa.cpp:64-74,157
			cmd.check(gr_fsr_sharpen(cmd.get_context(), cmd.get_stream(), &in.get_view(), &out.get_view(), sharpness), "fsr sharpen");
		});
	}
	return true;
}

unsigned smaa_search_steps(PostAAType type)
{
	switch (type)
	{
	case PostAAType::SMAA_Low: return 4;
	case PostAAType::SMAA_Medium: return 8;
	case PostAAType::SMAA_High: return 16;
	case PostAAType::SMAA_Ultra: return 32;
	default: return 0;
	}
}

bool setup_after_post_chain_antialiasing(PostAAType type, RenderGraph &graph, TemporalJitter &jitter, float scaling_factor, const std::string &input,
                                         const std::string &input_depth, const std::string &output, const StripPlan *strip)
{
	switch (type)
	{
	case PostAAType::None:
		jitter.init(TemporalJitter::Type::None, vec2(0.0f));
		return false;
	case PostAAType::FXAA:
		setup_fxaa_postprocess(graph, input, output, VK_FORMAT_UNDEFINED, strip);
		return true;
	case PostAAType::SMAA_Low:
		setup_smaa_postprocess(graph, jitter, scaling_factor, input, input_depth, output, SMAAPreset::Low, strip);
		return true;
	case PostAAType::SMAA_Medium:
		setup_smaa_postprocess(graph, jitter, scaling_factor, input, input_depth, output, SMAAPreset::Medium, strip);
		return true;
	case PostAAType::SMAA_High:
		setup_smaa_postprocess(graph, jitter, scaling_factor, input, input_depth, output, SMAAPreset::High, strip);
		return true;
	case PostAAType::SMAA_Ultra:
		setup_smaa_postprocess(graph, jitter, scaling_factor, input, input_depth, output, SMAAPreset::Ultra, strip);
		return true;
	case PostAAType::FXAA_2Phase:
	case PostAAType::SMAA_Ultra_T2X:
		throw std::logic_error("fxaa2phase / smaaUltraT2X reference shaders do not compile (SURVEY.md §2.2); not provided.");
	default:
		return false;
	}
}

PostAAType string_to_post_antialiasing_type(const char *type)
{
	static const struct { const char *name; PostAAType type; } table[] = {
		{"fxaa", PostAAType::FXAA}, {"fxaa2phase", PostAAType::FXAA_2Phase}, {"smaaLow", PostAAType::SMAA_Low},
		{"smaaMedium", PostAAType::SMAA_Medium}, {"smaaHigh", PostAAType::SMAA_High}, {"smaaUltra", PostAAType::SMAA_Ultra},
		{"smaaUltraT2X", PostAAType::SMAA_Ultra_T2X}, {"taaLow", PostAAType::TAA_Low}, {"taaMedium", PostAAType::TAA_Medium},
		{"taaHigh", PostAAType::TAA_High}, {"taaFSR2", PostAAType::TAA_FSR2}, {"none", PostAAType::None},
	};
	if (type)
		for (auto &e : table)
			if (strcmp(e.name, type) == 0)
				return e.type;
	return PostAAType::None;
}
} // namespace Granite
