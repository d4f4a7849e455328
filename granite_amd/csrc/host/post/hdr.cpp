// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// renderer/post/hdr.cpp restated on the HIP executor: identical pass / resource names, formats, size classes and push
// constants; each recorded dispatch or full-screen quad becomes one C-ABI kernel launch.
#include "hdr.hpp"
#include <cstring>
#include <cmath>

namespace Granite
{
namespace
{
const gr_luminance_data *luminance_ptr(RenderGraph &graph, const RenderBufferResource *res)
{
	return res ? static_cast<const gr_luminance_data *>(graph.get_physical_buffer_resource(*res).get_device_pointer()) : nullptr;
}

// luminance_build_compute / luminance_build_render_pass (hdr.cpp:35-98)
void record_luminance(HIP::CommandBuffer &cmd, const FrameParameters &frame, RenderGraph &graph, const RenderBufferResource &lum,
                      const RenderTextureResource &d3)
{
	auto &input = graph.get_physical_texture_resource(d3);
	auto &output = graph.get_physical_buffer_resource(lum);
	gr_push_luminance push = {};
	push.size[0] = input.get_width() / 2;
	push.size[1] = input.get_height() / 2;
	push.lerp = float(1.0 - std::pow(0.5, frame.frame_time));
	push.min_loglum = -3.0f;
	push.max_loglum = 2.0f;
	cmd.check(gr_luminance(cmd.get_context(), cmd.get_stream(), &input.get_view(),
	                       static_cast<gr_luminance_data *>(output.get_device_pointer()), &push),
	          "luminance");
}

// bloom_threshold_build_compute / _render_pass (hdr.cpp:100-144)
void record_threshold(HIP::CommandBuffer &cmd, RenderGraph &graph, const RenderTextureResource &threshold, const RenderTextureResource &hdr,
                      const RenderBufferResource *ubo, const RowRange *range = nullptr)
{
	gr_rows rows;
	if (!to_rows(range, rows))
		return;
	auto &output = graph.get_physical_texture_resource(threshold);
	auto &input = graph.get_physical_texture_resource(hdr);
	gr_push_bloom_threshold push = {};
	push.threads[0] = output.get_width();
	push.threads[1] = output.get_height();
	push.inv_output_size[0] = 1.0f / float(push.threads[0]);
	push.inv_output_size[1] = 1.0f / float(push.threads[1]);
	cmd.check(gr_bloom_threshold_rows(cmd.get_context(), cmd.get_stream(), &input.get_view(), &output.get_view(), luminance_ptr(graph, ubo),
	                                  &push, &rows),
	          "bloom_threshold");
}

// bloom_downsample_build_compute / _render_pass (hdr.cpp:146-187,218-270)
void record_downsample(HIP::CommandBuffer &cmd, const FrameParameters &frame, RenderGraph &graph, const RenderTextureResource &output_res,
                       const RenderTextureResource &input_res, const RenderTextureResource *feedback, const RowRange *range = nullptr)
{
	gr_rows rows;
	if (!to_rows(range, rows))
		return;
	auto &output = graph.get_physical_texture_resource(output_res);
	auto &input = graph.get_physical_texture_resource(input_res);
	HIP::ImageView *history = feedback ? graph.get_physical_history_texture_resource(*feedback) : nullptr; // null on frame 0

	gr_push_bloom_downsample push = {};
	push.threads[0] = output.get_width();
	push.threads[1] = output.get_height();
	push.inv_output_size[0] = 1.0f / float(push.threads[0]);
	push.inv_output_size[1] = 1.0f / float(push.threads[1]);
	push.inv_input_size[0] = 1.0f / float(input.get_width());
	push.inv_input_size[1] = 1.0f / float(input.get_height());
	push.lerp = float(1.0 - std::pow(0.001, frame.frame_time));
	cmd.check(gr_bloom_downsample_rows(cmd.get_context(), cmd.get_stream(), &input.get_view(), &output.get_view(),
	                                   history ? &history->get_view() : nullptr, &push, &rows),
	          "bloom_downsample");
}

// bloom_upsample_build_compute / _render_pass (hdr.cpp:189-216,272-281)
void record_upsample(HIP::CommandBuffer &cmd, RenderGraph &graph, const RenderTextureResource &output_res, const RenderTextureResource &input_res,
                     const RowRange *range = nullptr)
{
	gr_rows rows;
	if (!to_rows(range, rows))
		return;
	auto &output = graph.get_physical_texture_resource(output_res);
	auto &input = graph.get_physical_texture_resource(input_res);
	gr_push_bloom_upsample push = {};
	push.threads[0] = output.get_width();
	push.threads[1] = output.get_height();
	push.inv_output_size[0] = 1.0f / float(push.threads[0]);
	push.inv_output_size[1] = 1.0f / float(push.threads[1]);
	push.inv_input_size[0] = 1.0f / float(input.get_width());
	push.inv_input_size[1] = 1.0f / float(input.get_height());
	cmd.check(gr_bloom_upsample_rows(cmd.get_context(), cmd.get_stream(), &input.get_view(), &output.get_view(), &push, &rows),
	          "bloom_upsample");
}

gr_push_bloom_downsample downsample_push(const FrameParameters &frame, HIP::ImageView &output, HIP::ImageView &input)
{
	gr_push_bloom_downsample push = {};
	push.threads[0] = output.get_width();
	push.threads[1] = output.get_height();
	push.inv_output_size[0] = 1.0f / float(push.threads[0]);
	push.inv_output_size[1] = 1.0f / float(push.threads[1]);
	push.inv_input_size[0] = 1.0f / float(input.get_width());
	push.inv_input_size[1] = 1.0f / float(input.get_height());
	push.lerp = float(1.0 - std::pow(0.001, frame.frame_time));
	return push;
}

gr_push_bloom_upsample upsample_push(HIP::ImageView &output, HIP::ImageView &input)
{
	gr_push_bloom_upsample push = {};
	push.threads[0] = output.get_width();
	push.threads[1] = output.get_height();
	push.inv_output_size[0] = 1.0f / float(push.threads[0]);
	push.inv_output_size[1] = 1.0f / float(push.threads[1]);
	push.inv_input_size[0] = 1.0f / float(input.get_width());
	push.inv_input_size[1] = 1.0f / float(input.get_height());
	return push;
}

// The dispatches hdr.cpp:364-377 records for downsample-2, downsample-3, the luminance reduction, upsample-2 and upsample-1,
// as the two fused launches of the C ABI when the pyramid qualifies (gr_bloom_tail_supported); returns false otherwise and
// records nothing.  Same values in every level either way.
// u0_res (whole-image frames only): upsample-0 joins the second launch when the frame qualifies (gr_bloom_up_all_supported); *u0_done says
// whether it did.
bool record_pyramid_tail(HIP::CommandBuffer &cmd, const FrameParameters &frame, RenderGraph &graph, const RenderTextureResource &d1_res,
                         const RenderTextureResource &d2_res, const RenderTextureResource &d3_res, const RenderTextureResource &u2_res,
                         const RenderTextureResource &u1_res, const RenderBufferResource *lum_res, const RenderTextureResource *u0_res = nullptr,
                         bool *u0_done = nullptr, bool busy_frame = false)
{
	if (u0_done)
		*u0_done = false;
	auto &d1 = graph.get_physical_texture_resource(d1_res);
	auto &d2 = graph.get_physical_texture_resource(d2_res);
	auto &d3 = graph.get_physical_texture_resource(d3_res);
	auto &u2 = graph.get_physical_texture_resource(u2_res);
	auto &u1 = graph.get_physical_texture_resource(u1_res);
	HIP::ImageView *history = graph.get_physical_history_texture_resource(d3_res); // null on frame 0
	if (!history)
		return false;
	const gr_push_bloom_downsample push_d2 = downsample_push(frame, d2, d1), push_d3 = downsample_push(frame, d3, d2);
	const gr_push_bloom_upsample push_u2 = upsample_push(u2, d3), push_u1 = upsample_push(u1, u2);
	if (!gr_bloom_tail_supported(&d1.get_view(), &d2.get_view(), &d3.get_view(), &u2.get_view(), &u1.get_view(), &push_d2, &push_d3, &push_u2, &push_u1))
		return false;
	cmd.check(gr_bloom_down_tail(cmd.get_context(), cmd.get_stream(), &d1.get_view(), &d2.get_view(), &d3.get_view(), &history->get_view(), &push_d2,
	                             &push_d3),
	          "bloom_down_tail");
	cmd.barrier(VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_ACCESS_2_SHADER_STORAGE_WRITE_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT,
	            VK_ACCESS_2_SHADER_SAMPLED_READ_BIT);
	gr_push_luminance push_lum = {};
	gr_luminance_data *lum = nullptr;
	if (lum_res)
	{
		push_lum.size[0] = d3.get_width() / 2;
		push_lum.size[1] = d3.get_height() / 2;
		push_lum.lerp = float(1.0 - std::pow(0.5, frame.frame_time));
		push_lum.min_loglum = -3.0f;
		push_lum.max_loglum = 2.0f;
		lum = static_cast<gr_luminance_data *>(graph.get_physical_buffer_resource(*lum_res).get_device_pointer());
	}
	if (u0_res)
	{
		auto &u0 = graph.get_physical_texture_resource(*u0_res);
		const gr_push_bloom_upsample push_u0 = upsample_push(u0, u1);
		if (gr_bloom_up_all_supported(&d3.get_view(), &u2.get_view(), &u1.get_view(), &u0.get_view(), &push_u2, &push_u1, &push_u0))
		{
			cmd.check(gr_bloom_up_all(cmd.get_context(), cmd.get_stream(), &d3.get_view(), &u2.get_view(), &u1.get_view(), &u0.get_view(), lum, &push_u2,
			                          &push_u1, &push_u0, lum ? &push_lum : nullptr, busy_frame ? GR_BLOOM_BUSY_FRAME_BIT : 0u),
			          "bloom_up_all");
			if (u0_done)
				*u0_done = true;
			return true;
		}
	}
	cmd.check(gr_bloom_up_tail(cmd.get_context(), cmd.get_stream(), &d3.get_view(), &u2.get_view(), &u1.get_view(), lum, &push_u2, &push_u1,
	                           lum ? &push_lum : nullptr),
	          "bloom_up_tail");
	return true;
}

// The dispatches of downsample-0 and downsample-1 (hdr.cpp:358-362) as one fused launch of the C ABI when the levels qualify; returns
// false otherwise and records nothing.  Under row bands the launch is restricted to this rank's rows of downsample-1; downsample-0
// is written under their taps (StripPlan::d0 is that footprint).
bool record_pyramid_middle(HIP::CommandBuffer &cmd, const FrameParameters &frame, RenderGraph &graph, const RenderTextureResource &t_res,
                           const RenderTextureResource &d0_res, const RenderTextureResource &d1_res, const RowRange *rows_d1)
{
	auto &t = graph.get_physical_texture_resource(t_res);
	auto &d0 = graph.get_physical_texture_resource(d0_res);
	auto &d1 = graph.get_physical_texture_resource(d1_res);
	const gr_push_bloom_downsample push_d0 = downsample_push(frame, d0, t), push_d1 = downsample_push(frame, d1, d0);
	if (!gr_bloom_down_mid_supported(&t.get_view(), &d0.get_view(), &d1.get_view(), &push_d0, &push_d1))
		return false;
	gr_rows rows;
	if (to_rows(rows_d1, rows))
		cmd.check(gr_bloom_down_mid(cmd.get_context(), cmd.get_stream(), &t.get_view(), &d0.get_view(), &d1.get_view(), &push_d0, &push_d1, &rows),
		          "bloom_down_mid");
	return true;
}

// The threshold dispatch and the dispatches of downsample-0 and downsample-1 (hdr.cpp:354-362) as ONE fused launch of the C ABI when the
// frame qualifies (gr_bloom_down_head_supported: every level exactly half of its input, up to 1440p); returns false otherwise and
// records nothing.  Whole images only: row bands keep the band-limited launches.
bool record_pyramid_head(HIP::CommandBuffer &cmd, const FrameParameters &frame, RenderGraph &graph, const RenderTextureResource &hdr_res,
                         const RenderTextureResource &t_res, const RenderTextureResource &d0_res, const RenderTextureResource &d1_res,
                         const RenderBufferResource *ubo)
{
	auto &hdr = graph.get_physical_texture_resource(hdr_res);
	auto &t = graph.get_physical_texture_resource(t_res);
	auto &d0 = graph.get_physical_texture_resource(d0_res);
	auto &d1 = graph.get_physical_texture_resource(d1_res);
	gr_push_bloom_threshold push_t = {};
	push_t.threads[0] = t.get_width();
	push_t.threads[1] = t.get_height();
	push_t.inv_output_size[0] = 1.0f / float(push_t.threads[0]);
	push_t.inv_output_size[1] = 1.0f / float(push_t.threads[1]);
	const gr_push_bloom_downsample push_d0 = downsample_push(frame, d0, t), push_d1 = downsample_push(frame, d1, d0);
	if (!gr_bloom_down_head_supported(&hdr.get_view(), &t.get_view(), &d0.get_view(), &d1.get_view(), &push_t, &push_d0, &push_d1))
		return false;
	cmd.check(gr_bloom_down_head(cmd.get_context(), cmd.get_stream(), &hdr.get_view(), &t.get_view(), &d0.get_view(), &d1.get_view(),
	                             luminance_ptr(graph, ubo), &push_t, &push_d0, &push_d1),
	          "bloom_down_head");
	return true;
}

// Every dispatch of the pass (hdr.cpp:354-379) as ONE launch of the C ABI when the frame qualifies (gr_bloom_pyramid_supported: what the fused
// head, tail and upsample launches require, up to a 640 x 384 frame); returns false otherwise and records nothing.  Whole images only.
bool record_pyramid_whole(HIP::CommandBuffer &cmd, const FrameParameters &frame, RenderGraph &graph, const RenderTextureResource &hdr_res,
                          const RenderTextureResource &t_res, const RenderTextureResource &d0_res, const RenderTextureResource &d1_res,
                          const RenderTextureResource &d2_res, const RenderTextureResource &d3_res, const RenderTextureResource &u2_res,
                          const RenderTextureResource &u1_res, const RenderTextureResource &u0_res, const RenderBufferResource *lum_res)
{
	HIP::ImageView *history = graph.get_physical_history_texture_resource(d3_res); // null on frame 0
	if (!history)
		return false;
	auto &hdr = graph.get_physical_texture_resource(hdr_res);
	auto &t = graph.get_physical_texture_resource(t_res);
	auto &d0 = graph.get_physical_texture_resource(d0_res);
	auto &d1 = graph.get_physical_texture_resource(d1_res);
	auto &d2 = graph.get_physical_texture_resource(d2_res);
	auto &d3 = graph.get_physical_texture_resource(d3_res);
	auto &u2 = graph.get_physical_texture_resource(u2_res);
	auto &u1 = graph.get_physical_texture_resource(u1_res);
	auto &u0 = graph.get_physical_texture_resource(u0_res);
	gr_bloom_pyramid_args a = {};
	a.hdr = hdr.get_view(), a.threshold = t.get_view(), a.d0 = d0.get_view(), a.d1 = d1.get_view(), a.d2 = d2.get_view(), a.d3 = d3.get_view();
	a.history = history->get_view(), a.u2 = u2.get_view(), a.u1 = u1.get_view(), a.u0 = u0.get_view();
	a.push_threshold.threads[0] = t.get_width();
	a.push_threshold.threads[1] = t.get_height();
	a.push_threshold.inv_output_size[0] = 1.0f / float(a.push_threshold.threads[0]);
	a.push_threshold.inv_output_size[1] = 1.0f / float(a.push_threshold.threads[1]);
	a.push_d0 = downsample_push(frame, d0, t), a.push_d1 = downsample_push(frame, d1, d0);
	a.push_d2 = downsample_push(frame, d2, d1), a.push_d3 = downsample_push(frame, d3, d2);
	a.push_u2 = upsample_push(u2, d3), a.push_u1 = upsample_push(u1, u2), a.push_u0 = upsample_push(u0, u1);
	if (lum_res)
	{
		a.lum = static_cast<gr_luminance_data *>(graph.get_physical_buffer_resource(*lum_res).get_device_pointer());
		a.push_luminance.size[0] = d3.get_width() / 2;
		a.push_luminance.size[1] = d3.get_height() / 2;
		a.push_luminance.lerp = float(1.0 - std::pow(0.5, frame.frame_time));
		a.push_luminance.min_loglum = -3.0f;
		a.push_luminance.max_loglum = 2.0f;
	}
	if (!gr_bloom_pyramid_supported(&a))
		return false;
	cmd.check(gr_bloom_pyramid(cmd.get_context(), cmd.get_stream(), &a), "bloom_pyramid");
	return true;
}

// tonemap_build_render_pass (hdr.cpp:283-306)
void record_tonemap(RenderPass &pass, HIP::CommandBuffer &cmd, const RenderTextureResource &hdr_res, const RenderTextureResource &bloom_res,
                    const RenderBufferResource *ubo, const HDRDynamicExposureInterface *iface, const StripPlan *strip = nullptr)
{
	auto &graph = pass.get_graph();
	auto &hdr = graph.get_physical_texture_resource(hdr_res);
	auto &bloom = graph.get_physical_texture_resource(bloom_res);
	auto &output = graph.get_physical_texture_resource(*pass.get_color_outputs()[0]);
	gr_push_tonemap push = {iface ? iface->get_exposure() : 1.0f};
	gr_rows rows;
	const bool final_pass = !strip || !strip->post_aa(); // a post-tonemap AA pass owns the output image and its gather
	if (strip && strip->acquire_output && final_pass)
		strip->acquire_output(cmd, output);
	if (to_rows(strip ? &strip->tonemap : nullptr, rows))
		cmd.check(gr_tonemap_rows(cmd.get_context(), cmd.get_stream(), &hdr.get_view(), &bloom.get_view(), &output.get_view(),
		                          luminance_ptr(graph, ubo), &push, &rows),
		          "tonemap");
	// Row-band tiling: the tonemapped bands of all ranks meet in every rank's output image.
	if (!final_pass)
		return;
	if (strip && strip->exchange_output)
		strip->exchange_output(cmd, output, strip->out_chunk_rows, "tonemapped");
	else if (strip && strip->exchange)
		strip->exchange(cmd, output, strip->out_chunk_rows, "tonemapped");
}

AttachmentInfo bloom_level_info(const std::string &input, float scale)
{
	AttachmentInfo info;
	info.format = VK_FORMAT_R16G16B16A16_SFLOAT;
	info.size_x = scale;
	info.size_y = scale;
	info.size_class = SizeClass::InputRelative;
	info.size_relative_name = input;
	info.aux_usage = VK_IMAGE_USAGE_SAMPLED_BIT;
	return info;
}
} // namespace

void setup_hdr_postprocess_compute(RenderGraph &graph, const FrameParameters &frame, const std::string &input, const std::string &output,
                                   const HDROptions &options, const HDRDynamicExposureInterface *iface)
{
	BufferInfo buffer_info;
	buffer_info.size = 3 * sizeof(float);
	buffer_info.usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT | VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT;

	auto &bloom_pass = graph.add_pass("bloom-compute", RenderGraph::get_default_compute_queue());
	auto &t = bloom_pass.add_storage_texture_output("threshold", bloom_level_info(input, 0.5f));
	auto &d0 = bloom_pass.add_storage_texture_output("downsample-0", bloom_level_info(input, 0.25f));
	auto &u0 = bloom_pass.add_storage_texture_output("upsample-0", bloom_level_info(input, 0.25f));
	auto &d1 = bloom_pass.add_storage_texture_output("downsample-1", bloom_level_info(input, 0.125f));
	auto &u1 = bloom_pass.add_storage_texture_output("upsample-1", bloom_level_info(input, 0.125f));
	auto &d2 = bloom_pass.add_storage_texture_output("downsample-2", bloom_level_info(input, 0.0625f));
	auto &u2 = bloom_pass.add_storage_texture_output("upsample-2", bloom_level_info(input, 0.0625f));
	auto &d3 = bloom_pass.add_storage_texture_output("downsample-3", bloom_level_info(input, 0.03125f));

	const RenderBufferResource *lum = nullptr;
	if (options.dynamic_exposure)
		lum = &bloom_pass.add_storage_output("average-luminance", buffer_info);

	auto &hdr = bloom_pass.add_texture_input(input);
	bloom_pass.add_history_input("downsample-3");

	// Recorded order = hdr.cpp:354-379.  The threshold reads LAST frame's exposure (same buffer, updated later in this
	// pass); cmd.barrier() between dispatches is stream order here.
	// Row-band tiling (options.strip): the band-limited dispatches compute this rank's part of threshold / d0 / d1, the
	// 1/8 level is all-gathered, everything coarser is replicated, u0 is computed where the tonemap band samples it.
	// (A one-rank plan with an exchange installed still runs the exchange points: that is how the transport is tested.)
	const StripPlan *plan = options.strip;
	const bool busy_frame = options.busy_frame;
	bloom_pass.set_build_render_pass([&graph, &frame, &t, &d0, &d1, &d2, &d3, &u0, &u1, &u2, &hdr, ubo = lum, plan, busy_frame](HIP::CommandBuffer &cmd) {
		const StripPlan *strip = plan && (plan->active() || plan->exchange) ? plan : nullptr;
		const auto record = [&]() {
			const auto compute_to_compute = [&cmd]() {
				cmd.barrier(VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_ACCESS_2_SHADER_STORAGE_WRITE_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT,
				            VK_ACCESS_2_SHADER_SAMPLED_READ_BIT);
			};
			if (!strip && record_pyramid_whole(cmd, frame, graph, hdr, t, d0, d1, d2, d3, u2, u1, u0, ubo))
				return;
			if (strip || !record_pyramid_head(cmd, frame, graph, hdr, t, d0, d1, ubo))
			{
				record_threshold(cmd, graph, t, hdr, ubo, strip ? &strip->threshold : nullptr);
				compute_to_compute();
				if (!record_pyramid_middle(cmd, frame, graph, t, d0, d1, strip ? &strip->d1 : nullptr))
				{
					record_downsample(cmd, frame, graph, d0, t, nullptr, strip ? &strip->d0 : nullptr);
					compute_to_compute();
					record_downsample(cmd, frame, graph, d1, d0, nullptr, strip ? &strip->d1 : nullptr);
				}
			}
			if (strip && strip->exchange)
				strip->exchange(cmd, graph.get_physical_texture_resource(d1), strip->d1_chunk_rows, "downsample-1");
			compute_to_compute();
			bool u0_done = false;
			if (!record_pyramid_tail(cmd, frame, graph, d1, d2, d3, u2, u1, ubo, strip ? nullptr : &u0, &u0_done, busy_frame))
			{
				record_downsample(cmd, frame, graph, d2, d1, nullptr);
				compute_to_compute();
				record_downsample(cmd, frame, graph, d3, d2, &d3);
				compute_to_compute();
				if (ubo)
					record_luminance(cmd, frame, graph, *ubo, d3);
				record_upsample(cmd, graph, u2, d3);
				compute_to_compute();
				record_upsample(cmd, graph, u1, u2);
			}
			if (!u0_done)
			{
				compute_to_compute();
				record_upsample(cmd, graph, u0, u1, strip ? &strip->u0 : nullptr);
			}
		};
		if (strip)
		{
			record(); // the band exchange in the middle of the sequence is not a launch
			return;
		}
		// Six launches with nothing between them and every argument a function of the attachments: replayed as one
		// pre-recorded sequence once a key comes back (two keys in steady state: the feedback history ping-pongs).
		HIP::CommandBuffer::LaunchKey key;
		for (const RenderTextureResource *res : {&t, &d0, &d1, &d2, &d3, &u0, &u1, &u2})
			key.add(graph.get_physical_texture_resource(*res).get_view());
		key.add(graph.get_physical_texture_resource(hdr).get_view());
		const HIP::ImageView *history = graph.get_physical_history_texture_resource(d3);
		key.add(history ? history->get_device_pointer() : nullptr);
		key.add(ubo ? graph.get_physical_buffer_resource(*ubo).get_device_pointer() : nullptr);
		key.add(frame.frame_time);
		cmd.replayable("bloom-compute", key,
		               {"bloom_threshold", "bloom_downsample", "bloom_down_head", "bloom_down_mid", "bloom_down_tail", "bloom_up_tail", "bloom_up_all", "bloom_pyramid", "luminance", "bloom_upsample"}, record);
	});

	{
		AttachmentInfo tonemap_info;
		tonemap_info.flags |= ATTACHMENT_INFO_SUPPORTS_PREROTATE_BIT;
		tonemap_info.size_class = SizeClass::InputRelative;
		tonemap_info.size_relative_name = input;
		auto &tonemap = graph.add_pass("tonemap", RenderGraph::get_default_post_graphics_queue());
		tonemap.add_color_output(output, tonemap_info);
		auto &hdr_res = tonemap.add_texture_input(input);
		auto &bloom_res = tonemap.add_texture_input("upsample-0");
		const RenderBufferResource *ubo_res = nullptr;
		if (options.dynamic_exposure)
			ubo_res = &tonemap.add_uniform_input("average-luminance");
		tonemap.set_build_render_pass([&tonemap, &hdr_res, &bloom_res, iface, ubo = ubo_res, plan](HIP::CommandBuffer &cmd) {
			const StripPlan *strip = plan && (plan->active() || plan->exchange) ? plan : nullptr;
			record_tonemap(tonemap, cmd, hdr_res, bloom_res, ubo, iface, strip);
		});
	}
}

void setup_hdr_postprocess(RenderGraph &graph, const FrameParameters &frame, const std::string &input, const std::string &output,
                           const HDROptions &options, const HDRDynamicExposureInterface *iface)
{
	BufferInfo buffer_info;
	buffer_info.size = 3 * sizeof(float);
	buffer_info.usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT | VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT;

	if (options.dynamic_exposure)
	{
		// "average-luminance" has no writer inside a frame: last frame's value, read by the threshold pass and then
		// read-modify-written into its alias "average-luminance-updated" (hdr.cpp:411-431).
		graph.get_buffer_resource("average-luminance").set_buffer_info(buffer_info);
		auto &adapt_pass = graph.add_pass("adapt-luminance", RenderGraph::get_default_compute_queue());
		auto &output_res = adapt_pass.add_storage_output("average-luminance-updated", buffer_info, "average-luminance");
		auto &input_res = adapt_pass.add_texture_input("bloom-downsample-3");
		adapt_pass.set_build_render_pass([&graph, &frame, &output_res, &input_res](HIP::CommandBuffer &cmd) {
			record_luminance(cmd, frame, graph, output_res, input_res);
		});
	}

	{
		auto &threshold = graph.add_pass("bloom-threshold", RenderGraph::get_default_post_graphics_queue());
		auto info = bloom_level_info(input, 0.5f);
		info.aux_usage = 0;
		auto &out = threshold.add_color_output("threshold", info);
		auto &input_res = threshold.add_texture_input(input);
		const RenderBufferResource *ubo_res = nullptr;
		if (options.dynamic_exposure)
			ubo_res = &threshold.add_uniform_input("average-luminance");
		threshold.set_build_render_pass([&graph, &out, &input_res, ubo = ubo_res](HIP::CommandBuffer &cmd) {
			record_threshold(cmd, graph, out, input_res, ubo);
		});
	}

	struct Level
	{
		const char *pass_name;
		const char *source;
		float scale;
		bool upsample;
		bool feedback;
	};
	static const Level levels[] = {
		{"bloom-downsample-0", "threshold", 0.25f, false, false},
		{"bloom-downsample-1", "bloom-downsample-0", 0.125f, false, false},
		{"bloom-downsample-2", "bloom-downsample-1", 0.0625f, false, false},
		{"bloom-downsample-3", "bloom-downsample-2", 0.03125f, false, true},
		{"bloom-upsample-0", "bloom-downsample-3", 0.0625f, true, false},
		{"bloom-upsample-1", "bloom-upsample-0", 0.125f, true, false},
		{"bloom-upsample-2", "bloom-upsample-1", 0.25f, true, false},
	};
	for (auto &level : levels)
	{
		auto &pass = graph.add_pass(level.pass_name, RenderGraph::get_default_post_graphics_queue());
		auto info = bloom_level_info(input, level.scale);
		info.aux_usage = 0;
		auto &out = pass.add_color_output(level.pass_name, info);
		auto &in = pass.add_texture_input(level.source);
		RenderTextureResource *feedback = level.feedback ? &pass.add_history_input(level.pass_name) : nullptr;
		if (level.upsample)
			pass.set_build_render_pass([&graph, &out, &in](HIP::CommandBuffer &cmd) { record_upsample(cmd, graph, out, in); });
		else
			pass.set_build_render_pass([&graph, &frame, &out, &in, feedback](HIP::CommandBuffer &cmd) {
				record_downsample(cmd, frame, graph, out, in, feedback);
			});
		// Colour outputs without an input are LOAD_OP_CLEAR candidates in the reference; these quads overwrite every
		// pixel, so no clear is requested.
	}

	{
		AttachmentInfo tonemap_info;
		tonemap_info.size_class = SizeClass::InputRelative;
		tonemap_info.size_relative_name = input;
		auto &tonemap = graph.add_pass("tonemap", RenderGraph::get_default_post_graphics_queue());
		tonemap.add_color_output(output, tonemap_info);
		auto &hdr_res = tonemap.add_texture_input(input);
		auto &bloom_res = tonemap.add_texture_input("bloom-upsample-2");
		const RenderBufferResource *ubo_res = nullptr;
		if (options.dynamic_exposure)
			ubo_res = &tonemap.add_uniform_input("average-luminance-updated");
		tonemap.set_build_render_pass([&tonemap, &hdr_res, &bloom_res, iface, ubo = ubo_res](HIP::CommandBuffer &cmd) {
			record_tonemap(tonemap, cmd, hdr_res, bloom_res, ubo, iface);
		});
	}
}

// ---- HDR10 output (hdr.cpp:562-658) --------------------------------------------------------------------------------------------
namespace
{
struct Mat3
{
	float c[3][3]; // c[column][row]
};

void primary_to_xyz(const float xy[2], float out[3])
{
	out[0] = xy[0] / xy[1];
	out[1] = 1.0f;
	out[2] = (1.0f - xy[0] - xy[1]) / xy[1];
}

Mat3 inverse3(const Mat3 &m)
{
	// adjugate / determinant (muglm::inverse(mat3) has the same form)
	const float a = m.c[0][0], b = m.c[1][0], c = m.c[2][0];
	const float d = m.c[0][1], e = m.c[1][1], f = m.c[2][1];
	const float g = m.c[0][2], h = m.c[1][2], i = m.c[2][2];
	const float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
	const float inv = 1.0f / det;
	Mat3 r;
	r.c[0][0] = (e * i - f * h) * inv;
	r.c[1][0] = (c * h - b * i) * inv;
	r.c[2][0] = (b * f - c * e) * inv;
	r.c[0][1] = (f * g - d * i) * inv;
	r.c[1][1] = (a * i - c * g) * inv;
	r.c[2][1] = (c * d - a * f) * inv;
	r.c[0][2] = (d * h - e * g) * inv;
	r.c[1][2] = (b * g - a * h) * inv;
	r.c[2][2] = (a * e - b * d) * inv;
	return r;
}

Mat3 mul3(const Mat3 &a, const Mat3 &b)
{
	Mat3 r;
	for (int col = 0; col < 3; col++)
		for (int row = 0; row < 3; row++)
			r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2];
	return r;
}

Mat3 xyz_matrix(const float red[2], const float green[2], const float blue[2], const float white_point[2])
{
	Mat3 primaries;
	float white[3];
	primary_to_xyz(red, primaries.c[0]);
	primary_to_xyz(green, primaries.c[1]);
	primary_to_xyz(blue, primaries.c[2]);
	primary_to_xyz(white_point, white);
	const Mat3 inv = inverse3(primaries);
	float scale[3];
	for (int row = 0; row < 3; row++)
		scale[row] = inv.c[0][row] * white[0] + inv.c[1][row] * white[1] + inv.c[2][row] * white[2];
	Mat3 r;
	for (int col = 0; col < 3; col++)
		for (int row = 0; row < 3; row++)
			r.c[col][row] = primaries.c[col][row] * scale[col];
	return r;
}
} // namespace

void compute_xyz_matrix(const float red[2], const float green[2], const float blue[2], const float white_point[2], float out9[9])
{
	const Mat3 m = xyz_matrix(red, green, blue, white_point);
	memcpy(out9, m.c, sizeof(m.c));
}

void compute_rec709_to_st2020(const HdrMetadata &metadata, float out9[9])
{
	// sRGB in Vulkan uses BT.709 primaries with a D65 white point (hdr.cpp:582-589).
	const float r709[2] = {0.640f, 0.330f}, g709[2] = {0.3f, 0.6f}, b709[2] = {0.150f, 0.060f}, d65[2] = {0.3127f, 0.3290f};
	const Mat3 srgb_to_xyz = xyz_matrix(r709, g709, b709, d65);
	const Mat3 xyz_to_display = inverse3(xyz_matrix(metadata.display_primary_red, metadata.display_primary_green, metadata.display_primary_blue,
	                                                metadata.white_point));
	const Mat3 m = mul3(xyz_to_display, srgb_to_xyz);
	memcpy(out9, m.c, sizeof(m.c));
}

void setup_hdr10_pq_encoding(RenderGraph &graph, const std::string &output, const std::string &hdr_input, const std::string &ui_input,
                             const HDR10PQEncodingConfig &config, const HdrMetadata &static_metadata)
{
	struct PQEncoder : RenderPassInterface
	{
		RenderGraph *graph = nullptr;
		RenderTextureResource *hdr = nullptr, *ui = nullptr, *out = nullptr;
		gr_push_pq10 push = {};

		bool get_clear_color(unsigned, VkClearColorValue *) const override { return false; }

		void build_render_pass(HIP::CommandBuffer &cmd) override
		{
			auto &hdr_view = graph->get_physical_texture_resource(*hdr);
			auto &ui_view = graph->get_physical_texture_resource(*ui);
			auto &target = graph->get_physical_texture_resource(*out);
			cmd.check(gr_pq10_encode(cmd.get_context(), cmd.get_stream(), &hdr_view.get_view(), &ui_view.get_view(), &target.get_view(), &push),
			          "pq10 encode");
		}
	};

	auto &pq10 = graph.add_pass("pq10", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
	AttachmentInfo att; // swapchain size and format
	auto pass = std::make_shared<PQEncoder>();
	pass->graph = &graph;
	float conversion[9];
	compute_rec709_to_st2020(static_metadata, conversion);
	for (int col = 0; col < 3; col++) // mat4(mat3)
		for (int row = 0; row < 3; row++)
			pass->push.primary_conversion[4 * col + row] = conversion[3 * col + row];
	pass->push.primary_conversion[15] = 1.0f;
	pass->push.hdr_pre_exposure = config.hdr_pre_exposure;
	pass->push.ui_pre_exposure = config.ui_pre_exposure;
	pass->push.max_light_level = static_metadata.max_content_light_level;
	pass->push.inv_max_light_level = 1.0f / static_metadata.max_content_light_level; // pre-Reinhard scaling, hdr.cpp:643-644
	pass->out = &pq10.add_color_output(output, att);
	pass->hdr = &pq10.add_texture_input(hdr_input);
	pass->ui = &pq10.add_texture_input(ui_input);
	pq10.set_render_pass_interface(std::move(pass));
}
} // namespace Granite
