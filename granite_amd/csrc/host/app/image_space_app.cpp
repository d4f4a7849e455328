// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
#include "image_space_app.hpp"
#include "../timeline_trace.hpp"
#include <hip/hip_runtime_api.h>
#include "../gtx.hpp"
#include "../post/spd.hpp"
#include "../post/ssr.hpp"
#include <chrono>
#include <cmath>
#include <cstring>

namespace Granite
{
static std::string tagcat(const std::string &a, const std::string &b)
{
	return a + "-" + b;
}

static PostAAType to_post_aa_type(int32_t v)
{
	switch (v)
	{
	case GRA_POST_AA_FXAA: return PostAAType::FXAA;
	case GRA_POST_AA_SMAA_LOW: return PostAAType::SMAA_Low;
	case GRA_POST_AA_SMAA_MEDIUM: return PostAAType::SMAA_Medium;
	case GRA_POST_AA_SMAA_HIGH: return PostAAType::SMAA_High;
	case GRA_POST_AA_SMAA_ULTRA: return PostAAType::SMAA_Ultra;
	case GRA_POST_AA_TAA_LOW: return PostAAType::TAA_Low;
	case GRA_POST_AA_TAA_MEDIUM: return PostAAType::TAA_Medium;
	case GRA_POST_AA_TAA_HIGH: return PostAAType::TAA_High;
	default: return PostAAType::None;
	}
}

ImageSpaceApplication::ImageSpaceApplication(const gra_config &config_) : config(config_)
{
	if (config.device >= 0)
		device_holder = std::make_unique<HIP::Device>(config.device);
	if (!config.width || !config.height)
		throw std::logic_error("Backbuffer dimensions must be non-zero.");
	if (config.hdr10 && (!config.enable_lighting || config.hdr_bloom || config.post_aa != GRA_POST_AA_NONE || config.pre_aa != GRA_POST_AA_NONE ||
	                     config.strip_count > 1 || (config.resolution_scale > 0.0f && config.resolution_scale < 1.0f)))
		throw std::logic_error("hdr10 needs enable_lighting and excludes hdr_bloom, anti-aliasing, resolution scaling and row bands.");
	if (config.hdr_packed_float && (config.ssr || config.hdr10 || config.aa_bench))
		throw std::logic_error("hdr_packed_float (renderTargetFp16 = false) is not available with ssr / hdr10 / aa_bench: those passes take the lit target as RGBA16F.");
	if (config.aa_bench && (config.enable_lighting || config.hdr_bloom || config.hdr10 || config.ssr || config.strip_count > 1 || config.depth_hierarchy))
		throw std::logic_error("aa_bench is a graph of its own: no lighting, bloom, hdr10, SSR, depth hierarchy or row bands.");
	if (config.resolution_scale < 0.0f || config.resolution_scale > 1.0f)
		throw std::logic_error("resolution_scale must be in (0, 1].");
	render_width = config.width;
	render_height = config.height;
	if (scaled())
	{
		// ceil(size * scale), as the graph resolves SwapchainRelative sizes (render_graph.cpp:3158-3170)
		render_width = std::max(unsigned(std::ceil(float(config.width) * config.resolution_scale)), 1u);
		render_height = std::max(unsigned(std::ceil(float(config.height) * config.resolution_scale)), 1u);
		if (config.strip_count > 1)
			throw std::logic_error("Row-band tiling and resolution scaling cannot be combined.");
	}
	if (!config.cluster_res[0])
	{
		// SceneViewerApplication: cluster->set_resolution(128, 64, 4096) (scene_viewer_application.cpp:407)
		config.cluster_res[0] = 128;
		config.cluster_res[1] = 64;
		config.cluster_res[2] = 4096;
	}
	cluster.set_resolution(config.cluster_res[0], config.cluster_res[1], config.cluster_res[2]);
	cluster.set_base_render_context(&context);
	cluster.set_scene_lights(&light_list);

	lighting.directional.color = vec3(config.directional_color[0], config.directional_color[1], config.directional_color[2]);
	lighting.directional.direction = vec3(config.directional_direction[0], config.directional_direction[1], config.directional_direction[2]);
	lighting.cluster = &cluster;
	context.set_lighting_parameters(&lighting);
	hdr_options.dynamic_exposure = config.dynamic_exposure != 0;

	if (config.strip_count == 0)
		config.strip_count = 1;
	if (config.strip_count > 1)
	{
		if (config.strip_index >= config.strip_count)
			throw std::logic_error("strip_index must be below strip_count.");
		if (!config.enable_lighting || !config.hdr_bloom || !config.compute_post || config.rmw_emissive || config.ssr)
			throw std::logic_error("Row-band tiling needs the deferred compute-post graph without SSR and without the RMW emissive declaration.");
	}
	// What the anti-aliasing passes reach for around a band: FXAA / SMAA after the tonemap, TAA before it.
	StripAA strip_aa;
	const PostAAType post_type = to_post_aa_type(config.post_aa), pre_type = to_post_aa_type(config.pre_aa);
	if (post_type == PostAAType::FXAA)
		strip_aa.post = StripAA::Post::FXAA;
	else if (smaa_search_steps(post_type))
	{
		strip_aa.post = StripAA::Post::SMAA;
		strip_aa.smaa_search_steps = smaa_search_steps(post_type);
	}
	strip_aa.temporal = pre_type == PostAAType::TAA_Low || pre_type == PostAAType::TAA_Medium || pre_type == PostAAType::TAA_High;
	strip_aa.taa_history_reach = strip_aa.temporal ? config.taa_history_reach_rows : 0u;
	strip_plan = StripPlan::build(config.strip_index, config.strip_count, config.width, config.height, strip_aa);
	hdr_options.strip = &strip_plan;

	// Default camera of the survey's synthetic scene; gra_set_camera / gra_set_render_parameters override it.
	set_base_camera(perspective(1.0471975512f, float(config.width) / float(config.height), 0.1f, 100.0f),
	                look_at(vec3(0.0f, 2.0f, 8.0f), vec3(0.0f, 1.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f)));

	if (config.aa_bench)
		set_base_camera(mat4(1.0f), mat4(1.0f)); // AABenchApplication::render_frame: jitter.step(mat4(1.0f), mat4(1.0f))

	graph.enable_timestamps(config.enable_timestamps != 0 || config.aa_bench != 0); // aa_bench.cpp:155
	if (!device_holder)
		return;
	auto &device = *device_holder;
	device.set_image_row_granularity(config.strip_count);
	if (strip_plan.taa_exchange_rows)
	{
		void *word = nullptr;
		if (hipHostMalloc(&word, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
			throw std::runtime_error("hipHostMalloc failed");
		taa_reach_flag = static_cast<uint32_t *>(word);
		*taa_reach_flag = 0;
		strip_plan.taa_reach_flag = taa_reach_flag;
	}

	// External swapchain: 4 images R8G8B8A8_SRGB, cycled per frame.
	for (unsigned i = 0; i < 4; i++)
		swapchain.push_back(device.create_image(config.width, config.height, backbuffer_format(), "swapchain-" + std::to_string(i)));

	src_emissive = device.create_image(render_width, render_height, hdr_target_format(), "src-emissive");
	if (config.enable_lighting)
	{
		src_albedo = device.create_image(render_width, render_height, VK_FORMAT_R8G8B8A8_SRGB, "src-albedo");
		src_normal = device.create_image(render_width, render_height, VK_FORMAT_A2B10G10R10_UNORM_PACK32, "src-normal");
		src_pbr = device.create_image(render_width, render_height, VK_FORMAT_R8G8_UNORM, "src-pbr");
		src_depth = device.create_image(render_width, render_height, VK_FORMAT_D32_SFLOAT, "src-depth");
	}
}

ImageSpaceApplication::~ImageSpaceApplication()
{
	cluster.invalidate_prefetch(); // no helper-thread job may outlive the light objects it reads
	if (device_holder)
		device_holder->wait_idle();
	for (auto &e : output_gather_done)
		(void)hipEventDestroy(static_cast<hipEvent_t>(e.second));
	if (output_ready_event)
		(void)hipEventDestroy(static_cast<hipEvent_t>(output_ready_event));
	if (taa_reach_flag)
		(void)hipHostFree(taa_reach_flag);
}

void ImageSpaceApplication::check_taa_history_reach()
{
	if (taa_reach_flag && *static_cast<volatile uint32_t *>(taa_reach_flag))
		throw std::runtime_error("taa-resolve: a pixel's reprojection fetched history rows this rank does not hold (reach of more than " +
		                         std::to_string(config.taa_history_reach_rows) +
		                         " rows): frames since then are not the single-device frames.  Raise gra_config.taa_history_reach_rows, or set it to 0 "
		                         "(whole history bands are all-gathered).");
}

// ---- the frame's output bands on the wire --------------------------------------------------------------------------------------
// Every rank ends up with every band of the finished frame (collective B).  Over point-to-point xGMI each rank's band
// crosses one link per peer, and that -- 33 MB per 4K band and frame -- bounds the multi-GPU frame rate, not the kernels.  The
// alpha byte of a tonemapped (or FXAA'd) frame is 255 everywhere, so the bands travel as RGB888: pack the own band, all-gather
// the 24-bit buffer, unpack the other ranks' rows into the output image.  Three quarters of the bytes per link.
bool ImageSpaceApplication::output_packs(const HIP::Image &image, const char *tag) const
{
	if (config.output_gather_rgba || !tag)
		return false;
	const bool output_tag = strcmp(tag, "tonemapped") == 0 || strcmp(tag, "fxaa") == 0; // alpha is written as 1.0 by both passes
	const VkFormat format = image.get_format();
	return output_tag && (format == VK_FORMAT_R8G8B8A8_SRGB || format == VK_FORMAT_R8G8B8A8_UNORM) && (image.get_view().pitch_bytes & 15u) == 0;
}

void ImageSpaceApplication::pack_output_band(HIP::CommandBuffer &cmd, HIP::Image &image, uint32_t chunk_rows)
{
	auto &buffer = packed_output[image.get_device_pointer()];
	const size_t bytes = size_t(strip_plan.count) * chunk_rows * image.get_width() * 3u;
	if (!buffer || buffer->get_size() < bytes)
		buffer = get_device().create_buffer(bytes, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT, "packed-output");
	gr_rows own = {strip_plan.index * chunk_rows, chunk_rows};
	if (own.first < image.get_height())
		cmd.check(gr_pack_rgb8_rows(cmd.get_context(), cmd.get_stream(), &image.get_view(), &own, buffer->get_device_pointer()), "pack_rgb8");
}

void ImageSpaceApplication::gather_packed_output(HIP::Image &image, uint32_t chunk_rows, void *stream, const BandTransport &transport)
{
	auto &buffer = packed_output[image.get_device_pointer()];
	transport(buffer->get_device_pointer(), size_t(chunk_rows) * image.get_width() * 3u, stream);
	auto *ctx = get_device().get_context();
	const uint32_t first = std::min(strip_plan.index * chunk_rows, image.get_height());
	const uint32_t end = std::min(first + chunk_rows, image.get_height());
	const gr_rows above = {0, first}, below = {end, image.get_height() - end};
	if (above.count && gr_unpack_rgb8_rows(ctx, stream, buffer->get_device_pointer(), &image.get_view(), &above) < 0)
		throw std::runtime_error(gr_last_error(ctx));
	if (below.count && gr_unpack_rgb8_rows(ctx, stream, buffer->get_device_pointer(), &image.get_view(), &below) < 0)
		throw std::runtime_error(gr_last_error(ctx));
}

void ImageSpaceApplication::set_exchange_callback(gra_exchange_fn fn, void *user)
{
	if (!fn)
	{
		strip_plan.exchange = nullptr;
		return;
	}
	const uint32_t ranks = strip_plan.count;
	strip_plan.exchange = [this, fn, user, ranks](HIP::CommandBuffer &cmd, HIP::Image &image, uint32_t chunk_rows, const char *tag) {
		if (output_packs(image, tag))
		{
			pack_output_band(cmd, image, chunk_rows);
			gather_packed_output(image, chunk_rows, cmd.get_stream(), [&](void *base, size_t chunk_bytes, void *stream) {
				fn(user, tag, base, chunk_bytes, ranks, stream);
			});
		}
		else
			fn(user, tag, image.get_device_pointer(), uint64_t(chunk_rows) * image.get_view().pitch_bytes, ranks, cmd.get_stream());
	};
}

void ImageSpaceApplication::init_collective(const uint8_t *id128, int rank, int ranks)
{
	if (unsigned(ranks) != strip_plan.count || unsigned(rank) != strip_plan.index)
		throw std::logic_error("Collective rank / size must match the strip plan of this instance.");
	get_device().make_current(); // the communicator binds to the calling thread's current device
	collective.init(id128, rank, ranks);
	{
		// Every rank must pack (or not pack) the finished bands the same way: a first, tiny all-gather of the setting itself.
		auto &device = get_device();
		auto words = device.create_buffer(size_t(ranks) * 4u, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT, "collective-settings");
		const uint32_t mine = config.output_gather_rgba ? 1u : 0u;
		auto *ctx = device.get_context();
		auto *base = static_cast<uint8_t *>(words->get_device_pointer());
		std::vector<uint32_t> all(size_t(ranks), 0u);
		if (gr_upload(ctx, nullptr, base + size_t(rank) * 4u, &mine, 4) < 0 || gr_sync(ctx, nullptr) < 0)
			throw std::runtime_error(gr_last_error(ctx));
		collective.all_gather_in_place(base, 4, nullptr);
		if (gr_sync(ctx, nullptr) < 0 || gr_download(ctx, nullptr, all.data(), base, size_t(ranks) * 4u) < 0)
			throw std::runtime_error(gr_last_error(ctx));
		for (uint32_t v : all)
			if (v != mine)
				throw std::logic_error("Row bands: the ranks disagree on output_gather_rgba (every rank must transport the finished bands the same way).");
	}
	strip_plan.exchange = [this](HIP::CommandBuffer &cmd, HIP::Image &image, uint32_t chunk_rows, const char *tag) {
		// With the output gather on its own communicator and stream (init_output_collective), the kernels of the two communicators
		// must run in ONE order on every rank, or ranks can wait on each other across communicators (RCCL / NCCL: concurrent
		// collectives of different communicators on one device are only safe in a consistent order).  The order is made explicit:
		// an in-frame collective of frame N + 1 runs behind the output gather of frame N (this wait), the output gather of frame
		// N + 1 behind the band's last pass and with it behind every in-frame collective of that frame (its own event hand-over).
		if (last_output_gather_event && hipEventQuery(static_cast<hipEvent_t>(last_output_gather_event)) != hipSuccess)
			if (hipStreamWaitEvent(static_cast<hipStream_t>(cmd.get_stream()), static_cast<hipEvent_t>(last_output_gather_event), 0) != hipSuccess)
				throw std::runtime_error("hipStreamWaitEvent failed");
		// device time of the collective on its stream, when somebody asked (gr_timing_*: "inframe_gather")
		void *span = nullptr;
		(void)gr_timing_span_begin(cmd.get_context(), cmd.get_stream(), "inframe_gather", &span);
		if (output_packs(image, tag))
		{
			pack_output_band(cmd, image, chunk_rows);
			gather_packed_output(image, chunk_rows, cmd.get_stream(),
			                     [this](void *base, size_t chunk_bytes, void *stream) { collective.all_gather_in_place(base, chunk_bytes, stream); });
		}
		else
			collective.all_gather_in_place(image.get_device_pointer(), size_t(chunk_rows) * image.get_view().pitch_bytes, cmd.get_stream());
		(void)gr_timing_span_end(cmd.get_context(), cmd.get_stream(), span);
	};
}

void ImageSpaceApplication::init_output_collective(const uint8_t *id128, int rank, int ranks)
{
	if (unsigned(ranks) != strip_plan.count || unsigned(rank) != strip_plan.index)
		throw std::logic_error("Collective rank / size must match the strip plan of this instance.");
	if (!collective.is_initialized())
		throw std::logic_error("init_output_collective: initialise the in-frame communicator (gra_comm_init) first.");
	auto &device = get_device();
	device.make_current();
	output_collective.init(id128, rank, ranks);
	hipEvent_t ready;
	if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess)
		throw std::runtime_error("hipEventCreate failed");
	output_ready_event = ready;

	// The pass that writes an output image first waits for the gather that last touched it (four swapchain images rotate, so
	// that gather is normally long finished) ...
	strip_plan.acquire_output = [this](HIP::CommandBuffer &cmd, HIP::Image &image) {
		auto itr = output_gather_done.find(image.get_device_pointer());
		output_acquires++;
		if (itr != output_gather_done.end() && hipEventQuery(static_cast<hipEvent_t>(itr->second)) != hipSuccess)
		{
			output_acquire_waits++; // the gather of this image was still in flight when its next writer was enqueued: not hidden
			if (hipStreamWaitEvent(static_cast<hipStream_t>(cmd.get_stream()), static_cast<hipEvent_t>(itr->second), 0) != hipSuccess)
				throw std::runtime_error("hipStreamWaitEvent failed");
		}
	};
	// ... and the gather itself runs on the collective stream behind the band's last pass, beside whatever the executor's
	// streams do next (the following frames' cluster build, lighting and bloom chain).
	strip_plan.exchange_output = [this](HIP::CommandBuffer &cmd, HIP::Image &image, uint32_t chunk_rows, const char *tag) {
		auto gather_stream = static_cast<hipStream_t>(get_device().get_collective_stream());
		auto ready_event = static_cast<hipEvent_t>(output_ready_event);
		const bool packs = output_packs(image, tag);
		if (packs)
			pack_output_band(cmd, image, chunk_rows); // behind the band's last pass, on its stream
		if (hipEventRecord(ready_event, static_cast<hipStream_t>(cmd.get_stream())) != hipSuccess ||
		    hipStreamWaitEvent(gather_stream, ready_event, 0) != hipSuccess)
			throw std::runtime_error("output gather: event hand-over failed");
		const BandTransport rccl = [this](void *base, size_t chunk_bytes, void *stream) { output_collective.all_gather_in_place(base, chunk_bytes, stream); };
		void *span = nullptr;
		(void)gr_timing_span_begin(cmd.get_context(), gather_stream, "output_gather", &span);
		if (packs)
			gather_packed_output(image, chunk_rows, gather_stream, rccl);
		else
			rccl(image.get_device_pointer(), size_t(chunk_rows) * image.get_view().pitch_bytes, gather_stream);
		(void)gr_timing_span_end(cmd.get_context(), gather_stream, span);
		void *&done = output_gather_done[image.get_device_pointer()];
		if (!done)
		{
			hipEvent_t e;
			if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
				throw std::runtime_error("hipEventCreate failed");
			done = e;
		}
		if (hipEventRecord(static_cast<hipEvent_t>(done), gather_stream) != hipSuccess)
			throw std::runtime_error("hipEventRecord failed");
		last_output_gather_event = done; // what the next in-frame collective orders itself behind
	};
}

void ImageSpaceApplication::set_base_camera(const mat4 &projection, const mat4 &view)
{
	base_projection = projection;
	base_view = view;
	has_base_camera = true;
	context.set_camera(projection, view);
}

void ImageSpaceApplication::set_camera_motion(const vec3 &translation_per_frame)
{
	if (!has_base_camera)
		throw std::logic_error("Camera motion needs gra_set_camera (projection + view), not verbatim render parameters.");
	camera_motion = translation_per_frame;
	camera_moves = translation_per_frame.x != 0.0f || translation_per_frame.y != 0.0f || translation_per_frame.z != 0.0f;
	cluster.invalidate_prefetch();
}

mat4 ImageSpaceApplication::get_taa_reprojection() const
{
	return translate(vec3(0.5f, 0.5f, 0.0f)) * scale(vec3(0.5f, 0.5f, 1.0f)) * jitter.get_history_view_proj(1) * jitter.get_history_inv_view_proj(0);
}


void ImageSpaceApplication::set_lights(const gra_light_desc *descs, uint32_t count)
{
	cluster.invalidate_prefetch(); // the helper thread may be packing the current list for the next frame
	light_objects.clear();
	light_transforms.resize(count);
	light_list.clear();
	for (uint32_t i = 0; i < count; i++)
	{
		auto &d = descs[i];
		mat_affine &t = light_transforms[i];
		for (int r = 0; r < 3; r++)
			t[r] = vec4(d.transform[4 * r + 0], d.transform[4 * r + 1], d.transform[4 * r + 2], d.transform[4 * r + 3]);
		std::unique_ptr<PositionalLight> light;
		if (d.type == 0)
		{
			auto spot = std::make_unique<SpotLight>();
			spot->set_spot_parameters(d.inner_cone, d.outer_cone);
			light = std::move(spot);
		}
		else
			light = std::make_unique<PointLight>();
		light->set_color(vec3(d.color[0], d.color[1], d.color[2]));
		light->set_maximum_range(d.cutoff_range);
		light_list.push_back({light.get(), &light_transforms[i]});
		light_objects.push_back(std::move(light));
	}
}

void ImageSpaceApplication::upload_gbuffer(const void *emissive, const void *albedo, const void *normal, const void *pbr, const void *depth,
                                           const void *mv)
{
	auto &device = get_device();
	device.wait_idle();
	auto upload = [&](HIP::ImageHandle &img, const void *src, const char *what) {
		if (!src)
			return;
		if (!img)
			throw std::logic_error(std::string("This graph has no ") + what + " attachment.");
		if (gr_upload(device.get_context(), nullptr, img->get_device_pointer(), src, img->get_size_bytes()) < 0 ||
		    gr_sync(device.get_context(), nullptr) < 0)
			throw std::runtime_error(gr_last_error(device.get_context()));
	};
	upload(src_emissive, emissive, "emissive");
	upload(src_albedo, albedo, "albedo");
	upload(src_normal, normal, "normal");
	upload(src_pbr, pbr, "pbr");
	upload(src_depth, depth, "depth");
	if (mv && !src_mv)
		src_mv = device.create_image(render_width, render_height, VK_FORMAT_R16G16_SFLOAT, "src-mv");
	upload(src_mv, mv, "motion-vector");
	gbuffer_dirty = true;
	filled_targets.clear();
}

void ImageSpaceApplication::upload_ambient_occlusion(const void *ao_r8)
{
	if (!config.ambient_occlusion || !config.enable_lighting)
		throw std::logic_error("This graph has no ambient-occlusion input (config.ambient_occlusion).");
	if (!ao_r8)
		throw std::logic_error("upload_ambient_occlusion: null image");
	auto &device = get_device();
	device.wait_idle();
	if (!src_ao)
		src_ao = device.create_image(render_width, render_height, VK_FORMAT_R8_UNORM, "src-ssao");
	if (gr_upload(device.get_context(), nullptr, src_ao->get_device_pointer(), ao_r8, src_ao->get_size_bytes()) < 0 ||
	    gr_sync(device.get_context(), nullptr) < 0)
		throw std::runtime_error(gr_last_error(device.get_context()));
	filled_targets.clear();
}

void ImageSpaceApplication::upload_aa_bench_images(const void *first, const void *second, uint32_t width, uint32_t height)
{
	if (!config.aa_bench)
		throw std::logic_error("This graph takes no benchmark images (config.aa_bench).");
	if (!first || !second || !width || !height)
		throw std::logic_error("upload_aa_bench_images: two images and their size are needed.");
	auto &device = get_device();
	device.wait_idle();
	const void *src[2] = {first, second};
	for (int i = 0; i < 2; i++)
	{
		bench_images[i] = device.create_image(width, height, VK_FORMAT_R8G8B8A8_SRGB, "aa-bench-input-" + std::to_string(i));
		if (gr_upload(device.get_context(), nullptr, bench_images[i]->get_device_pointer(), src[i], size_t(width) * height * 4u) < 0 ||
		    gr_sync(device.get_context(), nullptr) < 0)
			throw std::runtime_error(gr_last_error(device.get_context()));
	}
}

void ImageSpaceApplication::upload_gbuffer_gtx(const char *const paths[6])
{
	struct Slot
	{
		const char *what;
		VkFormat formats[2];
	};
	static const Slot slots[6] = {
		{"emissive", {hdr_target_format(), hdr_target_format()}},
		{"albedo", {VK_FORMAT_R8G8B8A8_SRGB, VK_FORMAT_R8G8B8A8_UNORM}},
		{"normal", {VK_FORMAT_A2B10G10R10_UNORM_PACK32, VK_FORMAT_A2B10G10R10_UNORM_PACK32}},
		{"pbr", {VK_FORMAT_R8G8_UNORM, VK_FORMAT_R8G8_UNORM}},
		{"depth", {VK_FORMAT_D32_SFLOAT, VK_FORMAT_R32_SFLOAT}},
		{"motion-vector", {VK_FORMAT_R16G16_SFLOAT, VK_FORMAT_R16G16_SFLOAT}},
	};
	GtxImage images[6];
	const void *level0[6] = {};
	for (int i = 0; i < 6; i++)
	{
		if (!paths[i])
			continue;
		images[i] = gtx_load(paths[i]);
		auto &img = images[i];
		if (img.type != 1 || img.layers != 1 || img.depth != 1)
			throw std::runtime_error(std::string(paths[i]) + ": a 2-D single-layer image is expected for " + slots[i].what + ".");
		if (img.format != slots[i].formats[0] && img.format != slots[i].formats[1])
			throw std::runtime_error(std::string(paths[i]) + ": wrong format for the " + slots[i].what + " attachment.");
		if (img.width != render_width || img.height != render_height)
			throw std::runtime_error(std::string(paths[i]) + ": " + std::to_string(img.width) + " x " + std::to_string(img.height) +
			                         " does not match the configured frame.");
		level0[i] = img.payload.data() + img.level_offset(0);
	}
	upload_gbuffer(level0[0], level0[1], level0[2], level0[3], level0[4], level0[5]);
}

void ImageSpaceApplication::save_image_gtx(HIP::Image &image, const std::string &path)
{
	auto &device = get_device();
	device.wait_idle();
	GtxImage out;
	out.format = image.get_format();
	out.width = image.get_width();
	out.height = image.get_height();
	out.levels = image.get_levels();
	out.payload.assign(out.required_payload_size(), 0);
	// The executor packs mip levels back to back; GTX starts each one on a 16-byte boundary.
	for (unsigned level = 0; level < out.levels; level++)
	{
		const gr_image view = image.get_level_view(level);
		if (gr_download(device.get_context(), nullptr, out.payload.data() + out.level_offset(level), view.ptr, out.level_size(level)) < 0)
			throw std::runtime_error(gr_last_error(device.get_context()));
	}
	gtx_save(out, path);
}

// Config-1 style graph head: an "HDR-main" colour target filled from the uploaded HDR image (tools/aa_bench.cpp:80-101
// does the same with a blit of a PNG).
void ImageSpaceApplication::add_hdr_input_pass(const std::string &tag)
{
	AttachmentInfo hdr;
	hdr.format = hdr_target_format();
	hdr.flags |= ATTACHMENT_INFO_INTERNAL_RETAINED_BIT; // filled once (needs_fill), never aliased
	if (scaled())
		hdr.size_x = hdr.size_y = config.resolution_scale;
	auto &pass = graph.add_pass(tagcat("hdr-input", tag), RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
	auto &out = pass.add_color_output(tagcat("HDR", tag), hdr);
	// A conditional pass (RenderPassInterface::need_render_pass, render_graph.cpp:2218): once every copy of the target holds the image the pass
	// has nothing to record, and a pass that is not recorded costs its stream neither an event record nor the queries that follow one -- three
	// of the seven runtime calls of a post-only frame (BASELINE config 1).
	struct FillOnce : RenderPassInterface
	{
		ImageSpaceApplication *app;
		const RenderTextureResource *out;
		bool render_pass_is_conditional() const override { return true; }
		bool need_render_pass() const override { return !app->is_filled(app->graph.get_physical_texture_resource(*out)); }
		void build_render_pass(HIP::CommandBuffer &cmd) override
		{
			auto &target = app->graph.get_physical_texture_resource(*out);
			if (app->needs_fill(target))
				cmd.copy_image(target, *app->src_emissive);
		}
	};
	auto fill = std::make_shared<FillOnce>();
	fill->app = this;
	fill->out = &out;
	pass.set_render_pass_interface(std::move(fill));
}

// tools/aa_bench.cpp:76-117: the "main" pass of the AA benchmark -- HDR-main (B10G11R11 there, RGBA16F here, as the TAA output)
// and depth-main at `scale`, colour = one of the two input images through blit.frag with LinearClamp, depth cleared to 0.
void ImageSpaceApplication::add_aa_bench_main_pass(const std::string &tag)
{
	AttachmentInfo main_output, main_depth;
	main_output.format = VK_FORMAT_R16G16B16A16_SFLOAT;
	main_depth.format = VK_FORMAT_D32_SFLOAT;
	if (scaled())
	{
		main_output.size_x = main_output.size_y = config.resolution_scale;
		main_depth.size_x = main_depth.size_y = config.resolution_scale;
	}
	auto &pass = graph.add_pass(tag, RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
	auto &color = pass.add_color_output(tagcat("HDR", tag), main_output);
	auto &depth = pass.set_depth_stencil_output(tagcat("depth", tag), main_depth);
	pass.set_build_render_pass([this, &color, &depth](HIP::CommandBuffer &cmd) {
		auto &target = graph.get_physical_texture_resource(color);
		cmd.clear_image(graph.get_physical_texture_resource(depth)); // set_get_clear_depth_stencil: depth = 0.0
		HIP::Image *img = bench_images[(bench_input_index++) & 1].get();
		if (img)
			cmd.check(gr_blit(cmd.get_context(), cmd.get_stream(), &img->get_view(), &target.get_view(), 1), "blit");
		else
			cmd.clear_image(target); // set_get_clear_color: zero
	});
}

// Motion vectors come from the scene renderer in Granite (add_mv_pass, scene_viewer_application.cpp:1010-1060); here the
// uploaded synthetic RG16F image (zeros when none was uploaded) is copied into "mv-<tag>".
void ImageSpaceApplication::add_mv_pass(const std::string &tag)
{
	AttachmentInfo mv;
	mv.format = VK_FORMAT_R16G16_SFLOAT;
	mv.flags |= ATTACHMENT_INFO_INTERNAL_RETAINED_BIT; // filled once (needs_fill), never aliased
	if (scaled())
		mv.size_x = mv.size_y = config.resolution_scale;
	auto &pass = graph.add_pass(tagcat("mv", tag), RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
	auto &out = pass.add_color_output(tagcat("mv", tag), mv);
	pass.set_build_render_pass([this, &out](HIP::CommandBuffer &cmd) {
		auto &target = graph.get_physical_texture_resource(out);
		if (!needs_fill(target))
			return;
		if (src_mv)
			cmd.copy_image(target, *src_mv);
		else
			cmd.clear_image(target);
	});
}

void ImageSpaceApplication::add_main_pass_deferred(const std::string &tag)
{
	AttachmentInfo emissive, albedo, normal, pbr, depth;
	emissive.format = hdr_target_format(); // renderTargetFp16 ? RGBA16F : B10G11R11_UFLOAT_PACK32 (scene_viewer_application.cpp:881-883)
	albedo.format = VK_FORMAT_R8G8B8A8_SRGB;
	normal.format = VK_FORMAT_A2B10G10R10_UNORM_PACK32;
	pbr.format = VK_FORMAT_R8G8_UNORM;
	depth.format = VK_FORMAT_D32_SFLOAT;
	if (scaled()) // scene_viewer_application.cpp:758-761,888-889
		for (auto *info : {&emissive, &albedo, &normal, &pbr, &depth})
			info->size_x = info->size_y = config.resolution_scale;
	// Filled once per target and kept (needs_fill): the memory must stay theirs whatever streams the passes land on.
	// Emissive under the RMW declaration is rewritten every frame and stays an ordinary attachment.
	for (auto *info : {&albedo, &normal, &pbr, &depth})
		info->flags |= ATTACHMENT_INFO_INTERNAL_RETAINED_BIT;
	if (!config.rmw_emissive)
		emissive.flags |= ATTACHMENT_INFO_INTERNAL_RETAINED_BIT;

	// The G-buffer producer: Granite rasterises the scene here; the harness copies the synthetic attachments in.
	// Attachments persist across frames, so only what a later pass clobbers (emissive under the RMW declaration) is
	// restored every frame.
	auto &gbuffer = graph.add_pass(tagcat("gbuffer", tag), RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
	auto &g_emissive = gbuffer.add_color_output(tagcat("emissive", tag), emissive);
	auto &g_albedo = gbuffer.add_color_output(tagcat("albedo", tag), albedo);
	auto &g_normal = gbuffer.add_color_output(tagcat("normal", tag), normal);
	auto &g_pbr = gbuffer.add_color_output(tagcat("pbr", tag), pbr);
	auto &g_depth = gbuffer.set_depth_stencil_output(tagcat("depth-transient", tag), depth);
	gbuffer.set_build_render_pass([this, &g_emissive, &g_albedo, &g_normal, &g_pbr, &g_depth](HIP::CommandBuffer &cmd) {
		auto fill = [&](RenderTextureResource &res, HIP::ImageHandle &src, bool always) {
			auto &target = graph.get_physical_texture_resource(res);
			if (needs_fill(target) || always)
				cmd.copy_image(target, *src);
		};
		fill(g_emissive, src_emissive, config.rmw_emissive != 0);
		fill(g_albedo, src_albedo, false);
		fill(g_normal, src_normal, false);
		fill(g_pbr, src_pbr, false);
		fill(g_depth, src_depth, false);
	});

	ssao_output = nullptr;
	if (config.ambient_occlusion)
	{
		// setup_ffx_cacao(graph, context, "ssao-output-<tag>", depth, normal) (scene_viewer_application.cpp:950-954; output
		// declared at ssao.cpp:48-58: R8_UNORM storage image the size of the depth input).  CACAO itself exists only as
		// SPIR-V blobs in the reference; the pass here copies the uploaded image in (white = unoccluded until one arrives).
		auto &ssao = graph.add_pass(tagcat("ssao", tag), RENDER_GRAPH_QUEUE_COMPUTE_BIT);
		AttachmentInfo ao;
		ao.format = VK_FORMAT_R8_UNORM;
		ao.flags |= ATTACHMENT_INFO_INTERNAL_RETAINED_BIT; // filled once (needs_fill), never aliased
		ao.size_class = SizeClass::InputRelative;
		ao.size_relative_name = tagcat("depth-transient", tag);
		auto &ao_out = ssao.add_storage_texture_output(tagcat("ssao-output", tag), ao);
		ssao.add_texture_input(tagcat("depth-transient", tag));
		ssao.add_texture_input(tagcat("normal", tag));
		ssao.set_build_render_pass([this, &ao_out](HIP::CommandBuffer &cmd) {
			auto &target = graph.get_physical_texture_resource(ao_out);
			if (!needs_fill(target))
				return;
			if (src_ao)
				cmd.copy_image(target, *src_ao);
			else
				cmd.check(gr_fill_byte(cmd.get_context(), cmd.get_stream(), target.get_device_pointer(), 0xff, target.get_size_bytes()), "ssao fill");
		});
	}

	auto &lighting_pass = graph.add_pass(tagcat("lighting", tag), RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
	RenderTextureResource *hdr_out;
	RenderTextureResource *emissive_in = nullptr;
	if (config.rmw_emissive)
		hdr_out = &lighting_pass.add_color_output(tagcat("HDR", tag), emissive, tagcat("emissive", tag)); // reference form
	else
	{
		AttachmentInfo hdr_info = emissive;
		hdr_info.flags &= ~ATTACHMENT_INFO_INTERNAL_RETAINED_BIT; // the lit target is rewritten every frame: an ordinary attachment
		hdr_out = &lighting_pass.add_color_output(tagcat("HDR", tag), hdr_info);
		emissive_in = &lighting_pass.add_attachment_input(tagcat("emissive", tag));
	}
	auto &in_albedo = lighting_pass.add_attachment_input(tagcat("albedo", tag));
	auto &in_normal = lighting_pass.add_attachment_input(tagcat("normal", tag));
	auto &in_pbr = lighting_pass.add_attachment_input(tagcat("pbr", tag));
	auto &in_depth = lighting_pass.add_attachment_input(tagcat("depth-transient", tag));
	lighting_pass.set_depth_stencil_input(tagcat("depth-transient", tag));
	lighting_pass.add_fake_resource_write_alias(tagcat("depth-transient", tag), tagcat("depth", tag));

	if (config.ambient_occlusion)
		ssao_output = &lighting_pass.add_texture_input(tagcat("ssao-output", tag));

	lighting_pass.set_build_render_pass([this, hdr_out, emissive_in, &in_albedo, &in_normal, &in_pbr, &in_depth](HIP::CommandBuffer &cmd) {
		// lighting.ambient_occlusion = graph.maybe_get_physical_texture_resource(ssao_output) (scene_viewer_application.cpp:1571)
		lighting.ambient_occlusion = ssao_output ? &graph.get_physical_texture_resource(*ssao_output) : nullptr;
		DeferredLightAttachments att;
		att.base_color = &graph.get_physical_texture_resource(in_albedo);
		att.normal = &graph.get_physical_texture_resource(in_normal);
		att.pbr = &graph.get_physical_texture_resource(in_pbr);
		att.depth = &graph.get_physical_texture_resource(in_depth);
		att.hdr = &graph.get_physical_texture_resource(*hdr_out);
		att.emissive = emissive_in ? &graph.get_physical_texture_resource(*emissive_in) : att.hdr;
		att.rows = strip_plan.active() ? &strip_plan.lighting : nullptr;
		// the resolve of a temporal pre-AA and the SMAA passes of the frame before run beside this launch on the generic stream
		const PostAAType pre = to_post_aa_type(config.pre_aa), post = to_post_aa_type(config.post_aa);
		const bool heavy_neighbours = pre == PostAAType::TAA_Low || pre == PostAAType::TAA_Medium || pre == PostAAType::TAA_High ||
		                              post == PostAAType::SMAA_Low || post == PostAAType::SMAA_Medium || post == PostAAType::SMAA_High ||
		                              post == PostAAType::SMAA_Ultra;
		DeferredLightRenderer::render_light(cmd, context, att, heavy_neighbours ? DeferredLightRenderer::SHARE_REGISTERS_BIT : 0u);
	});

	// Scene::add_render_pass_dependencies(graph, lighting_pass, LIGHTING_BIT)
	cluster.setup_render_pass_dependencies(graph, lighting_pass, RenderPassCreator::LIGHTING_BIT);
}

void ImageSpaceApplication::bake_render_graph()
{
	GRANITE_SCOPED_TIMELINE_EVENT("bake-render-graph");
	// Keep feedback buffers (average luminance) alive across re-bakes (scene_viewer_application.cpp:1169,1315).
	auto physical_buffers = graph.consume_physical_buffers();
	graph.reset();
	filled_targets.clear(); // keyed by device pointer: a re-baked graph may place a new image at a recycled address
	graph.set_device(device_holder.get());
	if (device_holder)
		device_holder->reset_launch_cache(); // pre-recorded launch sequences hold the old graph's pointers
	graph.set_alias_disjoint_images(!config.disable_image_aliasing);
	// The frame's tail (post-tonemap anti-aliasing) on its own stream: one device rendering whole frames only -- under row bands the tail's
	// output meets the other ranks' on the collective stream, whose ordering against the frame is stated for three executor streams.
	// GRANITE_SPLIT_TAIL=0: A/B switch for measurements (frames identical either way: tests/test_gpu_app.py).
	{
		static const char *env = getenv("GRANITE_SPLIT_TAIL");
		if (env)
		{
			static bool told = false;
			if (!told)
				fprintf(stderr, "[granite-hip] note: GRANITE_SPLIT_TAIL=%s is set (results unchanged, timing differs)\n", env);
			told = true;
		}
		graph.set_split_tail(config.strip_count <= 1 && !(env && env[0] == '0'));
	}

	ResourceDimensions dim;
	dim.width = config.width;
	dim.height = config.height;
	dim.format = backbuffer_format();
	graph.set_backbuffer_dimensions(dim);

	const std::string tag = "main";
	if (config.aa_bench)
		add_aa_bench_main_pass(tag);
	else if (config.enable_lighting)
	{
		cluster.add_render_passes(graph);
		add_main_pass_deferred(tag);
	}
	else
		add_hdr_input_pass(tag);

	std::string light_output = tagcat("HDR", tag);
	if (config.ssr)
	{
		// scene_viewer_application.cpp:1206-1212
		if (!config.enable_lighting)
			throw std::logic_error("SSR needs the G-buffer of the deferred graph.");
		if (!ssr_tables_installed())
			throw std::logic_error("SSR: install the blue-noise and BRDF tables first (gra_install_ssr_tables).");
		setup_ssr_pass(graph, context, tagcat("depth-transient", tag), tagcat("albedo", tag), tagcat("normal", tag), tagcat("pbr", tag), light_output, "SSR");
		light_output = "SSR";
	}
	std::string ui_source = light_output;
	const PostAAType pre_aa = to_post_aa_type(config.pre_aa);
	const PostAAType post_aa = to_post_aa_type(config.post_aa);
	jitter.init(TemporalJitter::Type::None, vec2(0.0f));

	bool temporal = pre_aa == PostAAType::TAA_Low || pre_aa == PostAAType::TAA_Medium || pre_aa == PostAAType::TAA_High;
	if (temporal)
	{
		if (!config.enable_lighting && !config.aa_bench)
			throw std::logic_error("TAA needs the depth attachment of the deferred graph.");
		add_mv_pass(tag); // aa_bench.cpp hands the resolve an unnamed motion-vector input; here: the zero image
	}

	if (config.hdr_bloom)
	{
		bool resolved = setup_before_post_chain_antialiasing(pre_aa, graph, jitter, context, 1.0f, light_output, tagcat("depth", tag),
		                                                     tagcat("mv", tag), "HDR-resolved", &strip_plan);
		const std::string hdr_source = resolved ? "HDR-resolved" : light_output;
		{
			// the same question the lighting pass asks (heavy_neighbours): a temporal resolve in front of the bloom pass, SMAA behind the tonemap
			const PostAAType pre = to_post_aa_type(config.pre_aa), post = to_post_aa_type(config.post_aa);
			hdr_options.busy_frame = pre == PostAAType::TAA_Low || pre == PostAAType::TAA_Medium || pre == PostAAType::TAA_High || post == PostAAType::SMAA_Low ||
			                         post == PostAAType::SMAA_Medium || post == PostAAType::SMAA_High || post == PostAAType::SMAA_Ultra;
		}
		if (config.compute_post)
			setup_hdr_postprocess_compute(graph, frame, hdr_source, "tonemapped", hdr_options);
		else
			setup_hdr_postprocess(graph, frame, hdr_source, "tonemapped", hdr_options);
		ui_source = "tonemapped";
	}

	if (config.aa_bench)
	{
		// aa_bench.cpp:127-147: TAA in front, then "tonemap" = blit.frag with NearestClamp into a swapchain-sized target.
		const float scale = scaled() ? config.resolution_scale : 1.0f;
		bool resolved = setup_before_post_chain_antialiasing(pre_aa, graph, jitter, context, scale, light_output, tagcat("depth", tag),
		                                                     tagcat("mv", tag), "HDR-resolved");
		auto &tonemap = graph.add_pass("tonemap", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		AttachmentInfo swapchain_output;
		auto &tonemap_out = tonemap.add_color_output("tonemap", swapchain_output);
		auto &tonemap_res = tonemap.add_texture_input(resolved ? "HDR-resolved" : light_output);
		tonemap.set_build_render_pass([this, &tonemap_out, &tonemap_res](HIP::CommandBuffer &cmd) {
			auto &input = graph.get_physical_texture_resource(tonemap_res);
			auto &output = graph.get_physical_texture_resource(tonemap_out);
			cmd.check(gr_blit(cmd.get_context(), cmd.get_stream(), &input.get_view(), &output.get_view(), 0), "blit");
		});
		ui_source = "tonemap";
	}

	if (post_aa != PostAAType::None)
	{
		// PostAAType is a single enum in the viewer; the API composes TAA before and FXAA/SMAA after the chain, but the
		// later jitter.init() wins (smaa.cpp:60-67), so the temporal table is restored afterwards.
		TemporalJitter saved = jitter;
		if (setup_after_post_chain_antialiasing(post_aa, graph, jitter, 1.0f, ui_source, tagcat("depth", tag), "post-aa-output", &strip_plan))
			ui_source = "post-aa-output";
		if (temporal)
			jitter = saved;
	}

	if (config.hdr10)
	{
		// scene_viewer_application.cpp:1270-1288: a UI layer cleared to (0, 0, 0, 1), then the PQ encoder as the frame's last pass.
		auto &ui = graph.add_pass("ui", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		AttachmentInfo ui_info;
		ui_info.format = VK_FORMAT_R8G8B8A8_SRGB;
		auto &ui_layer = ui.add_color_output("ui-temporary", ui_info);
		ui.set_build_render_pass([this, &ui_layer](HIP::CommandBuffer &cmd) {
			auto &target = graph.get_physical_texture_resource(ui_layer);
			cmd.check(gr_fill_u32(cmd.get_context(), cmd.get_stream(), target.get_device_pointer(), 0xff000000u, target.get_size_bytes() / 4), "ui clear");
		});
		HDR10PQEncodingConfig hdr10_config = {};
		hdr10_config.hdr_pre_exposure = 500.0f;
		hdr10_config.ui_pre_exposure = 400.0f;
		setup_hdr10_pq_encoding(graph, "ui-output", ui_source, "ui-temporary", hdr10_config, HdrMetadata());
		ui_source = "ui-output";
	}

	if (scaled()) // scene_viewer_application.cpp:1263-1268
	{
		if (setup_after_post_chain_upscaling(graph, ui_source, "post-scale-output", config.resolution_scale_sharpen != 0, config.fsr_fp32 == 0))
			ui_source = "post-scale-output";
	}

	if (config.depth_hierarchy)
	{
		if (!config.enable_lighting)
			throw std::logic_error("The depth hierarchy needs the depth attachment of the deferred graph.");
		setup_depth_hierarchy_pass(graph, tagcat("depth", tag), "depth-hiz", &context, config.depth_hierarchy == 2);
		// Its consumers are outside the graph: a proxy keeps the pass alive and orders it before the end of the frame.
		constexpr VkPipelineStageFlags2 compute_stage = VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT;
		graph.find_pass("depth-hiz")->add_proxy_output("depth-hiz-ready", compute_stage, 0);
		for (unsigned writer : graph.get_texture_resource(ui_source).get_write_passes())
			graph.get_pass(writer).add_proxy_input("depth-hiz-ready", compute_stage, 0);
	}

	graph.set_backbuffer_source(ui_source);
	cluster.setup_render_pass_dependencies(graph);
	graph.bake();
	graph.install_physical_buffers(std::move(physical_buffers));
	need_bake = false;
	gbuffer_dirty = true;
}

void ImageSpaceApplication::render_frame()
{
	GRANITE_SCOPED_TIMELINE_EVENT("render-frame");
	auto &device = get_device();
	const auto host_t0 = std::chrono::steady_clock::now();
	check_taa_history_reach();
	if (need_bake)
		bake_render_graph();

	frame.frame_time = config.frame_time;
	frame.elapsed_time = elapsed;
	elapsed += config.frame_time;
	context.set_frame_parameters(frame);

	HIP::Image *backbuffer = swapchain[swapchain_index].get();
	swapchain_index = (swapchain_index + 1) % unsigned(swapchain.size());

	if (camera_moves)
		base_view = base_view * translate(-camera_motion); // the eye moves by +motion: view' = view * T(-motion)
	if (jitter.get_jitter_type() != TemporalJitter::Type::None)
	{
		if (!has_base_camera)
			throw std::logic_error("Temporal AA needs gra_set_camera (projection + view), not verbatim render parameters.");
		jitter.step(base_projection, base_view);
		context.set_camera(jitter.get_jittered_projection(), base_view);
	}
	else if (camera_moves)
		context.set_camera(base_projection, base_view);

	graph.setup_attachments(device, backbuffer);
	if (config.enable_lighting)
	{
		cluster.setup_render_pass_resources(graph);
		cluster.refresh(context, composer);
	}
	if (config.enable_lighting)
	{
		// Sort + pack of the NEXT frame's lights on the clusterer's helper threads while this frame is enqueued below, with the
		// parameters that frame is going to use (the next jitter phase when a temporal AA is active).  A camera or light
		// change in between simply makes the next refresh() pack on this thread as before.
		const mat4 next_view = camera_moves ? base_view * translate(-camera_motion) : base_view;
		if (jitter.get_jitter_type() != TemporalJitter::Type::None && has_base_camera)
		{
			TemporalJitter next_jitter = jitter;
			next_jitter.step(base_projection, next_view);
			RenderContext next_context = context;
			next_context.set_camera(next_jitter.get_jittered_projection(), next_view);
			cluster.prefetch(next_context.get_render_parameters());
		}
		else if (camera_moves)
		{
			RenderContext next_context = context;
			next_context.set_camera(base_projection, next_view);
			cluster.prefetch(next_context.get_render_parameters());
		}
		else
			cluster.prefetch(context.get_render_parameters());
	}
	graph.enqueue_render_passes(device, composer);
	gbuffer_dirty = false;
	last_backbuffer = backbuffer;
	host_frames++;
	host_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - host_t0).count();
}

void ImageSpaceApplication::set_directional_light(const float direction[3], const float color[3])
{
	lighting.directional.color = vec3(color[0], color[1], color[2]);
	lighting.directional.direction = normalize(vec3(direction[0], direction[1], direction[2]));
}

void ImageSpaceApplication::set_fog(const float color[3], float falloff)
{
	lighting.fog.color = vec3(color[0], color[1], color[2]);
	lighting.fog.falloff = falloff;
}

// A fresh application has no physical resources before its first frame: bake and allocate them now, so that inherited state can be
// written.  The extra setup_attachments() only rotates empty slots; the first frame's own swap then turns what was written into
// an image with a history copy into that frame's history.
void ImageSpaceApplication::prepare_resources_for_write()
{
	if (need_bake)
		bake_render_graph();
	if (host_frames == 0 && !resources_prepared)
	{
		graph.setup_attachments(get_device(), swapchain[swapchain_index].get());
		resources_prepared = true;
	}
}

void ImageSpaceApplication::get_frame_state(gra_frame_state &state) const
{
	state = {};
	state.frames = host_frames;
	state.elapsed = elapsed;
	state.swapchain_index = swapchain_index;
	memcpy(state.base_view, base_view.data(), sizeof(state.base_view));
	const TemporalJitter::State j = jitter.get_state();
	state.jitter_phase = j.phase;
	memcpy(state.jittered_projection, j.jittered_projection.data(), sizeof(state.jittered_projection));
	for (size_t i = 0; i < j.view_proj.size() && i < 16; i++)
	{
		memcpy(state.view_proj[i], j.view_proj[i].data(), 64);
		memcpy(state.inv_view_proj[i], j.inv_view_proj[i].data(), 64);
		memcpy(state.jittered_view_proj[i], j.jittered_view_proj[i].data(), 64);
	}
}

void ImageSpaceApplication::set_frame_state(const gra_frame_state &state)
{
	if (state.swapchain_index >= swapchain.size())
		throw std::logic_error("gra_set_frame_state: swapchain index out of range");
	// baking re-initialises the jitter sequence (as the reference's setup_*_postprocess do): bake before the saved phase goes in
	if (need_bake)
		bake_render_graph();
	elapsed = state.elapsed;
	swapchain_index = state.swapchain_index;
	// state.frames is informational (host_frames counts what THIS instance rendered; prepare_resources_for_write keys on it).  A
	// restore is all or nothing: prepare_resources_for_write allocates and rotates EVERY attachment with a history, so the first frame
	// after it takes the history paths (TAA resolve, bloom feedback, exposure) -- the caller writes back every inherited resource of
	// the saved frame (gra_write_resource: HDR-resolved-history, downsample-3, average-luminance ...), not a subset; what is left
	// unwritten is a zero history, which is neither frame 1 nor frame N + 1 of the original run.
	memcpy(base_view.data(), state.base_view, sizeof(state.base_view));
	TemporalJitter::State j = jitter.get_state();
	j.phase = state.jitter_phase;
	memcpy(j.jittered_projection.data(), state.jittered_projection, sizeof(state.jittered_projection));
	for (size_t i = 0; i < j.view_proj.size() && i < 16; i++)
	{
		memcpy(j.view_proj[i].data(), state.view_proj[i], 64);
		memcpy(j.inv_view_proj[i].data(), state.inv_view_proj[i], 64);
		memcpy(j.jittered_view_proj[i].data(), state.jittered_view_proj[i], 64);
	}
	jitter.set_state(j);
	if (has_base_camera)
		context.set_camera(jitter.get_jitter_type() != TemporalJitter::Type::None ? jitter.get_jittered_projection() : base_projection, base_view);
	cluster.invalidate_prefetch();
}
} // namespace Granite
