// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// ImageSpaceApplication — headless composition + frame loop of the image-space chain on the HIP executor.
// Graph composition follows SceneViewerApplication::bake_render_graph / add_main_pass_deferred
// (application/scene_viewer_application.cpp:876-991,1167-1318); the frame loop follows render_frame (:1540-1611) and the
// headless platform's external 4-image swapchain (application/platforms/application_headless.cpp:145-148,207-229).
#pragma once
#include <functional>
#include <memory>
#include <unordered_map>
#include <unordered_set>
#include <string>
#include <vector>
#include "../../../../include/granite_app.h"
#include "../collective.hpp"
#include "../lights/clusterer.hpp"
#include "../post/aa.hpp"
#include "../post/hdr.hpp"
#include "../render_context.hpp"
#include "../render_graph.hpp"
#include "../renderer.hpp"

namespace Granite
{
class ImageSpaceApplication
{
public:
	explicit ImageSpaceApplication(const gra_config &config);
	~ImageSpaceApplication();

	void set_lights(const gra_light_desc *descs, uint32_t count);
	void upload_gbuffer(const void *emissive, const void *albedo, const void *normal, const void *pbr, const void *depth, const void *mv);
	void render_frame();
	void wait_idle()
	{
		if (device_holder)
			device_holder->wait_idle();
		check_taa_history_reach();
	}
	// Row bands with a bounded TAA history reach: throws once a resolve has reported a fetch outside the rows this rank holds.
	void check_taa_history_reach();
	// Composes and bakes the graph without touching a GPU (config.device < 0 creates no device): used by CPU tests.
	void bake_only() { bake_render_graph(); }

	HIP::Device &get_device()
	{
		if (!device_holder)
			throw std::logic_error("Application was created without a device (dry mode).");
		return *device_holder;
	}
	// DirectionalLightComponent of the scene (read_lights, scene_viewer_application.cpp:58-77); direction need not be normalised.
	void set_directional_light(const float direction[3], const float color[3]);
	void set_fog(const float color[3], float falloff);
	// checkpoint / replay (include/granite_app.h: gra_write_resource, gra_frame_state)
	void prepare_resources_for_write();
	void get_frame_state(gra_frame_state &state) const;
	void set_frame_state(const gra_frame_state &state);
	RenderGraph &get_graph() { return graph; }
	RenderContext &get_context() { return context; }
	TemporalJitter &get_jitter() { return jitter; }
	HIP::Collective &get_collective() { return collective; }
	// Camera as the application sees it (un-jittered); with a temporal AA active each frame renders with
	// jitter.get_jittered_projection() like SceneViewerApplication::update_scene (scene_viewer_application.cpp:1431-1433).
	void set_base_camera(const mat4 &projection, const mat4 &view);
	mat4 get_taa_reprojection() const;
	LightClusterer &get_clusterer() { return cluster; }
	HIP::Image *get_last_backbuffer() { return last_backbuffer; }
	const gra_config &get_config() const { return config; }
	unsigned get_render_width() const { return render_width; }
	unsigned get_render_height() const { return render_height; }
	const StripPlan &get_strip_plan() const { return strip_plan; }
	void set_exchange_callback(gra_exchange_fn fn, void *user);
	void init_collective(const uint8_t *id128, int rank, int ranks);
	// Second communicator: the all-gather of the tonemapped bands moves to the device's collective stream and overlaps the
	// next frame ("overlap B with the next frame's lighting").
	void init_output_collective(const uint8_t *id128, int rank, int ranks);
	// Host-side cost of the frame loop: frames rendered, wall seconds inside render_frame(), of which blocked on the GPU.
	size_t get_allocated_bytes() const { return device_holder ? device_holder->get_allocated_bytes() : 0; }
	void get_host_stats(double out[3]) const
	{
		out[0] = double(host_frames);
		out[1] = host_seconds;
		out[2] = device_holder ? device_holder->get_blocked_seconds() : 0.0;
	}
	// The output gather beside the frame (init_output_collective): out[0] = times a pass was about to overwrite an output image, out[1] =
	// of those, how often the gather that last read the image was still in flight (the pass then waits: the gather was NOT hidden).
	void get_output_gather_stats(uint64_t out[2]) const
	{
		out[0] = output_acquires;
		out[1] = output_acquire_waits;
	}
	// G-buffer attachments from .gtx files (any may be null); the frame written back as .gtx.
	void upload_gbuffer_gtx(const char *const paths[6]);
	void set_camera_motion(const vec3 &translation_per_frame);
	void upload_ambient_occlusion(const void *ao_r8);
	void upload_aa_bench_images(const void *first, const void *second, uint32_t width, uint32_t height);
	void save_image_gtx(HIP::Image &image, const std::string &path);
	std::string last_error;

private:
	gra_config config;
	unsigned render_width = 0, render_height = 0; // backbuffer size x resolution_scale
	bool scaled() const { return config.resolution_scale > 0.0f && config.resolution_scale < 1.0f; }
	VkFormat backbuffer_format() const { return config.hdr10 ? VK_FORMAT_A2B10G10R10_UNORM_PACK32 : VK_FORMAT_R8G8B8A8_SRGB; }
	std::unique_ptr<HIP::Device> device_holder;
	RenderGraph graph;
	RenderContext context;
	FrameParameters frame;
	LightingParameters lighting;
	LightClusterer cluster;
	TaskComposer composer;
	HDROptions hdr_options;
	StripPlan strip_plan;
	HIP::Collective collective, output_collective;
	// per output image: the event its last beside-the-frame gather records on the collective stream
	std::unordered_map<const void *, void *> output_gather_done;
	void *last_output_gather_event = nullptr; // hipEvent_t of the newest output gather (one of output_gather_done)
	// 24-bit transport form of the output bands (one buffer per output image, rank_count * chunk_rows rows of width * 3 bytes)
	std::unordered_map<const void *, HIP::BufferHandle> packed_output;
	using BandTransport = std::function<void(void *base, size_t chunk_bytes, void *stream)>;
	bool output_packs(const HIP::Image &image, const char *tag) const;
	void pack_output_band(HIP::CommandBuffer &cmd, HIP::Image &image, uint32_t chunk_rows);
	void gather_packed_output(HIP::Image &image, uint32_t chunk_rows, void *stream, const BandTransport &transport);
	void *output_ready_event = nullptr;
	uint32_t *taa_reach_flag = nullptr; // pinned host word the band resolve writes (strip_plan.taa_reach_flag)
	TemporalJitter jitter;
	mat4 base_projection, base_view;
	bool has_base_camera = false;

	// "Scene": light objects + node transforms, and the synthetic G-buffer sources.
	std::vector<std::unique_ptr<PositionalLight>> light_objects;
	std::vector<mat_affine> light_transforms;
	PositionalLightList light_list;
	HIP::ImageHandle src_emissive, src_albedo, src_normal, src_pbr, src_depth, src_mv, src_ao;
	// scene_viewer_application.cpp:881-883: renderTargetFp16 ? R16G16B16A16_SFLOAT : B10G11R11_UFLOAT_PACK32
	VkFormat hdr_target_format() const { return config.hdr_packed_float ? VK_FORMAT_B10G11R11_UFLOAT_PACK32 : VK_FORMAT_R16G16B16A16_SFLOAT; }
	vec3 camera_motion = vec3(0.0f); // eye translation per frame (world units)
	bool camera_moves = false;
	HIP::ImageHandle bench_images[2]; // aa_bench: the two input images, alternating per frame
	unsigned bench_input_index = 0;
	void add_aa_bench_main_pass(const std::string &tag);
	RenderTextureResource *ssao_output = nullptr;
	bool gbuffer_dirty = true;
	// Physical targets that already hold the current synthetic upload (an attachment the executor double-buffers has two).
	std::unordered_set<const void *> filled_targets;
	bool needs_fill(const HIP::Image &target) { return filled_targets.insert(target.get_device_pointer()).second; }
	bool is_filled(const HIP::Image &target) const { return filled_targets.count(target.get_device_pointer()) != 0; }

	std::vector<HIP::ImageHandle> swapchain;
	unsigned swapchain_index = 0;
	HIP::Image *last_backbuffer = nullptr;
	bool need_bake = true;
	bool resources_prepared = false; // prepare_resources_for_write() ran before the first frame
	double elapsed = 0.0;
	uint64_t host_frames = 0;
	uint64_t output_acquires = 0, output_acquire_waits = 0;
	double host_seconds = 0.0;

	void bake_render_graph();
	void add_main_pass_deferred(const std::string &tag);
	void add_hdr_input_pass(const std::string &tag);
	void add_mv_pass(const std::string &tag);
};
} // namespace Granite
