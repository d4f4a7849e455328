// extern "C" surface of the harness (include/granite_app.h).  Exceptions from the C++ layer stop here.
#include "image_space_app.hpp"
#include "../timeline_trace.hpp"
#include "../post/ssr.hpp"
#include "../post/spd.hpp"
#include "../gtx.hpp"
#include "../post/hdr.hpp"
#include <cstdio>
#include <cstring>

using namespace Granite;

struct gra_app
{
	std::unique_ptr<ImageSpaceApplication> app;
	std::string error;
};

template <typename Fn>
static int guarded(gra_app *app, Fn &&fn)
{
	if (!app)
		return -1;
	try
	{
		fn();
		return 0;
	}
	catch (const std::exception &e)
	{
		app->error = e.what();
		return -1;
	}
}

static void unpack_mat4(mat4 &m, const float *src)
{
	memcpy(m.data(), src, 16 * sizeof(float));
}

template <typename Fn>
static int guarded_message(char *error, size_t error_size, Fn &&fn)
{
	try
	{
		fn();
		return 0;
	}
	catch (const std::exception &e)
	{
		if (error && error_size)
			snprintf(error, error_size, "%s", e.what());
		return -1;
	}
}

static void fill_gtx_info(gra_gtx_info *info, const GtxImage &img)
{
	info->type = img.type;
	info->format = uint32_t(img.format);
	info->width = img.width;
	info->height = img.height;
	info->depth = img.depth;
	info->layers = img.layers;
	info->levels = img.levels;
	info->flags = img.flags;
	info->payload_size = img.payload.size();
}

extern "C" {

gra_app *gra_create(const gra_config *config, char *error, size_t error_size)
{
	try
	{
		if (!config)
			throw std::logic_error("null config");
		auto *handle = new gra_app;
		handle->app = std::make_unique<ImageSpaceApplication>(*config);
		return handle;
	}
	catch (const std::exception &e)
	{
		if (error && error_size)
			snprintf(error, error_size, "%s", e.what());
		return nullptr;
	}
}

void gra_destroy(gra_app *app)
{
	delete app;
	Granite::TimelineTrace::get().flush(); // GRANITE_TIMELINE_TRACE: what the threads recorded goes to the file
}

const char *gra_last_error(gra_app *app)
{
	return app ? app->error.c_str() : "null app";
}

int gra_set_camera(gra_app *app, const float *projection16, const float *view16)
{
	return guarded(app, [&]() {
		mat4 p, v;
		unpack_mat4(p, projection16);
		unpack_mat4(v, view16);
		app->app->set_base_camera(p, v);
	});
}

int gra_set_camera_motion(gra_app *app, const float translation[3])
{
	return guarded(app, [&]() {
		if (!translation)
			throw std::logic_error("gra_set_camera_motion: null translation");
		app->app->set_camera_motion(vec3(translation[0], translation[1], translation[2]));
	});
}

int gra_set_render_parameters(gra_app *app, const float *f)
{
	return guarded(app, [&]() {
		RenderParameters rp = app->app->get_context().get_render_parameters();
		unpack_mat4(rp.projection, f);
		unpack_mat4(rp.view, f + 16);
		unpack_mat4(rp.view_projection, f + 32);
		unpack_mat4(rp.inv_projection, f + 48);
		unpack_mat4(rp.inv_view, f + 64);
		unpack_mat4(rp.inv_view_projection, f + 80);
		rp.unjittered_view_projection = rp.view_projection;
		rp.unjittered_inv_view_projection = rp.inv_view_projection;
		rp.camera_position = vec3(f[96], f[97], f[98]);
		rp.camera_front = vec3(f[99], f[100], f[101]);
		rp.z_near = f[102];
		rp.z_far = f[103];
		app->app->get_context().set_render_parameters(rp);
	});
}

int gra_get_render_parameters(gra_app *app, float *f)
{
	return guarded(app, [&]() {
		auto &rp = app->app->get_context().get_render_parameters();
		const mat4 *mats[6] = {&rp.projection, &rp.view, &rp.view_projection, &rp.inv_projection, &rp.inv_view, &rp.inv_view_projection};
		for (int i = 0; i < 6; i++)
			memcpy(f + 16 * i, mats[i]->data(), 16 * sizeof(float));
		for (int i = 0; i < 3; i++)
		{
			f[96 + i] = rp.camera_position[i];
			f[99 + i] = rp.camera_front[i];
		}
		f[102] = rp.z_near;
		f[103] = rp.z_far;
	});
}

int gra_set_lights(gra_app *app, const gra_light_desc *lights, uint32_t count)
{
	return guarded(app, [&]() { app->app->set_lights(lights, count); });
}

int gra_upload_gbuffer(gra_app *app, const void *emissive, const void *albedo, const void *normal, const void *pbr, const void *depth,
                       const void *mv)
{
	return guarded(app, [&]() { app->app->upload_gbuffer(emissive, albedo, normal, pbr, depth, mv); });
}

int gra_compute_rec709_to_display(const float *primaries8, float *out9)
{
	if (!primaries8 || !out9)
		return -1;
	HdrMetadata md;
	memcpy(md.display_primary_red, primaries8, 8);
	memcpy(md.display_primary_green, primaries8 + 2, 8);
	memcpy(md.display_primary_blue, primaries8 + 4, 8);
	memcpy(md.white_point, primaries8 + 6, 8);
	compute_rec709_to_st2020(md, out9);
	return 0;
}

int gra_gtx_probe(const char *path, gra_gtx_info *info, char *error, size_t error_size)
{
	return guarded_message(error, error_size, [&]() {
		if (!path || !info)
			throw std::logic_error("gra_gtx_probe: null argument");
		fill_gtx_info(info, gtx_load(path));
	});
}

int gra_gtx_read(const char *path, void *payload, uint64_t payload_capacity, char *error, size_t error_size)
{
	return guarded_message(error, error_size, [&]() {
		if (!path || !payload)
			throw std::logic_error("gra_gtx_read: null argument");
		auto img = gtx_load(path);
		if (img.payload.size() > payload_capacity)
			throw std::logic_error("gra_gtx_read: destination is smaller than the payload");
		memcpy(payload, img.payload.data(), img.payload.size());
	});
}

int gra_gtx_write(const char *path, const gra_gtx_info *info, const void *payload, char *error, size_t error_size)
{
	return guarded_message(error, error_size, [&]() {
		if (!path || !info || !payload)
			throw std::logic_error("gra_gtx_write: null argument");
		GtxImage img;
		img.type = info->type;
		img.format = VkFormat(info->format);
		img.width = info->width;
		img.height = info->height;
		img.depth = info->depth;
		img.layers = info->layers;
		img.levels = info->levels;
		img.flags = info->flags;
		if (info->payload_size != img.required_payload_size())
			throw std::logic_error("gra_gtx_write: payload_size does not match the layout of the described image");
		img.payload.assign(static_cast<const uint8_t *>(payload), static_cast<const uint8_t *>(payload) + info->payload_size);
		gtx_save(img, path);
	});
}

int gra_upload_aa_bench_images(gra_app *app, const void *rgba8_first, const void *rgba8_second, uint32_t width, uint32_t height)
{
	return guarded(app, [&]() { app->app->upload_aa_bench_images(rgba8_first, rgba8_second, width, height); });
}

int gra_upload_ambient_occlusion(gra_app *app, const void *ao_r8)
{
	return guarded(app, [&]() { app->app->upload_ambient_occlusion(ao_r8); });
}

int gra_upload_gbuffer_gtx(gra_app *app, const char *emissive, const char *albedo, const char *normal, const char *pbr,
                           const char *depth, const char *motion_vectors)
{
	return guarded(app, [&]() {
		const char *paths[6] = {emissive, albedo, normal, pbr, depth, motion_vectors};
		app->app->upload_gbuffer_gtx(paths);
	});
}

int gra_get_render_size(gra_app *app, uint32_t *width, uint32_t *height)
{
	return guarded(app, [&]() {
		if (!width || !height)
			throw std::logic_error("gra_get_render_size: null output");
		*width = app->app->get_render_width();
		*height = app->app->get_render_height();
	});
}

int gra_render_frames(gra_app *app, uint32_t count, int32_t sync)
{
	return guarded(app, [&]() {
		for (uint32_t i = 0; i < count; i++)
			app->app->render_frame();
		if (sync)
			app->app->wait_idle();
	});
}

int gra_set_exchange_callback(gra_app *app, gra_exchange_fn fn, void *user)
{
	return guarded(app, [&]() { app->app->set_exchange_callback(fn, user); });
}

int gra_comm_create_unique_id(uint8_t *id128)
{
	if (!id128)
		return -1;
	try
	{
		HIP::Collective::create_unique_id(id128);
		return 0;
	}
	catch (const std::exception &e)
	{
		fprintf(stderr, "gra_comm_create_unique_id: %s\n", e.what());
		return -1;
	}
}

int gra_comm_init(gra_app *app, const uint8_t *id128, int32_t rank, int32_t ranks)
{
	return guarded(app, [&]() {
		if (!id128)
			throw std::logic_error("gra_comm_init: null id");
		app->app->init_collective(id128, rank, ranks);
	});
}

int gra_comm_info(gra_app *app, int32_t *nranks, int32_t *version, int32_t *stand_in)
{
	return guarded(app, [&]() {
		auto &c = app->app->get_collective();
		if (!c.is_initialized())
			throw std::logic_error("gra_comm_info: no communicator (gra_comm_init)");
		if (nranks)
			*nranks = c.communicator_ranks();
		if (version)
			*version = HIP::Collective::library_version();
		if (stand_in)
			*stand_in = HIP::Collective::is_stand_in() ? 1 : 0;
	});
}

int gra_comm_init_output(gra_app *app, const uint8_t *id128, int32_t rank, int32_t ranks)
{
	return guarded(app, [&]() {
		if (!id128)
			throw std::logic_error("gra_comm_init_output: null id");
		app->app->init_output_collective(id128, rank, ranks);
	});
}

int gra_get_strip_plan(gra_app *app, uint32_t *out24)
{
	return guarded(app, [&]() {
		if (!out24)
			throw std::logic_error("gra_get_strip_plan: null output");
		auto &p = app->app->get_strip_plan();
		uint32_t *o = out24;
		*o++ = p.index; *o++ = p.count; *o++ = p.width; *o++ = p.height;
		for (const Granite::RowRange *r : {&p.lighting, &p.threshold, &p.d0, &p.d1, &p.u0, &p.tonemap})
		{
			*o++ = r->whole ? 1u : 0u;
			*o++ = r->first;
			*o++ = r->count;
		}
		*o++ = p.d1_chunk_rows;
		*o++ = p.out_chunk_rows;
	});
}

int gra_get_strip_plan_aa(gra_app *app, uint32_t *out12)
{
	return guarded(app, [&]() {
		if (!out12)
			throw std::logic_error("gra_get_strip_plan_aa: null output");
		auto &p = app->app->get_strip_plan();
		uint32_t *o = out12;
		for (const Granite::RowRange *r : {&p.taa, &p.smaa_edges, &p.smaa_weights, &p.aa_out})
		{
			*o++ = r->whole ? 1u : 0u;
			*o++ = r->first;
			*o++ = r->count;
		}
	});
}

int gra_get_strip_plan_taa_history(gra_app *app, uint32_t *out5)
{
	return guarded(app, [&]() {
		if (!out5)
			throw std::logic_error("gra_get_strip_plan_taa_history: null output");
		auto &p = app->app->get_strip_plan();
		out5[0] = p.aa.taa_history_reach;
		out5[1] = p.taa_exchange_rows;
		out5[2] = p.taa_history_held.whole ? 1u : 0u;
		out5[3] = p.taa_history_held.first;
		out5[4] = p.taa_history_held.count;
	});
}

int gra_get_host_stats(gra_app *app, double *out3)
{
	return guarded(app, [&]() {
		if (!out3)
			throw std::logic_error("gra_get_host_stats: null output");
		app->app->get_host_stats(out3);
	});
}

int gra_get_output_gather_stats(gra_app *app, uint64_t *out2)
{
	return guarded(app, [&]() {
		if (!out2)
			throw std::logic_error("gra_get_output_gather_stats: null output");
		app->app->get_output_gather_stats(out2);
	});
}

int gra_install_ssr_tables(const uint8_t *blue_noise_128x128_rg8, const uint16_t *brdf_lut_rg16f, uint32_t brdf_width, uint32_t brdf_height)
{
	try
	{
		Granite::ssr_install_tables(blue_noise_128x128_rg8, brdf_lut_rg16f, brdf_width, brdf_height);
		return 0;
	}
	catch (const std::exception &e)
	{
		fprintf(stderr, "gra_install_ssr_tables: %s\n", e.what());
		return -1;
	}
}

int gra_get_prefetched_refreshes(gra_app *app, uint64_t *out)
{
	return guarded(app, [&]() {
		if (!out)
			throw std::logic_error("gra_get_prefetched_refreshes: null output");
		*out = app->app->get_clusterer().get_prefetch_hits();
	});
}

int gra_get_launch_graph_replays(gra_app *app, uint64_t *out)
{
	return guarded(app, [&]() {
		if (!out)
			throw std::logic_error("gra_get_launch_graph_replays: null output");
		*out = app->app->get_device().get_launch_graph_replays();
	});
}

int gra_get_allocated_bytes(gra_app *app, uint64_t *out)
{
	return guarded(app, [&]() {
		if (!out)
			throw std::logic_error("gra_get_allocated_bytes: null output");
		*out = app->app->get_allocated_bytes();
	});
}

int gra_sync(gra_app *app)
{
	return guarded(app, [&]() {
		app->app->wait_idle();
		Granite::TimelineTrace::get().flush();
	});
}

static void fill_image_info(gra_resource_info *info, HIP::Image &img, int phys)
{
	info->device_ptr = img.get_device_pointer();
	info->width = img.get_width();
	info->height = img.get_height();
	info->format = img.get_format();
	info->size_bytes = img.get_size_bytes();
	info->physical_index = phys;
	info->levels = img.get_levels();
}

static void lookup_resource(gra_app *app, const char *name, gra_resource_info *info)
{
	auto &graph = app->app->get_graph();
	*info = {};
	// Buffers and textures share one name table; try texture first.
	try
	{
		auto &tex = graph.get_texture_resource(name);
		if (tex.get_physical_index() == RenderResource::Unused)
			throw std::logic_error(std::string("resource not part of the baked graph: ") + name);
		fill_image_info(info, graph.get_physical_texture_resource(tex), int(tex.get_physical_index()));
		return;
	}
	catch (const std::logic_error &)
	{
	}
	auto &buf = graph.get_buffer_resource(name);
	if (buf.get_physical_index() == RenderResource::Unused)
		throw std::logic_error(std::string("resource not part of the baked graph: ") + name);
	auto &phys = graph.get_physical_buffer_resource(buf);
	info->device_ptr = phys.get_device_pointer();
	info->size_bytes = phys.get_size();
	info->physical_index = int(buf.get_physical_index());
}

int gra_get_resource(gra_app *app, const char *name, gra_resource_info *info)
{
	return guarded(app, [&]() { lookup_resource(app, name, info); });
}

int gra_read_resource(gra_app *app, const char *name, void *dst_host, uint64_t size_bytes)
{
	return guarded(app, [&]() {
		gra_resource_info info;
		lookup_resource(app, name, &info);
		if (size_bytes > info.size_bytes)
			throw std::logic_error("read size exceeds resource size");
		app->app->wait_idle();
		auto *ctx = app->app->get_device().get_context();
		if (gr_download(ctx, nullptr, dst_host, info.device_ptr, size_bytes) < 0)
			throw std::runtime_error(gr_last_error(ctx));
	});
}

int gra_write_resource(gra_app *app, const char *name, const void *src_host, uint64_t size_bytes)
{
	return guarded(app, [&]() {
		if (!src_host)
			throw std::logic_error("gra_write_resource: null source");
		app->app->prepare_resources_for_write();
		gra_resource_info info;
		lookup_resource(app, name, &info);
		if (size_bytes != info.size_bytes)
			throw std::logic_error("gra_write_resource: the size must be the resource's (" + std::to_string(info.size_bytes) + " bytes)");
		app->app->wait_idle();
		auto *ctx = app->app->get_device().get_context();
		if (gr_upload(ctx, nullptr, info.device_ptr, src_host, size_bytes) < 0 || gr_sync(ctx, nullptr) < 0)
			throw std::runtime_error(gr_last_error(ctx));
	});
}

int gra_get_frame_state(gra_app *app, gra_frame_state *state)
{
	return guarded(app, [&]() {
		if (!state)
			throw std::logic_error("gra_get_frame_state: null argument");
		app->app->get_frame_state(*state);
	});
}

int gra_set_frame_state(gra_app *app, const gra_frame_state *state)
{
	return guarded(app, [&]() {
		if (!state)
			throw std::logic_error("gra_set_frame_state: null argument");
		app->app->set_frame_state(*state);
	});
}

int gra_save_resource_gtx(gra_app *app, const char *name, const char *path)
{
	return guarded(app, [&]() {
		if (!path)
			throw std::logic_error("gra_save_resource_gtx: null path");
		if (!name)
		{
			auto *bb = app->app->get_last_backbuffer();
			if (!bb)
				throw std::logic_error("no frame rendered yet");
			app->app->save_image_gtx(*bb, path);
			return;
		}
		auto &graph = app->app->get_graph();
		auto &tex = graph.get_texture_resource(name);
		app->app->save_image_gtx(graph.get_physical_texture_resource(tex), path);
	});
}

int gra_get_backbuffer(gra_app *app, gra_resource_info *info)
{
	return guarded(app, [&]() {
		auto *bb = app->app->get_last_backbuffer();
		if (!bb)
			throw std::logic_error("no frame rendered yet");
		fill_image_info(info, *bb, -1);
	});
}

int gra_read_backbuffer(gra_app *app, void *dst_host, uint64_t size_bytes)
{
	return guarded(app, [&]() {
		auto *bb = app->app->get_last_backbuffer();
		if (!bb)
			throw std::logic_error("no frame rendered yet");
		if (size_bytes > bb->get_size_bytes())
			throw std::logic_error("read size exceeds backbuffer size");
		app->app->wait_idle();
		auto *ctx = app->app->get_device().get_context();
		if (gr_download(ctx, nullptr, dst_host, bb->get_device_pointer(), size_bytes) < 0)
			throw std::runtime_error(gr_last_error(ctx));
	});
}

int gra_get_cluster_state(gra_app *app, void *lights48, void *models48, uint32_t *type_mask128, void *params176, uint32_t *light_ranges)
{
	if (!app)
		return -1;
	try
	{
		auto &c = app->app->get_clusterer();
		auto &lights = c.get_packed_lights();
		if (lights48)
			memcpy(lights48, lights.data(), lights.size() * sizeof(lights[0]));
		if (models48)
			memcpy(models48, c.get_packed_models().data(), lights.size() * sizeof(mat_affine));
		if (type_mask128)
			memcpy(type_mask128, c.get_type_mask(), 128 * sizeof(uint32_t));
		if (params176)
			memcpy(params176, &c.get_cluster_parameters_bindless(), sizeof(ClustererParametersBindless));
		if (light_ranges)
			memcpy(light_ranges, c.get_volume_index_range().data(), c.get_volume_index_range().size() * sizeof(uvec2));
		return int(lights.size());
	}
	catch (const std::exception &e)
	{
		app->error = e.what();
		return -1;
	}
}

size_t gra_dump_graph(gra_app *app, char *buffer, size_t size)
{
	if (!app)
		return 0;
	std::string json;
	try
	{
		if (app->app->get_graph().get_baked_pass_order().empty())
			app->app->bake_only();
		json = app->app->get_graph().dump_json();
	}
	catch (const std::exception &e)
	{
		app->error = e.what();
		return 0;
	}
	if (buffer && size)
	{
		size_t n = std::min(size - 1, json.size());
		memcpy(buffer, json.data(), n);
		buffer[n] = '\0';
	}
	return json.size() + 1;
}

int gra_generate_mipmaps(gra_app *app, const void *level0_rgba16f, uint32_t width, uint32_t height, uint32_t levels,
                         uint32_t components, const float *filter_mods, void *chain_rgba16f)
{
	return guarded(app, [&]() {
		if (!level0_rgba16f || !chain_rgba16f || width == 0 || height == 0 || levels < 2 || levels > Granite::MaxSPDMips + 1)
			throw std::logic_error("gra_generate_mipmaps: bad argument");
		auto &device = app->app->get_device();
		device.wait_idle();
		auto image = device.create_image(width, height, VK_FORMAT_R16G16B16A16_SFLOAT, "mipmapped", levels);
		auto *ctx = device.get_context();
		auto stream = device.get_stream(HIP::CommandBuffer::Type::Generic);
		if (gr_upload(ctx, stream, image->get_device_pointer(), level0_rgba16f, size_t(width) * height * 8) < 0)
			throw std::runtime_error(gr_last_error(ctx));

		// The reference's use of the downsampler (renderer/ocean.cpp:579-601): level 0 is the source, levels 1.. the outputs.
		const gr_image source = image->get_level_view(0);
		std::vector<gr_image> views;
		std::vector<const gr_image *> mips;
		for (unsigned l = 1; l < levels; l++)
			views.push_back(image->get_level_view(l));
		for (auto &v : views)
			mips.push_back(&v);
		std::vector<Granite::vec4> mods;
		if (filter_mods)
			for (unsigned l = 0; l + 1 < levels; l++)
				mods.emplace_back(filter_mods[4 * l], filter_mods[4 * l + 1], filter_mods[4 * l + 2], filter_mods[4 * l + 3]);

		Granite::SPDInfo info = {};
		info.input = &source;
		info.output_mips = mips.data();
		info.num_mips = unsigned(mips.size());
		info.num_components = components;
		info.filter_mod = filter_mods ? mods.data() : nullptr;
		info.mode = Granite::ReductionMode::Color;
		HIP::CommandBuffer cmd{device, stream, HIP::CommandBuffer::Type::Generic};
		Granite::emit_single_pass_downsample(cmd, info);
		if (gr_download(ctx, stream, chain_rgba16f, image->get_device_pointer(), image->get_size_bytes()) < 0 || gr_sync(ctx, stream) < 0)
			throw std::runtime_error(gr_last_error(ctx));
	});
}

int gra_reset_timestamps(gra_app *app)
{
	if (!app)
		return -1;
	try
	{
		app->app->get_graph().reset_timestamps();
		return 0;
	}
	catch (const std::exception &e)
	{
		app->error = e.what();
		return -1;
	}
}

int gra_set_directional_light(gra_app *app, const float direction[3], const float color[3])
{
	if (!app)
		return -1;
	if (!direction || !color)
	{
		app->error = "gra_set_directional_light: null argument";
		return -1;
	}
	app->app->set_directional_light(direction, color);
	return 0;
}

int gra_set_fog(gra_app *app, const float color[3], float falloff)
{
	if (!app)
		return -1;
	if (!color)
	{
		app->error = "gra_set_fog: null argument";
		return -1;
	}
	app->app->set_fog(color, falloff);
	return 0;
}

int gra_collect_timestamps(gra_app *app, gra_timestamp *entries, int max_entries)
{
	if (!app)
		return -1;
	try
	{
		auto reports = app->app->get_graph().collect_timestamps();
		int n = 0;
		for (auto &r : reports)
		{
			if (n >= max_entries)
				break;
			snprintf(entries[n].tag, sizeof(entries[n].tag), "%s", r.tag.c_str());
			entries[n].count = r.count;
			entries[n].total_ms = r.total_ms;
			n++;
		}
		return n;
	}
	catch (const std::exception &e)
	{
		app->error = e.what();
		return -1;
	}
}

int gra_get_taa_reprojection(gra_app *app, float *reproj16)
{
	return guarded(app, [&]() {
		mat4 m = app->app->get_taa_reprojection();
		memcpy(reproj16, m.data(), 16 * sizeof(float));
	});
}

int gra_set_smaa_luts(gra_app *app, const void *area_rg8, const void *search_r8)
{
	return guarded(app, [&]() {
		auto *ctx = app->app->get_device().get_context();
		if (gr_smaa_set_luts(ctx, area_rg8, search_r8) < 0)
			throw std::runtime_error(gr_last_error(ctx));
	});
}

void *gra_get_kernel_context(gra_app *app)
{
	return app ? app->app->get_device().get_context() : nullptr;
}

void *gra_get_stream(gra_app *app)
{
	return app ? app->app->get_device().get_stream(HIP::CommandBuffer::Type::Generic) : nullptr;
}
}
