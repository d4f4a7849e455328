// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
#include "clusterer.hpp"
#include "../timeline_trace.hpp"
#include <cstdlib>
#include <algorithm>
#include <cstring>

namespace Granite
{
// ---- graph wiring (clusterer.cpp:83-116,1575-1619) -------------------------------------------------------------------
void LightClusterer::add_render_passes(RenderGraph &graph)
{
	BufferInfo att;
	att.usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT | VK_BUFFER_USAGE_TRANSFER_DST_BIT;
	auto &pass = graph.add_pass("clustering-bindless", RENDER_GRAPH_QUEUE_COMPUTE_BIT);

	// Always sized for the 4096-light maximum, one bit per light per cell.
	att.size = resolution_x * resolution_y * (MaxLightsBindless / 8);
	pass.add_storage_output("cluster-bitmask", att);
	pass.add_storage_output("cluster-bitmask-decal", att);

	att.size = resolution_z * sizeof(ivec2);
	pass.add_storage_output("cluster-range", att);
	pass.add_storage_output("cluster-range-decal", att);

	att.size = GR_TRANSFORMS_SIZE; // sizeof(ClustererBindlessTransforms)
	pass.add_transfer_output("cluster-transforms", att);

	att.size = GR_CULL_SETUP_BYTES_PER_LIGHT * MaxLightsBindless;
	pass.add_storage_output("cluster-cull-setup", att);

	att.size = GR_TRANSFORMED_SPOT_BYTES_PER_LIGHT * MaxLightsBindless;
	pass.add_storage_output("cluster-transformed-spot", att);

	pass.set_build_render_pass([this](HIP::CommandBuffer &cmd) { build_cluster_bindless_gpu(cmd); });
}

void LightClusterer::setup_render_pass_dependencies(RenderGraph &, RenderPass &target, DependencyFlags dep_flags)
{
	if ((dep_flags & RenderPassCreator::LIGHTING_BIT) != 0)
	{
		target.add_storage_read_only_input("cluster-bitmask");
		target.add_storage_read_only_input("cluster-range");
		target.add_storage_read_only_input("cluster-transforms");
	}
}

void LightClusterer::setup_render_pass_resources(RenderGraph &graph)
{
	bindless.bitmask_buffer = &graph.get_physical_buffer_resource(graph.get_buffer_resource("cluster-bitmask"));
	bindless.range_buffer = &graph.get_physical_buffer_resource(graph.get_buffer_resource("cluster-range"));
	bindless.transforms_buffer = &graph.get_physical_buffer_resource(graph.get_buffer_resource("cluster-transforms"));
	bindless.transformed_spots = &graph.get_physical_buffer_resource(graph.get_buffer_resource("cluster-transformed-spot"));
	bindless.cull_data = &graph.get_physical_buffer_resource(graph.get_buffer_resource("cluster-cull-setup"));
	if (!bindless.light_ranges)
		bindless.light_ranges = graph.get_device().create_buffer(MaxLightsBindless * sizeof(uvec2), VK_BUFFER_USAGE_STORAGE_BUFFER_BIT,
		                                                         "cluster-light-ranges");
}

// ---- per-frame CPU refresh (clusterer.cpp:656-703,781-827,1133-1176; threaded_scene.cpp:141-150) -----------------------
float LightClusterer::get_z_slice_extent(const RenderParameters &rp) const
{
	return std::min(0.5f, rp.z_far / float(resolution_z));
}

static bool same_parameters(const RenderParameters &a, const RenderParameters &b)
{
	auto eq = [](const auto &x, const auto &y) { return memcmp(&x, &y, sizeof(x)) == 0; };
	return eq(a.projection, b.projection) && eq(a.view, b.view) && eq(a.view_projection, b.view_projection) &&
	       eq(a.inv_projection, b.inv_projection) && eq(a.camera_position, b.camera_position) && eq(a.camera_front, b.camera_front) &&
	       a.z_near == b.z_near && a.z_far == b.z_far;
}

void LightClusterer::refresh(const RenderContext &context_, TaskComposer &)
{
	GRANITE_SCOPED_TIMELINE_EVENT("clusterer-refresh");
	const RenderParameters &rp = context_.get_render_parameters();
	{
		std::unique_lock<std::mutex> holder{ahead.lock};
		wait_for_workers(holder);
		if (ahead.result_valid && same_parameters(ahead.parameters, rp))
		{
			// The helper threads have already sorted and packed this frame's lights.
			std::swap(packed.parameters, ahead.result.parameters);
			packed.lights.swap(ahead.result.lights);
			packed.model.swap(ahead.result.model);
			packed.volume_index_range.swap(ahead.result.volume_index_range);
			memcpy(packed.type_mask, ahead.result.type_mask, sizeof(packed.type_mask));
			ahead.result_valid = false;
			prefetch_hits++;
			return;
		}
		ahead.result_valid = false;
	}
	GRANITE_SCOPED_TIMELINE_EVENT("light-sort-and-pack");
	sort_and_pack(rp, sort_state, packed);
}

void LightClusterer::sort_and_pack(const RenderParameters &rp, SortState &sort, PackedLights &out) const
{
	sort_lights(rp, sort);
	begin_pack(rp, sort, out);
	const unsigned chunks = (unsigned(out.parameters.num_lights) + PackChunk - 1) / PackChunk;
	for (unsigned c = 0; c < chunks; c++)
		pack_chunk(rp, sort, out, c);
}

void LightClusterer::prefetch(const RenderParameters &next_parameters)
{
	// Waking three helper threads and collecting their result costs the submitting thread 12-15 us per frame (futex round trips;
	// measured on the device-less frame loop, tests/test_host_frame_loop_cpu.py); sorting and packing costs ~22 ns per light.  Below
	// about a thousand lights the hand-over is the larger part: refresh() packs in place then.
	// GRANITE_LIGHT_PREFETCH_MIN=<n> moves the threshold (tests run the threaded path on small scenes with 0).
	static const size_t min_lights = []() {
		const char *env = getenv("GRANITE_LIGHT_PREFETCH_MIN");
		return env ? size_t(strtoul(env, nullptr, 10)) : size_t(PrefetchMinLights);
	}();
	if (!scene_lights || scene_lights->size() < min_lights)
		return;
	std::unique_lock<std::mutex> holder{ahead.lock};
	if (ahead.threads.empty())
	{
		const unsigned hw = std::thread::hardware_concurrency();
		const unsigned count = hw >= 8 ? 3u : (hw >= 4 ? 2u : 1u);
		for (unsigned i = 0; i < count; i++)
			ahead.threads.emplace_back([this, i]() { worker_main(i); });
	}
	wait_for_workers(holder);
	ahead.parameters = next_parameters;
	ahead.result_valid = false;
	ahead.in_flight = true;
	ahead.sorted.store(0, std::memory_order_relaxed);
	ahead.next_chunk.store(0, std::memory_order_relaxed);
	ahead.workers_left.store(int(ahead.threads.size()), std::memory_order_relaxed);
	ahead.generation++;
	ahead.wake.notify_all();
}

void LightClusterer::invalidate_prefetch()
{
	std::unique_lock<std::mutex> holder{ahead.lock};
	wait_for_workers(holder);
	ahead.result_valid = false;
}

void LightClusterer::wait_for_workers(std::unique_lock<std::mutex> &holder)
{
	if (!ahead.in_flight)
		return;
	// Usually the job finished long ago: it had the whole submission of a frame to run beside.
	ahead.done.wait(holder, [this]() { return ahead.workers_left.load(std::memory_order_acquire) == 0; });
	ahead.in_flight = false;
	ahead.result_valid = true;
}

void LightClusterer::worker_main(unsigned id)
{
	static const char *const names[] = {"light-worker-0", "light-worker-1", "light-worker-2", "light-worker-3"};
	if (TimelineTrace::get().enabled())
		TimelineTrace::get().set_thread_name(names[id & 3]);
	uint64_t seen = 0;
	for (;;)
	{
		RenderParameters rp;
		{
			std::unique_lock<std::mutex> holder{ahead.lock};
			ahead.wake.wait(holder, [&]() { return ahead.generation != seen || ahead.quit; });
			if (ahead.quit)
				return;
			seen = ahead.generation;
			rp = ahead.parameters;
		}
		// scene_lights is only modified after invalidate_prefetch() has seen this job finish
		GRANITE_SCOPED_TIMELINE_EVENT("prefetch-next-frame-lights");
		if (id == 0)
		{
			GRANITE_SCOPED_TIMELINE_EVENT("light-sort");
			sort_lights(rp, ahead.sort_state);
			begin_pack(rp, ahead.sort_state, ahead.result);
			ahead.num_chunks = int((unsigned(ahead.result.parameters.num_lights) + PackChunk - 1) / PackChunk);
			ahead.sorted.store(1, std::memory_order_release);
		}
		else
			while (ahead.sorted.load(std::memory_order_acquire) == 0) // thread 0's sort: a few microseconds
				std::this_thread::yield();
		for (;;)
		{
			const int chunk = ahead.next_chunk.fetch_add(1, std::memory_order_relaxed);
			if (chunk >= ahead.num_chunks)
				break;
			pack_chunk(rp, ahead.sort_state, ahead.result, unsigned(chunk));
		}
		if (ahead.workers_left.fetch_sub(1, std::memory_order_acq_rel) == 1)
		{
			std::unique_lock<std::mutex> holder{ahead.lock};
			ahead.done.notify_all();
		}
	}
}

LightClusterer::~LightClusterer()
{
	{
		std::unique_lock<std::mutex> holder{ahead.lock};
		wait_for_workers(holder);
		ahead.quit = true;
		ahead.wake.notify_all();
	}
	for (auto &t : ahead.threads)
		if (t.joinable())
			t.join();
}

void LightClusterer::sort_lights(const RenderParameters &rp, SortState &sort) const
{
	// Visible positional lights, nearest first along the view direction, so that the per-slice [first, last] index
	// window produced by the z-range kernel is tight.  Same result as a stable sort of the scene list by
	// dot(translation, front) (the reference sorts the list with that comparator), done as: one key per light, then
	// either "last frame's order is still sorted" (static lights + camera: O(n) check) or a stable LSD radix sort.
	auto &sort_keys = sort.sort_keys;
	auto &sort_order = sort.sort_order;
	auto &sort_scratch = sort.sort_scratch;
	const size_t count = scene_lights ? scene_lights->size() : 0;
	const vec3 front = rp.camera_front;
	sort_keys.resize(count);
	for (size_t i = 0; i < count; i++)
	{
		float key = dot((*scene_lights)[i].transform->get_translation(), front);
		uint32_t bits;
		memcpy(&bits, &key, sizeof(bits));
		// order-preserving map of IEEE floats onto unsigned integers (-0 and +0 stay distinct but adjacent)
		sort_keys[i] = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
	}

	bool sorted = sort_order.size() == count;
	for (size_t i = 1; sorted && i < count; i++)
	{
		const uint32_t a = sort_order[i - 1], b = sort_order[i];
		sorted = sort_keys[a] < sort_keys[b] || (sort_keys[a] == sort_keys[b] && a < b);
	}
	if (sorted)
		return;
	sort_order.resize(count);
	sort_scratch.resize(count);
	for (size_t i = 0; i < count; i++)
		sort_order[i] = uint32_t(i);
	// 3 passes of 11 bits, least significant first; each pass is stable, ties keep scene order.
	uint32_t *src = sort_order.data(), *dst = sort_scratch.data();
	for (unsigned shift = 0; shift < 33; shift += 11)
	{
		uint32_t histogram[2048] = {};
		for (size_t i = 0; i < count; i++)
			histogram[(sort_keys[src[i]] >> shift) & 2047u]++;
		uint32_t sum = 0;
		for (auto &h : histogram)
		{
			uint32_t c = h;
			h = sum;
			sum += c;
		}
		for (size_t i = 0; i < count; i++)
			dst[histogram[(sort_keys[src[i]] >> shift) & 2047u]++] = src[i];
		std::swap(src, dst);
	}
	if (src != sort_order.data())
		sort_order.swap(sort_scratch);
}

// Sizes the outputs and fills the cluster parameters; the per-light records follow chunk by chunk (pack_chunk).
void LightClusterer::begin_pack(const RenderParameters &rp, const SortState &sort, PackedLights &out) const
{
	// lights beyond the bindless maximum are dropped
	const size_t count = std::min(sort.sort_order.size(), size_t(MaxLightsBindless));
	memset(out.type_mask, 0, sizeof(out.type_mask));
	out.lights.resize(count);
	out.model.resize(count);
	// The z-range kernel must still run with no lights so that the range buffer is cleared to "empty".
	out.volume_index_range.resize(std::max(count, size_t(1)));
	if (count == 0)
		out.volume_index_range[0] = uvec2(~0u, 0u);

	auto &p = out.parameters;
	p = {};
	p.num_lights = int32_t(count);
	p.num_lights_32 = (p.num_lights + 31) / 32;
	p.clip_scale[0] = rp.projection[0][0];
	p.clip_scale[1] = -rp.projection[1][1];
	p.clip_scale[2] = rp.inv_projection[0][0];
	p.clip_scale[3] = -rp.inv_projection[1][1];

	// translate(0.5, 0.5, 0) * scale(0.5, 0.5, 1) * view_projection, written out per column.
	for (int c = 0; c < 4; c++)
	{
		const vec4 &col = rp.view_projection[c];
		p.transform[4 * c + 0] = 0.5f * col.x + 0.5f * col.w;
		p.transform[4 * c + 1] = 0.5f * col.y + 0.5f * col.w;
		p.transform[4 * c + 2] = col.z;
		p.transform[4 * c + 3] = col.w;
	}
	for (int i = 0; i < 3; i++)
	{
		p.camera_front[i] = rp.camera_front[i];
		p.camera_base[i] = rp.camera_position[i];
	}
	p.xy_scale[0] = float(resolution_x);
	p.xy_scale[1] = float(resolution_y);
	p.resolution_xy[0] = int32_t(resolution_x);
	p.resolution_xy[1] = int32_t(resolution_y);
	p.inv_resolution_xy[0] = 1.0f / float(resolution_x);
	p.inv_resolution_xy[1] = 1.0f / float(resolution_y);
	p.z_scale = 1.0f / get_z_slice_extent(rp);
	p.z_max_index = int32_t(resolution_z) - 1;
}

// Lights [chunk * PackChunk, (chunk + 1) * PackChunk) of the sorted order: shader records, cull volumes, type bits and
// the per-light slice intervals of the z-range kernel (clusterer.cpp:656-698,1265-1275).
void LightClusterer::pack_chunk(const RenderParameters &rp, const SortState &sort, PackedLights &out, unsigned chunk) const
{
	const unsigned begin = chunk * PackChunk;
	const unsigned end = std::min(begin + unsigned(PackChunk), unsigned(out.parameters.num_lights));
	for (unsigned index = begin; index < end; index++)
	{
		const PositionalLightInfo &entry = (*scene_lights)[sort.sort_order[index]];
		vec2 range;
		if (entry.light->get_type() == PositionalLight::Type::Spot)
		{
			auto &spot = static_cast<SpotLight &>(*entry.light);
			out.lights[index] = spot.get_shader_info(*entry.transform);
			out.model[index] = spot.build_model_matrix(*entry.transform);
			range = spot_light_z_range(rp, out.model[index]);
		}
		else
		{
			auto &point = static_cast<PointLight &>(*entry.light);
			const auto info = point.get_shader_info(*entry.transform);
			out.lights[index] = info;
			mat_affine m;
			m[0] = vec4(info.position[0], info.position[1], info.position[2], 1.0f / info.inv_radius);
			m[1] = vec4(0.0f);
			m[2] = vec4(0.0f);
			out.model[index] = m;
			out.type_mask[index >> 5] |= 1u << (index & 31u);
			range = point_light_z_range(rp, vec3(info.position[0], info.position[1], info.position[2]), 1.0f / info.inv_radius);
		}
		out.volume_index_range[index] = compute_uint_range(rp, range);
	}
}

uvec2 LightClusterer::compute_uint_range(const RenderParameters &rp, vec2 range) const
{
	float extent = get_z_slice_extent(rp);
	range = vec2(range.x / extent, range.y / extent);
	if (range.y < 0.0f)
		return uvec2(0xffffffffu, 0u);
	range.x = std::max(range.x, 0.0f);
	uvec2 urange(uint32_t(range.x), uint32_t(range.y));
	urange.y = std::min(urange.y, resolution_z - 1);
	return urange;
}

// ---- GPU build (clusterer.cpp:1178-1207,1277-1346,1463-1573) ----------------------------------------------------------------
void LightClusterer::update_bindless_data(HIP::CommandBuffer &cmd)
{
	// The reference records three cmd.update_buffer here and a fourth (the per-light slice intervals) before the z-range
	// dispatch (clusterer.cpp:1178-1207,1302).  All four are staged now and uploaded by ONE kernel: nothing on the GPU
	// reads any of them before this point, and a copy-engine transfer in the middle of the pass costs more than the pass.
	uint32_t count = uint32_t(packed.parameters.num_lights);
	auto &ranges = packed.volume_index_range;
	const HIP::CommandBuffer::BufferUpdate updates[] = {
		{bindless.transforms_buffer, GR_TRANSFORMS_OFFSET_LIGHTS, count * sizeof(PositionalFragmentInfo), packed.lights.data()},
		{bindless.transforms_buffer, GR_TRANSFORMS_OFFSET_MODEL, count * sizeof(mat_affine), packed.model.data()},
		{bindless.transforms_buffer, GR_TRANSFORMS_OFFSET_TYPE_MASK, packed.parameters.num_lights_32 * sizeof(uint32_t), packed.type_mask},
		{bindless.light_ranges.get(), 0, ranges.size() * sizeof(uvec2), ranges.data()},
	};
	cmd.update_buffers(updates, 4);
}

void LightClusterer::update_bindless_mask_buffer_gpu(HIP::CommandBuffer &cmd)
{
	uint32_t local_count = uint32_t(packed.parameters.num_lights);
	if (local_count == 0)
		return;
	auto &rp = context->get_render_parameters();
	void *transforms = bindless.transforms_buffer->get_device_pointer();
	void *spots = bindless.transformed_spots->get_device_pointer();
	void *cull = bindless.cull_data->get_device_pointer();

	gr_push_spot_transform spot_push = {};
	memcpy(spot_push.vp, rp.view_projection.data(), sizeof(spot_push.vp));
	for (int i = 0; i < 3; i++)
	{
		spot_push.camera_pos[i] = rp.camera_position[i];
		spot_push.camera_front[i] = rp.camera_front[i];
	}
	spot_push.num_lights = local_count;
	spot_push.z_near = rp.z_near;
	spot_push.z_far = rp.z_far;
	cmd.check(gr_cluster_spot_transform(cmd.get_context(), cmd.get_stream(), transforms, spots, &spot_push), "cluster_spot_transform");
	cmd.barrier(VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_ACCESS_2_SHADER_STORAGE_WRITE_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT,
	            VK_ACCESS_2_SHADER_STORAGE_READ_BIT);

	gr_push_cluster_setup setup_push = {};
	memcpy(setup_push.view, rp.view.data(), sizeof(setup_push.view));
	setup_push.num_lights = local_count;
	cmd.check(gr_cluster_setup(cmd.get_context(), cmd.get_stream(), transforms, spots, cull, &packed.parameters, &setup_push), "cluster_setup");
	cmd.barrier(VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_ACCESS_2_SHADER_STORAGE_WRITE_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT,
	            VK_ACCESS_2_SHADER_STORAGE_READ_BIT);

	if ((resolution_x & 7) != 0 || (resolution_y & 7) != 0)
		throw std::logic_error("Cluster resolution must be a multiple of 8 in X and Y.");
	// MI355X executes wave64 only: this is the SUBGROUPS path with an 8x8 cell tile per wave (clusterer.cpp:1546-1552).
	cmd.check(gr_cluster_binning(cmd.get_context(), cmd.get_stream(), transforms, cull,
	                             static_cast<uint32_t *>(bindless.bitmask_buffer->get_device_pointer()), &packed.parameters),
	          "cluster_binning");
}

void LightClusterer::update_bindless_range_buffer_gpu(HIP::CommandBuffer &cmd)
{
	if ((resolution_z & 63) != 0)
		throw std::logic_error("Cluster Z resolution must be a multiple of 64.");

	auto &ranges = packed.volume_index_range; // uploaded by update_bindless_data
	gr_push_z_range push = {};
	push.num_volumes = uint32_t(ranges.size());
	push.num_volumes_128 = (push.num_volumes + 127) / 128;
	push.num_ranges = resolution_z;
	cmd.check(gr_cluster_z_range(cmd.get_context(), cmd.get_stream(), static_cast<const uint32_t *>(bindless.light_ranges->get_device_pointer()),
	                             static_cast<uint32_t *>(bindless.range_buffer->get_device_pointer()), &push),
	          "cluster_z_range");
}

void LightClusterer::build_cluster_bindless_gpu(HIP::CommandBuffer &cmd)
{
	uint32_t local_count = uint32_t(packed.parameters.num_lights);
	static const bool separate = getenv("GRANITE_CLUSTER_SEPARATE_LAUNCHES") != nullptr; // A/B: the reference's dispatch-by-dispatch form
	if (separate || local_count == 0 || packed.volume_index_range.empty())
	{
		update_bindless_data(cmd); // reads this frame's slot of the pinned staging ring: launched directly
		cmd.barrier(VK_PIPELINE_STAGE_2_COPY_BIT, VK_ACCESS_TRANSFER_WRITE_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_ACCESS_2_SHADER_STORAGE_READ_BIT);
		update_bindless_mask_buffer_gpu(cmd);
		update_bindless_range_buffer_gpu(cmd);
		return;
	}
	// Three launches instead of the reference's four transfers + four dispatches (clusterer.cpp:1178-1207,1277-1346,1463-1562): one
	// upload kernel for the four CPU-packed arrays, then spot_transform, setup and z_range -- which depend on those arrays only --
	// as one grid (gr_cluster_front), then binning, which needs every light's cull data.  (gr_cluster_front can also read the
	// arrays from the pinned staging ring itself and save the upload launch; measured, that form is slower: each of its 256
	// z-range workgroups then fetches the 32 KB of slice intervals across PCIe.)
	// Up to 512 lights (4 KB of slice intervals, 48 KB of records) gr_cluster_front reads the staged arrays itself and the upload launch is
	// saved: such frames are paced by the host's runtime calls (config 2: eight launches a frame), not by what crosses PCIe.
	static const uint32_t read_staged_below = []() {
		const char *env = getenv("GRANITE_CLUSTER_READ_STAGED_MAX_LIGHTS"); // A/B switch; 0 = always upload first (round 5)
		return env ? uint32_t(atoi(env)) : 512u;
	}();
	const bool read_staged = local_count <= read_staged_below;
	if (!read_staged)
	{
		update_bindless_data(cmd);
		cmd.barrier(VK_PIPELINE_STAGE_2_COPY_BIT, VK_ACCESS_TRANSFER_WRITE_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_ACCESS_2_SHADER_STORAGE_READ_BIT);
	}
	if ((resolution_z & 63) != 0)
		throw std::logic_error("Cluster Z resolution must be a multiple of 64.");
	if ((resolution_x & 7) != 0 || (resolution_y & 7) != 0)
		throw std::logic_error("Cluster resolution must be a multiple of 8 in X and Y.");
	auto &rp = context->get_render_parameters();
	auto &ranges = packed.volume_index_range;
	gr_push_spot_transform spot_push = {};
	memcpy(spot_push.vp, rp.view_projection.data(), sizeof(spot_push.vp));
	for (int i = 0; i < 3; i++)
	{
		spot_push.camera_pos[i] = rp.camera_position[i];
		spot_push.camera_front[i] = rp.camera_front[i];
	}
	spot_push.num_lights = local_count;
	spot_push.z_near = rp.z_near;
	spot_push.z_far = rp.z_far;
	gr_push_cluster_setup setup_push = {};
	memcpy(setup_push.view, rp.view.data(), sizeof(setup_push.view));
	setup_push.num_lights = local_count;
	gr_push_z_range z_push = {};
	z_push.num_volumes = uint32_t(ranges.size());
	z_push.num_volumes_128 = (z_push.num_volumes + 127) / 128;
	z_push.num_ranges = resolution_z;

	gr_cluster_front_args front = {};
	front.transforms = bindless.transforms_buffer->get_device_pointer();
	front.transformed_spots = bindless.transformed_spots->get_device_pointer();
	front.cull_setup = bindless.cull_data->get_device_pointer();
	front.params = &packed.parameters;
	front.spot_push = &spot_push;
	front.setup_push = &setup_push;
	front.light_ranges = static_cast<uint32_t *>(bindless.light_ranges->get_device_pointer());
	front.range_out = static_cast<uint32_t *>(bindless.range_buffer->get_device_pointer());
	front.z_push = &z_push;
	if (read_staged)
	{
		front.src_lights = cmd.stage(packed.lights.data(), local_count * sizeof(PositionalFragmentInfo));
		front.src_models = cmd.stage(packed.model.data(), local_count * sizeof(mat_affine));
		front.src_type_mask = cmd.stage(packed.type_mask, packed.parameters.num_lights_32 * sizeof(uint32_t));
		front.src_ranges = cmd.stage(ranges.data(), ranges.size() * sizeof(uvec2));
	}
	cmd.check(gr_cluster_front(cmd.get_context(), cmd.get_stream(), &front), "cluster_front");
	cmd.barrier(VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_ACCESS_2_SHADER_STORAGE_WRITE_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT,
	            VK_ACCESS_2_SHADER_STORAGE_READ_BIT);
	cmd.check(gr_cluster_binning(cmd.get_context(), cmd.get_stream(), front.transforms, front.cull_setup,
	                             static_cast<uint32_t *>(bindless.bitmask_buffer->get_device_pointer()), &packed.parameters),
	          "cluster_binning");
}
} // namespace Granite
