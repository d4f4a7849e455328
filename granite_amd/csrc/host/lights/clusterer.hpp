// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// LightClusterer — the bindless clustered-light path of renderer/lights/clusterer.{hpp,cpp} on the HIP executor.
// Shadows, decals, volumetric diffuse / fog and the legacy (non-bindless) clusterer need scene geometry and are out of
// scope; the class keeps the RenderPassCreator / PerFrameRefreshable surface used by the hot path.
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "../render_graph.hpp"
#include "../render_context.hpp"
#include "lights.hpp"

namespace Granite
{
using ClustererParametersBindless = gr_cluster_params; // math/render_parameters.hpp:90-108
static_assert(sizeof(ClustererParametersBindless) == 176, "ClustererParametersBindless must match the std140 UBO.");

struct RenderPassCreator
{
	enum DependencyBits
	{
		GEOMETRY_BIT = 1 << 0,
		MATERIAL_BIT = 1 << 1,
		LIGHTING_BIT = 1 << 2
	};
	using DependencyFlags = uint32_t;
	virtual ~RenderPassCreator() = default;
	virtual void add_render_passes(RenderGraph &graph) = 0;
	virtual void set_base_render_context(const RenderContext *context) = 0;
	virtual void setup_render_pass_dependencies(RenderGraph &graph, RenderPass &target, DependencyFlags dep_flags) = 0;
	virtual void setup_render_pass_dependencies(RenderGraph &graph) = 0;
	virtual void setup_render_pass_resources(RenderGraph &graph) = 0;
};

struct PerFrameRefreshable
{
	virtual ~PerFrameRefreshable() = default;
	virtual void refresh(const RenderContext &context, TaskComposer &composer) = 0;
};

// One visible positional light: what Scene::gather_visible_positional_lights hands the clusterer
// (renderer/scene.hpp PositionalLightInfo {light, transform}).
struct PositionalLightInfo
{
	PositionalLight *light;
	const mat_affine *transform;
};
using PositionalLightList = std::vector<PositionalLightInfo>;

class LightClusterer : public RenderPassCreator, public PerFrameRefreshable
{
public:
	enum { MaxLightsBindless = GR_MAX_LIGHTS_BINDLESS, MaxLightsGlobal = 32 };

	void set_resolution(unsigned x, unsigned y, unsigned z)
	{
		resolution_x = x;
		resolution_y = y;
		resolution_z = z;
	}
	// Stand-in for set_scene(): the visible-light list the per-frame refresh gathers from.
	void set_scene_lights(const PositionalLightList *lights_)
	{
		invalidate_prefetch();
		scene_lights = lights_;
	}
	~LightClusterer() override;

	// The reference runs its per-frame refreshes as TaskComposer tasks on worker threads, beside command recording
	// (threaded_scene.cpp:141-150, clusterer.cpp:1133-1176).  Here the sort + pack of the NEXT frame's lights runs on one
	// helper thread while this frame is being enqueued: prefetch() is called with the render parameters the next frame is
	// going to use, refresh() adopts the result if those are what it is then given (and the light list has not been touched),
	// and packs synchronously otherwise.  The work per frame is the same; it is off the submitting thread.
	void prefetch(const RenderParameters &next_parameters);
	// Call before the scene's lights or their transforms are modified.
	void invalidate_prefetch();
	uint64_t get_prefetch_hits() const { return prefetch_hits; }

	void add_render_passes(RenderGraph &graph) override;
	void set_base_render_context(const RenderContext *context_) override { context = context_; }
	void setup_render_pass_dependencies(RenderGraph &graph, RenderPass &target, DependencyFlags dep_flags) override;
	void setup_render_pass_dependencies(RenderGraph &) override {}
	void setup_render_pass_resources(RenderGraph &graph) override;
	void refresh(const RenderContext &context, TaskComposer &composer) override;

	const ClustererParametersBindless &get_cluster_parameters_bindless() const { return packed.parameters; }
	const HIP::Buffer *get_cluster_transform_buffer() const { return bindless.transforms_buffer; }
	const HIP::Buffer *get_cluster_bitmask_buffer() const { return bindless.bitmask_buffer; }
	const HIP::Buffer *get_cluster_range_buffer() const { return bindless.range_buffer; }
	bool clusterer_has_volumetric_diffuse() const { return false; }

	// Introspection for the parity tests: CPU-side packed state of the last refresh.
	const std::vector<PositionalFragmentInfo> &get_packed_lights() const { return packed.lights; }
	const std::vector<mat_affine> &get_packed_models() const { return packed.model; }
	const uint32_t *get_type_mask() const { return packed.type_mask; }
	const std::vector<uvec2> &get_volume_index_range() const { return packed.volume_index_range; }

private:
	const RenderContext *context = nullptr;
	const PositionalLightList *scene_lights = nullptr;
	unsigned resolution_x = 64, resolution_y = 32, resolution_z = 16; // clusterer.hpp:127; the viewer sets 128x64x4096

	// CPU-side result of one refresh: what the cluster pass uploads and what the lighting pass is parameterised with.
	struct PackedLights
	{
		ClustererParametersBindless parameters = {};
		std::vector<PositionalFragmentInfo> lights;
		std::vector<mat_affine> model;
		uint32_t type_mask[MaxLightsBindless / 32] = {};
		std::vector<uvec2> volume_index_range;
	};
	// Scratch of the depth sort; the previous order is kept to recognise an unchanged one.
	struct SortState
	{
		std::vector<uint32_t> sort_keys, sort_order, sort_scratch;
	};
	PackedLights packed;
	SortState sort_state;

	struct
	{
		const HIP::Buffer *bitmask_buffer = nullptr;
		const HIP::Buffer *range_buffer = nullptr;
		const HIP::Buffer *transforms_buffer = nullptr;
		const HIP::Buffer *transformed_spots = nullptr;
		const HIP::Buffer *cull_data = nullptr;
		HIP::BufferHandle light_ranges; // uvec2[MaxLightsBindless] upload target for the z-range kernel
	} bindless;

	// One-frame-ahead refresh on helper threads: thread 0 sorts, then all of them pack chunks of PackChunk lights.
	enum { PackChunk = 256 }; // a multiple of 32: a chunk owns whole words of the type mask
	enum { PrefetchMinLights = 1024 }; // below this the hand-over to the helper threads costs more than the packing (prefetch())
	struct
	{
		std::vector<std::thread> threads;
		std::mutex lock;
		std::condition_variable wake, done;
		uint64_t generation = 0; // bumped (under the lock) for every job
		bool quit = false;
		std::atomic<int> sorted{0}, next_chunk{0}, workers_left{0};
		int num_chunks = 0;
		bool in_flight = false;      // a job has been posted and not yet collected
		RenderParameters parameters; // what the job was (or is being) computed for
		PackedLights result;
		SortState sort_state;
		bool result_valid = false;
	} ahead;
	uint64_t prefetch_hits = 0;
	void worker_main(unsigned id);
	void wait_for_workers(std::unique_lock<std::mutex> &holder);

	void sort_lights(const RenderParameters &rp, SortState &sort) const;
	void begin_pack(const RenderParameters &rp, const SortState &sort, PackedLights &out) const;
	void pack_chunk(const RenderParameters &rp, const SortState &sort, PackedLights &out, unsigned chunk) const;
	void sort_and_pack(const RenderParameters &rp, SortState &sort, PackedLights &out) const;
	float get_z_slice_extent(const RenderParameters &rp) const;
	uvec2 compute_uint_range(const RenderParameters &rp, vec2 range) const;
	bool bindless_light_is_point(unsigned index) const { return (packed.type_mask[index >> 5] & (1u << (index & 31))) != 0; }
	void build_cluster_bindless_gpu(HIP::CommandBuffer &cmd);
	void update_bindless_data(HIP::CommandBuffer &cmd);
	void update_bindless_mask_buffer_gpu(HIP::CommandBuffer &cmd);
	void update_bindless_range_buffer_gpu(HIP::CommandBuffer &cmd);
};
} // namespace Granite
