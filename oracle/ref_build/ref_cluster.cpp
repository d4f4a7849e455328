// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own cluster-build compute shaders on the CPU (lights/clusterer_bindless_{spot_transform,setup,binning,
// z_range,z_range_opt}.comp, re-spelled into gen/ at build time): push constants and buffers as LightClusterer's
// update_bindless_mask_buffer_gpu / update_bindless_range_buffer_gpu set them (renderer/lights/clusterer.cpp:1277-1346,
// 1463-1562).  Workgroups that communicate (binning: shared memory + barrier in the plain form, subgroupBallot in the
// SUBGROUPS form) run as teams of real threads, one per invocation; a subgroup is one team.
#include <atomic>
#include <barrier>
#include <thread>
#include <vector>
#include "glsl_cpu.hpp"

using namespace glsl;

namespace
{
std::barrier<> *team_barrier = nullptr;
uint team_ballot_bits[4];
uvec4 team_exchange[128]; // one slot per invocation of the workgroup (subgroupShuffleXor)
}
static inline void barrier()
{
	if (team_barrier)
		team_barrier->arrive_and_wait();
}
static inline void memoryBarrierShared() {}
// subgroupBallot over the team: every lane contributes its bit, then all read the same mask.
static inline uvec4 subgroupBallot(bool value)
{
	const uint lane = gl_SubgroupInvocationID;
	if (lane == 0)
		for (auto &w : team_ballot_bits)
			w = 0;
	barrier();
	if (value)
		__atomic_fetch_or(&team_ballot_bits[lane >> 5], 1u << (lane & 31u), __ATOMIC_SEQ_CST);
	barrier();
	const uvec4 result(team_ballot_bits[0], team_ballot_bits[1], team_ballot_bits[2], team_ballot_bits[3]);
	barrier();
	return result;
}

// subgroupShuffleXor: every invocation of the workgroup publishes its value, then reads its partner's -- the lane of ITS OWN
// subgroup whose index differs by `mask` (GL_KHR_shader_subgroup_shuffle).  Called in uniform control flow by the shader.
static inline uvec4 subgroupShuffleXor(const uvec4 &value, uint mask)
{
	const uint self = gl_LocalInvocationIndex;
	team_exchange[self] = value;
	barrier();
	const uvec4 result = team_exchange[gl_SubgroupID * gl_SubgroupSize + ((gl_SubgroupInvocationID ^ mask) % gl_SubgroupSize)];
	barrier();
	return result;
}

namespace spot_transform
{
#include "gen/clusterer_bindless_spot_transform.inc"
}
namespace setup
{
#include "gen/clusterer_bindless_setup.inc"
}
#define SUBGROUPS 0
namespace binning_plain
{
#include "gen/clusterer_bindless_binning.inc"
}
#undef SUBGROUPS
#define SUBGROUPS 1
namespace binning_subgroups
{
#include "gen/clusterer_bindless_binning.inc"
}
#undef SUBGROUPS
namespace z_range
{
#include "gen/clusterer_bindless_z_range.inc"
}

// The shader the reference actually dispatches on subgroup-capable devices (clusterer.cpp:1305-1314): 128-thread workgroups,
// 128 x 128 bit-matrix transpose through shared memory and subgroupShuffleXor.
namespace z_range_opt
{
static constexpr struct
{
	unsigned x = 128, y = 1, z = 1;
} gl_WorkGroupSize; // layout(local_size_x = 128) in;
#include "gen/clusterer_bindless_z_range_opt.inc"
}

namespace
{
struct LightInfo
{
	float color[3];
	uint32_t spot_scale_bias;
	float position[3];
	uint32_t offset_radius;
	float direction[3];
	float inv_radius;
};
struct ClusterParams
{
	float transform[16];
	float clip_scale[4];
	float camera_base[3], pad0;
	float camera_front[3], pad1;
	float xy_scale[2];
	int32_t resolution_xy[2];
	float inv_resolution_xy[2];
	int32_t num_lights, num_lights_32, num_decals, num_decals_32, decals_texture_offset, z_max_index;
	float z_scale;
	float pad2[3];
};
struct RenderParams
{
	float projection[16], view[16], view_projection[16], inv_projection[16], inv_view[16], inv_view_projection[16];
	float camera_position[3], camera_front[3];
	float z_near, z_far;
};

mat4 load_mat4(const float *m)
{
	mat4 r;
	for (int c = 0; c < 4; c++)
		r.c[c] = vec4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
	return r;
}
vec3 ld3(const float *v) { return vec3(v[0], v[1], v[2]); }

template <typename Params>
void load_cluster_params(Params &dst, const ClusterParams &cl)
{
	dst.transform = load_mat4(cl.transform);
	dst.clip_scale = vec4(cl.clip_scale[0], cl.clip_scale[1], cl.clip_scale[2], cl.clip_scale[3]);
	dst.camera_base = ld3(cl.camera_base);
	dst.camera_front = ld3(cl.camera_front);
	dst.xy_scale = vec2(cl.xy_scale[0], cl.xy_scale[1]);
	dst.resolution_xy = ivec2(cl.resolution_xy[0], cl.resolution_xy[1]);
	dst.inv_resolution_xy = vec2(cl.inv_resolution_xy[0], cl.inv_resolution_xy[1]);
	dst.num_lights = cl.num_lights;
	dst.num_lights_32 = cl.num_lights_32;
	dst.num_decals = cl.num_decals;
	dst.num_decals_32 = cl.num_decals_32;
	dst.decals_texture_offset = cl.decals_texture_offset;
	dst.z_max_index = cl.z_max_index;
	dst.z_scale = cl.z_scale;
}

template <typename Transforms>
void load_lights(Transforms &dst, const LightInfo *lights, const float *model12, const uint32_t *type_mask, int num_lights)
{
	for (int i = 0; i < num_lights; i++)
	{
		if (lights)
		{
			auto &l = dst.lights[i];
			l.color = ld3(lights[i].color);
			l.spot_scale_bias = lights[i].spot_scale_bias;
			l.position = ld3(lights[i].position);
			l.offset_radius = lights[i].offset_radius;
			l.direction = ld3(lights[i].direction);
			l.inv_radius = lights[i].inv_radius;
		}
		if (model12)
			for (int r = 0; r < 3; r++)
				dst.model[i].rows[r] = vec4(model12[12 * i + 4 * r], model12[12 * i + 4 * r + 1], model12[12 * i + 4 * r + 2], model12[12 * i + 4 * r + 3]);
	}
	if (type_mask)
		for (int i = 0; i < 128; i++)
			dst.type_mask[i] = type_mask[i];
}

// One workgroup = `size` threads running main() with barriers between them.
template <typename Main>
void run_team(unsigned size, const uvec3 &workgroup, unsigned subgroup_size, Main main_fn)
{
	std::barrier<> sync(size);
	team_barrier = &sync;
	std::vector<std::thread> threads;
	for (unsigned i = 0; i < size; i++)
		threads.emplace_back([=]() {
			gl_WorkGroupID = workgroup;
			gl_LocalInvocationID = uvec3(i, 0u, 0u);
			gl_LocalInvocationIndex = i;
			gl_GlobalInvocationID = uvec3(workgroup.x * size + i, workgroup.y, workgroup.z);
			gl_SubgroupSize = subgroup_size;
			gl_NumSubgroups = size / subgroup_size;
			gl_SubgroupID = i / subgroup_size;
			gl_SubgroupInvocationID = i % subgroup_size;
			main_fn();
		});
	for (auto &t : threads)
		t.join();
	team_barrier = nullptr;
}
} // namespace

extern "C" {

// out: 24 floats per light (TransformedSpot: clip[5], z).
void ref_cluster_spot_transform(const RenderParams *rp, const float *model12, int num_lights, float *out)
{
	namespace s = spot_transform;
	load_lights(s::cluster_transforms, nullptr, model12, nullptr, num_lights);
	s::registers.vp = load_mat4(rp->view_projection);
	s::registers.camera_pos = ld3(rp->camera_position);
	s::registers.num_lights = uint(num_lights);
	s::registers.camera_front = ld3(rp->camera_front);
	s::registers.z_near = rp->z_near;
	s::registers.z_far = rp->z_far;
	static_assert(sizeof(s::TransformedSpot) == 96, "TransformedSpot layout");
	s::transformed.spots = reinterpret_cast<s::TransformedSpot *>(out);
	for (int i = 0; i < ((num_lights + 63) & ~63); i++)
	{
		gl_GlobalInvocationID = uvec3(uint(i), 0u, 0u);
		s::main();
	}
}

// cull_setup: 128 floats per light, zero-initialised by the caller (the graph zero-fills buffers).
void ref_cluster_setup(const RenderParams *rp, const ClusterParams *prm, const LightInfo *lights, const uint32_t *type_mask,
                       const float *transformed_spots, int num_lights, float *cull_setup)
{
	namespace s = setup;
	load_lights(s::cluster_transforms, lights, nullptr, type_mask, num_lights);
	load_cluster_params(s::parameters, *prm);
	s::registers.view = load_mat4(rp->view);
	s::registers.num_lights = uint(num_lights);
	static_assert(sizeof(s::CullSetup) == 512, "CullSetup layout");
	s::transformed.spots = reinterpret_cast<s::TransformedSpot *>(const_cast<float *>(transformed_spots));
	s::culling_setup.data = reinterpret_cast<s::CullSetup *>(cull_setup);
	for (int i = 0; i < ((num_lights + 63) & ~63); i++)
	{
		gl_GlobalInvocationID = uvec3(uint(i), 0u, 0u);
		s::main();
	}
}

// subgroup_size 0: the plain form (32 threads per cell and chunk, shared mask); otherwise the SUBGROUPS form with one
// subgroup of that many lanes per workgroup (tile 8 x subgroup_size / 8 cells), dispatched as clusterer.cpp:1533-1556 does.
void ref_cluster_binning(const ClusterParams *prm, const uint32_t *type_mask, const float *cull_setup, uint32_t *bitmask, int subgroup_size)
{
	const int res_x = prm->resolution_xy[0], res_y = prm->resolution_xy[1], n32 = prm->num_lights_32;
	if (subgroup_size == 0)
	{
		namespace s = binning_plain;
		load_lights(s::cluster_transforms, nullptr, nullptr, type_mask, 0);
		load_cluster_params(s::parameters, *prm);
		s::culling_setup.data = reinterpret_cast<s::CullSetup *>(const_cast<float *>(cull_setup));
		s::bitmask = bitmask;
		for (int y = 0; y < res_y; y++)
			for (int x = 0; x < res_x; x++)
				for (int chunk = 0; chunk < n32; chunk++)
					run_team(32, uvec3(uint(chunk), uint(x), uint(y)), 32, s::main);
	}
	else
	{
		namespace s = binning_subgroups;
		load_lights(s::cluster_transforms, nullptr, nullptr, type_mask, 0);
		load_cluster_params(s::parameters, *prm);
		s::culling_setup.data = reinterpret_cast<s::CullSetup *>(const_cast<float *>(cull_setup));
		s::bitmask = bitmask;
		const int tile_h = subgroup_size / 8;
		for (int ty = 0; ty < res_y / tile_h; ty++)
			for (int tx = 0; tx < res_x / 8; tx++)
				for (int chunk = 0; chunk < n32; chunk++)
					run_team(unsigned(subgroup_size), uvec3(uint(chunk), uint(tx), uint(ty)), unsigned(subgroup_size), s::main);
	}
}

void ref_cluster_z_range(const uint32_t *light_ranges, int num_lights, int num_ranges, uint32_t *out)
{
	namespace s = z_range;
	s::registers.num_lights = uint(num_lights);
	s::z_ranges = reinterpret_cast<uvec2 *>(const_cast<uint32_t *>(light_ranges));
	s::light_ranges = reinterpret_cast<uvec2 *>(out);
	for (int z = 0; z < num_ranges; z++)
	{
		gl_GlobalInvocationID = uvec3(uint(z), 0u, 0u);
		s::main();
	}
}

// clusterer_bindless_z_range_opt.comp as update_bindless_range_buffer_gpu dispatches it (clusterer.cpp:1277-1346): push constants
// {num_lights, num_lights_128 = ceil(num_lights / 128), num_ranges}, (num_ranges + 127) / 128 workgroups of 128 invocations, run as
// teams of real threads in subgroups of `subgroup_size` lanes (32, 64 or 128: the sizes the shader's shuffle network is valid for).
void ref_cluster_z_range_opt(const uint32_t *light_ranges, int num_lights, int num_ranges, uint32_t *out, int subgroup_size)
{
	namespace s = z_range_opt;
	s::num_lights = num_lights;
	s::num_lights_128 = (num_lights + 127) / 128;
	s::num_ranges = uint(num_ranges);
	s::z_ranges = reinterpret_cast<uvec2 *>(const_cast<uint32_t *>(light_ranges));
	s::light_ranges = reinterpret_cast<uvec2 *>(out);
	for (int group = 0; group < (num_ranges + 127) / 128; group++)
		run_team(128, uvec3(uint(group), 0u, 0u), unsigned(subgroup_size), s::main);
}

} // extern "C"
