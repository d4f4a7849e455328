// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of Granite's screen-space reflection pass: renderer/post/ssr.cpp:84-323 and
// assets/shaders/post/ffx-sssr/{classify,build_indirect,trace_primary}.comp, apply.frag, sssr_util.h,
// inc/project_direction.h.  The arithmetic inside is AMD FidelityFX SSSR (hierarchical depth-buffer traversal, GGX VNDF
// sampling by Heitz) as the reference vendors it; every function below follows the vendored GLSL statement by statement,
// fp32, uncontracted.
//
// What the GPU leaves open and this file fixes (the HIP kernels make the same choices):
//   * ray order.  classify.comp appends rays with atomicAdd, so the order of the list -- and with it which 64 rays share a
//     wave in trace_primary.comp -- is whatever the hardware did.  Here: tiles in row-major order, inside a tile the
//     shader's Z-order lane index; one valid outcome of the reference.
//   * wave-level early exit.  The traversal leaves a ray when no more than 4 lanes of its wave are still marching
//     (subgroupBallotBitCount, trace_primary.comp:164-166): emulated with 64 consecutive rays in lockstep.
//   * texelFetch outside a mip level returns 0 (robust image access).
//   * copy conflicts.  On a denoised (non-mirror) surface each 2 x 2 quad shoots two rays (the checkerboard phase of
//     is_base_ray) and both copy their result into BOTH other pixels of the quad: one horizontally, one vertically.  The shader
//     stores the horizontal copies first and the vertical ones after them (trace_primary.comp:284-298), so with the two rays
//     in one wave the vertical copy is what stays; across waves it is a race.  Here the vertical copy always wins: a ray
//     skips its horizontal copy when its diagonal neighbour is a listed ray (which then copies vertically into that pixel).
//   * `confidence` of a ray whose depth is exactly 1.0 is an uninitialised `out` in the shader; 0 here (unreachable:
//     classify.comp only lists pixels with depth < 1).
// trace_fallback.comp only runs with a volumetric-diffuse probe set bound (ssr.cpp:141), which is outside this path.
//
// PINNED: oracle/ref_build/ref_ssr.cpp executes the reference's own four shaders (re-spelled at build time, 64 real threads
// per workgroup) under the same four statements; tests/test_reference_shaders_cpu.py::test_sssr_shaders_bit_for_bit requires
// ray list, counters, traced colour, ray length, confidence and the blended target to be identical, bit for bit.
#include "oracle_common.h"
#include <vector>

using namespace orc;

extern "C" {
struct OrcSSRArgs
{
	int32_t width, height;
	const float *hier;   // depth hierarchy, R32F, levels packed back to back (level l: max(w0 >> l, 1) x max(h0 >> l, 1))
	int32_t hier_w, hier_h, hier_levels;
	const uint16_t *pbr;    // R8G8_UNORM
	const uint32_t *normal; // A2B10G10R10_UNORM
	const uint16_t *light;  // RGBA16F
	const uint16_t *noise;  // R8G8_UNORM, 128 x 128 x 64 layers
	int32_t frame;
	const float *view_projection;     // column-major
	const float *inv_view_projection; // column-major
	float camera_position[3];
	// outputs
	uint16_t *output;     // RGBA16F
	uint16_t *ray_length; // R16F
	uint8_t *confidence;  // R8_UNORM
	uint32_t *ray_list;   // width * height entries
	uint32_t *ray_counter; // 6 dwords: indirect.xyzw, atomic_count, copied_count
};
}

namespace
{
struct Ctx
{
	const OrcSSRArgs *a;
	mat4 vp, inv_vp;
	vec2 float_resolution, inv_resolution;
	vec3 camera;
};

mat4 load_mat(const float *m)
{
	mat4 r;
	for (int c = 0; c < 4; c++)
		r.c[c] = V4(m[4 * c + 0], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
	return r;
}

size_t level_offset(int w, int h, int level)
{
	size_t o = 0;
	for (int l = 0; l < level; l++)
		o += size_t(std::max(w >> l, 1)) * size_t(std::max(h >> l, 1));
	return o;
}

// FFX_SSSR_LoadDepth: texelFetch(uDepth, coord, lod).x; 0 outside the level.
float load_depth(const Ctx &c, int x, int y, int lod)
{
	const auto *a = c.a;
	if (lod < 0 || lod >= a->hier_levels)
		return 0.0f;
	const int w = std::max(a->hier_w >> lod, 1), h = std::max(a->hier_h >> lod, 1);
	if (x < 0 || y < 0 || x >= w || y >= h)
		return 0.0f;
	return a->hier[level_offset(a->hier_w, a->hier_h, lod) + size_t(y) * w + x];
}

vec3 load_normal(const Ctx &c, int x, int y) // FFX_SSSR_LoadWorldSpaceNormal: texel.xyz * 2 - 1
{
	if (x < 0 || y < 0 || x >= c.a->width || y >= c.a->height)
		return V3(-1.0f);
	const uint32_t v = c.a->normal[size_t(y) * c.a->width + x];
	const vec3 n = V3(float(v & 1023u) / 1023.0f, float((v >> 10) & 1023u) / 1023.0f, float((v >> 20) & 1023u) / 1023.0f);
	return n * 2.0f - V3(1.0f);
}

vec2 load_pbr(const Ctx &c, int x, int y)
{
	if (x < 0 || y < 0 || x >= c.a->width || y >= c.a->height)
		return V2(0.0f, 0.0f);
	const uint16_t v = c.a->pbr[size_t(y) * c.a->width + x];
	return V2(float(v & 255u) / 255.0f, float(v >> 8) / 255.0f);
}

vec3 load_light(const Ctx &c, int x, int y)
{
	if (x < 0 || y < 0 || x >= c.a->width || y >= c.a->height)
		return V3(0.0f);
	const vec4 t = load_rgba16f(c.a->light, c.a->width, x, y);
	return V3(t.x, t.y, t.z);
}

vec3 screen_to_world(const Ctx &c, vec3 ndc) // FFX_SSSR_ScreenSpaceToWorldSpace
{
	const vec4 w = mul(c.inv_vp, V4(ndc, 1.0f));
	return V3(w.x, w.y, w.z) / w.w;
}

vec3 cross3(vec3 a, vec3 b) { return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
vec3 reflect3(vec3 i, vec3 n) { return i - n * (2.0f * dot(n, i)); } // GLSL reflect: I - 2 dot(N, I) N

// ---- sssr_util.h:55-143 ---------------------------------------------------------------------------------------------------
const float M_PI_SIC = 3.1415628f;

vec3 sample_ggx_vndf(vec3 Ve, float alpha_x, float alpha_y, float U1, float U2)
{
	const vec3 Vh = normalize(V3(alpha_x * Ve.x, alpha_y * Ve.y, Ve.z));
	const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
	const vec3 T1 = lensq > 0.0f ? V3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : V3(1.0f, 0.0f, 0.0f);
	const vec3 T2 = cross3(Vh, T1);
	const float r = sqrtf(U1);
	const float phi = 2.0f * M_PI_SIC * U2;
	const float t1 = r * cosf(phi);
	float t2 = r * sinf(phi);
	const float s = 0.5f * (1.0f + Vh.z);
	t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
	const vec3 Nh = t1 * T1 + t2 * T2 + sqrtf(std::max(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
	return normalize(V3(alpha_x * Nh.x, alpha_y * Nh.y, std::max(0.0f, Nh.z)));
}

struct mat3 { vec3 c[3]; };
mat3 create_tbn(vec3 N)
{
	vec3 U;
	if (fabsf(N.z) > 0.0f)
	{
		const float k = sqrtf(N.y * N.y + N.z * N.z);
		U = V3(0.0f, -N.z / k, N.y / k);
	}
	else
	{
		const float k = sqrtf(N.x * N.x + N.y * N.y);
		U = V3(N.y / k, -N.x / k, 0.0f);
	}
	return {{U, cross3(N, U), N}};
}
vec3 vec_times_mat(vec3 v, const mat3 &m) { return V3(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2])); }
vec3 mat_times_vec(const mat3 &m, vec3 v)
{
	vec3 r = m.c[0] * v.x;
	r = r + m.c[1] * v.y;
	r = r + m.c[2] * v.z;
	return r;
}

vec2 sample_noise(const Ctx &c, int px, int py) // SampleRandomVector2D
{
	const uint16_t v = c.a->noise[(size_t(c.a->frame) * 128 + size_t(py & 127)) * 128 + size_t(px & 127)];
	return V2(float(v & 255u) / 255.0f, float(v >> 8) / 255.0f);
}

vec3 sample_reflection_vector(const Ctx &c, vec3 view_direction, vec3 normal, float roughness, int px, int py)
{
	const mat3 tbn = create_tbn(normal);
	const vec3 view_tbn = vec_times_mat(-view_direction, tbn);
	const vec2 u = sample_noise(c, px, py);
	const vec3 sampled_normal_tbn = sample_ggx_vndf(view_tbn, roughness, roughness, u.x, u.y);
	const vec3 reflected_tbn = reflect3(-view_tbn, sampled_normal_tbn);
	return mat_times_vec(tbn, reflected_tbn);
}

// inc/project_direction.h:45-46
vec3 project_direction_to_clip_space(const Ctx &c, vec3 clip, vec3 world_direction)
{
	const vec4 clip_d = mul(c.vp, V4(world_direction, 0.0f));
	return normalize(V3(clip_d.x, clip_d.y, clip_d.z) - clip * clip_d.w);
}

// ---- trace_primary.comp:54-205: the traversal, 64 rays in lockstep --------------------------------------------------------
const float FLOAT_MAX = 3.402823466e+38f;

struct Ray
{
	// inputs
	vec3 origin, direction;
	bool is_mirror;
	// state
	vec3 inv_direction, position;
	vec2 mip_resolution, mip_resolution_inv, uv_offset, floor_offset;
	float current_t;
	int mip, i;
	bool exit_low, active, listed;
};

void ray_begin(const Ctx &c, Ray &r, int most_detailed_mip)
{
	const vec3 d = r.direction;
	r.inv_direction = V3(d.x != 0.0f ? 1.0f / d.x : FLOAT_MAX, d.y != 0.0f ? 1.0f / d.y : FLOAT_MAX, d.z != 0.0f ? 1.0f / d.z : FLOAT_MAX);
	r.mip = most_detailed_mip;
	r.mip_resolution = c.float_resolution * ldexpf(1.0f, -r.mip);
	r.mip_resolution_inv = V2(1.0f / r.mip_resolution.x, 1.0f / r.mip_resolution.y);
	vec2 uv_offset = c.inv_resolution * (0.005f * exp2f(float(most_detailed_mip)));
	r.uv_offset = V2(d.x < 0.0f ? -uv_offset.x : uv_offset.x, d.y < 0.0f ? -uv_offset.y : uv_offset.y);
	r.floor_offset = V2(d.x < 0.0f ? 0.0f : 1.0f, d.y < 0.0f ? 0.0f : 1.0f);
	// FFX_SSSR_InitialAdvanceRay
	const vec2 mip_position = r.mip_resolution * V2(r.origin.x, r.origin.y);
	vec2 xy_plane = V2(floorf(mip_position.x), floorf(mip_position.y)) + r.floor_offset;
	xy_plane = xy_plane * r.mip_resolution_inv + r.uv_offset;
	const vec2 t = xy_plane * V2(r.inv_direction.x, r.inv_direction.y) - V2(r.origin.x, r.origin.y) * V2(r.inv_direction.x, r.inv_direction.y);
	r.current_t = std::min(t.x, t.y);
	r.position = r.origin + r.current_t * r.direction;
	r.exit_low = false;
	r.i = 0;
}

void ray_step(const Ctx &c, Ray &r, uint32_t active_lanes, uint32_t min_occupancy)
{
	const vec2 mip_position = r.mip_resolution * V2(r.position.x, r.position.y);
	const float surface_z = load_depth(c, int(mip_position.x), int(mip_position.y), r.mip);
	r.exit_low = !r.is_mirror && active_lanes <= min_occupancy;
	// FFX_SSSR_AdvanceRay
	vec2 xy_plane = V2(floorf(mip_position.x), floorf(mip_position.y)) + r.floor_offset;
	xy_plane = xy_plane * r.mip_resolution_inv + r.uv_offset;
	const vec3 boundary = V3(xy_plane.x, xy_plane.y, surface_z);
	vec3 t = boundary * r.inv_direction - r.origin * r.inv_direction;
	t.z = r.direction.z > 0.0f ? t.z : FLOAT_MAX;
	const float t_min = std::min(std::min(t.x, t.y), t.z);
	const bool above_surface = surface_z > r.position.z;
	const bool skipped_tile = f2u(t_min) != f2u(t.z) && above_surface;
	r.current_t = above_surface ? t_min : r.current_t;
	r.position = r.origin + r.current_t * r.direction;
	r.mip += skipped_tile ? 1 : -1;
	r.mip_resolution = r.mip_resolution * (skipped_tile ? 0.5f : 2.0f);
	r.mip_resolution_inv = r.mip_resolution_inv * (skipped_tile ? 2.0f : 0.5f);
	r.i++;
}

float validate_hit(const Ctx &c, vec3 hit, vec2 uv, vec3 world_ray_direction, float thickness)
{
	if (hit.x < 0.0f || hit.y < 0.0f || 1.0f < hit.x || 1.0f < hit.y)
		return 0.0f;
	const vec2 manhattan = V2(fabsf(hit.x - uv.x), fabsf(hit.y - uv.y));
	if (manhattan.x < 2.0f * c.inv_resolution.x && manhattan.y < 2.0f * c.inv_resolution.y)
		return 0.0f;
	const int tx = int(c.float_resolution.x * hit.x), ty = int(c.float_resolution.y * hit.y);
	const float surface_z = load_depth(c, tx / 2, ty / 2, 1);
	if (surface_z == 1.0f)
		return 1.0f;
	const vec3 hit_normal = load_normal(c, tx, ty);
	if (dot(hit_normal, world_ray_direction) > 0.0f)
		return 0.0f;
	const vec3 surface = screen_to_world(c, V3(hit.x, hit.y, surface_z));
	const vec3 hit_world = screen_to_world(c, hit);
	const float dist = length(surface - hit_world);
	const vec2 fov = V2(c.float_resolution.y * c.inv_resolution.x, 1.0f) * 0.05f;
	const vec2 border = V2(smoothstep(0.0f, fov.x, hit.x) * (1.0f - smoothstep(1.0f - fov.x, 1.0f, hit.x)),
	                       smoothstep(0.0f, fov.y, hit.y) * (1.0f - smoothstep(1.0f - fov.y, 1.0f, hit.y)));
	const float vignette = border.x * border.y;
	float confidence = 1.0f - smoothstep(0.0f, thickness, dist);
	confidence *= confidence;
	return vignette * confidence;
}

// classify.comp:33-48 for one pixel: does it shoot a ray?
bool pixel_needs_ray(const Ctx &c, int x, int y)
{
	if (x < 0 || y < 0 || x >= c.a->width || y >= c.a->height)
		return false;
	const float roughness = load_pbr(c, x, y).y;
	bool ray = roughness < 0.2f && load_depth(c, x, y, 0) < 1.0f;
	const bool needs_denoiser = ray && !(roughness < 0.0001f);
	const bool base_ray = ((uint32_t(x) ^ (uint32_t(c.a->frame) & 1u)) & 1u) == (uint32_t(y) & 1u);
	return ray && (!needs_denoiser || base_ray);
}

uint32_t pack_ray(uint32_t x, uint32_t y, bool ch, bool cv, bool cd)
{
	return x | (y << 14u) | (uint32_t(ch) << 28u) | (uint32_t(cv) << 29u) | (uint32_t(cd) << 30u);
}

void unpack_z_order(uint32_t l, uint32_t &x, uint32_t &y)
{
	x = ((l >> 0) & 1u) | (((l >> 2) & 1u) << 1) | (((l >> 4) & 1u) << 2);
	y = ((l >> 1) & 1u) | (((l >> 3) & 1u) << 1) | (((l >> 5) & 1u) << 2);
}

void store_result(const OrcSSRArgs *a, int x, int y, vec3 color, float length_value, float confidence)
{
	if (x < 0 || y < 0 || x >= a->width || y >= a->height)
		return; // imageStore outside the image is dropped
	store_rgba16f(a->output, a->width, x, y, V4(color, 0.0f));
	a->ray_length[size_t(y) * a->width + x] = float_to_half_rne(length_value);
	a->confidence[size_t(y) * a->width + x] = float_to_unorm8(confidence);
}
} // namespace

extern "C" {

// classify.comp + build_indirect.comp: clears the output / confidence images, fills the ray list (deterministic order) and the
// counter buffer as build_indirect leaves it.
void orc_ssr_classify(const OrcSSRArgs *a)
{
	Ctx c{a, {}, {}, V2(float(a->width), float(a->height)), V2(1.0f / float(a->width), 1.0f / float(a->height)), V3(0.0f)};
	uint32_t count = 0;
	const int tiles_x = (a->width + 7) / 8, tiles_y = (a->height + 7) / 8;
	for (int ty = 0; ty < tiles_y; ty++)
		for (int tx = 0; tx < tiles_x; tx++)
		{
			bool needs_ray[64], require_copy[64], base_ray[64];
			uint32_t gx[64], gy[64];
			for (uint32_t lane = 0; lane < 64; lane++)
			{
				uint32_t lx, ly;
				unpack_z_order(lane, lx, ly);
				gx[lane] = uint32_t(tx) * 8u + lx;
				gy[lane] = uint32_t(ty) * 8u + ly;
				const bool inside = int(gx[lane]) < a->width && int(gy[lane]) < a->height;
				const float roughness = load_pbr(c, int(gx[lane]), int(gy[lane])).y;
				bool ray = inside;
				const bool reflective = load_depth(c, int(gx[lane]), int(gy[lane]), 0) < 1.0f;
				const bool glossy = roughness < 0.2f;
				ray = ray && glossy && reflective;
				const bool needs_denoiser = ray && !(roughness < 0.0001f);
				base_ray[lane] = ((gx[lane] ^ (uint32_t(a->frame) & 1u)) & 1u) == (gy[lane] & 1u);
				ray = ray && (!needs_denoiser || base_ray[lane]);
				needs_ray[lane] = ray;
				require_copy[lane] = !ray && needs_denoiser;
				if (inside)
				{
					store_rgba16f(a->output, a->width, int(gx[lane]), int(gy[lane]), V4(0.0f));
					a->confidence[size_t(gy[lane]) * a->width + gx[lane]] = 0;
				}
			}
			for (uint32_t lane = 0; lane < 64; lane++)
			{
				if (!needs_ray[lane])
					continue;
				const bool ch = base_ray[lane] && require_copy[lane ^ 1u];
				const bool cv = base_ray[lane] && require_copy[lane ^ 2u];
				const bool cd = base_ray[lane] && require_copy[lane ^ 3u];
				a->ray_list[count++] = pack_ray(gx[lane], gy[lane], ch, cv, cd);
			}
		}
	a->ray_counter[0] = (count + 63u) / 64u;
	a->ray_counter[1] = 1;
	a->ray_counter[2] = 1;
	a->ray_counter[3] = 0;
	a->ray_counter[4] = 0;     // atomic_count, reset by build_indirect
	a->ray_counter[5] = count; // copied_count
}

// trace_primary.comp over the ray list orc_ssr_classify left.
void orc_ssr_trace(const OrcSSRArgs *a)
{
	Ctx c{a, load_mat(a->view_projection), load_mat(a->inv_view_projection), V2(float(a->width), float(a->height)),
	      V2(1.0f / float(a->width), 1.0f / float(a->height)), V3(a->camera_position[0], a->camera_position[1], a->camera_position[2])};
	const uint32_t count = a->ray_counter[5];
	const int most_detailed_mip = 1;
	const uint32_t min_occupancy = 4, max_intersections = 128;
	const float thickness = 0.05f;
	const long waves = long((count + 63u) / 64u);
#pragma omp parallel for schedule(dynamic, 16)
	for (long wave = 0; wave < waves; wave++)
	{
		Ray rays[64];
		int cx[64], cy[64];
		bool copy_h[64], copy_v[64], copy_d[64], early_out[64];
		vec3 world_pos[64], reflected[64];
		vec2 uv[64];
		for (uint32_t lane = 0; lane < 64; lane++)
		{
			Ray &r = rays[lane];
			const uint32_t index = uint32_t(wave) * 64u + lane;
			r.listed = index < count;
			r.active = false;
			early_out[lane] = false;
			if (!r.listed)
				continue;
			const uint32_t word = a->ray_list[index];
			cx[lane] = int(word & 0x3fffu);
			cy[lane] = int((word >> 14) & 0x3fffu);
			copy_h[lane] = ((word >> 28) & 1u) != 0;
			copy_v[lane] = ((word >> 29) & 1u) != 0;
			copy_d[lane] = ((word >> 30) & 1u) != 0;
			// trace_inner up to the traversal
			uv[lane] = (V2(float(cx[lane]), float(cy[lane])) + V2(0.5f, 0.5f)) * c.inv_resolution;
			const vec2 clip_uv = 2.0f * uv[lane] - V2(1.0f, 1.0f);
			const float clip_depth = load_depth(c, cx[lane], cy[lane], 0);
			if (clip_depth == 1.0f)
			{
				early_out[lane] = true;
				continue;
			}
			const float roughness = load_pbr(c, cx[lane], cy[lane]).y;
			world_pos[lane] = screen_to_world(c, V3(clip_uv.x, clip_uv.y, clip_depth));
			const vec3 V = normalize(c.camera - world_pos[lane]);
			const vec3 N = load_normal(c, cx[lane], cy[lane]);
			reflected[lane] = sample_reflection_vector(c, -V, N, roughness, cx[lane], cy[lane]);
			vec3 dir = project_direction_to_clip_space(c, V3(clip_uv.x, clip_uv.y, clip_depth), reflected[lane]);
			dir.x *= 0.5f;
			dir.y *= 0.5f;
			r.origin = V3(uv[lane].x, uv[lane].y, clip_depth);
			r.direction = dir;
			r.is_mirror = roughness < 0.0001f;
			ray_begin(c, r, most_detailed_mip);
			r.active = true;
		}
		// the while loop of FFX_SSSR_HierarchicalRaymarch, all lanes of the wave in lockstep
		for (;;)
		{
			uint32_t active_lanes = 0;
			bool in_loop[64];
			for (uint32_t lane = 0; lane < 64; lane++)
			{
				const Ray &r = rays[lane];
				in_loop[lane] = r.active && uint32_t(r.i) < max_intersections && r.mip >= most_detailed_mip && !r.exit_low;
				active_lanes += in_loop[lane] ? 1u : 0u;
			}
			if (active_lanes == 0)
				break;
			for (uint32_t lane = 0; lane < 64; lane++)
				if (in_loop[lane])
					ray_step(c, rays[lane], active_lanes, min_occupancy);
		}
		for (uint32_t lane = 0; lane < 64; lane++)
		{
			const Ray &r = rays[lane];
			if (!r.listed)
				continue;
			float confidence = 0.0f, ray_len = 0.0f;
			vec3 color = V3(0.0f);
			if (!early_out[lane])
			{
				const bool valid_hit = uint32_t(r.i) <= max_intersections;
				vec3 result = r.position;
				confidence = valid_hit ? validate_hit(c, result, uv[lane], reflected[lane], thickness) : 0.0f;
				if (confidence > 0.0f)
				{
					const int tx = int(c.float_resolution.x * result.x), ty = int(c.float_resolution.y * result.y);
					color = load_light(c, tx, ty) * confidence;
					result.x = result.x * 2.0f - 1.0f;
					result.y = result.y * 2.0f - 1.0f;
					const vec3 hit_pos = screen_to_world(c, result);
					ray_len = length(world_pos[lane] - hit_pos);
				}
			}
			color = color + load_light(c, cx[lane], cy[lane]);
			store_result(a, cx[lane], cy[lane], color, ray_len, confidence);
			if (copy_h[lane] && !pixel_needs_ray(c, cx[lane] ^ 1, cy[lane] ^ 1)) // the diagonal ray's vertical copy owns that pixel
				store_result(a, cx[lane] ^ 1, cy[lane], color, ray_len, confidence);
			if (copy_v[lane])
				store_result(a, cx[lane], cy[lane] ^ 1, color, ray_len, confidence);
			if (copy_d[lane])
				store_result(a, cx[lane] ^ 1, cy[lane] ^ 1, color, ray_len, confidence);
		}
	}
}

// apply.frag (+ apply.vert, blend ONE / ONE, depth test NOT_EQUAL against the quad's z = 1): hdr += reflected * (F brdf.x + brdf.y).
void orc_ssr_apply(int width, int height, const uint16_t *reflected, const uint32_t *albedo_srgb, const uint32_t *normal, const uint16_t *pbr,
                   const float *depth, const uint16_t *brdf_lut_rg16f, int lut_w, int lut_h, const float *inv_view_projection,
                   const float *camera_position, uint16_t *hdr)
{
	const mat4 inv_vp = load_mat(inv_view_projection);
	const vec3 camera = V3(camera_position[0], camera_position[1], camera_position[2]);
#pragma omp parallel for schedule(static)
	for (int y = 0; y < height; y++)
		for (int x = 0; x < width; x++)
		{
			const size_t i = size_t(y) * width + x;
			const float clip_depth = depth[i];
			if (clip_depth == 1.0f)
				continue; // depth test NOT_EQUAL
			const uint16_t mr = pbr[i];
			const float metallic = float(mr & 255u) / 255.0f, roughness = float(mr >> 8) / 255.0f;
			const vec2 vuv = (V2(float(x), float(y)) + V2(0.5f, 0.5f)) * V2(1.0f / float(width), 1.0f / float(height));
			const vec2 clip_uv = vuv * 2.0f - V2(1.0f, 1.0f);
			const vec4 wc = mul(inv_vp, V4(clip_uv.x, clip_uv.y, clip_depth, 1.0f));
			const vec3 world_pos = V3(wc.x, wc.y, wc.z) / wc.w;
			const vec3 V = normalize(camera - world_pos);
			const uint32_t nv = normal[i];
			const vec3 N = normalize(V3(float(nv & 1023u) / 1023.0f, float((nv >> 10) & 1023u) / 1023.0f, float((nv >> 20) & 1023u) / 1023.0f) * 2.0f - V3(1.0f));
			const float NoV = clampf(dot(N, V), 0.0f, 1.0f);
			const uint32_t al = albedo_srgb[i];
			const vec3 base = V3(srgb8_to_float(uint8_t(al & 255u)), srgb8_to_float(uint8_t((al >> 8) & 255u)), srgb8_to_float(uint8_t((al >> 16) & 255u)));
			const vec3 F0 = mix(V3(0.04f), base, metallic); // compute_F0 (pbr.h)
			// fresnel_ibl (pbr.h): F0 + (max(vec3(1 - roughness), F0) - F0) * pow(1 - cos_theta, 5)
			const vec3 F = F0 + (max3(V3(1.0f - roughness), F0) - F0) * powf(1.0f - NoV, 5.0f);
			// textureLod(uBRDFLut, vec2(NoV, roughness), 0): LinearClamp on RG16F
			float wa, wb;
			int ix, iy;
			linear_axis(NoV * float(lut_w) - 0.5f, ix, wa);
			linear_axis(roughness * float(lut_h) - 0.5f, iy, wb);
			const int x0 = clampi(ix, 0, lut_w - 1), x1 = clampi(ix + 1, 0, lut_w - 1);
			const int y0 = clampi(iy, 0, lut_h - 1), y1 = clampi(iy + 1, 0, lut_h - 1);
			auto lut = [&](int lx, int ly) {
				const uint16_t *p = brdf_lut_rg16f + (size_t(ly) * lut_w + lx) * 2;
				return V2(half_to_float(p[0]), half_to_float(p[1]));
			};
			const vec2 brdf = linear_combine(lut(x0, y0), lut(x1, y0), lut(x0, y1), lut(x1, y1), wa, wb);
			const vec4 r = load_rgba16f(reflected, width, x, y); // NearestClamp at the pixel centre
			const vec3 color = V3(r.x, r.y, r.z) * (F * brdf.x + V3(brdf.y));
			// blend ONE / ONE into the RGBA16F target; alpha untouched (the shader writes a vec3)
			uint16_t *p = hdr + i * 4;
			p[0] = float_to_half_rne(half_to_float(p[0]) + color.x);
			p[1] = float_to_half_rne(half_to_float(p[1]) + color.y);
			p[2] = float_to_half_rne(half_to_float(p[2]) + color.z);
		}
}
}
