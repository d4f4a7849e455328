// Light-cluster build kernels for gfx950 and their C-ABI launchers (include/granite_hip.h).
//
// Replaces assets/shaders/lights/clusterer_bindless_{spot_transform,setup,binning,z_range}.comp as dispatched by
// LightClusterer::update_bindless_mask_buffer_gpu / update_bindless_range_buffer_gpu
// (renderer/lights/clusterer.cpp:1277-1320,1463-1562).
//
// The outputs are integer bit masks / index ranges, so this translation unit is compiled with -ffp-contract=off and
// uses only correctly-rounded fp32 +,-,*,/,sqrt in a fixed association order: the masks are reproducible bit for bit.
#include "ctx.hpp"
#include "device_common.hpp"
#include "device_vec.hpp"

namespace
{
// column-major mat4 * vec4, left-to-right column accumulation (GLSL M * v).
__device__ __forceinline__ v4 mul_mat4(const float *m, v4 v)
{
	v4 r = mk4(m[0], m[1], m[2], m[3]) * v.x;
	r = r + mk4(m[4], m[5], m[6], m[7]) * v.y;
	r = r + mk4(m[8], m[9], m[10], m[11]) * v.z;
	r = r + mk4(m[12], m[13], m[14], m[15]) * v.w;
	return r;
}

struct TransformedSpot { v4 clip[5]; v4 z; };
struct CullSetup { v4 data[32]; };

__device__ __forceinline__ const gr_light_info *lights_of(const void *transforms)
{
	return reinterpret_cast<const gr_light_info *>(static_cast<const uint8_t *>(transforms) + GR_TRANSFORMS_OFFSET_LIGHTS);
}
__device__ __forceinline__ const gr_mat_affine *models_of(const void *transforms)
{
	return reinterpret_cast<const gr_mat_affine *>(static_cast<const uint8_t *>(transforms) + GR_TRANSFORMS_OFFSET_MODEL);
}
__device__ __forceinline__ const uint32_t *type_mask_of(const void *transforms)
{
	return reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(transforms) + GR_TRANSFORMS_OFFSET_TYPE_MASK);
}

// ---- spot_transform.comp:38-73 -------------------------------------------------------------------------------------
__device__ __forceinline__ TransformedSpot spot_transform_of(const gr_mat_affine &m, const gr_push_spot_transform &push)
{
	TransformedSpot result;
	v3 p[5];
	p[0] = mk3(m.rows[0][3], m.rows[1][3], m.rows[2][3]);
	const v3 pz = p[0] + mk3(-m.rows[0][2], -m.rows[1][2], -m.rows[2][2]);
	const v3 right = mk3(m.rows[0][0], m.rows[1][0], m.rows[2][0]);
	const v3 up = mk3(m.rows[0][1], m.rows[1][1], m.rows[2][1]);
	p[1] = pz + right + up;
	p[2] = pz - right + up;
	p[3] = pz - right - up;
	p[4] = pz + right - up;

	const v3 cam = mk3(push.camera_pos[0], push.camera_pos[1], push.camera_pos[2]);
	const v3 front = mk3(push.camera_front[0], push.camera_front[1], push.camera_front[2]);
	float z_lo, z_hi;
	z_lo = z_hi = dot3(p[0] - cam, front);
#pragma unroll
	for (int i = 1; i < 5; i++)
	{
		const float z = dot3(p[i] - cam, front);
		z_lo = fminf(z_lo, z);
		z_hi = fmaxf(z_hi, z);
	}
	float cull;
	if (z_lo <= push.z_near && z_hi >= push.z_far)
		cull = 0.0f;
	else if (z_lo <= push.z_near)
		cull = -1.0f;
	else
		cull = 1.0f;
#pragma unroll
	for (int i = 0; i < 5; i++)
		result.clip[i] = mul_mat4(push.vp, mk4(p[i].x, p[i].y, p[i].z, 1.0f));
	result.z = mk4(cull, z_lo, z_hi, 0.0f);
	return result;
}

__global__ __launch_bounds__(64) void k_spot_transform(const void *transforms, TransformedSpot *out, gr_push_spot_transform push)
{
	const uint32_t index = blockIdx.x * 64u + threadIdx.x;
	if (index >= push.num_lights)
		return;
	out[index] = spot_transform_of(models_of(transforms)[index], push);
}

// ---- setup.comp ------------------------------------------------------------------------------------------------------
__device__ v2 project_sphere_flat(float view_xy, float view_z, float radius)
{
	const float len = len2(mk2(view_xy, view_z));
	const float sin_xy = radius / len;
	v2 result;
	if (sin_xy < 0.999f)
	{
		const float cos_xy = sqrtf(1.0f - sin_xy * sin_xy);
		v2 rot_lo = mk2(cos_xy * view_xy + (-sin_xy) * view_z, sin_xy * view_xy + cos_xy * view_z);
		v2 rot_hi = mk2(cos_xy * view_xy + sin_xy * view_z, (-sin_xy) * view_xy + cos_xy * view_z);
		if (rot_lo.y <= 0.0f)
			rot_lo = mk2(-1.0f, 0.0f);
		if (rot_hi.y <= 0.0f)
			rot_hi = mk2(+1.0f, 0.0f);
		result = mk2(rot_lo.x / rot_lo.y, rot_hi.x / rot_hi.y);
	}
	else
		result = mk2(-__builtin_inff(), __builtin_inff());
	return result;
}

struct Tri2 { v2 c[3]; };
struct Tri3 { v3 c[3]; };

__device__ void emit_triangle(CullSetup &cs, uint32_t &num_triangles, const Tri2 &t, float cull)
{
	const v2 c0 = t.c[0], c1 = t.c[1], c2 = t.c[2];
	const v2 ab = c1 - c0, bc = c2 - c1, ca = c0 - c2;
	const float z = cross_2d(ab, -ca);
	if (fabsf(z) < 0.000001f || signf(cull) == signf(z))
		return;
	const float inv_z = 1.0f / z;
	if (num_triangles < 8u)
	{
		const uint32_t o = 4u * num_triangles;
		cs.data[o] = mk4(inv_z * cross_2d(ab, -c0), inv_z * cross_2d(bc, -c1), inv_z * cross_2d(ca, -c2), 0.0f);
		cs.data[o + 1u] = mk4(inv_z * (-ab.y), inv_z * (-bc.y), inv_z * (-ca.y), z);
		cs.data[o + 2u] = mk4(inv_z * ab.x, inv_z * bc.x, inv_z * ca.x, inv_z);
		cs.data[o + 3u] = mk4(fminf(fminf(c0.x, c1.x), c2.x), fminf(fminf(c0.y, c1.y), c2.y), fmaxf(fmaxf(c0.x, c1.x), c2.x),
		                      fmaxf(fmaxf(c0.y, c1.y), c2.y));
	}
	num_triangles++;
}

// Clip against z = 0 (second overload family in setup.comp:150-199).
__device__ void clip_near_z(CullSetup &cs, uint32_t &num_triangles, const Tri3 &t, float cull)
{
	v3 c0 = t.c[0], c1 = t.c[1], c2 = t.c[2];
	const uint32_t code = uint32_t(c0.z < 0.0f) + uint32_t(c1.z < 0.0f) * 2u + uint32_t(c2.z < 0.0f) * 4u;
	if (code == 7u)
		return;
	if (code == 0u)
	{
		emit_triangle(cs, num_triangles, Tri2{{xy(c0), xy(c1), xy(c2)}}, cull);
		return;
	}
	// rotate so the reference's argument order (c0,c1,c2) / (c1,c2,c0) / (c2,c0,c1) applies
	v3 a, b, c;
	bool dual;
	switch (code)
	{
	case 1u: a = c0; b = c1; c = c2; dual = true; break;
	case 2u: a = c1; b = c2; c = c0; dual = true; break;
	case 4u: a = c2; b = c0; c = c1; dual = true; break;
	case 3u: a = c0; b = c1; c = c2; dual = false; break;
	case 5u: a = c2; b = c0; c = c1; dual = false; break;
	default: a = c1; b = c2; c = c0; dual = false; break; // 6
	}
	const float target = 0.0f;
	if (dual)
	{
		const float l_ab = (target - a.z) / (b.z - a.z);
		const float l_ac = (target - a.z) / (c.z - a.z);
		const v3 ab = mix3(a, b, l_ab);
		const v3 ac = mix3(a, c, l_ac);
		emit_triangle(cs, num_triangles, Tri2{{xy(ab), xy(b), xy(ac)}}, cull);
		emit_triangle(cs, num_triangles, Tri2{{xy(ac), xy(b), xy(c)}}, cull);
	}
	else
	{
		const float la = (target - a.z) / (c.z - a.z);
		const float lb = (target - b.z) / (c.z - b.z);
		a = mix3(a, c, la);
		b = mix3(b, c, lb);
		emit_triangle(cs, num_triangles, Tri2{{xy(a), xy(b), xy(c)}}, cull);
	}
}

// Clip against w = MIN_W, perspective divide (setup.comp:201-250).
__device__ void clip_w_and_project(CullSetup &cs, uint32_t &num_triangles, v4 c0, v4 c1, v4 c2, float cull)
{
	const float MIN_W = 1.0f / 1024.0f;
	const uint32_t code = uint32_t(c0.w < MIN_W) + uint32_t(c1.w < MIN_W) * 2u + uint32_t(c2.w < MIN_W) * 4u;
	if (code == 7u)
		return;
	if (code == 0u)
	{
		clip_near_z(cs, num_triangles, Tri3{{xyz(c0) / c0.w, xyz(c1) / c1.w, xyz(c2) / c2.w}}, cull);
		return;
	}
	v4 a, b, c;
	bool dual;
	switch (code)
	{
	case 1u: a = c0; b = c1; c = c2; dual = true; break;
	case 2u: a = c1; b = c2; c = c0; dual = true; break;
	case 4u: a = c2; b = c0; c = c1; dual = true; break;
	case 3u: a = c0; b = c1; c = c2; dual = false; break;
	case 5u: a = c2; b = c0; c = c1; dual = false; break;
	default: a = c1; b = c2; c = c0; dual = false; break; // 6
	}
	if (dual)
	{
		const float l_ab = (MIN_W - a.w) / (b.w - a.w);
		const float l_ac = (MIN_W - a.w) / (c.w - a.w);
		const v4 ab = mix4(a, b, l_ab);
		const v4 ac = mix4(a, c, l_ac);
		clip_near_z(cs, num_triangles, Tri3{{xyz(ab) / MIN_W, xyz(b) / b.w, xyz(ac) / MIN_W}}, cull);
		clip_near_z(cs, num_triangles, Tri3{{xyz(ac) / MIN_W, xyz(b) / b.w, xyz(c) / c.w}}, cull);
	}
	else
	{
		const float la = (MIN_W - a.w) / (c.w - a.w);
		const float lb = (MIN_W - b.w) / (c.w - b.w);
		a = mix4(a, c, la);
		b = mix4(b, c, lb);
		clip_near_z(cs, num_triangles, Tri3{{xyz(a) / MIN_W, xyz(b) / MIN_W, xyz(c) / c.w}}, cull);
	}
}

__device__ __forceinline__ void cluster_setup_of(bool point, const gr_light_info &li, const TransformedSpot &spot, CullSetup &cs,
                                                 const gr_cluster_params &params, const gr_push_cluster_setup &push)
{
	if (point)
	{
		const float radius = 1.0f / li.inv_radius;
		const v4 v4d = mul_mat4(push.view, mk4(li.position[0], li.position[1], li.position[2], 1.0f));
		const v3 view = mk3(v4d.x, -v4d.y, -v4d.z);
		const v2 r0 = project_sphere_flat(view.x, view.z, radius);
		const v2 r1 = project_sphere_flat(view.y, view.z, radius);
		const float xy_length = len2(mk2(view.x, view.y));
		v2 ct0, ct1;
		if (xy_length < 0.00001f)
		{
			ct0 = mk2(1.0f, 0.0f);
			ct1 = mk2(0.0f, 1.0f);
		}
		else
		{
			const float inv_xy_length = 1.0f / xy_length;
			ct0 = mk2(view.x, -view.y) * inv_xy_length;
			ct1 = mk2(view.y, view.x) * inv_xy_length;
		}
		const v2 txy = mk2(ct0.x * view.x + ct1.x * view.y, ct0.y * view.x + ct1.y * view.y);
		const v2 t0 = project_sphere_flat(txy.x, view.z, radius);
		const v2 t1 = project_sphere_flat(txy.y, view.z, radius);
		const bool ellipsis = !isinf(t0.x) && !isinf(t0.y) && !isinf(t1.x) && !isinf(t1.y);
		const v2 center = (mk2(t0.x, t1.x) + mk2(t0.y, t1.y)) * 0.5f;
		const v2 ellipse_radius = mk2(t0.y, t1.y) - center;
		cs.data[0] = mk4(r0.x * params.clip_scale[0], r1.x * params.clip_scale[1], r0.y * params.clip_scale[0],
		                 r1.y * params.clip_scale[1]);
		cs.data[1] = mk4(t0.x, t0.y, t1.x, t1.y);
		cs.data[2] = mk4(ct0.x, ct0.y, ct1.x, ct1.y);
		cs.data[3] = mk4(ellipsis ? 1.0f : 0.0f, 1.0f / ellipse_radius.x, 1.0f / ellipse_radius.y, 0.0f);
	}
	else
	{
		const v4 z = spot.z;
		if (z.x != 0.0f)
		{
			uint32_t n = 0u;
			const v4 c0 = spot.clip[0], c1 = spot.clip[1], c2 = spot.clip[2];
			const v4 c3 = spot.clip[3], c4 = spot.clip[4];
			clip_w_and_project(cs, n, c0, c1, c2, z.x);
			clip_w_and_project(cs, n, c0, c2, c3, z.x);
			clip_w_and_project(cs, n, c0, c3, c4, z.x);
			clip_w_and_project(cs, n, c0, c4, c1, z.x);
			clip_w_and_project(cs, n, c2, c1, c3, z.x);
			clip_w_and_project(cs, n, c4, c3, c1, z.x);
			cs.data[0].w = __uint_as_float(n);
		}
		else
			cs.data[0].w = __uint_as_float(0xffffffffu);
	}
}

__global__ __launch_bounds__(64) void k_cluster_setup(const void *transforms, const TransformedSpot *spots, CullSetup *setup,
                                                      gr_cluster_params params, gr_push_cluster_setup push)
{
	const uint32_t index = blockIdx.x * 64u + threadIdx.x;
	if (index >= push.num_lights)
		return;
	const bool point = (type_mask_of(transforms)[index >> 5u] & (1u << (index & 31u))) != 0u;
	cluster_setup_of(point, lights_of(transforms)[index], spots[index], setup[index], params, push);
}

// ---- binning.comp -------------------------------------------------------------------------------------------------------
__device__ bool test_point_light(const gr_cluster_params &prm, const CullSetup &cs, v2 uv, v2 uv_stride)
{
	const v4 eir = cs.data[3];
	if (eir.x != 0.0f)
	{
		const v4 tr = cs.data[1];
		const v4 ct = cs.data[2];
		const v2 center = (mk2(tr.x, tr.z) + mk2(tr.y, tr.w)) * 0.5f;
		const v2 cszw = mk2(prm.clip_scale[2], prm.clip_scale[3]);
		const v2 lo = uv * cszw;
		const v2 hi = (uv + uv_stride) * cszw;
		const v2 inv_r = mk2(eir.y, eir.z);
		const v2 d00 = (mk2(ct.x * lo.x + ct.z * lo.y, ct.y * lo.x + ct.w * lo.y) - center) * inv_r;
		const v2 d01 = (mk2(ct.x * lo.x + ct.z * hi.y, ct.y * lo.x + ct.w * hi.y) - center) * inv_r;
		const v2 d10 = (mk2(ct.x * hi.x + ct.z * lo.y, ct.y * hi.x + ct.w * lo.y) - center) * inv_r;
		const v2 d11 = (mk2(ct.x * hi.x + ct.z * hi.y, ct.y * hi.x + ct.w * hi.y) - center) * inv_r;
		const float max_diag = fmaxf(dist2(d00, d11), dist2(d01, d10));
		float min_sq_dist = 1.0f + max_diag;
		min_sq_dist *= min_sq_dist;
		return dot2(d00, d00) < min_sq_dist && dot2(d01, d01) < min_sq_dist && dot2(d10, d10) < min_sq_dist &&
		       dot2(d11, d11) < min_sq_dist;
	}
	const v4 bb = cs.data[0];
	const v2 hi = uv + uv_stride;
	return hi.x > bb.x && hi.y > bb.y && uv.x < bb.z && uv.y < bb.w;
}

__device__ bool test_spot_light(const CullSetup &cs, v2 uv, v2 uv_stride)
{
	const uint32_t num_triangles = __float_as_uint(cs.data[0].w);
	if (num_triangles > 8u)
		return true;
	const v2 hi = uv + uv_stride;
	for (uint32_t i = 0; i < num_triangles; i++)
	{
		const v4 bb = cs.data[4u * i + 3u];
		if (hi.x > bb.x && hi.y > bb.y && uv.x < bb.z && uv.y < bb.w)
		{
			v3 base = xyz(cs.data[4u * i]);
			const v3 dx = xyz(cs.data[4u * i + 1u]);
			const v3 dy = xyz(cs.data[4u * i + 2u]);
			base = base + dx * uv.x;
			base = base + dy * uv.y;
			base = base + mk3(dx.x > 0.0f ? uv_stride.x * dx.x : 0.0f, dx.y > 0.0f ? uv_stride.x * dx.y : 0.0f,
			                  dx.z > 0.0f ? uv_stride.x * dx.z : 0.0f);
			base = base + mk3(dy.x > 0.0f ? uv_stride.y * dy.x : 0.0f, dy.y > 0.0f ? uv_stride.y * dy.y : 0.0f,
			                  dy.z > 0.0f ? uv_stride.y * dy.z : 0.0f);
			if (base.x > 0.0f && base.y > 0.0f && base.z > 0.0f)
				return true;
		}
	}
	return false;
}

// One wave64 per (32-light chunk, 8x8 cell tile): lanes 0..31 run the coarse tile test for their light, the ballot is
// then walked wave-uniformly while each of the 64 lanes tests its own cell (binning.comp:136-179 at gl_SubgroupSize 64).
__global__ __launch_bounds__(64) void k_cluster_binning(const void *transforms, const CullSetup *setup, uint32_t *bitmask,
                                                        gr_cluster_params prm)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t chunk = blockIdx.x;
	const uint32_t tile_x = blockIdx.y, tile_y = blockIdx.z;
	const v2 inv_res = mk2(prm.inv_resolution_xy[0], prm.inv_resolution_xy[1]);
	const uint32_t type_mask = type_mask_of(transforms)[chunk];

	const v2 tile_uv = mk2(2.0f * float(tile_x * 8u), 2.0f * float(tile_y * 8u)) * inv_res - mk2(1.0f, 1.0f);
	const v2 tile_stride = mk2(2.0f * 8.0f, 2.0f * 8.0f) * inv_res;

	bool passed = false;
	if (lane < 32u)
	{
		const uint32_t light_index = 32u * chunk + lane;
		if (light_index < uint32_t(prm.num_lights))
		{
			if ((type_mask >> lane) & 1u)
				passed = test_point_light(prm, setup[light_index], tile_uv, tile_stride);
			else
				passed = test_spot_light(setup[light_index], tile_uv, tile_stride);
		}
	}
	uint32_t ballot = uint32_t(__ballot(passed));

	const uint32_t px = tile_x * 8u + (lane & 7u), py = tile_y * 8u + (lane >> 3u);
	const v2 uv = mk2(2.0f * float(px), 2.0f * float(py)) * inv_res - mk2(1.0f, 1.0f);
	const v2 uv_stride = inv_res * 2.0f;
	uint32_t pixel_mask = 0u;
	while (ballot != 0u)
	{
		const int lsb = __builtin_ctz(ballot);
		ballot &= ballot - 1u;
		const uint32_t light_index = chunk * 32u + uint32_t(lsb);
		bool hit;
		if ((type_mask >> lsb) & 1u)
			hit = test_point_light(prm, setup[light_index], uv, uv_stride);
		else
			hit = test_spot_light(setup[light_index], uv, uv_stride);
		if (hit)
			pixel_mask |= 1u << lsb;
	}
	const uint32_t linear_coord = py * uint32_t(prm.resolution_xy[0]) + px;
	bitmask[size_t(linear_coord) * uint32_t(prm.num_lights_32) + chunk] = pixel_mask;
}

// ---- z_range.comp ----------------------------------------------------------------------------------------------------------
// out[z] = (first light whose [lo,hi] slice interval covers z, last such light); empty = (0xffffffff, 0).
// The reference scans every interval for every slice (O(slices x lights)).  Same result with pruning: intervals are staged in
// LDS together with the union interval of each group of 64 consecutive lights; for a slice, one ballot over the group
// bounds says which groups can contain a covering light, and the first (last) covering light is found by walking those
// groups from the front (back) with one ballot per group -- lights arrive sorted by depth, so it is almost always the
// first group tried.  Exact for any input order (the group bounds only prune), integer-only.
constexpr int ZR_THREADS = 256;
constexpr int ZR_SLICES_PER_WAVE = 4;
constexpr int ZR_SLICES = (ZR_THREADS / 64) * ZR_SLICES_PER_WAVE;

__device__ __forceinline__ uint32_t zr_wave_min(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v = min(v, uint32_t(__shfl_xor(int(v), off, 64)));
	return v;
}
__device__ __forceinline__ uint32_t zr_wave_max(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v = max(v, uint32_t(__shfl_xor(int(v), off, 64)));
	return v;
}

// The work of one z-range workgroup (ZR_SLICES slices from slice_block * ZR_SLICES); `copy_out`, when given, also receives the
// intervals (the fused launch reads them from the pinned staging area and keeps the "light-ranges" buffer in step).
__device__ __forceinline__ void z_range_block(const uint2 *light_ranges, uint2 *out, const gr_push_z_range &push, uint32_t slice_block, uint8_t *smem,
                                              uint2 *copy_out)
{
	const uint32_t num_groups = (push.num_volumes + 63u) / 64u;
	uint2 *ranges = reinterpret_cast<uint2 *>(smem);        // num_groups * 64 entries, padded with empty intervals
	uint2 *group_bounds = ranges + size_t(num_groups) * 64u; // num_groups entries
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6u;

	for (uint32_t g = wave; g < num_groups; g += ZR_THREADS / 64)
	{
		const uint32_t i = g * 64u + lane;
		const uint2 r = i < push.num_volumes ? light_ranges[i] : make_uint2(0xffffffffu, 0u);
		ranges[i] = r;
		if (copy_out && i < push.num_volumes)
			copy_out[i] = r;
		// an empty interval (lo > hi) must not widen the group's bounds
		const bool valid = r.x <= r.y;
		const uint32_t glo = zr_wave_min(valid ? r.x : 0xffffffffu), ghi = zr_wave_max(valid ? r.y : 0u);
		if (lane == 0)
			group_bounds[g] = make_uint2(glo, ghi);
	}
	__syncthreads();

	for (uint32_t s = 0; s < ZR_SLICES_PER_WAVE; s++)
	{
		const uint32_t z = slice_block * ZR_SLICES + wave * ZR_SLICES_PER_WAVE + s;
		if (z >= push.num_ranges)
			break;
		uint32_t first = 0xffffffffu, last = 0u;
		// up to 64 groups per round (4096 lights = one round)
		for (uint32_t base = 0; base < num_groups; base += 64u)
		{
			const uint32_t g = base + lane;
			const uint2 gb = g < num_groups ? group_bounds[g] : make_uint2(0xffffffffu, 0u);
			const uint64_t candidates = __ballot(gb.x <= z && z <= gb.y);
			if (first == 0xffffffffu)
			{
				uint64_t walk = candidates;
				while (walk != 0ull)
				{
					const uint32_t gg = base + uint32_t(__builtin_ctzll(walk));
					walk &= walk - 1ull;
					const uint2 r = ranges[gg * 64u + lane];
					const uint64_t m = __ballot(r.x <= z && z <= r.y);
					if (m != 0ull)
					{
						first = gg * 64u + uint32_t(__builtin_ctzll(m));
						break;
					}
				}
			}
			uint64_t walk = candidates;
			while (walk != 0ull)
			{
				const uint32_t top = 63u - uint32_t(__builtin_clzll(walk));
				walk &= ~(1ull << top);
				const uint32_t gg = base + top;
				const uint2 r = ranges[gg * 64u + lane];
				const uint64_t m = __ballot(r.x <= z && z <= r.y);
				if (m != 0ull)
				{
					last = max(last, gg * 64u + 63u - uint32_t(__builtin_clzll(m)));
					break;
				}
			}
		}
		if (lane == 0)
			out[z] = make_uint2(first, last);
	}
}

__global__ __launch_bounds__(ZR_THREADS) void k_cluster_z_range(const uint2 *light_ranges, uint2 *out, gr_push_z_range push)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	z_range_block(light_ranges, out, push, blockIdx.x, smem, nullptr);
}

// ---- the front of the cluster build as ONE launch ------------------------------------------------------------------------------
// What clusterer.cpp:1178-1207,1302 (four cmd.update_buffer), :1463-1510 (spot_transform.comp, setup.comp) and :1277-1346
// (z_range.comp) record as four transfers and three dispatches: every one of them depends on this frame's CPU-packed light data
// only, so one grid does them all -- workgroups [0, light_blocks) take 256 lights each (copy the light's records from the pinned
// staging area into the transforms buffer, transform, set up the cull data), the following workgroups are z-range workgroups
// reading the slice intervals from staging.  Only the binning pass, which needs every light's cull data, stays a launch of its
// own.  Same device functions as the separate launches: the buffers come out bit-identical.
struct ClusterFrontArgs
{
	void *transforms;                 // HBM: lights, model matrices, type mask at their GR_TRANSFORMS_OFFSET_*
	const gr_light_info *src_lights;  // pinned staging (or the transforms buffer itself: nothing is copied then)
	const gr_mat_affine *src_models;
	const uint32_t *src_type_mask;
	TransformedSpot *spots;
	CullSetup *setup;
	gr_cluster_params params;
	gr_push_spot_transform spot_push;
	gr_push_cluster_setup setup_push;
	const uint2 *src_ranges; // pinned staging or the HBM buffer
	uint2 *light_ranges;     // HBM copy of the intervals (null or == src_ranges: not written)
	uint2 *range_out;
	gr_push_z_range z_push;
	uint32_t light_blocks;
};

__global__ __launch_bounds__(ZR_THREADS) void k_cluster_front(ClusterFrontArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	if (blockIdx.x >= a.light_blocks)
	{
		z_range_block(a.src_ranges, a.range_out, a.z_push, blockIdx.x - a.light_blocks, smem,
		              (blockIdx.x == a.light_blocks && a.light_ranges != a.src_ranges) ? a.light_ranges : nullptr);
		return;
	}
	const uint32_t index = blockIdx.x * ZR_THREADS + threadIdx.x;
	uint8_t *t = static_cast<uint8_t *>(a.transforms);
	uint32_t *dst_mask = reinterpret_cast<uint32_t *>(t + GR_TRANSFORMS_OFFSET_TYPE_MASK);
	if (a.src_type_mask != dst_mask && index < uint32_t(a.params.num_lights_32))
		dst_mask[index] = a.src_type_mask[index];
	if (index >= a.spot_push.num_lights)
		return;
	const gr_light_info li = a.src_lights[index];
	const gr_mat_affine m = a.src_models[index];
	gr_light_info *dst_lights = reinterpret_cast<gr_light_info *>(t + GR_TRANSFORMS_OFFSET_LIGHTS);
	gr_mat_affine *dst_models = reinterpret_cast<gr_mat_affine *>(t + GR_TRANSFORMS_OFFSET_MODEL);
	if (a.src_lights != dst_lights)
		dst_lights[index] = li;
	if (a.src_models != dst_models)
		dst_models[index] = m;
	const bool point = (a.src_type_mask[index >> 5u] & (1u << (index & 31u))) != 0u;
	const TransformedSpot spot = spot_transform_of(m, a.spot_push);
	a.spots[index] = spot;
	cluster_setup_of(point, li, spot, a.setup[index], a.params, a.setup_push);
}
} // namespace

extern "C" {

int gr_cluster_spot_transform(gr_ctx *ctx, gr_stream stream, const void *transforms, void *transformed_spots,
                              const gr_push_spot_transform *push)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, transforms && transformed_spots && push);
	GR_CHECK_ARG(ctx, push->num_lights <= GR_MAX_LIGHTS_BINDLESS);
	if (push->num_lights == 0)
		return GR_OK;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "cluster_spot_transform"};
	hipLaunchKernelGGL(k_spot_transform, dim3(gr_div_up(push->num_lights, 64)), dim3(64), 0, gr_to_stream(stream), transforms,
	                   static_cast<TransformedSpot *>(transformed_spots), *push);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_cluster_setup(gr_ctx *ctx, gr_stream stream, const void *transforms, const void *transformed_spots, void *cull_setup,
                     const gr_cluster_params *params, const gr_push_cluster_setup *push)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, transforms && transformed_spots && cull_setup && params && push);
	GR_CHECK_ARG(ctx, push->num_lights <= GR_MAX_LIGHTS_BINDLESS);
	if (push->num_lights == 0)
		return GR_OK;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "cluster_setup"};
	hipLaunchKernelGGL(k_cluster_setup, dim3(gr_div_up(push->num_lights, 64)), dim3(64), 0, gr_to_stream(stream), transforms,
	                   static_cast<const TransformedSpot *>(transformed_spots), static_cast<CullSetup *>(cull_setup), *params,
	                   *push);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_cluster_front(gr_ctx *ctx, gr_stream stream, const gr_cluster_front_args *args)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, args && args->transforms && args->transformed_spots && args->cull_setup && args->params && args->spot_push && args->setup_push);
	GR_CHECK_ARG(ctx, args->light_ranges && args->range_out && args->z_push);
	const uint32_t n = args->spot_push->num_lights;
	GR_CHECK_ARG(ctx, n >= 1 && n <= GR_MAX_LIGHTS_BINDLESS && n == args->setup_push->num_lights && int(n) == args->params->num_lights);
	GR_CHECK_ARG(ctx, args->z_push->num_volumes >= 1 && args->z_push->num_volumes <= GR_MAX_LIGHTS_BINDLESS && args->z_push->num_ranges > 0);
	uint8_t *t = static_cast<uint8_t *>(args->transforms);
	ClusterFrontArgs a = {};
	a.transforms = args->transforms;
	a.src_lights = args->src_lights ? static_cast<const gr_light_info *>(args->src_lights) : reinterpret_cast<const gr_light_info *>(t + GR_TRANSFORMS_OFFSET_LIGHTS);
	a.src_models = args->src_models ? static_cast<const gr_mat_affine *>(args->src_models) : reinterpret_cast<const gr_mat_affine *>(t + GR_TRANSFORMS_OFFSET_MODEL);
	a.src_type_mask = args->src_type_mask ? static_cast<const uint32_t *>(args->src_type_mask) : reinterpret_cast<const uint32_t *>(t + GR_TRANSFORMS_OFFSET_TYPE_MASK);
	a.spots = static_cast<TransformedSpot *>(args->transformed_spots);
	a.setup = static_cast<CullSetup *>(args->cull_setup);
	a.params = *args->params;
	a.spot_push = *args->spot_push;
	a.setup_push = *args->setup_push;
	a.light_ranges = reinterpret_cast<uint2 *>(args->light_ranges);
	a.src_ranges = args->src_ranges ? static_cast<const uint2 *>(args->src_ranges) : a.light_ranges;
	a.range_out = reinterpret_cast<uint2 *>(args->range_out);
	a.z_push = *args->z_push;
	a.light_blocks = gr_div_up(n, ZR_THREADS);
	const uint32_t num_groups = (a.z_push.num_volumes + 63u) / 64u;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "cluster_front"};
	hipLaunchKernelGGL(k_cluster_front, dim3(a.light_blocks + gr_div_up(a.z_push.num_ranges, ZR_SLICES)), dim3(ZR_THREADS),
	                   size_t(num_groups) * 65u * sizeof(uint2), gr_to_stream(stream), a);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_cluster_binning(gr_ctx *ctx, gr_stream stream, const void *transforms, const void *cull_setup, uint32_t *bitmask,
                       const gr_cluster_params *params)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, transforms && cull_setup && bitmask && params);
	GR_CHECK_ARG(ctx, params->resolution_xy[0] > 0 && params->resolution_xy[1] > 0);
	// clusterer.cpp:1513-1514 asserts the same.
	GR_CHECK_ARG(ctx, (params->resolution_xy[0] & 7) == 0 && (params->resolution_xy[1] & 7) == 0);
	GR_CHECK_ARG(ctx, params->num_lights >= 0 && params->num_lights <= GR_MAX_LIGHTS_BINDLESS);
	GR_CHECK_ARG(ctx, params->num_lights_32 == (params->num_lights + 31) / 32);
	if (params->num_lights == 0)
		return GR_OK;
	dim3 grid(params->num_lights_32, params->resolution_xy[0] / 8, params->resolution_xy[1] / 8);
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "cluster_binning"};
	hipLaunchKernelGGL(k_cluster_binning, grid, dim3(64), 0, gr_to_stream(stream), transforms,
	                   static_cast<const CullSetup *>(cull_setup), bitmask, *params);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_cluster_z_range(gr_ctx *ctx, gr_stream stream, const uint32_t *light_ranges, uint32_t *out, const gr_push_z_range *push)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, light_ranges && out && push);
	GR_CHECK_ARG(ctx, push->num_volumes >= 1 && push->num_volumes <= GR_MAX_LIGHTS_BINDLESS);
	GR_CHECK_ARG(ctx, push->num_ranges > 0);
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "cluster_z_range"};
	const uint32_t num_groups = (push->num_volumes + 63u) / 64u;
	hipLaunchKernelGGL(k_cluster_z_range, dim3(gr_div_up(push->num_ranges, ZR_SLICES)), dim3(ZR_THREADS),
	                   size_t(num_groups) * 65u * sizeof(uint2), gr_to_stream(stream), reinterpret_cast<const uint2 *>(light_ranges),
	                   reinterpret_cast<uint2 *>(out), *push);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}
}
