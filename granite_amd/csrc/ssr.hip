// Screen-space reflections for gfx950: SSRState::build_render_pass and the apply pass of renderer/post/ssr.cpp:84-323 with
// assets/shaders/post/ffx-sssr/{classify,build_indirect,trace_primary}.comp, apply.frag, sssr_util.h and
// inc/project_direction.h.  The arithmetic is AMD FidelityFX SSSR as the reference vendors it (hierarchical depth-buffer
// traversal; GGX VNDF sampling after Heitz); this file restates it for wave64, fp32 with IEEE division / square roots and no
// contraction (-ffp-contract=off), so that it decides every traversal step as the oracle does.
//
// Mapping.
//   * classify: a wave owns an 8 x 8 tile, lanes in the shader's Z-order (bit 0 = x0, bit 1 = y0, ...: a quad is lanes 4q..4q+3),
//     so subgroupQuadSwap{Horizontal,Vertical,Diagonal} are xor-shuffles by 1 / 2 / 3.  The shader appends rays with one
//     atomicAdd per ray, which makes the list order -- and with it which rays share a wave in the trace pass -- arbitrary; here
//     the list is in tile order (row-major), Z-order inside a tile: three short launches (count per tile, exclusive scan over the
//     tiles, emit at the scanned offsets), no atomics, the same list every run.  build_indirect.comp's result (the indirect
//     arguments, copied_count, atomic_count = 0) is written by the scan.
//   * trace: lane = ray, 64 consecutive rays per wave as in the shader.  The traversal loop runs under the hardware's own
//     divergence: the lanes still marching are the exec mask, so subgroupBallotBitCount(subgroupBallot(true)) is
//     popcount(__ballot(1)).  No indirect dispatch: a grid for the largest possible list is launched and waves beyond
//     copied_count leave at once.
//   * texelFetch outside a mip level returns 0 (robust image access), imageStore outside the image is dropped.
//   * On a denoised surface the two rays of a 2 x 2 quad both copy into both other pixels (one horizontally, one vertically); the
//     shader stores horizontal copies before vertical ones, so inside a wave the vertical copy stays, across waves it is a race.
//     Here the vertical copy always wins: a ray skips its horizontal copy when its diagonal neighbour is a listed ray.
//   * apply: one thread per pixel; the reference's blend ONE / ONE into the RGBA16F target is hdr = rne16(hdr + colour).
// trace_fallback.comp only runs with a volumetric-diffuse probe set bound (ssr.cpp:141), which is outside this path.
#include <cmath>
#include "ctx.hpp"
#include "device_common.hpp"

namespace
{
struct float3_ { float x, y, z; };
__device__ __forceinline__ float3_ f3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ float3_ operator+(float3_ a, float3_ b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float3_ operator-(float3_ a, float3_ b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float3_ operator*(float3_ a, float3_ b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float3_ operator*(float3_ a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float3_ operator*(float s, float3_ a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float3_ operator/(float3_ a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ float3_ operator-(float3_ a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float dot3(float3_ a, float3_ b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float length3(float3_ a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ float3_ normalize3(float3_ a) { const float l = length3(a); return {a.x / l, a.y / l, a.z / l}; }
__device__ __forceinline__ float3_ cross3(float3_ a, float3_ b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
__device__ __forceinline__ float smoothstepf(float e0, float e1, float x)
{
	const float t = fminf(fmaxf((x - e0) / (e1 - e0), 0.0f), 1.0f);
	return t * t * (3.0f - 2.0f * t);
}

struct SSRParams
{
	int width, height;
	const float *hier; // R32F chain
	int hier_w, hier_h, hier_levels;
	uint32_t hier_offset[16]; // float offset of level l
	DevImage pbr, normal, light;
	const uint16_t *noise; // RG8, 128 x 128 x 64
	const float2 *azimuth; // (cos, sin) of 2 pi u / 255 for the 256 values of the dither texture's second byte, from the host's libm
	int frame;
	float vp[16], inv_vp[16]; // column-major
	float camera[3];
	DevImageRW output, ray_length, confidence;
	uint32_t *ray_list, *ray_counter;
	uint32_t *tile_count, *tile_offset; // scratch: one dword per 8 x 8 tile each
	uint32_t *block_base;               // scratch: one dword per 1024 tiles
	int tiles_x, tiles_y;
};

__device__ __forceinline__ float load_depth(const SSRParams &p, int x, int y, int lod)
{
	if (lod < 0 || lod >= p.hier_levels)
		return 0.0f;
	const int w = max(p.hier_w >> lod, 1), h = max(p.hier_h >> lod, 1);
	if (x < 0 || y < 0 || x >= w || y >= h)
		return 0.0f;
	return p.hier[p.hier_offset[lod] + uint32_t(y) * uint32_t(w) + uint32_t(x)];
}

__device__ __forceinline__ float3_ load_normal(const SSRParams &p, int x, int y)
{
	if (x < 0 || y < 0 || x >= p.width || y >= p.height)
		return f3(-1.0f, -1.0f, -1.0f);
	const uint32_t v = *reinterpret_cast<const uint32_t *>(p.normal.ptr + size_t(y) * p.normal.pitch + size_t(x) * 4u);
	const float3_ n = f3(float(v & 1023u) / 1023.0f, float((v >> 10) & 1023u) / 1023.0f, float((v >> 20) & 1023u) / 1023.0f);
	return n * 2.0f - f3(1.0f, 1.0f, 1.0f);
}

__device__ __forceinline__ float load_roughness(const SSRParams &p, int x, int y)
{
	if (x < 0 || y < 0 || x >= p.width || y >= p.height)
		return 0.0f;
	const uint16_t v = *reinterpret_cast<const uint16_t *>(p.pbr.ptr + size_t(y) * p.pbr.pitch + size_t(x) * 2u);
	return float(v >> 8) / 255.0f;
}

__device__ __forceinline__ float3_ load_light(const SSRParams &p, int x, int y)
{
	if (x < 0 || y < 0 || x >= p.width || y >= p.height)
		return f3(0.0f, 0.0f, 0.0f);
	const f16x4 t = *reinterpret_cast<const f16x4 *>(p.light.ptr + size_t(y) * p.light.pitch + size_t(x) * 8u);
	return f3(float(t.x), float(t.y), float(t.z));
}

// GLSL M * v = sum of column_i * v_i, left to right
__device__ __forceinline__ void mul_mat4(const float *m, float x, float y, float z, float w, float out[4])
{
#pragma unroll
	for (int r = 0; r < 4; r++)
	{
		float v = m[r] * x;
		v = v + m[4 + r] * y;
		v = v + m[8 + r] * z;
		v = v + m[12 + r] * w;
		out[r] = v;
	}
}

__device__ __forceinline__ float3_ screen_to_world(const SSRParams &p, float3_ ndc)
{
	float w[4];
	mul_mat4(p.inv_vp, ndc.x, ndc.y, ndc.z, 1.0f, w);
	return f3(w[0], w[1], w[2]) / w[3];
}

__device__ __forceinline__ void unpack_z_order(uint32_t l, uint32_t &x, uint32_t &y)
{
	x = ((l >> 0) & 1u) | (((l >> 2) & 1u) << 1) | (((l >> 4) & 1u) << 2);
	y = ((l >> 1) & 1u) | (((l >> 3) & 1u) << 1) | (((l >> 5) & 1u) << 2);
}

// classify.comp:33-48 for one pixel: needs_ray, require_copy, is_base_ray
struct Classified
{
	bool needs_ray, require_copy, base_ray, inside;
	uint32_t gx, gy;
};
__device__ __forceinline__ Classified classify_pixel(const SSRParams &p, uint32_t tile, uint32_t lane)
{
	Classified c;
	uint32_t lx, ly;
	unpack_z_order(lane, lx, ly);
	c.gx = (tile % uint32_t(p.tiles_x)) * 8u + lx;
	c.gy = (tile / uint32_t(p.tiles_x)) * 8u + ly;
	c.inside = int(c.gx) < p.width && int(c.gy) < p.height;
	const float roughness = load_roughness(p, int(c.gx), int(c.gy));
	const bool reflective = load_depth(p, int(c.gx), int(c.gy), 0) < 1.0f; // IsReflective
	const bool glossy = roughness < 0.2f;                                   // IsGlossy
	bool ray = c.inside && glossy && reflective;
	const bool needs_denoiser = ray && !(roughness < 0.0001f);
	c.base_ray = ((c.gx ^ (uint32_t(p.frame) & 1u)) & 1u) == (c.gy & 1u);
	ray = ray && (!needs_denoiser || c.base_ray);
	c.needs_ray = ray;
	c.require_copy = !ray && needs_denoiser;
	return c;
}

__device__ __forceinline__ bool pixel_needs_ray(const SSRParams &p, int x, int y)
{
	if (x < 0 || y < 0 || x >= p.width || y >= p.height)
		return false;
	const float roughness = load_roughness(p, x, y);
	const bool ray = roughness < 0.2f && load_depth(p, x, y, 0) < 1.0f;
	const bool needs_denoiser = ray && !(roughness < 0.0001f);
	const bool base_ray = ((uint32_t(x) ^ (uint32_t(p.frame) & 1u)) & 1u) == (uint32_t(y) & 1u);
	return ray && (!needs_denoiser || base_ray);
}

// one wave per tile, four tiles per workgroup
__global__ __launch_bounds__(256) void k_ssr_classify_count(SSRParams p)
{
	const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (tile >= uint32_t(p.tiles_x * p.tiles_y))
		return;
	const Classified c = classify_pixel(p, tile, lane);
	const uint64_t rays = __ballot(c.needs_ray);
	if (lane == 0)
		p.tile_count[tile] = uint32_t(__popcll(rays));
	// "Clear out confidence texture here."
	if (c.inside)
	{
		*reinterpret_cast<uint2 *>(p.output.ptr + size_t(c.gy) * p.output.pitch + size_t(c.gx) * 8u) = make_uint2(0u, 0u);
		p.confidence.ptr[size_t(c.gy) * p.confidence.pitch + c.gx] = 0;
	}
}

// Exclusive scan of the tile counts in two levels, every access coalesced: k_ssr_scan_blocks scans 1024 consecutive counts per
// workgroup (local offsets + one block sum), k_ssr_scan_top scans the block sums in one workgroup and is build_indirect.comp.
constexpr uint32_t SCAN_BLOCK = 1024; // counts per workgroup: 256 threads x 4
constexpr uint32_t SCAN_MAX_BLOCKS = 1024;

// Exclusive scan of one value per thread over a 256-thread workgroup; returns the exclusive prefix, `total` = the sum.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t value, uint32_t *wave_sums, uint32_t &total)
{
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t inclusive = value;
#pragma unroll
	for (int step = 1; step < 64; step <<= 1)
	{
		const uint32_t up = __shfl_up(inclusive, step, 64);
		if (lane >= uint32_t(step))
			inclusive += up;
	}
	if (lane == 63u)
		wave_sums[wave] = inclusive;
	__syncthreads();
	uint32_t base = 0;
	for (uint32_t w = 0; w < wave; w++)
		base += wave_sums[w];
	total = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
	return base + inclusive - value;
}

__global__ __launch_bounds__(256) void k_ssr_scan_blocks(SSRParams p)
{
	__shared__ uint32_t wave_sums[4];
	const uint32_t tiles = uint32_t(p.tiles_x * p.tiles_y);
	const uint32_t first = blockIdx.x * SCAN_BLOCK + threadIdx.x * 4u;
	uint32_t c[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
		c[k] = first + k < tiles ? p.tile_count[first + k] : 0u;
	uint32_t total;
	uint32_t running = block_exclusive_scan_256(c[0] + c[1] + c[2] + c[3], wave_sums, total);
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		if (first + k < tiles)
			p.tile_offset[first + k] = running; // local to the block; the emit pass adds block_base
		running += c[k];
	}
	if (threadIdx.x == 0)
		p.block_base[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_ssr_scan_top(SSRParams p, uint32_t blocks)
{
	__shared__ uint32_t wave_sums[4];
	const uint32_t first = threadIdx.x * 4u;
	uint32_t c[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
		c[k] = first + k < blocks ? p.block_base[first + k] : 0u;
	uint32_t count;
	uint32_t running = block_exclusive_scan_256(c[0] + c[1] + c[2] + c[3], wave_sums, count);
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		if (first + k < blocks)
			p.block_base[first + k] = running;
		running += c[k];
	}
	if (threadIdx.x == 0)
	{
		p.ray_counter[0] = (count + 63u) / 64u; // indirect
		p.ray_counter[1] = 1u;
		p.ray_counter[2] = 1u;
		p.ray_counter[3] = 0u;
		p.ray_counter[4] = 0u;    // atomic_count, reset
		p.ray_counter[5] = count; // copied_count
	}
}

__global__ __launch_bounds__(256) void k_ssr_classify_emit(SSRParams p)
{
	const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (tile >= uint32_t(p.tiles_x * p.tiles_y))
		return;
	const Classified c = classify_pixel(p, tile, lane);
	// subgroupQuadSwap{Horizontal, Vertical, Diagonal}(require_copy)
	const int rc = c.require_copy ? 1 : 0;
	const bool horiz = __shfl_xor(rc, 1, 64) != 0, vert = __shfl_xor(rc, 2, 64) != 0, diag = __shfl_xor(rc, 3, 64) != 0;
	const uint64_t rays = __ballot(c.needs_ray);
	if (c.needs_ray)
	{
		const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(rays >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(rays), 0u));
		// PackRay
		p.ray_list[p.block_base[tile / SCAN_BLOCK] + p.tile_offset[tile] + rank] = c.gx | (c.gy << 14u) | (uint32_t(c.base_ray && horiz) << 28u) | (uint32_t(c.base_ray && vert) << 29u) |
		                                         (uint32_t(c.base_ray && diag) << 30u);
	}
}

// ---- sssr_util.h:55-143 ---------------------------------------------------------------------------------------------------

// U2 enters only through cos / sin of phi = 2 pi U2, and U2 is a byte / 255: `trig` is (cos phi, sin phi) from the host-built table,
// so no device approximation of a transcendental is on the path that decides where a ray goes (sqrtf and / are correctly rounded).
__device__ __forceinline__ float3_ sample_ggx_vndf(float3_ Ve, float alpha_x, float alpha_y, float U1, float2 trig)
{
	const float3_ Vh = normalize3(f3(alpha_x * Ve.x, alpha_y * Ve.y, Ve.z));
	const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
	const float3_ T1 = lensq > 0.0f ? f3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : f3(1.0f, 0.0f, 0.0f);
	const float3_ T2 = cross3(Vh, T1);
	const float r = sqrtf(U1);
	const float t1 = r * trig.x;
	float t2 = r * trig.y;
	const float s = 0.5f * (1.0f + Vh.z);
	t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
	const float3_ Nh = t1 * T1 + t2 * T2 + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
	return normalize3(f3(alpha_x * Nh.x, alpha_y * Nh.y, fmaxf(0.0f, Nh.z)));
}

__device__ __forceinline__ float3_ sample_reflection_vector(const SSRParams &p, float3_ view_direction, float3_ N, float roughness, int px, int py)
{
	// CreateTBN
	float3_ U;
	if (fabsf(N.z) > 0.0f)
	{
		const float k = sqrtf(N.y * N.y + N.z * N.z);
		U = f3(0.0f, -N.z / k, N.y / k);
	}
	else
	{
		const float k = sqrtf(N.x * N.x + N.y * N.y);
		U = f3(N.y / k, -N.x / k, 0.0f);
	}
	const float3_ c0 = U, c1 = cross3(N, U), c2 = N;
	const float3_ nv = -view_direction;
	const float3_ view_tbn = f3(dot3(nv, c0), dot3(nv, c1), dot3(nv, c2)); // vector * matrix
	const uint16_t noise = p.noise[(size_t(p.frame) * 128u + size_t(py & 127)) * 128u + size_t(px & 127)];
	const float u1 = float(noise & 255u) / 255.0f;
	const float3_ sampled = sample_ggx_vndf(view_tbn, roughness, roughness, u1, p.azimuth[noise >> 8]);
	const float3_ incident = -view_tbn;
	const float3_ reflected_tbn = incident - sampled * (2.0f * dot3(sampled, incident)); // reflect(I, N)
	float3_ r = c0 * reflected_tbn.x; // matrix * vector
	r = r + c1 * reflected_tbn.y;
	r = r + c2 * reflected_tbn.z;
	return r;
}

constexpr float SSR_FLOAT_MAX = 3.402823466e+38f;

__device__ __forceinline__ float validate_hit(const SSRParams &p, float3_ hit, float uvx, float uvy, float3_ ray_direction, float thickness,
                                              float inv_res_x, float inv_res_y)
{
	if (hit.x < 0.0f || hit.y < 0.0f || 1.0f < hit.x || 1.0f < hit.y)
		return 0.0f;
	if (fabsf(hit.x - uvx) < 2.0f * inv_res_x && fabsf(hit.y - uvy) < 2.0f * inv_res_y)
		return 0.0f;
	const int tx = int(float(p.width) * hit.x), ty = int(float(p.height) * hit.y);
	const float surface_z = load_depth(p, tx / 2, ty / 2, 1);
	if (surface_z == 1.0f)
		return 1.0f;
	const float3_ hit_normal = load_normal(p, tx, ty);
	if (dot3(hit_normal, ray_direction) > 0.0f)
		return 0.0f;
	const float3_ surface = screen_to_world(p, f3(hit.x, hit.y, surface_z));
	const float3_ hit_world = screen_to_world(p, hit);
	const float dist = length3(surface - hit_world);
	const float fov_x = float(p.height) * inv_res_x * 0.05f, fov_y = 0.05f;
	const float border_x = smoothstepf(0.0f, fov_x, hit.x) * (1.0f - smoothstepf(1.0f - fov_x, 1.0f, hit.x));
	const float border_y = smoothstepf(0.0f, fov_y, hit.y) * (1.0f - smoothstepf(1.0f - fov_y, 1.0f, hit.y));
	const float vignette = border_x * border_y;
	float confidence = 1.0f - smoothstepf(0.0f, thickness, dist);
	confidence *= confidence;
	return vignette * confidence;
}

__device__ __forceinline__ void store_result(const SSRParams &p, int x, int y, float3_ color, float ray_len, float confidence)
{
	if (x < 0 || y < 0 || x >= p.width || y >= p.height)
		return;
	f16x4 o;
	o.x = _Float16(color.x);
	o.y = _Float16(color.y);
	o.z = _Float16(color.z);
	o.w = _Float16(0.0f);
	*reinterpret_cast<f16x4 *>(p.output.ptr + size_t(y) * p.output.pitch + size_t(x) * 8u) = o;
	*reinterpret_cast<_Float16 *>(p.ray_length.ptr + size_t(y) * p.ray_length.pitch + size_t(x) * 2u) = _Float16(ray_len);
	// R8_UNORM store: round(clamp(v, 0, 1) * 255); NaN -> 0
	uint8_t q = 0;
	if (confidence > 0.0f)
		q = confidence >= 1.0f ? uint8_t(255) : uint8_t(int(confidence * 255.0f + 0.5f));
	p.confidence.ptr[size_t(y) * p.confidence.pitch + size_t(x)] = q;
}

__global__ __launch_bounds__(64) void k_ssr_trace(SSRParams p)
{
	const uint32_t count = p.ray_counter[5]; // copied_count
	const uint32_t index = blockIdx.x * 64u + threadIdx.x;
	if (blockIdx.x * 64u >= count)
		return;
	const bool listed = index < count;
	const float res_x = float(p.width), res_y = float(p.height);
	const float inv_res_x = 1.0f / res_x, inv_res_y = 1.0f / res_y;
	const int most_detailed_mip = 1;
	const uint32_t min_occupancy = 4, max_intersections = 128;
	const float thickness = 0.05f;

	int cx = 0, cy = 0;
	bool copy_h = false, copy_v = false, copy_d = false, marching = false, early_out = false, is_mirror = false;
	float uvx = 0.0f, uvy = 0.0f;
	float3_ world_pos = f3(0, 0, 0), reflected = f3(0, 0, 0), origin = f3(0, 0, 0), direction = f3(0, 0, 0);
	if (listed)
	{
		const uint32_t word = p.ray_list[index];
		cx = int(word & 0x3fffu);
		cy = int((word >> 14) & 0x3fffu);
		copy_h = ((word >> 28) & 1u) != 0;
		copy_v = ((word >> 29) & 1u) != 0;
		copy_d = ((word >> 30) & 1u) != 0;
		uvx = (float(cx) + 0.5f) * inv_res_x;
		uvy = (float(cy) + 0.5f) * inv_res_y;
		const float clip_x = 2.0f * uvx - 1.0f, clip_y = 2.0f * uvy - 1.0f;
		const float clip_depth = load_depth(p, cx, cy, 0);
		if (clip_depth == 1.0f)
			early_out = true;
		else
		{
			const float roughness = load_roughness(p, cx, cy);
			world_pos = screen_to_world(p, f3(clip_x, clip_y, clip_depth));
			const float3_ V = normalize3(f3(p.camera[0], p.camera[1], p.camera[2]) - world_pos);
			const float3_ N = load_normal(p, cx, cy);
			reflected = sample_reflection_vector(p, -V, N, roughness, cx, cy);
			// project_direction_to_clip_space
			float clip_d[4];
			mul_mat4(p.vp, reflected.x, reflected.y, reflected.z, 0.0f, clip_d);
			float3_ dir = normalize3(f3(clip_d[0], clip_d[1], clip_d[2]) - f3(clip_x, clip_y, clip_depth) * clip_d[3]);
			dir.x *= 0.5f;
			dir.y *= 0.5f;
			origin = f3(uvx, uvy, clip_depth);
			direction = dir;
			is_mirror = roughness < 0.0001f;
			marching = true;
		}
	}

	// ---- FFX_SSSR_HierarchicalRaymarch ----
	float3_ position = origin;
	int i = 0;
	if (marching)
	{
		const float3_ inv_direction = f3(direction.x != 0.0f ? 1.0f / direction.x : SSR_FLOAT_MAX, direction.y != 0.0f ? 1.0f / direction.y : SSR_FLOAT_MAX,
		                                 direction.z != 0.0f ? 1.0f / direction.z : SSR_FLOAT_MAX);
		int mip = most_detailed_mip;
		float mip_res_x = res_x * ldexpf(1.0f, -mip), mip_res_y = res_y * ldexpf(1.0f, -mip);
		float mip_res_inv_x = 1.0f / mip_res_x, mip_res_inv_y = 1.0f / mip_res_y;
		const float off = 0.005f * exp2f(float(most_detailed_mip));
		const float uv_off_x = direction.x < 0.0f ? -(inv_res_x * off) : inv_res_x * off;
		const float uv_off_y = direction.y < 0.0f ? -(inv_res_y * off) : inv_res_y * off;
		const float floor_off_x = direction.x < 0.0f ? 0.0f : 1.0f, floor_off_y = direction.y < 0.0f ? 0.0f : 1.0f;
		float current_t;
		{
			// FFX_SSSR_InitialAdvanceRay
			const float mx = mip_res_x * origin.x, my = mip_res_y * origin.y;
			const float px = (floorf(mx) + floor_off_x) * mip_res_inv_x + uv_off_x;
			const float py = (floorf(my) + floor_off_y) * mip_res_inv_y + uv_off_y;
			const float tx = px * inv_direction.x - origin.x * inv_direction.x;
			const float ty = py * inv_direction.y - origin.y * inv_direction.y;
			current_t = fminf(tx, ty);
			position = origin + current_t * direction;
		}
		bool exit_low = false;
		while (uint32_t(i) < max_intersections && mip >= most_detailed_mip && !exit_low)
		{
			const float mx = mip_res_x * position.x, my = mip_res_y * position.y;
			const float surface_z = load_depth(p, int(mx), int(my), mip);
			const uint32_t active_lanes = uint32_t(__popcll(__ballot(1))); // the lanes still inside this loop
			exit_low = !is_mirror && active_lanes <= min_occupancy;
			// FFX_SSSR_AdvanceRay
			const float px = (floorf(mx) + floor_off_x) * mip_res_inv_x + uv_off_x;
			const float py = (floorf(my) + floor_off_y) * mip_res_inv_y + uv_off_y;
			const float tx = px * inv_direction.x - origin.x * inv_direction.x;
			const float ty = py * inv_direction.y - origin.y * inv_direction.y;
			float tz = surface_z * inv_direction.z - origin.z * inv_direction.z;
			tz = direction.z > 0.0f ? tz : SSR_FLOAT_MAX;
			const float t_min = fminf(fminf(tx, ty), tz);
			const bool above_surface = surface_z > position.z;
			const bool skipped_tile = __builtin_bit_cast(uint32_t, t_min) != __builtin_bit_cast(uint32_t, tz) && above_surface;
			current_t = above_surface ? t_min : current_t;
			position = origin + current_t * direction;
			mip += skipped_tile ? 1 : -1;
			mip_res_x *= skipped_tile ? 0.5f : 2.0f;
			mip_res_y *= skipped_tile ? 0.5f : 2.0f;
			mip_res_inv_x *= skipped_tile ? 2.0f : 0.5f;
			mip_res_inv_y *= skipped_tile ? 2.0f : 0.5f;
			++i;
		}
	}
	if (!listed)
		return;

	float confidence = 0.0f, ray_len = 0.0f;
	float3_ color = f3(0.0f, 0.0f, 0.0f);
	if (!early_out)
	{
		const bool valid_hit = uint32_t(i) <= max_intersections;
		float3_ result = position;
		confidence = valid_hit ? validate_hit(p, result, uvx, uvy, reflected, thickness, inv_res_x, inv_res_y) : 0.0f;
		if (confidence > 0.0f)
		{
			const int tx = int(res_x * result.x), ty = int(res_y * result.y);
			color = load_light(p, tx, ty) * confidence;
			result.x = result.x * 2.0f - 1.0f;
			result.y = result.y * 2.0f - 1.0f;
			const float3_ hit_pos = screen_to_world(p, result);
			ray_len = length3(world_pos - hit_pos);
		}
	}
	color = color + load_light(p, cx, cy);
	store_result(p, cx, cy, color, ray_len, confidence);
	if (copy_h && !pixel_needs_ray(p, cx ^ 1, cy ^ 1)) // the diagonal ray's vertical copy owns that pixel
		store_result(p, cx ^ 1, cy, color, ray_len, confidence);
	if (copy_v)
		store_result(p, cx, cy ^ 1, color, ray_len, confidence);
	if (copy_d)
		store_result(p, cx ^ 1, cy ^ 1, color, ray_len, confidence);
}

struct ApplyParams
{
	int width, height;
	DevImage reflected, albedo, normal, pbr, depth;
	const uint16_t *brdf_lut; // RG16F
	int lut_w, lut_h;
	const float *srgb_lut;
	float inv_vp[16];
	float camera[3];
	DevImageRW hdr;
};

__global__ __launch_bounds__(256) void k_ssr_apply(ApplyParams p)
{
	const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
	if (x >= p.width || y >= p.height)
		return;
	const float clip_depth = *reinterpret_cast<const float *>(p.depth.ptr + size_t(y) * p.depth.pitch + size_t(x) * 4u);
	if (clip_depth == 1.0f)
		return; // depth test NOT_EQUAL against the quad at z = 1
	const uint16_t mr = *reinterpret_cast<const uint16_t *>(p.pbr.ptr + size_t(y) * p.pbr.pitch + size_t(x) * 2u);
	const float metallic = float(mr & 255u) / 255.0f, roughness = float(mr >> 8) / 255.0f;
	const float vu = (float(x) + 0.5f) * (1.0f / float(p.width)), vv = (float(y) + 0.5f) * (1.0f / float(p.height));
	const float clip_x = vu * 2.0f - 1.0f, clip_y = vv * 2.0f - 1.0f;
	float w[4];
	{
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
			float v = p.inv_vp[r] * clip_x;
			v = v + p.inv_vp[4 + r] * clip_y;
			v = v + p.inv_vp[8 + r] * clip_depth;
			v = v + p.inv_vp[12 + r] * 1.0f;
			w[r] = v;
		}
	}
	const float3_ world_pos = f3(w[0], w[1], w[2]) / w[3];
	const float3_ V = normalize3(f3(p.camera[0], p.camera[1], p.camera[2]) - world_pos);
	const uint32_t nv = *reinterpret_cast<const uint32_t *>(p.normal.ptr + size_t(y) * p.normal.pitch + size_t(x) * 4u);
	const float3_ N = normalize3(f3(float(nv & 1023u) / 1023.0f, float((nv >> 10) & 1023u) / 1023.0f, float((nv >> 20) & 1023u) / 1023.0f) * 2.0f - f3(1.0f, 1.0f, 1.0f));
	const float NoV = fminf(fmaxf(dot3(N, V), 0.0f), 1.0f);
	const uint32_t al = *reinterpret_cast<const uint32_t *>(p.albedo.ptr + size_t(y) * p.albedo.pitch + size_t(x) * 4u);
	const float3_ base = f3(p.srgb_lut[al & 255u], p.srgb_lut[(al >> 8) & 255u], p.srgb_lut[(al >> 16) & 255u]);
	// compute_F0: mix(vec3(0.04), base, metallic) = 0.04 (1 - m) + base m
	const float3_ F0 = f3(0.04f * (1.0f - metallic) + base.x * metallic, 0.04f * (1.0f - metallic) + base.y * metallic, 0.04f * (1.0f - metallic) + base.z * metallic);
	// fresnel_ibl
	const float omr = 1.0f - roughness;
	const float fp = powf(1.0f - NoV, 5.0f);
	const float3_ F = F0 + (f3(fmaxf(omr, F0.x), fmaxf(omr, F0.y), fmaxf(omr, F0.z)) - F0) * fp;
	// textureLod(uBRDFLut, vec2(NoV, roughness), 0): LinearClamp
	int ix, iy;
	float wa, wb;
	linear_axis(NoV * float(p.lut_w) - 0.5f, ix, wa);
	linear_axis(roughness * float(p.lut_h) - 0.5f, iy, wb);
	const int x0 = clampi(ix, 0, p.lut_w - 1), x1 = clampi(ix + 1, 0, p.lut_w - 1);
	const int y0 = clampi(iy, 0, p.lut_h - 1), y1 = clampi(iy + 1, 0, p.lut_h - 1);
	auto lut = [&](int lx, int ly, int c) { return float(__builtin_bit_cast(_Float16, p.brdf_lut[(size_t(ly) * p.lut_w + lx) * 2 + c])); };
	float brdf[2];
#pragma unroll
	for (int c = 0; c < 2; c++)
	{
		const float top = lut(x0, y0, c) * (1.0f - wa) + lut(x1, y0, c) * wa;
		const float bot = lut(x0, y1, c) * (1.0f - wa) + lut(x1, y1, c) * wa;
		brdf[c] = top * (1.0f - wb) + bot * wb;
	}
	const f16x4 r = *reinterpret_cast<const f16x4 *>(p.reflected.ptr + size_t(y) * p.reflected.pitch + size_t(x) * 8u);
	const float3_ color = f3(float(r.x), float(r.y), float(r.z)) * (F * brdf[0] + f3(brdf[1], brdf[1], brdf[1]));
	f16x4 *dst = reinterpret_cast<f16x4 *>(p.hdr.ptr + size_t(y) * p.hdr.pitch + size_t(x) * 8u);
	f16x4 d = *dst;
	d.x = _Float16(float(d.x) + color.x);
	d.y = _Float16(float(d.y) + color.y);
	d.z = _Float16(float(d.z) + color.z);
	*dst = d;
}

bool image_ok(const gr_image &img, uint32_t format, uint32_t bpp, uint32_t w, uint32_t h)
{
	return img.ptr && img.format == format && img.width == w && img.height == h && img.pitch_bytes >= w * bpp && (img.pitch_bytes % bpp) == 0;
}
DevImage dev(const gr_image &i) { return DevImage{static_cast<const uint8_t *>(i.ptr), int(i.width), int(i.height), i.pitch_bytes}; }
DevImageRW dev_rw(const gr_image &i) { return DevImageRW{static_cast<uint8_t *>(i.ptr), int(i.width), int(i.height), i.pitch_bytes}; }
} // namespace

extern "C" {

size_t gr_ssr_scratch_bytes(uint32_t width, uint32_t height)
{
	const size_t tiles = size_t((width + 7u) / 8u) * size_t((height + 7u) / 8u);
	return (tiles * 2u + (tiles + SCAN_BLOCK - 1) / SCAN_BLOCK) * sizeof(uint32_t);
}

int gr_ssr_trace(gr_ctx *ctx, gr_stream stream, const gr_ssr_args *args)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, args != nullptr);
	const uint32_t W = args->output.width, H = args->output.height;
	GR_CHECK_ARG(ctx, W != 0 && H != 0 && W < 16384u && H < 16384u); // PackRay: 14 bits per coordinate
	GR_CHECK_ARG(ctx, image_ok(args->output, GR_FORMAT_R16G16B16A16_SFLOAT, 8, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->ray_length, GR_FORMAT_R16_SFLOAT, 2, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->ray_confidence, GR_FORMAT_R8_UNORM, 1, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->light, GR_FORMAT_R16G16B16A16_SFLOAT, 8, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->normal, GR_FORMAT_A2B10G10R10_UNORM_PACK32, 4, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->pbr, GR_FORMAT_R8G8_UNORM, 2, W, H));
	GR_CHECK_ARG(ctx, args->depth_chain && args->chain_levels >= 1 && args->chain_levels <= 16 && args->chain_width >= W && args->chain_height >= H);
	GR_CHECK_ARG(ctx, args->dither_lut && args->ray_list && args->ray_counter && args->scratch);
	GR_CHECK_ARG(ctx, args->frame < 64u);

	SSRParams p = {};
	p.width = int(W);
	p.height = int(H);
	p.hier = static_cast<const float *>(args->depth_chain);
	p.hier_w = int(args->chain_width);
	p.hier_h = int(args->chain_height);
	p.hier_levels = int(args->chain_levels);
	for (uint32_t l = 0; l < args->chain_levels; l++)
		p.hier_offset[l] = uint32_t(gr_mip_chain_offset(args->chain_width, args->chain_height, 4, l) / 4);
	p.pbr = dev(args->pbr);
	p.normal = dev(args->normal);
	p.light = dev(args->light);
	p.noise = static_cast<const uint16_t *>(args->dither_lut);
	p.azimuth = ctx->ssr_azimuth_lut;
	p.frame = int(args->frame);
	for (int i = 0; i < 16; i++)
	{
		p.vp[i] = args->view_projection[i];
		p.inv_vp[i] = args->inv_view_projection[i];
	}
	for (int i = 0; i < 3; i++)
		p.camera[i] = args->camera_position[i];
	p.output = dev_rw(args->output);
	p.ray_length = dev_rw(args->ray_length);
	p.confidence = dev_rw(args->ray_confidence);
	p.ray_list = args->ray_list;
	p.ray_counter = args->ray_counter;
	p.tiles_x = int((W + 7u) / 8u);
	p.tiles_y = int((H + 7u) / 8u);
	const uint32_t tiles = uint32_t(p.tiles_x) * uint32_t(p.tiles_y);
	p.tile_count = static_cast<uint32_t *>(args->scratch);
	p.tile_offset = p.tile_count + tiles;
	p.block_base = p.tile_offset + tiles;
	const uint32_t scan_blocks = gr_div_up(tiles, SCAN_BLOCK);
	GR_CHECK_ARG(ctx, scan_blocks <= SCAN_MAX_BLOCKS); // 1 Mi tiles = 67 M pixels

	hipStream_t s = gr_to_stream(stream);
	{
		gr_scoped_timing timing{ctx, s, "ssr_classify"};
		hipLaunchKernelGGL(k_ssr_classify_count, dim3(gr_div_up(tiles, 4u)), dim3(256), 0, s, p);
		hipLaunchKernelGGL(k_ssr_scan_blocks, dim3(scan_blocks), dim3(256), 0, s, p);
		hipLaunchKernelGGL(k_ssr_scan_top, dim3(1), dim3(256), 0, s, p, scan_blocks);
		hipLaunchKernelGGL(k_ssr_classify_emit, dim3(gr_div_up(tiles, 4u)), dim3(256), 0, s, p);
	}
	{
		// dispatch_indirect(ray_counter): the list can hold at most one ray per pixel
		gr_scoped_timing timing{ctx, s, "ssr_trace"};
		hipLaunchKernelGGL(k_ssr_trace, dim3(gr_div_up(W * H, 64u)), dim3(64), 0, s, p);
	}
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_ssr_apply(gr_ctx *ctx, gr_stream stream, const gr_ssr_apply_args *args)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, args != nullptr);
	const uint32_t W = args->hdr.width, H = args->hdr.height;
	GR_CHECK_ARG(ctx, W != 0 && H != 0);
	GR_CHECK_ARG(ctx, image_ok(args->hdr, GR_FORMAT_R16G16B16A16_SFLOAT, 8, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->reflected, GR_FORMAT_R16G16B16A16_SFLOAT, 8, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->albedo, GR_FORMAT_R8G8B8A8_SRGB, 4, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->normal, GR_FORMAT_A2B10G10R10_UNORM_PACK32, 4, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->pbr, GR_FORMAT_R8G8_UNORM, 2, W, H));
	GR_CHECK_ARG(ctx, image_ok(args->depth, GR_FORMAT_D32_SFLOAT, 4, W, H));
	GR_CHECK_ARG(ctx, args->brdf_lut.ptr && args->brdf_lut.format == GR_FORMAT_R16G16_SFLOAT && args->brdf_lut.width && args->brdf_lut.height &&
	                      args->brdf_lut.pitch_bytes == args->brdf_lut.width * 4u);
	ApplyParams p = {};
	p.width = int(W);
	p.height = int(H);
	p.reflected = dev(args->reflected);
	p.albedo = dev(args->albedo);
	p.normal = dev(args->normal);
	p.pbr = dev(args->pbr);
	p.depth = dev(args->depth);
	p.brdf_lut = static_cast<const uint16_t *>(args->brdf_lut.ptr);
	p.lut_w = int(args->brdf_lut.width);
	p.lut_h = int(args->brdf_lut.height);
	p.srgb_lut = ctx->srgb_decode_lut;
	for (int i = 0; i < 16; i++)
		p.inv_vp[i] = args->inv_view_projection[i];
	for (int i = 0; i < 3; i++)
		p.camera[i] = args->camera_position[i];
	p.hdr = dev_rw(args->hdr);
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "ssr_apply"};
	hipLaunchKernelGGL(k_ssr_apply, dim3(gr_div_up(W, 32u), gr_div_up(H, 8u)), dim3(256), 0, gr_to_stream(stream), p);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}
}
