// Clustered deferred lighting kernel for gfx950 and its C-ABI launcher (include/granite_hip.h).
//
// Replaces DeferredLightRenderer::render_light (renderer/renderer.cpp:1004-1156): the directional full-screen quad
// (assets/shaders/lights/directional.frag + lighting.h) followed by the clustered quad (clustering.frag +
// clusterer_bindless.h + point.h + spot.h + pbr.h), both blended ONE/ONE into the RGBA16F HDR target with depth test
// NOT_EQUAL against z = 0.
//
// One kernel reads the four G-buffer attachments + emissive once (22 B/px) and writes HDR once (8 B/px); emissive and
// HDR may be the same buffer (the reference's blend read-modify-write) or distinct ones (same values, same bytes).  The two
// blend roundings of the reference are reproduced in registers: hdr = rne16(rne16(emissive + directional) + clustered).
//
// Wave mapping: a wave64 owns a 16x4 pixel tile (128 B HDR / 64 B albedo row segments).  The light loop is wave-uniform
// exactly like the reference's subgroup path (clusterer_bindless.h:49-81): per 32-light word the lanes' range-trimmed
// cell masks are OR-reduced across the wave, the union is walked with scalar bit ops, and each light record is fetched
// with scalar loads (SGPR-resident, no LDS traffic) while all 64 lanes shade it.
#include "ctx.hpp"
#include "device_common.hpp"

namespace
{
constexpr int LIGHT_TILE_W = 16;
constexpr int LIGHT_TILE_H = 4; // per wave
constexpr int LIGHT_WAVES = 4;  // waves per workgroup, stacked vertically -> 16x16 block tile

constexpr float PI_SIC = 3.1415628f; // assets/shaders/lights/pbr.h:4-6 (sic)

struct KernelArgs
{
	DevImage albedo, normal, pbr, depth, emissive;
	DevImageRW hdr;
	float inv_vp[16];
	float camera_pos[3];
	float dir_color[3];
	float dir_direction[3];
	float inv_resolution[2];
	// cluster UBO subset
	float cl_camera_base[3];
	float cl_camera_front[3];
	float cl_xy_scale[2];
	int cl_res_x, cl_res_y;
	int cl_num_lights, cl_num_lights_32, cl_z_max_index;
	float cl_z_scale;
	const gr_light_info *lights;
	const uint32_t *type_mask;
	const uint32_t *bitmask;
	const uint2 *range;
	const float *srgb_lut;
	uint32_t flags;
};

struct float3_ { float x, y, z; };
__device__ __forceinline__ float3_ f3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ float3_ operator+(float3_ a, float3_ b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float3_ operator-(float3_ a, float3_ b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float3_ operator*(float3_ a, float3_ b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float3_ operator*(float3_ a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(float3_ a, float3_ b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float rsq(float v) { return __builtin_amdgcn_rsqf(v); }

// Per-pixel material terms hoisted out of the light loop.
struct Surface
{
	float3_ pos, N, V, F0, diffuse; // diffuse = base * (1 - metallic) / PI
	float NoV, m2, k, Gv;
};

// Shared BRDF tail of compute_point_light / compute_spot_light / compute_lighting
// (point.h:119-142, spot.h:122-145, lighting.h:26-45): returns NoL * (spec + diffuse) for light direction L.
__device__ __forceinline__ float3_ shade(const Surface &s, float3_ L)
{
	float3_ H = s.V + L;
	H = H * rsq(dot(H, H));
	const float NoL = clampf(dot(s.N, L), 0.001f, 1.0f);
	const float HoV = clampf(dot(H, s.V), 0.001f, 1.0f);
	const float NoH = clampf(dot(s.N, H), 0.0001f, 1.0f);

	const float omh = 1.0f - HoV;
	const float omh2 = omh * omh;
	const float f = omh2 * omh2 * omh; // pow(1 - HoV, 5)
	const float omf = 1.0f - f;
	const float3_ F = f3(fmaf(s.F0.x, omf, f), fmaf(s.F0.y, omf, f), fmaf(s.F0.z, omf, f)); // mix(F0, 1, f)

	const float d = fmaf(fmaf(NoH, s.m2, -NoH), NoH, 1.0f);
	const float D = s.m2 * rcp(PI_SIC * d * d);
	const float Gl = fmaf(NoL, 1.0f - s.k, s.k);
	const float G = 0.25f * rcp(fmaxf(s.Gv * Gl, 0.001f));
	const float GD = G * D;

	return f3(NoL * fmaf(F.x, GD, (1.0f - F.x) * s.diffuse.x), NoL * fmaf(F.y, GD, (1.0f - F.y) * s.diffuse.y),
	          NoL * fmaf(F.z, GD, (1.0f - F.z) * s.diffuse.z));
}

__device__ __forceinline__ float smooth_falloff(float x)
{
	// 1 - smoothstep(0.9, 1.0, x)
	const float t = clampf((x - 0.9f) * (1.0f / (1.0f - 0.9f)), 0.0f, 1.0f);
	return 1.0f - t * t * (3.0f - 2.0f * t);
}

// clusterer_bindless_buffers.h:17-27
__device__ __forceinline__ uint32_t cluster_mask_range(uint32_t mask, uint2 range, uint32_t start_index)
{
	const uint32_t rx = min(max(range.x, start_index), start_index + 32u);
	const uint32_t ry = min(max(range.y + 1u, rx), start_index + 32u);
	const uint32_t num_bits = ry - rx;
	const uint32_t range_mask = num_bits == 32u ? 0xffffffffu : ((1u << num_bits) - 1u) << (rx - start_index);
	return mask & range_mask;
}

__global__ __launch_bounds__(64 * LIGHT_WAVES) void k_lighting(KernelArgs a)
{
	const int wave = threadIdx.x >> 6;
	const int lane = threadIdx.x & 63;
	const int x = blockIdx.x * LIGHT_TILE_W + (lane & (LIGHT_TILE_W - 1));
	const int y = (blockIdx.y * LIGHT_WAVES + wave) * LIGHT_TILE_H + (lane / LIGHT_TILE_W);
	const int W = a.hdr.w, H = a.hdr.h;
	const bool inside = x < W && y < H;

	float depth = 0.0f;
	if (inside)
		depth = *reinterpret_cast<const float *>(a.depth.ptr + size_t(y) * a.depth.pitch + size_t(x) * 4u);
	// depth test NOT_EQUAL against the quad at z = 0 (renderer.cpp:1056-1057): reverse-Z far plane untouched.
	const bool active = inside && depth != 0.0f;
	if (inside && !active && a.emissive.ptr != a.hdr.ptr)
	{
		// Far-plane pixel: the draws are depth-rejected, the target keeps the emissive value.
		*reinterpret_cast<f16x4 *>(a.hdr.ptr + size_t(y) * a.hdr.pitch + size_t(x) * 8u) =
		    *reinterpret_cast<const f16x4 *>(a.emissive.ptr + size_t(y) * a.emissive.pitch + size_t(x) * 8u);
	}
	if (!__any(active))
		return;

	uint32_t alb = 0, nrm = 0, mr = 0;
	f16x4 dst = {0, 0, 0, 0};
	if (active)
	{
		alb = *reinterpret_cast<const uint32_t *>(a.albedo.ptr + size_t(y) * a.albedo.pitch + size_t(x) * 4u);
		nrm = *reinterpret_cast<const uint32_t *>(a.normal.ptr + size_t(y) * a.normal.pitch + size_t(x) * 4u);
		mr = *reinterpret_cast<const uint16_t *>(a.pbr.ptr + size_t(y) * a.pbr.pitch + size_t(x) * 2u);
		dst = *reinterpret_cast<const f16x4 *>(a.emissive.ptr + size_t(y) * a.emissive.pitch + size_t(x) * 8u);
	}

	// ---- G-buffer decode (clustering.frag:31-35) ----
	const float3_ base = f3(a.srgb_lut[alb & 255u], a.srgb_lut[(alb >> 8) & 255u], a.srgb_lut[(alb >> 16) & 255u]);
	const float3_ N = f3(float(nrm & 1023u) * (2.0f / 1023.0f) - 1.0f, float((nrm >> 10) & 1023u) * (2.0f / 1023.0f) - 1.0f,
	                     float((nrm >> 20) & 1023u) * (2.0f / 1023.0f) - 1.0f);
	const float metallic = float(mr & 255u) * (1.0f / 255.0f);
	const float mat_roughness = float(mr >> 8) * (1.0f / 255.0f);

	// ---- position reconstruction (clustering.vert:10-13, clustering.frag:37-39); plain mul/add, no contraction,
	// so the cell / slice selection below is reproducible. ----
	const float ndc_x = __fsub_rn(__fmul_rn(2.0f, __fmul_rn(float(x) + 0.5f, a.inv_resolution[0])), 1.0f);
	const float ndc_y = __fsub_rn(__fmul_rn(2.0f, __fmul_rn(float(y) + 0.5f, a.inv_resolution[1])), 1.0f);
	float clip[4];
#pragma unroll
	for (int i = 0; i < 4; i++)
	{
		float v = __fmul_rn(a.inv_vp[i], ndc_x);
		v = __fadd_rn(v, __fmul_rn(a.inv_vp[4 + i], ndc_y));
		v = __fadd_rn(v, __fmul_rn(a.inv_vp[8 + i], 0.0f));
		v = __fadd_rn(v, a.inv_vp[12 + i]);
		clip[i] = __fadd_rn(v, __fmul_rn(depth, a.inv_vp[8 + i]));
	}
	const float clip_w = active ? clip[3] : 1.0f;
	const float3_ pos = f3(__fdiv_rn(clip[0], clip_w), __fdiv_rn(clip[1], clip_w), __fdiv_rn(clip[2], clip_w));

	Surface s;
	s.pos = pos;
	s.N = N;
	const float3_ cam = f3(a.camera_pos[0], a.camera_pos[1], a.camera_pos[2]);
	float3_ V = cam - pos;
	V = V * rsq(fmaxf(dot(V, V), 1e-30f));
	s.V = V;
	s.NoV = clampf(dot(N, V), 0.001f, 1.0f);
	s.F0 = f3(fmaf(base.x - 0.04f, metallic, 0.04f), fmaf(base.y - 0.04f, metallic, 0.04f), fmaf(base.z - 0.04f, metallic, 0.04f));
	const float roughness = fmaf(mat_roughness, 0.75f, 0.25f);
	const float m = roughness * roughness;
	s.m2 = m * m;
	const float r1 = roughness + 1.0f;
	s.k = r1 * r1 * (1.0f / 8.0f);
	s.Gv = fmaf(s.NoV, 1.0f - s.k, s.k);
	const float dscale = (1.0f - metallic) * (1.0f / PI_SIC);
	s.diffuse = base * dscale;

	float3_ accum = f3(float(dst.x), float(dst.y), float(dst.z));

	// ---- directional quad (directional.frag:41-65) ----
	if (a.flags & GR_LIGHTING_DIRECTIONAL_BIT)
	{
		const float3_ L = f3(a.dir_direction[0], a.dir_direction[1], a.dir_direction[2]);
		float3_ lit = f3(a.dir_color[0], a.dir_color[1], a.dir_color[2]) * shade(s, L);
		if (a.flags & GR_LIGHTING_AMBIENT_FALLBACK_BIT)
			lit = lit + base * 0.05f;
		// blend ONE/ONE, attachment store rounds to fp16
		accum = f3(float(_Float16(accum.x + lit.x)), float(_Float16(accum.y + lit.y)), float(_Float16(accum.z + lit.z)));
	}

	// ---- clustered quad (clusterer_bindless.h:29-84) ----
	if ((a.flags & GR_LIGHTING_CLUSTERED_BIT) && a.cl_num_lights > 0)
	{
		float3_ result = f3(0.0f, 0.0f, 0.0f);

		int ccx = int(__fmul_rn(__fmul_rn(float(x) + 0.5f, a.inv_resolution[0]), a.cl_xy_scale[0]));
		int ccy = int(__fmul_rn(__fmul_rn(float(y) + 0.5f, a.inv_resolution[1]), a.cl_xy_scale[1]));
		ccx = clampi(ccx, 0, a.cl_res_x - 1);
		ccy = clampi(ccy, 0, a.cl_res_y - 1);
		const int cluster_base = (ccy * a.cl_res_x + ccx) * a.cl_num_lights_32;

		const float dzx = __fsub_rn(pos.x, a.cl_camera_base[0]), dzy = __fsub_rn(pos.y, a.cl_camera_base[1]),
		            dzz = __fsub_rn(pos.z, a.cl_camera_base[2]);
		const float z = __fadd_rn(__fadd_rn(__fmul_rn(dzx, a.cl_camera_front[0]), __fmul_rn(dzy, a.cl_camera_front[1])),
		                          __fmul_rn(dzz, a.cl_camera_front[2]));
		int z_index = int(__fmul_rn(z, a.cl_z_scale));
		z_index = clampi(z_index, 0, a.cl_z_max_index);
		uint2 z_range = make_uint2(0xffffffffu, 0u);
		if (active)
			z_range = a.range[z_index];

		const int z_start = __builtin_amdgcn_readfirstlane(int(wave_min_u32(z_range.x) >> 5u));
		const int z_end = __builtin_amdgcn_readfirstlane(min(int(wave_max_u32(z_range.y) >> 5u), a.cl_num_lights_32 - 1));

		for (int i = z_start; i <= z_end; i++)
		{
			uint32_t mask = 0u;
			if (active)
				mask = cluster_mask_range(a.bitmask[cluster_base + i], z_range, 32u * uint32_t(i));
			uint32_t uni = __builtin_amdgcn_readfirstlane(wave_or(mask));
			const uint32_t type_mask = a.type_mask[i];

			while (uni != 0u)
			{
				const int bit = __builtin_ctz(uni);
				uni &= uni - 1u;
				const gr_light_info &li = a.lights[32 * i + bit]; // wave-uniform address -> scalar loads
				const float3_ lpos = f3(li.position[0], li.position[1], li.position[2]);
				float3_ Lf = lpos - pos;
				const float d2 = dot(Lf, Lf);
				const float inv_d = rsq(fmaxf(d2, 1e-30f));
				const float3_ L = Lf * inv_d;
				const float dist = fmaxf(0.1f, d2 * inv_d);
				float atten = smooth_falloff(dist * li.inv_radius);
				if (!((type_mask >> bit) & 1u))
				{
					// spot.h:41-46: cone = dot(normalize(world_pos - light_pos), direction) = -dot(L, direction)
					const float cone_angle = -dot(L, f3(li.direction[0], li.direction[1], li.direction[2]));
					const f16x2 sb = __builtin_bit_cast(f16x2, li.spot_scale_bias);
					float cone = clampf(fmaf(cone_angle, float(sb.x), float(sb.y)), 0.0f, 1.0f);
					atten *= cone * cone;
				}
				if (atten > 0.0f)
				{
					const float a2 = atten * rcp(dist * dist);
					const float3_ color = f3(li.color[0] * a2, li.color[1] * a2, li.color[2] * a2);
					if (color.x != 0.0f || color.y != 0.0f || color.z != 0.0f)
						result = result + color * shade(s, L);
				}
			}
		}
		accum = f3(float(_Float16(accum.x + result.x)), float(_Float16(accum.y + result.y)), float(_Float16(accum.z + result.z)));
	}

	if (active)
	{
		f16x4 o;
		o.x = _Float16(accum.x);
		o.y = _Float16(accum.y);
		o.z = _Float16(accum.z);
		o.w = dst.w;
		*reinterpret_cast<f16x4 *>(a.hdr.ptr + size_t(y) * a.hdr.pitch + size_t(x) * 8u) = o;
	}
}

static bool check_image(const gr_image &img, uint32_t format, uint32_t bpp, uint32_t w, uint32_t h)
{
	return img.ptr && img.format == format && img.width == w && img.height == h && img.pitch_bytes >= w * bpp &&
	       (img.pitch_bytes % bpp) == 0;
}
} // namespace

extern "C" {

int gr_lighting(gr_ctx *ctx, gr_stream stream, const gr_lighting_args *args)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, args != nullptr);
	const uint32_t W = args->hdr.width, H = args->hdr.height;
	GR_CHECK_ARG(ctx, W != 0 && H != 0);
	GR_CHECK_ARG(ctx, check_image(args->hdr, GR_FORMAT_R16G16B16A16_SFLOAT, 8, W, H));
	GR_CHECK_ARG(ctx, check_image(args->emissive, GR_FORMAT_R16G16B16A16_SFLOAT, 8, W, H));
	GR_CHECK_ARG(ctx, check_image(args->albedo, GR_FORMAT_R8G8B8A8_SRGB, 4, W, H));
	GR_CHECK_ARG(ctx, check_image(args->normal, GR_FORMAT_A2B10G10R10_UNORM_PACK32, 4, W, H));
	GR_CHECK_ARG(ctx, check_image(args->pbr, GR_FORMAT_R8G8_UNORM, 2, W, H));
	GR_CHECK_ARG(ctx, check_image(args->depth, GR_FORMAT_D32_SFLOAT, 4, W, H));
	const bool clustered = (args->flags & GR_LIGHTING_CLUSTERED_BIT) != 0 && args->cluster.num_lights > 0;
	if (clustered)
	{
		GR_CHECK_ARG(ctx, args->transforms && args->bitmask && args->range);
		GR_CHECK_ARG(ctx, args->cluster.num_lights <= GR_MAX_LIGHTS_BINDLESS);
		GR_CHECK_ARG(ctx, args->cluster.resolution_xy[0] > 0 && args->cluster.resolution_xy[1] > 0);
	}

	KernelArgs k{};
	auto dev = [](const gr_image &i) { return DevImage{static_cast<const uint8_t *>(i.ptr), int(i.width), int(i.height), i.pitch_bytes}; };
	k.albedo = dev(args->albedo);
	k.normal = dev(args->normal);
	k.pbr = dev(args->pbr);
	k.depth = dev(args->depth);
	k.emissive = dev(args->emissive);
	k.hdr = DevImageRW{static_cast<uint8_t *>(args->hdr.ptr), int(W), int(H), args->hdr.pitch_bytes};
	for (int i = 0; i < 16; i++)
		k.inv_vp[i] = args->inv_view_projection[i];
	for (int i = 0; i < 3; i++)
	{
		k.camera_pos[i] = args->clustering.camera_pos[i];
		k.dir_color[i] = args->directional.color[i];
		k.dir_direction[i] = args->directional.direction[i];
		k.cl_camera_base[i] = args->cluster.camera_base[i];
		k.cl_camera_front[i] = args->cluster.camera_front[i];
	}
	k.inv_resolution[0] = args->clustering.inv_resolution[0];
	k.inv_resolution[1] = args->clustering.inv_resolution[1];
	k.cl_xy_scale[0] = args->cluster.xy_scale[0];
	k.cl_xy_scale[1] = args->cluster.xy_scale[1];
	k.cl_res_x = args->cluster.resolution_xy[0];
	k.cl_res_y = args->cluster.resolution_xy[1];
	k.cl_num_lights = clustered ? args->cluster.num_lights : 0;
	k.cl_num_lights_32 = args->cluster.num_lights_32;
	k.cl_z_max_index = args->cluster.z_max_index;
	k.cl_z_scale = args->cluster.z_scale;
	if (clustered)
	{
		const uint8_t *t = static_cast<const uint8_t *>(args->transforms);
		k.lights = reinterpret_cast<const gr_light_info *>(t + GR_TRANSFORMS_OFFSET_LIGHTS);
		k.type_mask = reinterpret_cast<const uint32_t *>(t + GR_TRANSFORMS_OFFSET_TYPE_MASK);
		k.bitmask = args->bitmask;
		k.range = reinterpret_cast<const uint2 *>(args->range);
	}
	k.srgb_lut = ctx->srgb_decode_lut;
	k.flags = args->flags;

	dim3 grid(gr_div_up(W, LIGHT_TILE_W), gr_div_up(H, LIGHT_TILE_H * LIGHT_WAVES));
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "lighting"};
	hipLaunchKernelGGL(k_lighting, grid, dim3(64 * LIGHT_WAVES), 0, gr_to_stream(stream), k);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}
}
