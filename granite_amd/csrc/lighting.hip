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
// Mapping.  A wave64 owns an 8x8 pixel tile and never synchronises with another wave (wave-private LDS, no barriers);
// four waves side by side form a 32x8 workgroup so every row of the workgroup is one 256 B HDR / 128 B G-buffer segment.
// The light loop is wave-uniform like the reference's subgroup path (clusterer_bindless.h:49-81), in three steps per
// 64-light chunk of the tile's light-index window:
//   1. gather (one LIGHT per lane): lane l owns light 64*chunk + l.  Its bit is the OR of that bit over every cluster
//      cell the tile touches, trimmed to the wave's Z-slice index window [min first, max last] - a superset of the
//      reference's subgroupOr of per-lane trimmed masks.  What the superset adds lies outside its radius for the pixels
//      concerned and contributes exactly 0 (point.h:38, spot.h:45); the reference's subgroup footprint cannot matter
//      for the same reason (oracle: orc_lighting_bruteforce_clustered).
//   2. cull + stage: each lane tests its light's sphere (1.001 r) against the tile's bounding sphere, and a spot's cone
//      against the same sphere (most of a spot's sphere lies outside its cone); a survivor writes its record -- with per-light
//      constants (10 / r, (1.001 r)^2, fp32 spot scale / bias) computed once per light instead of once per pixel -- into
//      the slot of its own lane in a wave-private LDS list.  Two ballots name the survivors: point lights, and the lights
//      walked with the cone body (spots, and point lights small enough for the 0.1 distance floor to matter).
//   3. shade (one PIXEL per lane): the set bits of each ballot are walked in index order by a loop with a straight-line body
//      (no light-type branch) and broadcast ds_read_b128; every BRDF operand is a VGPR.
//      On gfx950 fp32 fma / mul / add issue at full rate only with VGPR / literal / inline operands (measured: 1.2 ns per
//      wave-instruction per SIMD vs 1.9 ns with an SGPR operand, 2.0 ns for min / max / med3 / cmp / cvt, 3.6 ns for
//      rcp / rsq), which is what bounds this kernel on the 4096-light config, not HBM.
// The BRDF is algebraically the reference's; normalisations are folded (dot products on the unnormalised light vector,
// H.V = |V+L|/2, one rcp for G*D) to cut the per-light VALU count.
#include <cstdlib>
#include <cstring>
#include "ctx.hpp"
#include "device_common.hpp"

namespace
{
constexpr int LIGHT_TILE = 8;  // wave tile edge
#ifndef LV_WAVES
#define LV_WAVES 4
#endif
constexpr int LIGHT_WAVES = LV_WAVES; // waves per workgroup, side by side: four 8 PX x 8 tiles in a row (A/B builds: -DLV_WAVES=1 / 2, slower in rounds 4 and 6)
constexpr int LIGHT_SLOT_BYTES = 64;

constexpr float PI_SIC = 3.1415628f; // assets/shaders/lights/pbr.h:4-6 (sic)
// Sphere culling margins: the shader's falloff is exactly 0 once dist * inv_radius >= 1.
constexpr float CULL_RADIUS_SCALE = 1.001f;
constexpr float CULL_SLACK = 1e-3f;

// Measurement build only (-DLV_STAMP, tools/lighting_stamps.py): shader-clock marks at the phase boundaries of a tile.
#ifdef LV_STAMP
#define LV_STAMP_PARAM , uint32_t (&stamp_marks__)[4]
#define LV_STAMP_ARG , stamp_scope__.marks
#define LV_STAMP_MARK(k) stamp_marks__[k] = uint32_t(__builtin_amdgcn_s_memtime())
#define LV_STAMP_LAP_BEGIN() uint32_t stamp_lap__ = uint32_t(__builtin_amdgcn_s_memtime())
#define LV_STAMP_LAP(k)                                                          \
	do                                                                           \
	{                                                                            \
		const uint32_t stamp_now__ = uint32_t(__builtin_amdgcn_s_memtime());     \
		stamp_marks__[k] += stamp_now__ - stamp_lap__;                           \
		stamp_lap__ = stamp_now__;                                               \
	} while (0)
#else
#define LV_STAMP_PARAM
#define LV_STAMP_ARG
#define LV_STAMP_MARK(k)
#define LV_STAMP_LAP_BEGIN()
#define LV_STAMP_LAP(k)
#endif

struct KernelArgs
{
	DevImage albedo, normal, pbr, depth, emissive, ao;
	DevImageRW hdr;
	float inv_vp[16];
	float clip_dx[4]; // inv_vp column 0 * 2 / width: clip-space step between horizontally adjacent pixels
	float camera_pos[3];
	float dir_color[3];
	float dir_direction[3];
	float inv_resolution[2];
	// cluster UBO subset
	float cl_xy_scale[2];
	int cl_res_x, cl_res_y;
	int cl_num_lights, cl_num_lights_32, cl_z_max_index;
	float cl_z_row[4]; // slice = int(dot(pos, cl_z_row.xyz) + cl_z_row.w): camera_front * z_scale and -dot(camera_base, camera_front) * z_scale
	const gr_light_info *__restrict__ lights;
	const uint32_t *__restrict__ type_mask;
	const uint32_t *__restrict__ bitmask;
	const uint2 *__restrict__ range;
	const float *__restrict__ srgb_lut;
	uint32_t flags;
	float fog_color[3], fog_falloff; // the fog quad behind the clustered one (renderer.cpp:1179-1196); falloff <= 0: none
	int row_first, row_end, block_row0; // render area rows [row_first, row_end); first block row = row_first / 8
	int list_words;                     // wide-window path: words of the cluster bitmask per span (0: off), 32 list entries of 2 B each per wave
};

struct float3_ { float x, y, z; };
__device__ __forceinline__ float3_ f3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ float3_ operator+(float3_ a, float3_ b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float3_ operator-(float3_ a, float3_ b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float3_ operator*(float3_ a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(float3_ a, float3_ b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float rsq(float v) { return __builtin_amdgcn_rsqf(v); }
__device__ __forceinline__ float med3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); } // clamp
__device__ __forceinline__ float sat(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); } // folds into a clamp modifier
// clamp(v, 0, hi) with a wave-uniform hi >= 0 as one v_med3_i32 (the backend forms it only for constant bounds: v_max + v_min otherwise)
__device__ __forceinline__ int clamp0_i32(int v, int hi)
{
	int r;
	asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
	return r;
}

// StockSampler::LinearClamp on an R8_UNORM image (the ambient-occlusion input).
__device__ __forceinline__ float sample_linear_r8(const DevImage &img, float u, float v)
{
	int ix, iy;
	float wx, wy;
	linear_axis(u * float(img.w) - 0.5f, ix, wx);
	linear_axis(v * float(img.h) - 0.5f, iy, wy);
	const int x0 = clampi(ix, 0, img.w - 1), x1 = clampi(ix + 1, 0, img.w - 1);
	const int y0 = clampi(iy, 0, img.h - 1), y1 = clampi(iy + 1, 0, img.h - 1);
	const uint8_t *r0 = img.ptr + size_t(y0) * img.pitch, *r1 = img.ptr + size_t(y1) * img.pitch;
	const float t00 = unorm8_to_float(r0[x0]), t10 = unorm8_to_float(r0[x1]), t01 = unorm8_to_float(r1[x0]), t11 = unorm8_to_float(r1[x1]);
	const float top = t00 * (1.0f - wx) + t10 * wx, bottom = t01 * (1.0f - wx) + t11 * wx;
	return top * (1.0f - wy) + bottom * wy;
}

// Per-pixel material terms hoisted out of the light loop.
struct Surface
{
	float3_ pos, N, V, F0, D1; // D1 = (1 - F0) * base * (1 - metallic) / PI: the diffuse term at f = 0
	float NdV, m2m1, gA, gB;   // NdV unclamped; m2m1 = m^2 - 1; Gv Gl / c0 = NoL gA + gB with Gv = NoV (1-k) + k, c0 = m^2 / (4 PI)
};

// Shared BRDF tail of compute_point_light / compute_spot_light / compute_lighting (point.h:119-142, spot.h:122-145,
// lighting.h:26-45) for a unit light direction L given through NdL = dot(N, L) and hh = |V + L|^2 (> 0):
//   H = (V + L) / |V + L|,  dot(H, V) = |V + L| / 2 (unit V, L),  dot(N, H) = (NdV + NdL) / |V + L|.
// Adds colour * scale * NoL * (F G D + (1 - F) diffuse) to acc.  With f = (1 - HoV)^5 and F = mix(F0, 1, f):
//   F GD + (1 - F) diffuse = (1 - f) (GD F0 + D1) + f GD,
// so the per-channel work is three fmas on two scalars that already carry scale * NoL.
// Every clamp of the reference is kept where it can change a bit of the result:
//   * HoV = clamp(|V + L| / 2, 0.001, 1): 1 - HoV = sat(1.001 - HoV) - 0.001 (the upper clamp is the sat modifier; HoV
//     exceeds 1 by rounding only, which leaves |1 - HoV| < 2e-7 and f < 1e-33).
//   * NoH = clamp(., 0.0001, 1) enters only as d = NoH^2 (m^2 - 1) + 1: with NoH <= 1e-4 the product is below 1e-8 <
//     half an fp32 ulp of 1, so d == 1.0f exactly whether the lower bound is 1e-4 or 0 -- the clamp is the free
//     sat modifier of the multiply.  (N is not renormalised, clustering.frag:35, so the upper bound does engage.)
//   * NoL = clamp(., 0.001, 1) is a med3: both bounds matter.
//   * max(Gv Gl, 0.001) never engages: roughness = 0.25 + 0.75 r >= 0.25 gives k = (roughness + 1)^2 / 8 >= 0.195, and
//     Gv, Gl = mix(k, 1, NoX) >= k, so Gv Gl >= 0.038.
__device__ __forceinline__ void brdf_accumulate(const Surface &s, float NdL, float hh, float scale, float3_ colour, float3_ &acc)
{
	const float NoL = med3(NdL, 0.001f, 1.0f);
	const float inv_h = rsq(hh);
	const float omh = sat(fmaf(-0.5f * hh, inv_h, 1.001f)) - 0.001f; // 1 - clamp(HoV, 0.001, 1)
	const float NoH = sat((s.NdV + NdL) * inv_h);

	const float omh2 = omh * omh;
	const float f = omh2 * omh2 * omh; // pow(1 - HoV, 5)

	const float d = fmaf(NoH * NoH, s.m2m1, 1.0f);     // (NoH m2 - NoH) NoH + 1
	const float g = fmaf(NoL, s.gA, s.gB);             // Gv Gl / c0, c0 = m^2 / (4 PI): G D = 1 / (d^2 g)
	const float GD = rcp(d * d * g);

	const float w = NoL * scale;
	const float cw = fmaf(-f, w, w); // (1 - f) w
	const float fw = f * GD * w;
	acc.x = fmaf(colour.x, fmaf(fmaf(GD, s.F0.x, s.D1.x), cw, fw), acc.x);
	acc.y = fmaf(colour.y, fmaf(fmaf(GD, s.F0.y, s.D1.y), cw, fw), acc.y);
	acc.z = fmaf(colour.z, fmaf(fmaf(GD, s.F0.z, s.D1.z), cw, fw), acc.z);
}

// clusterer_bindless_buffers.h:17-27 for one light index instead of one 32-bit word.
__device__ __forceinline__ bool index_in_range(uint32_t index, uint32_t range_x, uint32_t range_y)
{
	return index >= range_x && index <= range_y;
}

// Wave64 min / max -> SGPR in six DPP steps and one v_readlane: four steps reduce each row of 16 lanes (every lane of a row ends up with
// the row's result), row_bcast:15 folds rows 0 / 2 into rows 1 / 3, row_bcast:31 folds row 1 into rows 2 / 3, lane 63 holds the wave's.
// (Round 4 read the four row results back with four v_readlane and combined them through v_mov + v_max3: 13 VALU slots against 7.)
// `old` of the two broadcast steps is the operation's identity: rows outside the row mask then compute op(v, identity) = v, and the
// backend folds the v_mov_dpp into the v_min / v_max as its DPP source (it does so for an immediate identity only).
template <bool IS_MAX>
__device__ __forceinline__ uint32_t minmax_step(uint32_t v, uint32_t moved)
{
	return IS_MAX ? max(v, moved) : min(v, moved);
}
#define GR_DPP_STEP(IS_MAX, v, ctrl) v = minmax_step<IS_MAX>(v, uint32_t(__builtin_amdgcn_mov_dpp(int(v), ctrl, 0xf, 0xf, true)))
#define GR_DPP_BCAST(IS_MAX, v, ctrl, rows) \
	v = minmax_step<IS_MAX>(v, uint32_t(__builtin_amdgcn_update_dpp(IS_MAX ? 0 : -1, int(v), ctrl, rows, 0xf, false)))
template <bool IS_MAX>
__device__ __forceinline__ uint32_t wave_minmax_u32(uint32_t v)
{
	GR_DPP_STEP(IS_MAX, v, 0xB1);  // quad_perm [1,0,3,2]
	GR_DPP_STEP(IS_MAX, v, 0x4E);  // quad_perm [2,3,0,1]
	GR_DPP_STEP(IS_MAX, v, 0x141); // row_half_mirror
	GR_DPP_STEP(IS_MAX, v, 0x140); // row_mirror
	GR_DPP_BCAST(IS_MAX, v, 0x142, 0xa); // row_bcast:15 into rows 1, 3
	GR_DPP_BCAST(IS_MAX, v, 0x143, 0xc); // row_bcast:31 into rows 2, 3
	return uint32_t(__builtin_amdgcn_readlane(int(v), 63));
}
// min of one value and max of another, step by step side by side: each DPP read has the other reduction's instruction between it and
// the write it depends on (a DPP source needs two wait states after the VALU write).
__device__ __forceinline__ void wave_min_and_max_u32(uint32_t &lo, uint32_t &hi)
{
	GR_DPP_STEP(false, lo, 0xB1);
	GR_DPP_STEP(true, hi, 0xB1);
	GR_DPP_STEP(false, lo, 0x4E);
	GR_DPP_STEP(true, hi, 0x4E);
	GR_DPP_STEP(false, lo, 0x141);
	GR_DPP_STEP(true, hi, 0x141);
	GR_DPP_STEP(false, lo, 0x140);
	GR_DPP_STEP(true, hi, 0x140);
	GR_DPP_BCAST(false, lo, 0x142, 0xa);
	GR_DPP_BCAST(true, hi, 0x142, 0xa);
	GR_DPP_BCAST(false, lo, 0x143, 0xc);
	GR_DPP_BCAST(true, hi, 0x143, 0xc);
	lo = uint32_t(__builtin_amdgcn_readlane(int(lo), 63));
	hi = uint32_t(__builtin_amdgcn_readlane(int(hi), 63));
}
#undef GR_DPP_STEP
#undef GR_DPP_BCAST

// Wave64 inclusive prefix sum: four row_shr steps scan each row of 16 lanes, row_bcast:15 adds row 0's total to row 1 and row 2's to row 3,
// row_bcast:31 adds the total of rows 0 + 1 to rows 2 and 3.
__device__ __forceinline__ uint32_t wave_inclusive_add_u32(uint32_t v)
{
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, true)); // row_shr:1, lanes without a source add 0
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, true)); // row_shr:2
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, true)); // row_shr:4
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, true)); // row_shr:8
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1, 3
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2, 3
	return v;
}
// The bits of the 32-bit word that starts at light index `word_first` whose index is >= `index`.
__device__ __forceinline__ uint32_t bits_from(uint32_t index, uint32_t word_first)
{
	return index <= word_first ? 0xffffffffu : (index - word_first >= 32u ? 0u : 0xffffffffu << (index - word_first));
}

// The workgroup's dynamic LDS (behind the static arrays of k_lighting): the wide-window candidate lists, LIGHT_WAVES x list_words x 32 entries.
extern __shared__ __attribute__((aligned(16))) uint8_t lv_dynamic_lds[];

// One staged light = 4 x 16 B in wave-private LDS.
//   q0: position.xyz, (1.001 r)^2        q1: colour.xyz, 10 / r
//   q2: direction.xyz, -                 q3: spot scale, spot bias (fp32), -, -          (q2, q3 read by the second list only)
//
// PX pixels per lane (horizontally adjacent): the light record, the tile-level early-outs and all of the wave-level
// bookkeeping are shared by 64 * PX pixels, and the PX independent BRDF chains give the wave enough instruction-level
// parallelism to saturate the VALU at half the resident waves -- which is what leaves wave slots to the executor's other
// streams while this kernel runs (see gr_lighting).
//
// Two lists, two straight-line bodies:
//   CONE = false: point lights of radius >= 1 / 8.  light_dist = max(0.1, length) (point.h:36) feeds only the smoothstep
//     t = sat(light_dist * 10 / r - 9): with 10 / r <= 80 a length below 0.1 gives t = sat(negative) = 0 with or without the
//     floor, a length above it is untouched by it, so the half-rate v_max is not issued (bit-identical, by the argument above).
//   CONE = true: spot lights -- and the point lights of radius < 1 / 8, staged with the neutral cone (direction 0, scale 0,
//     bias 1: cone factor sat(-0 * 0 + 1)^2 = 1 exactly), for which the floor can reach the smoothstep.
template <int PX, bool CONE>
__device__ __forceinline__ void shade_positional(const Surface (&s)[PX], const f32x4 *slot, float3_ (&result)[PX])
{
	const f32x4 q0 = slot[0], q1 = slot[1];
	float3_ Lf[PX];
	float d2[PX];
	bool near_any = false;
#pragma unroll
	for (int p = 0; p < PX; p++)
	{
		Lf[p] = f3(q0.x, q0.y, q0.z) - s[p].pos;                                               // light_pos - world_pos
		d2[p] = fmaf(Lf[p].z, Lf[p].z, fmaf(Lf[p].y, Lf[p].y, fmaf(Lf[p].x, Lf[p].x, 1e-30f))); // > 0: no inf / nan downstream
		near_any = near_any || d2[p] < q0.w;
	}
	// Falloff is exactly 0 once dist * inv_radius >= 1: skip the light when no pixel of the tile is inside 1.001 r.
	if (!__any(near_any))
		return;

	float inv_d[PX], len[PX], inv_d2[PX], atten[PX];
#pragma unroll
	for (int p = 0; p < PX; p++)
	{
		inv_d[p] = rsq(d2[p]);
		len[p] = d2[p] * inv_d[p]; // length(light_dir_full)
		const float dist = CONE ? fmaxf(0.1f, len[p]) : len[p]; // light_dist = max(MIN_POINT_DIST, length) (point.h:36, spot.h:38)
		inv_d2[p] = inv_d[p] * inv_d[p];
		// 1 - smoothstep(0.9, 1.0, dist * inv_radius)
		const float t = sat(fmaf(dist, q1.w, -9.0f));
		atten[p] = fmaf(t * t, fmaf(2.0f, t, -3.0f), 1.0f); // 1 - t^2 (3 - 2 t), the sign carried by the inner fma: no negation to materialise
	}
	if (CONE)
	{
		// spot.h:41-46: cone = dot(normalize(world_pos - light_pos), direction) = -dot(Lf, direction) / |Lf|
		const f32x4 q2 = slot[2], q3 = slot[3];
		bool lit_any = false;
#pragma unroll
		for (int p = 0; p < PX; p++)
		{
			const float cone_angle = -dot(Lf[p], f3(q2.x, q2.y, q2.z)) * inv_d[p];
			const float cone = sat(fmaf(cone_angle, q3.x, q3.y));
			atten[p] *= cone * cone;
			lit_any = lit_any || atten[p] > 0.0f;
		}
		// What is left of a spot's bounding sphere after the staging-time cone test can still miss every pixel.
		if (!__any(lit_any))
			return;
	}
#pragma unroll
	for (int p = 0; p < PX; p++)
	{
		// colour = light colour * atten / light_dist^2, light_dist^2 = max(len, 0.1)^2
		const float a2 = atten[p] * fminf(inv_d2[p], 1.0f / (0.1f * 0.1f));
		const float NdL = dot(s[p].N, Lf[p]) * inv_d[p];
		// |V + L|^2 with L = Lf / len: |len V + Lf|^2 / len^2.  The vector sum keeps the relative error of hh at
		// fp32 level when L is nearly -V, where 2 + 2 dot(V, L) would cancel.
		const float3_ Hs = f3(fmaf(s[p].V.x, len[p], Lf[p].x), fmaf(s[p].V.y, len[p], Lf[p].y), fmaf(s[p].V.z, len[p], Lf[p].z));
		const float hh = fmaf(Hs.z, Hs.z, fmaf(Hs.y, Hs.y, fmaf(Hs.x, Hs.x, 1e-30f))) * inv_d2[p];
		brdf_accumulate(s[p], NdL, hh, a2, f3(q1.x, q1.y, q1.z), result[p]);
	}
}

// Forms of this walk that were built, measured slower and removed in round 4 (profiles/r04_lighting_variants_ab.txt, with the commits):
// the walk on float2 values (one v_pk_* for both pixels of the lane wherever the ISA has one: 237 us against 182 us -- a packed
// instruction whose operand is the result of the packed instruction before it waits); the colour accumulation on the matrix pipe
// (v_mfma_f32_4x4x1: 196 us against 182 us -- the pipe runs beside the VALU for OTHER waves, not for the wave that issued the instruction);
// persistent waves dealing themselves tiles from ticket queues (227 us with a queue per XCD, 766 us with a queue per workgroup and stealing,
// against 181 us for the static grid: the machine is VALU-bound, full wave slots lengthen every tile); XCD bands instead of screen order.

// What a lane reads of its PX pixels, as loaded (11 VGPRs for PX = 2, RGBA16F).
template <int PX, bool B10>
struct RawTile
{
	float depth[PX];
	uint32_t alb[PX], nrm[PX];
	uint32_t mr;                     // PX == 2: both pixels' (metallic, roughness) byte pairs
	uint32_t em[B10 ? PX : 2 * PX];  // emissive texels, packed as stored
};

// Attachment loads of one tile, all issued back to back.  Byte offsets are 32-bit (images < 4 GiB, checked by the launcher), so
// every load is base SGPR pair + one VGPR offset.  PX == 2 is only launched for even widths and pitches that keep the pair of
// texels naturally aligned: the lane's two pixels are inside or outside together and come in with one load per attachment.
// Lanes outside the target read the texels their coordinates clamp to (no exec-mask region, no zero-filled registers): what they
// hold is never stored, and their depth is not looked at (`inside` gates `active`).  The clamp is to the whole target, not to the render
// area: a band launch (gr_rows) may read rows of the attachments it does not shade -- every attachment has the target's full height
// (check_image in gr_lighting) -- and, with emissive aliased to the target, texels another wave or another band's launch is writing;
// the values are dropped, so this is a benign read of a location being written (a race detector would name it).
template <int PX, bool B10>
__device__ __forceinline__ void load_raw(const KernelArgs &a, int x0, int y, RawTile<PX, B10> &r)
{
	const uint32_t uy = uint32_t(min(y, a.hdr.h - 1)), ux = uint32_t(min(x0, a.hdr.w - PX));
	if constexpr (PX == 2)
	{
		const float2 d = *reinterpret_cast<const float2 *>(a.depth.ptr + (uy * a.depth.pitch + ux * 4u));
		const uint2 al = *reinterpret_cast<const uint2 *>(a.albedo.ptr + (uy * a.albedo.pitch + ux * 4u));
		const uint2 nr = *reinterpret_cast<const uint2 *>(a.normal.ptr + (uy * a.normal.pitch + ux * 4u));
		r.mr = *reinterpret_cast<const uint32_t *>(a.pbr.ptr + (uy * a.pbr.pitch + ux * 2u));
		if constexpr (B10)
		{
			const uint2 packed = *reinterpret_cast<const uint2 *>(a.emissive.ptr + (uy * a.emissive.pitch + ux * 4u));
			r.em[0] = packed.x, r.em[1] = packed.y;
		}
		else
		{
			const uint4 em = *reinterpret_cast<const uint4 *>(a.emissive.ptr + (uy * a.emissive.pitch + ux * 8u));
			r.em[0] = em.x, r.em[1] = em.y, r.em[2] = em.z, r.em[3] = em.w;
		}
		r.depth[0] = d.x, r.depth[PX - 1] = d.y;
		r.alb[0] = al.x, r.alb[PX - 1] = al.y;
		r.nrm[0] = nr.x, r.nrm[PX - 1] = nr.y;
	}
	else
	{
		r.depth[0] = *reinterpret_cast<const float *>(a.depth.ptr + (uy * a.depth.pitch + ux * 4u));
		r.alb[0] = *reinterpret_cast<const uint32_t *>(a.albedo.ptr + (uy * a.albedo.pitch + ux * 4u));
		r.nrm[0] = *reinterpret_cast<const uint32_t *>(a.normal.ptr + (uy * a.normal.pitch + ux * 4u));
		r.mr = *reinterpret_cast<const uint16_t *>(a.pbr.ptr + (uy * a.pbr.pitch + ux * 2u));
		if constexpr (B10)
			r.em[0] = *reinterpret_cast<const uint32_t *>(a.emissive.ptr + (uy * a.emissive.pitch + ux * 4u));
		else
		{
			const uint2 em = *reinterpret_cast<const uint2 *>(a.emissive.ptr + (uy * a.emissive.pitch + ux * 8u));
			r.em[0] = em.x, r.em[1] = em.y;
		}
	}
}

// One wave, one tile of 8 PX x 8 pixels whose attachment words are in `raw`: both quads, the fog quad, the store.
template <int PX, bool AO, bool B10>
__device__ __forceinline__ void shade_tile(const KernelArgs &a, const int tile_x0, const int tile_y0, const int wave, const int lane, const RawTile<PX, B10> &raw,
                                           f32x4 *const slots, const float *s_srgb LV_STAMP_PARAM)
{
	const int x0 = tile_x0 + (lane & (LIGHT_TILE - 1)) * PX;
	const int y = tile_y0 + (lane >> 3);
	const int W = a.hdr.w, H = a.hdr.h;
	const bool row_inside = y >= a.row_first && y < a.row_end; // row_end <= H

	bool inside[PX], active[PX];
	f16x4 dst[PX];
	float depth_v[PX];
	uint32_t alb_v[PX], nrm_v[PX], mr_v[PX];
#pragma unroll
	for (int p = 0; p < PX; p++)
	{
		inside[p] = row_inside && x0 + p < W;
		depth_v[p] = raw.depth[p];
		alb_v[p] = raw.alb[p];
		nrm_v[p] = raw.nrm[p];
		mr_v[p] = PX == 2 ? (p == 0 ? raw.mr & 0xffffu : raw.mr >> 16) : raw.mr;
		if constexpr (B10)
		{
			uint32_t rg, ba;
			expand_b10g11r11(raw.em[p], rg, ba);
			dst[p] = __builtin_bit_cast(f16x4, make_uint2(rg, ba));
		}
		else
			dst[p] = __builtin_bit_cast(f16x4, make_uint2(raw.em[2 * p], raw.em[2 * p + 1]));
	}

	// ---- position reconstruction (clustering.vert:10-13, clustering.frag:37-39): clip = invVP * (ndc.xy, depth, 1),
	// pos = clip.xyz / clip.w.  Evaluated with fused multiply-adds and a Newton-refined reciprocal; a pixel that lands
	// in the neighbouring Z slice because of the last-bit difference sees the same lights up to ones at the very edge
	// of their radius (falloff -> 0), cf. the conservative slice ranges of clusterer.cpp:1265-1275.  The part that does not
	// depend on depth is affine in the pixel coordinate: evaluated once per lane, stepped by clip_dx for the second pixel. ----
	float clip_base[4];
	{
		const float ndc_x = fmaf(2.0f * (float(x0) + 0.5f), a.inv_resolution[0], -1.0f);
		const float ndc_y = fmaf(2.0f * (float(y) + 0.5f), a.inv_resolution[1], -1.0f);
#pragma unroll
		for (int i = 0; i < 4; i++)
			clip_base[i] = fmaf(a.inv_vp[4 + i], ndc_y, fmaf(a.inv_vp[i], ndc_x, a.inv_vp[12 + i]));
	}

	Surface s[PX];
	float3_ base[PX], accum[PX];
	bool any_active = false;
#pragma unroll
	for (int p = 0; p < PX; p++)
	{
		const int x = x0 + p;
		const float depth = depth_v[p];
		const uint32_t alb = alb_v[p], nrm = nrm_v[p], mr = mr_v[p];
		// depth test NOT_EQUAL against the quad at z = 0 (renderer.cpp:1056-1057): reverse-Z far-plane pixels keep the
		// emissive value (both draws are depth-rejected).
		active[p] = inside[p] && depth != 0.0f;
		any_active = any_active || active[p];

		// ---- G-buffer decode (clustering.frag:31-35) ----
		base[p] = f3(s_srgb[alb & 255u], s_srgb[(alb >> 8) & 255u], s_srgb[(alb >> 16) & 255u]);
		const float3_ N = f3(float(nrm & 1023u) * (2.0f / 1023.0f) - 1.0f, float((nrm >> 10) & 1023u) * (2.0f / 1023.0f) - 1.0f,
		                     float((nrm >> 20) & 1023u) * (2.0f / 1023.0f) - 1.0f);
		const float metallic = float(mr & 255u) * (1.0f / 255.0f);
		const float mat_roughness = float(mr >> 8) * (1.0f / 255.0f);

		float clip[4];
#pragma unroll
		for (int i = 0; i < 4; i++)
			clip[i] = fmaf(depth, a.inv_vp[8 + i], p == 0 ? clip_base[i] : clip_base[i] + float(p) * a.clip_dx[i]);
		const float clip_w = active[p] ? clip[3] : 1.0f;
		float inv_w = rcp(clip_w);
		inv_w = inv_w * fmaf(-clip_w, inv_w, 2.0f);
		const float3_ pos = f3(clip[0] * inv_w, clip[1] * inv_w, clip[2] * inv_w);

		s[p].pos = pos;
		s[p].N = N;
		const float3_ cam = f3(a.camera_pos[0], a.camera_pos[1], a.camera_pos[2]);
		float3_ V = cam - pos;
		V = V * rsq(fmaxf(dot(V, V), 1e-30f));
		s[p].V = V;
		s[p].NdV = dot(N, V);
		const float NoV = med3(s[p].NdV, 0.001f, 1.0f);
		s[p].F0 = f3(fmaf(base[p].x - 0.04f, metallic, 0.04f), fmaf(base[p].y - 0.04f, metallic, 0.04f),
		             fmaf(base[p].z - 0.04f, metallic, 0.04f));
		const float roughness = fmaf(mat_roughness, 0.75f, 0.25f);
		const float m = roughness * roughness;
		const float m2 = m * m;
		s[p].m2m1 = m2 - 1.0f;
		const float r1 = roughness + 1.0f;
		const float k = r1 * r1 * (1.0f / 8.0f);
		const float omk = 1.0f - k;
		const float Gv_over_c0 = fmaf(NoV, omk, k) * rcp(m2 * (0.25f / PI_SIC)); // m2 >= 0.0039
		s[p].gA = Gv_over_c0 * omk;
		s[p].gB = Gv_over_c0 * k;
		const float kd = (1.0f - metallic) * (1.0f / PI_SIC);
		s[p].D1 = f3(fmaf(-s[p].F0.x, base[p].x, base[p].x) * kd, fmaf(-s[p].F0.y, base[p].y, base[p].y) * kd,
		             fmaf(-s[p].F0.z, base[p].z, base[p].z) * kd);

		accum[p] = f3(float(dst[p].x), float(dst[p].y), float(dst[p].z));

		// ---- directional quad (directional.frag:41-65) ----
		if (a.flags & GR_LIGHTING_DIRECTIONAL_BIT)
		{
			const float3_ L = f3(a.dir_direction[0], a.dir_direction[1], a.dir_direction[2]);
			const float3_ Hv = V + L;
			float3_ lit = f3(0.0f, 0.0f, 0.0f);
			brdf_accumulate(s[p], dot(N, L), fmaxf(dot(Hv, Hv), 1e-30f), 1.0f, f3(a.dir_color[0], a.dir_color[1], a.dir_color[2]), lit);
			// base_ambient * base_color * 0.05 (directional.frag:52-64); a scalar 0 when the fallback term is off
			float ambient = (a.flags & GR_LIGHTING_AMBIENT_FALLBACK_BIT) ? 0.05f : 0.0f;
			if (AO)
				ambient *= sample_linear_r8(a.ao, (float(x) + 0.5f) * a.inv_resolution[0], (float(y) + 0.5f) * a.inv_resolution[1]);
			lit = f3(fmaf(base[p].x, ambient, lit.x), fmaf(base[p].y, ambient, lit.y), fmaf(base[p].z, ambient, lit.z));
			// blend ONE/ONE, attachment store rounds to fp16 (or to the packed floats)
			if constexpr (B10)
				accum[p] = f3(round_to_ufloat<6>(accum[p].x + lit.x), round_to_ufloat<6>(accum[p].y + lit.y), round_to_ufloat<5>(accum[p].z + lit.z));
			else
				accum[p] = f3(float(_Float16(accum[p].x + lit.x)), float(_Float16(accum[p].y + lit.y)), float(_Float16(accum[p].z + lit.z)));
		}
	}

	LV_STAMP_MARK(0); // G-buffer decode, material terms and the directional quad are through
	// ---- clustered quad (clusterer_bindless.h:29-84) ----
	f16x4 out_h[PX];
	float3_ out_f[PX]; // B10: the sums themselves, rounded by the packed store below
	if ((a.flags & GR_LIGHTING_CLUSTERED_BIT) && a.cl_num_lights > 0)
	{
		float3_ result[PX];
		uint32_t lane_lo = 0xffffffffu, lane_hi = 0u;
		uint2 px_range[PX]; // the pixel's own light-index range, defined where active[p] (read by the wide-window path only)
#pragma unroll
		for (int p = 0; p < PX; p++)
		{
			result[p] = f3(0.0f, 0.0f, 0.0f);
			// Slice lookup (clusterer_bindless.h:43-47): int(dot(pos - camera_base, camera_front) * z_scale) with the scale and the
			// base folded into the row by the launcher (three fma; a last-bit difference moves a pixel that sits on a slice boundary
			// into the neighbouring slice, whose lights differ from its own only by ones at the very edge of their radius: see above).
			const float zf = fmaf(s[p].pos.z, a.cl_z_row[2], fmaf(s[p].pos.y, a.cl_z_row[1], fmaf(s[p].pos.x, a.cl_z_row[0], a.cl_z_row[3])));
			const uint32_t z_index = uint32_t(clamp0_i32(int(zf), a.cl_z_max_index)); // -> v_med3_i32
			if (active[p])
			{
				const uint2 z_range = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(a.range) + (z_index << 3u));
				px_range[p] = z_range;
				lane_lo = min(lane_lo, z_range.x);
				lane_hi = max(lane_hi, z_range.y);
			}
		}
		// The wave's light-index window.
		wave_min_and_max_u32(lane_lo, lane_hi);
		const uint32_t win_lo = lane_lo, win_hi = min(lane_hi, uint32_t(a.cl_num_lights - 1));
		// A tile across a depth discontinuity (foreground against background) has a window that spans every light between the two depths,
		// although the indices between the foreground's ranges and the background's are in no pixel's range -- the reference's per-lane
		// cluster_mask_range (clusterer_bindless_buffers.h:17-27) removes them from every lane's mask, so they are in no subgroupOr
		// either.  For a window of three chunks or more (the wide-window path below), with P = the smallest range end of the tile: every
		// pixel whose range starts at or below P ends at or below Q = the largest end among them, every other pixel's range starts at
		// or above G = the smallest start among those; the indices strictly between Q and G (if any) are trimmed from the window.
		const bool wide = a.list_words != 0 && win_lo <= win_hi && (win_hi >> 6u) - (win_lo >> 6u) >= 2u;
		uint32_t gap_lo = 0u, gap_hi = 0xffffffffu; // Q and G: indices i with gap_lo < i < gap_hi are in no pixel's range
		bool in_low[PX], in_high[PX];                // the pixel's range lies at or below Q / at or above G (neither: it has no range)
#pragma unroll
		for (int p = 0; p < PX; p++)
			in_low[p] = in_high[p] = false;
		if (wide)
		{
			uint32_t first_end = 0xffffffffu;
#pragma unroll
			for (int p = 0; p < PX; p++)
				first_end = min(first_end, active[p] && px_range[p].x <= px_range[p].y ? px_range[p].y : 0xffffffffu);
			first_end = wave_minmax_u32<false>(first_end);
#pragma unroll
			for (int p = 0; p < PX; p++)
			{
				const bool some = active[p] && px_range[p].x <= px_range[p].y, low = px_range[p].x <= first_end;
				in_low[p] = some && low, in_high[p] = some && !low;
				gap_lo = max(gap_lo, some && low ? px_range[p].y : 0u);
				gap_hi = min(gap_hi, some && !low ? px_range[p].x : 0xffffffffu);
			}
			wave_min_and_max_u32(gap_hi, gap_lo); // gap_hi <= gap_lo + 1: an empty gap, nothing is trimmed
		}
		if (win_lo <= win_hi)
		{
			// Cluster cells the tile touches (clusterer_bindless.h:39-42 at the tile's first and last pixel; the per-pixel formula is
			// monotonic, so every lane's cell lies in this rectangle).  There is no scalar float unit: lane 0 evaluates the formula for
			// the tile's first pixel, lane 63 for its last one (its second pixel, clamped into the image), and two v_readlane per axis
			// fetch them -- instead of four wave-uniform evaluations on the vector unit.
			auto cell = [](int p, float inv_res, float scale, int res) {
				return clamp0_i32(int(__fmul_rn(__fmul_rn(float(p) + 0.5f, inv_res), scale)), res - 1);
			};
			const int my_cx = cell(min(x0 + (lane >> 5) * (PX - 1), W - 1), a.inv_resolution[0], a.cl_xy_scale[0], a.cl_res_x);
			const int my_cy = cell(min(y, H - 1), a.inv_resolution[1], a.cl_xy_scale[1], a.cl_res_y);
			const int cx0 = __builtin_amdgcn_readlane(my_cx, 0), cx1 = __builtin_amdgcn_readlane(my_cx, 63);
			const int cy0 = __builtin_amdgcn_readlane(my_cy, 0), cy1 = __builtin_amdgcn_readlane(my_cy, 63);

			// Bounding sphere of a set of the tile's surface points (member[p]: a non-empty set): centre = its first pixel, radius = its
			// farthest pixel from that.
			auto bounding_sphere = [&](const bool (&member)[PX], float3_ &centre_out, float &radius_out) __attribute__((always_inline)) {
				bool any_member = false;
#pragma unroll
				for (int p = 0; p < PX; p++)
					any_member = any_member || member[p];
				const int first = __builtin_ctzll(__ballot(any_member));
				const float3_ mine = PX == 2 && !member[0] ? s[PX - 1].pos : s[0].pos; // a lane of the ballot has one of them in the set
				centre_out = f3(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.x), first)),
				                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.y), first)),
				                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.z), first)));
				float off2 = 0.0f;
#pragma unroll
				for (int p = 0; p < PX; p++)
				{
					const float3_ off = s[p].pos - centre_out;
					const float mine2 = member[p] ? dot(off, off) : 0.0f;
					off2 = p == 0 ? mine2 : fmaxf(off2, mine2);
				}
				// v_sqrt_f32 (1 ulp) is well inside the 1.0001 + CULL_SLACK margin
				radius_out = __builtin_amdgcn_sqrtf(__builtin_bit_cast(float, wave_minmax_u32<true>(__builtin_bit_cast(uint32_t, off2)))) * 1.0001f + CULL_SLACK;
			};
			float3_ centre;
			float tile_radius;
			bounding_sphere(active, centre, tile_radius); // the lit pixels of the tile

			LV_STAMP_MARK(1); // slice window, cells, bounding sphere
			LV_STAMP_LAP_BEGIN();
			// Two sources of candidate lights, one loop around the cull and the walks:
			//   * narrow window (one or two 64-light chunks, the usual case of a surface near the camera): a chunk per turn, lane l owns
			//     light 64 * chunk + l and reads the word of each touched cell that holds its bit;
			//   * wide window (a surface far from the camera: the ranges of its Z slices span the lights of a thick slab, 22 chunks at 30
			//     units in the 4096-light scenes, of whose lights three in a hundred touch the tile's cells): the window's words are read
			//     ONCE, lane j the j-th word of a span of `list_words` words, and their set bits are compacted in index order into a
			//     wave-private list of light indices (counts by v_bcnt, their prefix sums by a DPP scan); a turn takes the next 64 of the
			//     list -- one pair of memory round trips per 64 CANDIDATES instead of one per 64 indices.  The list lives in the
			//     workgroup's dynamic LDS, which the launcher sizes anyway to cap the kernel's residency (gr_lighting).
			const int chunk_lo = int(win_lo >> 6u), chunk_hi = int(win_hi >> 6u);
			// One turn: up to 64 candidate lights, one per lane, culled against the tile and staged; then the walks over the survivors.
			auto turn = [&](const uint32_t light_index, const bool candidate, const float3_ &centre, const float tile_radius) __attribute__((always_inline)) {
				// ---- cull: one light per lane ----
				bool keep = false;
				bool second = false; // walked with the cone body: spot lights, and point lights of radius < 1 / 8
				{
					if (candidate)
					{
						const f32x4 *rec = reinterpret_cast<const f32x4 *>(a.lights + light_index);
						const f32x4 c = rec[0], pq = rec[1], d = rec[2]; // colour|scale_bias, position|offset_radius, direction|inv_radius
						const float radius = CULL_RADIUS_SCALE * rcp(d.w);
						const float3_ to_light = f3(pq.x, pq.y, pq.z) - centre;
						const float reach = radius + tile_radius;
						const float dist2 = dot(to_light, to_light);
						keep = dist2 <= reach * reach;
						const bool is_spot = ((a.type_mask[light_index >> 5] >> (light_index & 31u)) & 1u) == 0u;
						// unpackHalf2x16(spot_scale_bias).  The lane is copied to a scalar first: clang (ROCm 7.2) evaluates
						// __builtin_bit_cast on a vector-component lvalue (v.w) at the address of the whole vector, i.e. as
						// lane .x (observed: spots shaded with colour.x as scale | bias).
						const float sb_lane = c.w;
						const uint32_t sb_bits = __builtin_bit_cast(uint32_t, sb_lane);
						const float spot_scale = float(__builtin_bit_cast(_Float16, uint16_t(sb_bits & 0xffffu)));
						const float spot_bias = float(__builtin_bit_cast(_Float16, uint16_t(sb_bits >> 16)));
						if (is_spot && keep && spot_scale > 0.0f)
						{
							// Cone vs the tile's bounding sphere.  spot.h:44-45: the cone factor sat(cone_angle * scale + bias) is
							// exactly 0 for cone_angle <= -bias / scale =: cos(theta).  With v = centre - light, a = dot(v, dir) and
							// p = distance of the centre from the axis, e = p cos(theta) - a sin(theta) = |v| sin(phi - theta) is the
							// distance of the centre from the cone's mantle line (never more than its distance from the cone), so
							// e > tile_radius (+ margin) puts every pixel of the tile outside the cone by an angle far above fp32
							// rounding: the light adds exactly 0 there.  (theta >= 90 degrees, cos(theta) <= 0, never culls.)
							const float cos_t = -spot_bias * rcp(spot_scale);
							if (cos_t > 0.0f && cos_t < 1.0f)
							{
								const float sin_t = __builtin_amdgcn_sqrtf(fmaf(-cos_t, cos_t, 1.0f));
								const float along = -dot(to_light, f3(d.x, d.y, d.z)); // dot(centre - light, direction)
								const float off_axis = __builtin_amdgcn_sqrtf(fmaxf(fmaf(-along, along, dist2), 0.0f));
								const float e = fmaf(off_axis, cos_t, -along * sin_t);
								keep = e <= tile_radius * 1.001f + CULL_SLACK;
							}
						}
						// inv_radius > 8: the 0.1 distance floor can reach this light's smoothstep (shade_positional)
						second = is_spot || d.w > 8.0f;
						// A survivor stages its record in the slot of its own lane, here, where its registers are: nothing of a light
						// lives across the ballots below (round 4 compacted the list with mbcnt after them, which kept sixteen
						// zero-filled registers per lane alive on every path).  The walks visit the set bits of the ballots.
						if (keep)
						{
							f32x4 *dst_slot = slots + lane * (LIGHT_SLOT_BYTES / 16);
							dst_slot[0] = f32x4{pq.x, pq.y, pq.z, radius * radius};
							dst_slot[1] = f32x4{c.x, c.y, c.z, 10.0f * d.w};
							if (second)
							{
								dst_slot[2] = is_spot ? f32x4{d.x, d.y, d.z, 0.0f} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
								dst_slot[3] = is_spot ? f32x4{spot_scale, spot_bias, 0.0f, 0.0f} : f32x4{0.0f, 1.0f, 0.0f, 0.0f};
							}
						}
					}
				}
				const uint64_t kept = __ballot(keep);
				const uint64_t seconds = __ballot(keep && second);
				__builtin_amdgcn_wave_barrier(); // LDS is wave-private: in-order DS execution is the only ordering needed
				LV_STAMP_LAP(2); // gather + cull

				// ---- shade: PX pixels per lane, lights broadcast from LDS; each list in index order, its body without a light-type branch ----
				// (the visited bit is cleared by s_bitset0_b64: one scalar instruction where `todo &= todo - 1` is three on 64 bits)
				for (uint64_t todo = kept & ~seconds; todo != 0ull;)
				{
					const int bit = __builtin_ctzll(todo);
					asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(bit));
					shade_positional<PX, false>(s, slots + bit * (LIGHT_SLOT_BYTES / 16), result);
				}
				for (uint64_t todo = seconds; todo != 0ull;)
				{
					const int bit = __builtin_ctzll(todo);
					asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(bit));
					shade_positional<PX, true>(s, slots + bit * (LIGHT_SLOT_BYTES / 16), result);
				}
				LV_STAMP_LAP(3); // the two walks
				__builtin_amdgcn_wave_barrier(); // the slots are rewritten by the next turn
			};
			if (!wide)
			{
				for (int chunk = chunk_lo; chunk <= chunk_hi; chunk++)
				{
					// ---- gather: lane l looks up the bit of light 64 * chunk + l ----
					const uint32_t light_index = uint32_t(chunk) * 64u + uint32_t(lane);
					const int my_word = chunk * 2 + (lane >> 5);
					bool candidate = false;
					if (index_in_range(light_index, win_lo, win_hi))
					{
						uint32_t word = 0u;
						for (int cy = cy0; cy <= cy1; cy++)
						{
#pragma clang loop vectorize(disable) unroll(disable)
							for (int cx = cx0; cx <= cx1; cx++)
								word |= a.bitmask[(cy * a.cl_res_x + cx) * a.cl_num_lights_32 + my_word];
						}
						candidate = ((word >> (uint32_t(lane) & 31u)) & 1u) != 0u;
					}
					turn(light_index, candidate, centre, tile_radius);
				}
			}
			else
			{
				// A tile with a gap holds two groups of pixels, those whose ranges lie below it and those above (foreground and background), and
				// ONE sphere around both is as wide as the scene is deep: no light would be culled by it.  A light of the lower part of the
				// window is in no range of the upper group (it adds exactly 0 there, as every light outside a pixel's range does), so it is
				// culled against the lower group's sphere alone, and the other way round.
				const bool two_groups = gap_hi != 0xffffffffu && gap_hi > gap_lo + 1u;
				float3_ centre_low = centre, centre_high = centre;
				float radius_low = tile_radius, radius_high = tile_radius;
				if (two_groups)
				{
					bounding_sphere(in_low, centre_low, radius_low);
					bounding_sphere(in_high, centre_high, radius_high);
				}
				uint16_t *const list = reinterpret_cast<uint16_t *>(lv_dynamic_lds) + wave * (a.list_words * 32);
				const int word_last = int(win_hi >> 5u);
				for (int span_word = int(win_lo >> 5u); span_word <= word_last; span_word += a.list_words)
				{
					// ---- gather: lane j reads word j of the span, the set bits of all words go to the list in index order ----
					const int my_word = span_word + lane;
					const uint32_t word_first = uint32_t(my_word) << 5u;
					uint32_t word = 0u;
					if (lane < a.list_words && my_word <= word_last)
					{
						for (int cy = cy0; cy <= cy1; cy++)
						{
#pragma clang loop vectorize(disable) unroll(disable)
							for (int cx = cx0; cx <= cx1; cx++)
								word |= a.bitmask[(cy * a.cl_res_x + cx) * a.cl_num_lights_32 + my_word];
						}
						// the window [win_lo, win_hi] without its gap, as bits of this word
						word &= bits_from(win_lo, word_first) & ~bits_from(win_hi + 1u, word_first) & ~(bits_from(gap_lo + 1u, word_first) & ~bits_from(gap_hi, word_first));
					}
					const uint32_t count = uint32_t(__builtin_popcount(word));
					const uint32_t upto = wave_inclusive_add_u32(count);
					const int list_count = __builtin_amdgcn_readlane(int(upto), 63);
					uint16_t *entry = list + (upto - count);
					for (; word != 0u; word &= word - 1u)
						*entry++ = uint16_t(word_first + uint32_t(__builtin_ctz(word)));
					__builtin_amdgcn_wave_barrier();
					for (int next = 0; next < list_count; next += 64)
					{
						const bool candidate = next + lane < list_count;
						const uint32_t light_index = candidate ? uint32_t(list[next + lane]) : 0u;
						const bool upper = light_index >= gap_hi; // (without two groups both spheres are the tile's)
						turn(light_index, candidate,
						     f3(upper ? centre_high.x : centre_low.x, upper ? centre_high.y : centre_low.y, upper ? centre_high.z : centre_low.z),
						     upper ? radius_high : radius_low);
					}
					__builtin_amdgcn_wave_barrier(); // the list is rewritten by the next span
				}
			}
		}
		// second blend: the attachment store rounds once more, straight into the halves that are written out
#pragma unroll
		for (int p = 0; p < PX; p++)
		{
			out_f[p] = f3(accum[p].x + result[p].x, accum[p].y + result[p].y, accum[p].z + result[p].z);
			out_h[p].x = _Float16(out_f[p].x);
			out_h[p].y = _Float16(out_f[p].y);
			out_h[p].z = _Float16(out_f[p].z);
		}
	}
	else
	{
#pragma unroll
		for (int p = 0; p < PX; p++)
		{
			out_f[p] = accum[p];
			out_h[p].x = _Float16(accum[p].x); // exact: accum holds fp16 values
			out_h[p].y = _Float16(accum[p].y);
			out_h[p].z = _Float16(accum[p].z);
		}
	}

	if (!inside[0])
		return;
	f16x4 o[PX];
#pragma unroll
	for (int p = 0; p < PX; p++)
	{
		o[p] = dst[p];
		if (active[p])
		{
			o[p].x = out_h[p].x;
			o[p].y = out_h[p].y;
			o[p].z = out_h[p].z;
		}
	}
	// ---- fog quad (fog.frag:17-25, fog.h:4-8): a third blend, src * (1 - src.a) + dst * src.a on colour and alpha, onto what the
	// second blend stored (its rounded value), rounded by the store once more.  Uniform branch: no fog, no instruction. ----
	if (a.fog_falloff > 0.0f)
	{
#pragma unroll
		for (int p = 0; p < PX; p++)
			if (active[p])
			{
				const float3_ eye = s[p].pos - f3(a.camera_pos[0], a.camera_pos[1], a.camera_pos[2]);
				const float f = __builtin_amdgcn_exp2f(-dot(eye, eye) * a.fog_falloff), omf = 1.0f - f;
				const float3_ lit = B10 ? f3(round_to_ufloat<6>(out_f[p].x), round_to_ufloat<6>(out_f[p].y), round_to_ufloat<5>(out_f[p].z))
				                        : f3(float(o[p].x), float(o[p].y), float(o[p].z));
				out_f[p] = f3(fmaf(lit.x, f, a.fog_color[0] * omf), fmaf(lit.y, f, a.fog_color[1] * omf), fmaf(lit.z, f, a.fog_color[2] * omf));
				o[p].x = _Float16(out_f[p].x);
				o[p].y = _Float16(out_f[p].y);
				o[p].z = _Float16(out_f[p].z);
				o[p].w = _Float16(fmaf(float(dst[p].w), f, f * omf));
			}
	}
	// In place (emissive == hdr) untouched pixels need no store; a pair with one lit pixel rewrites the other's own value.
	if constexpr (B10)
	{
		if (any_active || a.emissive.ptr != a.hdr.ptr)
		{
			// an untouched pixel keeps its packed value: repacking the exact expansion is the identity
			uint32_t words[PX];
#pragma unroll
			for (int p = 0; p < PX; p++)
				words[p] = active[p] ? pack_b10g11r11(out_f[p].x, out_f[p].y, out_f[p].z) : pack_b10g11r11(float(dst[p].x), float(dst[p].y), float(dst[p].z));
			uint8_t *out = a.hdr.ptr + (uint32_t(y) * a.hdr.pitch + uint32_t(x0) * 4u);
			if constexpr (PX == 2)
				*reinterpret_cast<uint2 *>(out) = make_uint2(words[0], words[PX - 1]);
			else
				*reinterpret_cast<uint32_t *>(out) = words[0];
		}
		return;
	}
	if (any_active || a.emissive.ptr != a.hdr.ptr)
	{
		uint8_t *out = a.hdr.ptr + (uint32_t(y) * a.hdr.pitch + uint32_t(x0) * 8u);
		if constexpr (PX == 2)
		{
			const uint2 lo = __builtin_bit_cast(uint2, o[0]), hi = __builtin_bit_cast(uint2, o[PX - 1]);
			*reinterpret_cast<uint4 *>(out) = make_uint4(lo.x, lo.y, hi.x, hi.y);
		}
		else
			*reinterpret_cast<f16x4 *>(out) = o[0];
	}
}

#ifdef LV_STAMP
// Measurement build only (make OUT=../lib_stamp EXTRA_lighting=-DLV_STAMP, tools/lighting_stamps.py): one record per wave tile,
// {start, end} of the 100 MHz s_memrealtime counter, the shader-clock cycles in between, and where the wave ran.
__device__ uint4 *g_lighting_stamps;
struct StampScope
{
	uint4 *rec;
	uint64_t t0, c0;
	uint32_t marks[4] = {0u, 0u, 0u, 0u};
	__device__ __forceinline__ StampScope(uint32_t tile)
	{
		rec = g_lighting_stamps ? g_lighting_stamps + 2u * tile : nullptr;
		t0 = __builtin_amdgcn_s_memrealtime();
		c0 = __builtin_amdgcn_s_memtime();
	}
	__device__ __forceinline__ ~StampScope()
	{
		__builtin_amdgcn_s_waitcnt(0); // the tile's stores have left
		const uint64_t t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
		uint32_t xcc, hw;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		if (rec && (threadIdx.x & 63u) == 0u)
		{
			rec[0] = make_uint4(uint32_t(t0), uint32_t(t1), uint32_t(c1 - c0), (xcc << 28) | (hw & 0x0fffffffu));
			// to the marks: cycles from the start of the tile (0: not passed); gather and walks: cycles summed over the chunks
			rec[1] = make_uint4(marks[0] ? marks[0] - uint32_t(c0) : 0u, marks[1] ? marks[1] - uint32_t(c0) : 0u, marks[2], marks[3]);
		}
	}
};
#define LV_STAMP_SCOPE(tile) StampScope stamp_scope__{uint32_t(tile)}
#else
#define LV_STAMP_SCOPE(tile)
#endif

// AO: the AMBIENT_OCCLUSION shader variant (renderer.cpp:1050-1051), a separate instantiation so that the default kernel keeps
// its register budget.
// B10: emissive and the HDR target are B10G11R11_UFLOAT_PACK32 (the reference's default, renderTargetFp16 = false): 4-byte
// texels, both blends round to the packed format (device_common.hpp: float_to_ufloat).
//
// The grid is the tile list in screen order: blockIdx.x = block column (four wave tiles side by side, a 32 PX x 8 block), blockIdx.y =
// block row.  Workgroups go to the XCDs round-robin in that order, so every XCD takes every eighth block and their finish times stay
// within 4 % (contiguous XCD bands differed by up to 20 % with the scene's light density: profiles/r04_lighting_tiles_*.txt).
// A/B builds: -DLV_WAVES_PER_EU=6 makes the backend fit the kernel into 80 registers (six waves per SIMD), spilling what it must.
#ifdef LV_WAVES_PER_EU
#define LV_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu(LV_WAVES_PER_EU, LV_WAVES_PER_EU)))
#else
#define LV_OCCUPANCY_ATTR
#endif
template <int PX, bool AO, bool B10 = false>
__global__ __launch_bounds__(64 * LIGHT_WAVES) LV_OCCUPANCY_ATTR void k_lighting(KernelArgs a)
{
	__shared__ __attribute__((aligned(16))) f32x4 s_lights[LIGHT_WAVES][64 * (LIGHT_SLOT_BYTES / 16)];
	__shared__ float s_srgb[256];
	constexpr int TILE_W = LIGHT_TILE * PX;
	// sRGB8 -> linear table into LDS (one entry per thread of the four-wave workgroup), the only workgroup-wide step.
#pragma unroll
	for (int i = 0; i < 256; i += 64 * LIGHT_WAVES)
		s_srgb[i + threadIdx.x] = a.srgb_lut[i + threadIdx.x];
	const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)); // wave-uniform: the slot and list addresses stay scalar
	const int lane = threadIdx.x & 63;
	const int tile_x0 = (int(blockIdx.x) * LIGHT_WAVES + wave) * TILE_W, tile_y0 = (a.block_row0 + int(blockIdx.y)) * LIGHT_TILE;
	LV_STAMP_SCOPE((int(blockIdx.y) * int(gridDim.x) + int(blockIdx.x)) * LIGHT_WAVES + wave);
	RawTile<PX, B10> raw;
	load_raw<PX, B10>(a, tile_x0 + (lane & (LIGHT_TILE - 1)) * PX, tile_y0 + (lane >> 3), raw);
	__syncthreads(); // s_srgb
	// the slot array's address stays a vector register: the walks form a light's address with one v_add (a scalar base costs a v_mov from
	// an SGPR per ds_read, two instructions per walked light)
	shade_tile<PX, AO, B10>(a, tile_x0, tile_y0, wave, lane, raw, s_lights[threadIdx.x >> 6], s_srgb LV_STAMP_ARG);
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
	// HDR target and emissive share a format: RGBA16F, or B10G11R11_UFLOAT_PACK32 (renderTargetFp16 = false in the reference)
	const bool b10 = args->hdr.format == GR_FORMAT_B10G11R11_UFLOAT_PACK32;
	const uint32_t hdr_format = b10 ? uint32_t(GR_FORMAT_B10G11R11_UFLOAT_PACK32) : uint32_t(GR_FORMAT_R16G16B16A16_SFLOAT), hdr_bpp = b10 ? 4u : 8u;
	GR_CHECK_ARG(ctx, check_image(args->hdr, hdr_format, hdr_bpp, W, H));
	GR_CHECK_ARG(ctx, check_image(args->emissive, hdr_format, hdr_bpp, W, H));
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
	if (args->flags & GR_LIGHTING_AMBIENT_OCCLUSION_BIT)
	{
		GR_CHECK_ARG(ctx, (args->flags & GR_LIGHTING_AMBIENT_FALLBACK_BIT) != 0);
		GR_CHECK_ARG(ctx, args->ambient_occlusion.ptr && args->ambient_occlusion.format == GR_FORMAT_R8_UNORM && args->ambient_occlusion.width &&
		                      args->ambient_occlusion.height && args->ambient_occlusion.pitch_bytes >= args->ambient_occlusion.width);
		k.ao = dev(args->ambient_occlusion);
	}
	k.hdr = DevImageRW{static_cast<uint8_t *>(args->hdr.ptr), int(W), int(H), args->hdr.pitch_bytes};
	for (int i = 0; i < 16; i++)
		k.inv_vp[i] = args->inv_view_projection[i];
	for (int i = 0; i < 4; i++)
		k.clip_dx[i] = k.inv_vp[i] * (2.0f * args->clustering.inv_resolution[0]);
	for (int i = 0; i < 3; i++)
	{
		k.camera_pos[i] = args->clustering.camera_pos[i];
		k.dir_color[i] = args->directional.color[i];
		k.dir_direction[i] = args->directional.direction[i];
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
	{
		// slice = int(dot(pos, front * zs) - dot(base, front) * zs) where the shader has dot(pos - base, front) * zs: the row is formed in
		// double here (its entries are correctly rounded), what differs in the kernel is that the two large terms meet in the last fma instead
		// of the subtraction coming first -- an absolute error of the slice coordinate of about |camera_base| * z_scale * 2^-23 (2e-4 of a slice
		// at the tests' camera, 8 units from the origin with 4096 slices over 100 units; a third of a slice at 1e4 units, where fp32 world
		// positions themselves are only good to 1e-3 units).  A pixel that changes slice by it sees lights at the edge of their range appear
		// or vanish, with falloff -> 0 there (see shade_tile).
		const float *front = args->cluster.camera_front, *camera = args->cluster.camera_base;
		const double z_scale = args->cluster.z_scale;
		for (int i = 0; i < 3; i++)
			k.cl_z_row[i] = float(double(front[i]) * z_scale);
		k.cl_z_row[3] = float(-(double(camera[0]) * front[0] + double(camera[1]) * front[1] + double(camera[2]) * front[2]) * z_scale);
	}
	if (clustered)
	{
		const uint8_t *t = static_cast<const uint8_t *>(args->transforms);
		k.lights = reinterpret_cast<const gr_light_info *>(t + GR_TRANSFORMS_OFFSET_LIGHTS);
		k.type_mask = reinterpret_cast<const uint32_t *>(t + GR_TRANSFORMS_OFFSET_TYPE_MASK);
		k.bitmask = args->bitmask;
		k.range = reinterpret_cast<const uint2 *>(args->range);
	}
	k.srgb_lut = ctx->srgb_decode_lut;
	k.flags = args->flags & (GR_LIGHTING_DIRECTIONAL_BIT | GR_LIGHTING_CLUSTERED_BIT | GR_LIGHTING_AMBIENT_FALLBACK_BIT | GR_LIGHTING_AMBIENT_OCCLUSION_BIT); // the shading bits; scheduling hints stay on the host
	for (int i = 0; i < 3; i++)
		k.fog_color[i] = args->fog_color[i];
	k.fog_falloff = args->fog_falloff > 0.0f ? args->fog_falloff : 0.0f;

	// Render area: tiles stay aligned to multiples of 8 rows of the full target, rows outside the band are masked.
	uint32_t row_first = 0, row_end = H;
	if (args->rows.count != 0)
	{
		row_first = args->rows.first < H ? args->rows.first : H;
		const uint64_t end = uint64_t(args->rows.first) + args->rows.count;
		row_end = end < H ? uint32_t(end) : H;
	}
	if (row_first >= row_end)
		return GR_OK;
	k.row_first = int(row_first);
	k.row_end = int(row_end);
	k.block_row0 = int(row_first / LIGHT_TILE);
	// Pixels per lane: 2 (GR_LIGHTING_PX=1 selects the one-pixel form for A/B measurements) whenever the lane's pair of
	// texels is one naturally aligned access in every attachment: even width, pitches that are multiples of two texels.
	// Odd-sized targets run the one-pixel kernel.
	static const int px_pref = []() {
		const char *env = gr_measurement_switch("GR_LIGHTING_PX");
		return env && atoi(env) == 1 ? 1 : 2;
	}();
	const bool pairs_aligned = (W & 1u) == 0 && (args->depth.pitch_bytes & 7u) == 0 && (args->albedo.pitch_bytes & 7u) == 0 &&
	                           (args->normal.pitch_bytes & 7u) == 0 && (args->pbr.pitch_bytes & 3u) == 0 &&
	                           (args->emissive.pitch_bytes & (2u * hdr_bpp - 1u)) == 0 && (args->hdr.pitch_bytes & (2u * hdr_bpp - 1u)) == 0 &&
	                           (reinterpret_cast<uintptr_t>(args->emissive.ptr) & (2u * hdr_bpp - 1u)) == 0 &&
	                           (reinterpret_cast<uintptr_t>(args->hdr.ptr) & (2u * hdr_bpp - 1u)) == 0 &&
	                           (reinterpret_cast<uintptr_t>(args->depth.ptr) & 7u) == 0 && (reinterpret_cast<uintptr_t>(args->albedo.ptr) & 7u) == 0 &&
	                           (reinterpret_cast<uintptr_t>(args->normal.ptr) & 7u) == 0 && (reinterpret_cast<uintptr_t>(args->pbr.ptr) & 3u) == 0;
	const int px = pairs_aligned ? px_pref : 1;
	// 32-bit byte offsets inside the kernel.
	GR_CHECK_ARG(ctx, uint64_t(args->hdr.pitch_bytes) * H <= 0xffffffffull && uint64_t(args->emissive.pitch_bytes) * H <= 0xffffffffull);
	const dim3 grid(gr_div_up(W, unsigned(LIGHT_TILE * px * LIGHT_WAVES)), gr_div_up(row_end, unsigned(LIGHT_TILE)) - unsigned(k.block_row0));
	// Residency cap.  The kernel is bound by the vector pipe; at full occupancy it owns every wave slot of the chip for the whole launch
	// and the executor's other streams (the previous frame's bloom / tonemap, the next frame's cluster build) cannot get a single wave
	// in.  Padding the workgroup's LDS footprint so that only `max_wgs` workgroups fit per CU leaves the remaining slots to them.  With
	// two pixels per lane (94 VGPRs) five workgroups per CU = five waves per SIMD: round 5, after the kernel's per-wave instruction
	// diet, 171.7 / 168.7 us alone against 175.7 / 170.8 with four, the frame 0.2005-0.2010 against 0.2069-0.2076 ms (four was
	// the round-2 choice, when a fifth wave measured nothing: profiles/r05_lighting_instruction_diet.txt).
	static const int max_wgs_env = []() {
		const char *env = gr_measurement_switch("GR_LIGHTING_WGS_PER_CU");
		const int v = env ? atoi(env) : 0;
		return v >= 1 && v <= 8 ? v : 0;
	}();
	const int max_wgs = max_wgs_env ? max_wgs_env : (px == 2 ? ((args->flags & GR_LIGHTING_SHARE_REGISTERS_BIT) ? 4 : 5) : 7);
	const size_t static_lds = sizeof(f32x4) * LIGHT_WAVES * 64 * (LIGHT_SLOT_BYTES / 16) + 256 * sizeof(float);
	// 8 KiB of the CU's 160 KiB stay free: back-of-frame kernels that use a little LDS (luminance, the fused pyramid tail)
	// must be able to start beside resident lighting workgroups instead of waiting for one to retire.
	const size_t per_wg = ((160u - 8u) * 1024u / unsigned(max_wgs * 4 / LIGHT_WAVES)) & ~size_t(1023); // max_wgs counts four-wave workgroups
	size_t pad_lds = max_wgs >= 8 || per_wg <= static_lds ? 0 : per_wg - static_lds;
	{
		// measurement switch: the pad in KiB whatever the cap asks for (with 96 registers the two-pixel kernel cannot exceed five waves per SIMD anyway)
		static const int pad_kib = []() { const char *env = gr_measurement_switch("GR_LIGHTING_PAD_KIB"); return env ? atoi(env) : -1; }();
		if (pad_kib >= 0)
			pad_lds = size_t(pad_kib) * 1024u;
	}
	// The pad is not idle: it holds the wide-window candidate lists (shade_tile), 32 two-byte entries per word of a span and wave.
	// 13 KiB at five workgroups per CU: spans of 52 words = 1664 light indices.  No pad (GR_LIGHTING_WGS_PER_CU=8): the chunk loop only.
	{
		const size_t words = pad_lds / (size_t(LIGHT_WAVES) * 32u * sizeof(uint16_t));
		k.list_words = words >= 8 ? int(words < 64 ? words : 64) : 0;
		static const bool narrow_only = []() { const char *env = gr_measurement_switch("GR_LIGHTING_NARROW_ONLY"); return env && atoi(env) != 0; }();
		if (narrow_only)
			k.list_words = 0;
	}
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "lighting"};
	const bool ao = (args->flags & GR_LIGHTING_AMBIENT_OCCLUSION_BIT) != 0;
	const dim3 block(64 * LIGHT_WAVES);
	const hipStream_t s = gr_to_stream(stream);
#define GR_LAUNCH_LIGHTING(PX_, AO_, B10_) hipLaunchKernelGGL((k_lighting<PX_, AO_, B10_>), grid, block, pad_lds, s, k)
	if (b10)
	{
		if (px == 2 && ao)
			GR_LAUNCH_LIGHTING(2, true, true);
		else if (px == 2)
			GR_LAUNCH_LIGHTING(2, false, true);
		else if (ao)
			GR_LAUNCH_LIGHTING(1, true, true);
		else
			GR_LAUNCH_LIGHTING(1, false, true);
	}
	else if (px == 2 && ao)
		GR_LAUNCH_LIGHTING(2, true, false);
	else if (px == 2)
		GR_LAUNCH_LIGHTING(2, false, false);
	else if (ao)
		GR_LAUNCH_LIGHTING(1, true, false);
	else
		GR_LAUNCH_LIGHTING(1, false, false);
#undef GR_LAUNCH_LIGHTING
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

#ifdef LV_STAMP
// Measurement build only: where the next launches write their per-tile records (nullptr: nowhere).
int gr_debug_lighting_stamps(gr_ctx *ctx, void *records, uint32_t count)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	uint4 *p = static_cast<uint4 *>(records);
	GR_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_lighting_stamps), &p, sizeof(p)));
	return GR_OK;
}
#endif
}
