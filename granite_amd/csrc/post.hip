// HDR post chain kernels for gfx950 and their C-ABI launchers (include/granite_hip.h).
//
// Replaces assets/shaders/post/{bloom_threshold,bloom_downsample,bloom_upsample,luminance}.comp and tonemap.frag as
// recorded by renderer/post/hdr.cpp:68-216,283-306.  All targets are linear row-major HBM buffers; every kernel is
// HBM-bound (no MFMA): one pass over its declared inputs, coalesced 8/16-byte accesses per lane.
#include <cstdlib>
#include "ctx.hpp"
#include "device_common.hpp"

namespace
{
constexpr int POST_BLOCK_X = 32;
constexpr int POST_BLOCK_Y = 8;

// Wave priority of the back-of-frame kernels.  They run beside the next frame's lighting kernel, which is VALU-issue-bound
// and whose (older) waves otherwise win the per-SIMD issue arbitration: these kernels are latency chains with little VALU
// work, so letting their few instructions go first shortens the critical path of the frame at next to no cost to lighting.
#ifndef GR_POST_WAVE_PRIORITY
#define GR_POST_WAVE_PRIORITY 0
#endif
__device__ __forceinline__ void post_wave_priority()
{
	if (GR_POST_WAVE_PRIORITY != 0)
		__builtin_amdgcn_s_setprio(GR_POST_WAVE_PRIORITY);
}

static inline DevImage to_dev(const gr_image *img)
{
	return {static_cast<const uint8_t *>(img->ptr), int(img->width), int(img->height), img->pitch_bytes};
}
static inline DevImageRW to_dev_rw(const gr_image *img)
{
	return {static_cast<uint8_t *>(img->ptr), int(img->width), int(img->height), img->pitch_bytes};
}

// ---- bloom threshold (bloom_threshold.comp:23-44) ------------------------------------------------------------------
template <bool DYNAMIC_EXPOSURE>
__global__ __launch_bounds__(POST_BLOCK_X *POST_BLOCK_Y) void k_bloom_threshold(DevImage hdr, DevImageRW out,
                                                                                const gr_luminance_data *lum,
                                                                                gr_push_bloom_threshold push, uint32_t y_first,
                                                                                uint32_t y_end, bool hdr_b10)
{
	post_wave_priority();
	const int x = blockIdx.x * POST_BLOCK_X + threadIdx.x;
	const int y = int(y_first) + blockIdx.y * POST_BLOCK_Y + threadIdx.y;
	if (uint32_t(x) >= push.threads[0] || uint32_t(y) >= y_end)
		return;

	const float u = (float(x) + 0.5f) * push.inv_output_size[0];
	const float v = (float(y) + 0.5f) * push.inv_output_size[1];
	const float4 c = hdr_b10 ? sample_linear_with([&hdr](int tx, int ty) { return load_b10g11r11(hdr, tx, ty); }, hdr.w, hdr.h, u, v)
	                         : sample_linear_rgba16f(hdr, u, v);

	float luminance = fmaxf(fmaxf(c.x, c.y), c.z) + 0.0001f;
	const float loglum = __log2f(luminance);
	const float inv = 1.0f / luminance;
	float r = c.x * inv, g = c.y * inv, b = c.z * inv;
	if (DYNAMIC_EXPOSURE)
		luminance -= 8.0f * lum->average_linear_luminance;
	else
		luminance -= 8.0f;
	store_rgba16f(out, x, y,
	              make_float4(fmaxf(r * luminance, 0.0f), fmaxf(g * luminance, 0.0f), fmaxf(b * luminance, 0.0f), loglum));
}

// 2:1 form (the threshold level of an even-sized HDR target): the LinearClamp tap at an output pixel's centre lands on the
// corner shared by the four HDR texels (2x, 2y) .. (2x + 1, 2y + 1), so the footprint is known without the sampler's index
// arithmetic.  The weights are NOT taken as 1/4: uv * size - 0.5 carries the rounding of its fp32 evaluation (up to
// 1e-4 at 4K), which the log-luminance channel -- log2 of a value near 1 over much of a frame -- is sensitive to, so they
// are computed exactly as the sampler does (uncontracted).  One lane makes two adjacent outputs from 2 x 32 contiguous bytes
// per row and stores 16 bytes; conversions ride on v_fma_mix_f32.
// fractional sampler coordinate of output p: ((p + 0.5) * inv_out) * size - 0.5, minus its floor (= 2 p)
__device__ __forceinline__ float threshold_fraction(uint32_t p, float inv_out, int size)
{
	const float f = __fsub_rn(__fmul_rn(__fmul_rn(float(p) + 0.5f, inv_out), float(size)), 0.5f);
	return f - floorf(f);
}
// One threshold texel from the 2 x 2 HDR texels under it (`top`, `bottom`: left and right texel of the row as RGBA16F dwords), the
// right / bottom weights of the sampler and the luminance to subtract.  Shared by the stand-alone kernel and the fused head of the
// pyramid (k_bloom_down_head): the bytes are the same.
__device__ __forceinline__ f16x4 threshold_of_quad(u32x4 top, u32x4 bottom, float wr, float wb, float threshold)
{
	const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	const float wl = 1.0f - wr, wt = 1.0f - wb;
	// sampler order: (t00 (1-a) + t10 a) (1-b) + (t01 (1-a) + t11 a) b
	const float4 row_t = fma_mix_texel(top.z, top.w, wr, fma_mix_texel(top.x, top.y, wl, zero));
	const float4 row_b = fma_mix_texel(bottom.z, bottom.w, wr, fma_mix_texel(bottom.x, bottom.y, wl, zero));
	const float4 c = fma4(row_b, wb, row_t * wt);
	float luminance = fmaxf(fmaxf(c.x, c.y), c.z) + 0.0001f;
	const float loglum = __log2f(luminance);
	float inv = __builtin_amdgcn_rcpf(luminance);
	inv = inv * fmaf(-luminance, inv, 2.0f); // one Newton step: the quotient to within an fp32 ulp
	luminance = __fsub_rn(luminance, threshold); // (and the caller's 8 x average as __fmul_rn: no contraction into one fma, wherever this is inlined)
	const float gain = inv * luminance;
	return pack_rgba16f(make_float4(fmaxf(c.x * gain, 0.0f), fmaxf(c.y * gain, 0.0f), fmaxf(c.z * gain, 0.0f), loglum));
}

template <bool DYNAMIC_EXPOSURE>
__global__ __launch_bounds__(POST_BLOCK_X *POST_BLOCK_Y) void k_bloom_threshold_2to1(DevImage hdr, DevImageRW out,
                                                                                     const gr_luminance_data *lum, uint32_t pairs_x,
                                                                                     float inv_out_w, float inv_out_h, uint32_t y_first,
                                                                                     uint32_t y_end, bool hdr_b10)
{
	post_wave_priority();
	const uint32_t xp = blockIdx.x * POST_BLOCK_X + threadIdx.x; // output pixels 2 xp, 2 xp + 1
	const uint32_t y = y_first + blockIdx.y * POST_BLOCK_Y + threadIdx.y;
	if (xp >= pairs_x || y >= y_end)
		return;
	u32x4 a0, a1, b0, b1; // four HDR texels per row as RGBA16F dwords
	if (hdr_b10)
	{
		// B10G11R11 target: 16 bytes per row hold the four texels; each expands exactly into its RGBA16F dwords
		const uint8_t *row0 = hdr.ptr + size_t(2u * y) * hdr.pitch + size_t(xp) * 16u;
		const u32x4 pa = *reinterpret_cast<const u32x4 *>(row0), pb = *reinterpret_cast<const u32x4 *>(row0 + hdr.pitch);
		a0 = expand_b10g11r11_pair(pa.x, pa.y), a1 = expand_b10g11r11_pair(pa.z, pa.w);
		b0 = expand_b10g11r11_pair(pb.x, pb.y), b1 = expand_b10g11r11_pair(pb.z, pb.w);
	}
	else
	{
		const uint8_t *row0 = hdr.ptr + size_t(2u * y) * hdr.pitch + size_t(xp) * 32u;
		a0 = *reinterpret_cast<const u32x4 *>(row0), a1 = *reinterpret_cast<const u32x4 *>(row0 + 16);
		b0 = *reinterpret_cast<const u32x4 *>(row0 + hdr.pitch), b1 = *reinterpret_cast<const u32x4 *>(row0 + hdr.pitch + 16);
	}
	const float wb = threshold_fraction(y, inv_out_h, hdr.h);
	const float threshold = DYNAMIC_EXPOSURE ? __fmul_rn(8.0f, lum->average_linear_luminance) : 8.0f;
	f16x4 result[2];
#pragma unroll
	for (int i = 0; i < 2; i++)
		result[i] = threshold_of_quad(i ? a1 : a0, i ? b1 : b0, threshold_fraction(2u * xp + uint32_t(i), inv_out_w, hdr.w), wb, threshold);
	const u32x2 lo = __builtin_bit_cast(u32x2, result[0]), hi = __builtin_bit_cast(u32x2, result[1]);
	*reinterpret_cast<u32x4 *>(out.ptr + size_t(y) * out.pitch + size_t(xp) * 16u) = u32x4{lo.x, lo.y, hi.x, hi.y};
}

// ---- 9-tap tent (bloom_downsample.comp:30-38 / bloom_upsample.comp:25-33) --------------------------------------------
// `sample(u, v)` is a LinearClamp fetch of the input level; the tap order is the shaders'.
template <typename Sample>
__device__ __forceinline__ float4 tent9_with(Sample sample, float u, float v, float ox, float oy)
{
#pragma clang fp contract(off)
	float4 value = sample(u, v) * 0.25f;
	value = fma4(sample(u - ox, v + oy), 0.0625f, value);
	value = fma4(sample(u, v + oy), 0.125f, value);
	value = fma4(sample(u + ox, v + oy), 0.0625f, value);
	value = fma4(sample(u - ox, v), 0.125f, value);
	value = fma4(sample(u + ox, v), 0.125f, value);
	value = fma4(sample(u - ox, v - oy), 0.0625f, value);
	value = fma4(sample(u, v - oy), 0.125f, value);
	value = fma4(sample(u + ox, v - oy), 0.0625f, value);
	return value;
}
__device__ __forceinline__ float4 tent9(const DevImage &in, float u, float v, float ox, float oy)
{
	return tent9_with([&in](float su, float sv) { return sample_linear_rgba16f(in, su, sv); }, u, v, ox, oy);
}
// Centre and tap offsets of output texel (x, y) of a level, from the pass's push constants -- stated once (and without contraction)
// for the separate kernels and the fused tails alike.  REACH: 1.75 input texels for the downsample, 0.875 for the upsample.
struct TentTaps
{
	float u, v, ox, oy;
};
__device__ __forceinline__ TentTaps tent_taps(int x, int y, const float inv_output_size[2], const float inv_input_size[2], float reach)
{
#pragma clang fp contract(off)
	return {(float(x) + 0.5f) * inv_output_size[0], (float(y) + 0.5f) * inv_output_size[1], reach * inv_input_size[0], reach * inv_input_size[1]};
}

// Temporal feedback of the last downsample level (hdr.cpp:160-166): NearestClamp fetch of the previous frame's level at the
// same texel, mix(history, value, vec4(lerp, lerp, lerp, 1)).
__device__ __forceinline__ float4 apply_feedback(float4 value, const DevImage &history, float u, float v, float l)
{
#pragma clang fp contract(off)
	const int hx = clampi(int(floorf(u * float(history.w))), 0, history.w - 1);
	const int hy = clampi(int(floorf(v * float(history.h))), 0, history.h - 1);
	const float4 h = load_rgba16f(history, hx, hy);
	return make_float4(h.x * (1.0f - l) + value.x * l, h.y * (1.0f - l) + value.y * l, h.z * (1.0f - l) + value.z * l, value.w);
}

template <bool FEEDBACK>
__global__ __launch_bounds__(POST_BLOCK_X *POST_BLOCK_Y) void k_bloom_downsample(DevImage in, DevImageRW out, DevImage history,
                                                                                 gr_push_bloom_downsample push, uint32_t y_first,
                                                                                 uint32_t y_end)
{
	post_wave_priority();
	const int x = blockIdx.x * POST_BLOCK_X + threadIdx.x;
	const int y = int(y_first) + blockIdx.y * POST_BLOCK_Y + threadIdx.y;
	if (uint32_t(x) >= push.threads[0] || uint32_t(y) >= y_end)
		return;
	const TentTaps t = tent_taps(x, y, push.inv_output_size, push.inv_input_size, 1.75f);
	float4 value = tent9(in, t.u, t.v, t.ox, t.oy);
	if (FEEDBACK)
		value = apply_feedback(value, history, t.u, t.v, push.lerp);
	store_rgba16f(out, x, y, value);
}

__global__ __launch_bounds__(POST_BLOCK_X *POST_BLOCK_Y) void k_bloom_upsample(DevImage in, DevImageRW out,
                                                                               gr_push_bloom_upsample push, uint32_t y_first,
                                                                               uint32_t y_end)
{
	post_wave_priority();
	const int x = blockIdx.x * POST_BLOCK_X + threadIdx.x;
	const int y = int(y_first) + blockIdx.y * POST_BLOCK_Y + threadIdx.y;
	if (uint32_t(x) >= push.threads[0] || uint32_t(y) >= y_end)
		return;
	const TentTaps t = tent_taps(x, y, push.inv_output_size, push.inv_input_size, 0.875f);
	store_rgba16f(out, x, y, tent9(in, t.u, t.v, t.ox, t.oy));
}

// ---- exact 2:1 / 1:2 forms of the tent filters -----------------------------------------------------------------------------
// When a level is exactly half (twice) its input in both axes -- every level of an even-sized pyramid, e.g. all of 4K down to
// 240x135 -- the nine LinearClamp taps of bloom_downsample.comp / bloom_upsample.comp land on fixed sub-texel phases, so the
// filter is a separable stencil with constant weights over clamped texel indices:
//   downsample: taps at 2i+0.5 and 2i+0.5 +- 1.75 input texels  ->  texels 2i-2 .. 2i+3, weights (1 3 4 4 3 1) / 16 per axis
//   upsample:   taps at i/2-0.25 and +- 0.875 input texels      ->  4 texels from k-2 (i = 2k) or k-1 (i = 2k+1),
//               weights (1 11 15 5) / 32 for even i, mirrored for odd i
// Same clamping as the sampler (each texel index clamped on its own); the products differ from the tap-by-tap evaluation only
// in fp32 summation order (covered by the per-level fp16 tolerance).  A 9-tap kernel spends ~60 VALU per tap on coordinates,
// weights and conversions; the stencils need 36 (16) texel fetches and one fma per texel and channel.
__device__ __forceinline__ float4 cvt4(f16x4 t) { return make_float4(float(t.x), float(t.y), float(t.z), float(t.w)); }
__device__ __forceinline__ float4 mul4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// Register budget of the back-of-frame kernels.  They run on the executor's generic stream WHILE the next frame's lighting
// kernel is resident with 4 waves x 104 VGPRs per SIMD, i.e. they live in the 96 registers per lane that are left: a kernel
// above that cannot co-reside at all and is served only between lighting workgroups (observed: a 140-VGPR downsample took
// 72 us instead of 8 us).  Under 56 registers one wave per SIMD always fits, under 48 two.
#define POST_VGPR_BUDGET /* documentation only: clang (ROCm 7.2) ignores amdgpu_num_vgpr below the 64-register occupancy step; the kernels are written to stay under 56 */

// One output texel of the 2:1 downsample: the 6 x 6 stencil over `in`, row by row.  ROWS_IN_FLIGHT = 1 keeps one row of six
// texels live at a time (the standalone kernel's register budget); the fused tail, whose few workgroups are latency-bound,
// unrolls the rows so that all 18 loads are in flight together.  Same arithmetic either way.
template <int ROWS_IN_FLIGHT = 1>
__device__ __forceinline__ float4 downsample_2to1_value(const DevImage &in, int x, int y)
{
	const float wt[6] = {0.0625f, 0.1875f, 0.25f, 0.25f, 0.1875f, 0.0625f};
	const int col0 = 2 * x - 2; // first of 6 input columns; even, so 16-byte aligned when inside the image
	const bool interior = col0 >= 0 && col0 + 5 < in.w;
	float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll ROWS_IN_FLIGHT
	for (int r = 0; r < 6; r++)
	{
		const int iy = clampi(2 * y - 2 + r, 0, in.h - 1);
		const uint8_t *row = in.ptr + size_t(iy) * in.pitch;
		float4 h;
		if (interior)
		{
			const u32x4 v0 = *reinterpret_cast<const u32x4 *>(row + size_t(col0) * 8u);
			const u32x4 v1 = *reinterpret_cast<const u32x4 *>(row + size_t(col0 + 2) * 8u);
			const u32x4 v2 = *reinterpret_cast<const u32x4 *>(row + size_t(col0 + 4) * 8u);
			// conversion folded into the multiply-add (fma_mix_texel): same values as cvt + fma
			h = fma_mix_texel(v0.x, v0.y, wt[0], make_float4(0.0f, 0.0f, 0.0f, 0.0f));
			h = fma_mix_texel(v0.z, v0.w, wt[1], h);
			h = fma_mix_texel(v1.x, v1.y, wt[2], h);
			h = fma_mix_texel(v1.z, v1.w, wt[3], h);
			h = fma_mix_texel(v2.x, v2.y, wt[4], h);
			h = fma_mix_texel(v2.z, v2.w, wt[5], h);
		}
		else
		{
			h = mul4(cvt4(*reinterpret_cast<const f16x4 *>(row + size_t(clampi(col0, 0, in.w - 1)) * 8u)), wt[0]);
#pragma unroll
			for (int c = 1; c < 6; c++)
				h = fma4(cvt4(*reinterpret_cast<const f16x4 *>(row + size_t(clampi(col0 + c, 0, in.w - 1)) * 8u)), wt[c], h);
		}
		const float wr = r == 0 || r == 5 ? 0.0625f : (r == 1 || r == 4 ? 0.1875f : 0.25f);
		acc = fma4(h, wr, acc);
	}
	return acc;
}

// Two vertically adjacent output texels (x, y) and (x, y + 1): their stencils share four of eight input rows, and a row's horizontal sum
// is the same value in both, so it is formed once -- 8 x 24 + 2 x 24 multiply-adds for two outputs instead of 2 x (6 x 24 + 24), and 24
// texel loads instead of 36.  Each output still accumulates its own six rows in the order of downsample_2to1_value, from zero: the same bits.
__device__ __forceinline__ void downsample_2to1_pair(const DevImage &in, int x, int y, float4 &upper, float4 &lower)
{
	const float wt[6] = {0.0625f, 0.1875f, 0.25f, 0.25f, 0.1875f, 0.0625f};
	const int col0 = 2 * x - 2;
	const bool interior = col0 >= 0 && col0 + 5 < in.w;
	upper = make_float4(0, 0, 0, 0);
	lower = make_float4(0, 0, 0, 0);
#pragma unroll 1
	for (int r = 0; r < 8; r++)
	{
		const int iy = clampi(2 * y - 2 + r, 0, in.h - 1);
		const uint8_t *row = in.ptr + size_t(iy) * in.pitch;
		float4 h;
		if (interior)
		{
			const u32x4 v0 = *reinterpret_cast<const u32x4 *>(row + size_t(col0) * 8u);
			const u32x4 v1 = *reinterpret_cast<const u32x4 *>(row + size_t(col0 + 2) * 8u);
			const u32x4 v2 = *reinterpret_cast<const u32x4 *>(row + size_t(col0 + 4) * 8u);
			h = fma_mix_texel(v0.x, v0.y, wt[0], make_float4(0.0f, 0.0f, 0.0f, 0.0f));
			h = fma_mix_texel(v0.z, v0.w, wt[1], h);
			h = fma_mix_texel(v1.x, v1.y, wt[2], h);
			h = fma_mix_texel(v1.z, v1.w, wt[3], h);
			h = fma_mix_texel(v2.x, v2.y, wt[4], h);
			h = fma_mix_texel(v2.z, v2.w, wt[5], h);
		}
		else
		{
			h = mul4(cvt4(*reinterpret_cast<const f16x4 *>(row + size_t(clampi(col0, 0, in.w - 1)) * 8u)), wt[0]);
#pragma unroll
			for (int c = 1; c < 6; c++)
				h = fma4(cvt4(*reinterpret_cast<const f16x4 *>(row + size_t(clampi(col0 + c, 0, in.w - 1)) * 8u)), wt[c], h);
		}
		// row r is row r of the upper texel's stencil and row r - 2 of the lower one's (input row 2 (y + 1) - 2 + (r - 2), clamped alike)
		const int ru = r, rl = r - 2;
		if (ru < 6)
			upper = fma4(h, ru == 0 || ru == 5 ? 0.0625f : (ru == 1 || ru == 4 ? 0.1875f : 0.25f), upper);
		if (rl >= 0)
			lower = fma4(h, rl == 0 || rl == 5 ? 0.0625f : (rl == 1 || rl == 4 ? 0.1875f : 0.25f), lower);
	}
}

// A thread makes two vertically adjacent outputs (rows y_first + 2 k and + 1 of the render area).
template <bool FEEDBACK>
__global__ __launch_bounds__(POST_BLOCK_X *POST_BLOCK_Y) POST_VGPR_BUDGET void k_bloom_downsample_2to1(DevImage in, DevImageRW out, DevImage history,
                                                                                      gr_push_bloom_downsample push, uint32_t y_first,
                                                                                      uint32_t y_end)
{
	post_wave_priority();
	const int x = blockIdx.x * POST_BLOCK_X + threadIdx.x;
	const int y = int(y_first) + 2 * int(blockIdx.y * POST_BLOCK_Y + threadIdx.y);
	if (uint32_t(x) >= push.threads[0] || uint32_t(y) >= y_end)
		return;
	float4 value[2];
	downsample_2to1_pair(in, x, y, value[0], value[1]);
#pragma unroll
	for (int i = 0; i < 2; i++)
	{
		if (uint32_t(y + i) >= y_end)
			break;
		if (FEEDBACK)
			value[i] = apply_feedback(value[i], history, (float(x) + 0.5f) * push.inv_output_size[0], (float(y + i) + 0.5f) * push.inv_output_size[1], push.lerp);
		store_rgba16f(out, x, y + i, value[i]);
	}
}

// One output texel of the 1:2 upsample: the 4 x 4 stencil; `texel(x, y)` returns the two dwords of an input texel at
// coordinates already clamped to the `w` x `h` input.
template <typename Texel>
__device__ __forceinline__ float4 upsample_1to2_value(Texel texel, int w, int h, int x, int y)
{
	// even output: texels k-2..k+1, weights (1 11 15 5)/32; odd output: texels k-1..k+2, weights (5 15 11 1)/32
	const bool odd_x = (x & 1) != 0, odd_y = (y & 1) != 0;
	const int sx = (x >> 1) - 2 + (odd_x ? 1 : 0), sy = (y >> 1) - 2 + (odd_y ? 1 : 0);
	const float wx[4] = {odd_x ? 0.15625f : 0.03125f, odd_x ? 0.46875f : 0.34375f, odd_x ? 0.34375f : 0.46875f, odd_x ? 0.03125f : 0.15625f};
	const float wy[4] = {odd_y ? 0.15625f : 0.03125f, odd_y ? 0.46875f : 0.34375f, odd_y ? 0.34375f : 0.46875f, odd_y ? 0.03125f : 0.15625f};
	int cx[4];
#pragma unroll
	for (int c = 0; c < 4; c++)
		cx[c] = clampi(sx + c, 0, w - 1);
	float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 4; r++)
	{
		const int cy = clampi(sy + r, 0, h - 1);
		float4 hsum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
		for (int c = 0; c < 4; c++)
		{
			const u32x2 t = texel(cx[c], cy);
			hsum = fma_mix_texel(t.x, t.y, wx[c], hsum); // conversion folded into the multiply-add
		}
		acc = fma4(hsum, wy[r], acc);
	}
	return acc;
}

__global__ __launch_bounds__(POST_BLOCK_X *POST_BLOCK_Y) POST_VGPR_BUDGET void k_bloom_upsample_1to2(DevImage in, DevImageRW out,
                                                                                    gr_push_bloom_upsample push, uint32_t y_first,
                                                                                    uint32_t y_end)
{
	post_wave_priority();
	const int x = blockIdx.x * POST_BLOCK_X + threadIdx.x;
	const int y = int(y_first) + blockIdx.y * POST_BLOCK_Y + threadIdx.y;
	if (uint32_t(x) >= push.threads[0] || uint32_t(y) >= y_end)
		return;
	const auto texel = [&in](int tx, int ty) { return *reinterpret_cast<const u32x2 *>(in.ptr + size_t(ty) * in.pitch + size_t(tx) * 8u); };
	store_rgba16f(out, x, y, upsample_1to2_value(texel, in.w, in.h, x, y));
}

// ---- average luminance (luminance.comp:25-67) ---------------------------------------------------------------------
// The reference walks the sample grid with ONE 8x8 workgroup (latency-bound serial loop).  Here a single 1024-thread
// workgroup strides the grid, reduces each wave64 with cross-lane shuffles, then the 16 wave partials through LDS.
// Summation order is fixed (deterministic across runs and ranks) but differs from the reference's tree: covered by the
// stated fp32 tolerance on LuminanceData.
constexpr int LUM_THREADS = 1024;
// Called by all NT threads of a workgroup (thread = 0 .. NT - 1); wave_partial: LUM_THREADS / 64 floats of LDS.  NT < LUM_THREADS: every thread
// takes LUM_THREADS / NT of the 1024 strided sums, wave by wave -- the same additions in the same order, so the result does not depend on NT.
template <int NT = LUM_THREADS>
__device__ __forceinline__ void luminance_block(const DevImage &in, gr_luminance_data *lum, const gr_push_luminance &push, int thread,
                                                float *wave_partial)
{
	static_assert(LUM_THREADS % NT == 0 && NT % 64 == 0, "whole waves of the 1024-thread order per pass");
	const int sx = int(push.size[0]), sy = int(push.size[1]);
	const float inv_x = 1.0f / float(sx), inv_y = 1.0f / float(sy);
	const int total = sx * sy;
#pragma unroll 1
	for (int first = 0; first < LUM_THREADS; first += NT)
	{
		float sum = 0.0f;
		for (int i = first + thread; i < total; i += LUM_THREADS)
		{
			const int py = i / sx, px = i - py * sx;
			sum += sample_linear_rgba16f(in, (float(px) + 0.5f) * inv_x, (float(py) + 0.5f) * inv_y).w;
		}
		sum = wave_sum(sum);
		if ((thread & 63) == 0)
			wave_partial[(first + thread) >> 6] = sum;
	}
	__syncthreads();
	if (thread == 0)
	{
		float loglum = 0.0f;
#pragma unroll
		for (int i = 0; i < LUM_THREADS / 64; i++)
			loglum += wave_partial[i];
		loglum *= inv_x * inv_y;
		loglum = fminf(fmaxf(loglum, push.min_loglum), push.max_loglum);
		const float prev = lum->average_log_luminance;
		const float new_log_luma = prev * (1.0f - push.lerp) + loglum * push.lerp;
		lum->average_log_luminance = new_log_luma;
		lum->average_linear_luminance = exp2f(new_log_luma);
		lum->average_inv_linear_luminance = exp2f(-new_log_luma);
	}
}

__global__ __launch_bounds__(LUM_THREADS) void k_luminance(DevImage in, gr_luminance_data *lum, gr_push_luminance push)
{
	post_wave_priority();
	__shared__ float wave_partial[LUM_THREADS / 64];
	luminance_block(in, lum, push, int(threadIdx.x), wave_partial);
}

// ---- the coarse end of the pyramid in two launches --------------------------------------------------------------------------
// downsample-2 -> downsample-3 (+ feedback) -> luminance -> upsample-2 -> upsample-1 touch under 3 MB at 4K, yet as five
// dependent launches they are a third of the back-of-frame chain's latency (each pays a dispatch, a fill of the machine and
// a drain for a few microseconds of work).  Fused through LDS, with every texel computed by the very functions the separate
// kernels use (so the values, their fp16 roundings between levels included, are the same):
//   k_bloom_down_pair (gr_bloom_down_tail): a workgroup makes an 8 x 8 tile of downsample-3; it first makes the patch of downsample-2 under that
//     tile's tent taps (2:1 stencil from downsample-1), stores it (LDS as fp16, and to the downsample-2 image: neighbouring
//     workgroups write identical values into the overlap), then filters the patch.
//   k_bloom_up_tail: a workgroup of 1024 threads makes a 32 x 32 tile of upsample-1 from the patch of upsample-2 under it (20 x 20
//     when the level is exactly twice its input, up to 24 x 24 under the generic taps), which it first makes from downsample-3;
//     workgroup 0 also runs the luminance reduction (same 1024-thread order as k_luminance).
// D2_EXACT / D3_EXACT / U2_EXACT / U1_EXACT mirror what the separate launchers would pick for that level (2:1 / 1:2 stencil where the
// level is exactly half / twice its input, the nine generic taps else: the odd level sizes of 1080p).
constexpr int TAIL_TILE = 8;
constexpr int TAIL_PATCH = 24; // rows / columns of downsample-2 under an 8 x 8 tile of downsample-3 (<= 2 * 8 + 4 + slack)
struct TailPatch
{
	const f16x4 *texels;
	int x0, y0, w, h; // patch origin and size inside the level
	__device__ __forceinline__ float4 fetch(int x, int y) const { return cvt4(texels[(y - y0) * w + (x - x0)]); }
};

// First and last input row (column) a run of outputs [lo, hi] of `out_n` samples through taps displaced by `reach` input
// texels in an `in_n`-sized level, with one texel of slack either side for the fp32 evaluation of the coordinates.
__device__ __forceinline__ void tap_span(int lo, int hi, int out_n, int in_n, float reach, int &first, int &last)
{
	const float scale = float(in_n) / float(out_n);
	first = clampi(int(floorf((float(lo) + 0.5f) * scale - 0.5f - reach)) - 1, 0, in_n - 1);
	last = clampi(int(floorf((float(hi) + 0.5f) * scale - 0.5f + reach)) + 2, 0, in_n - 1);
}

// the 2:1 stencil of downsample_2to1_value over a patch in LDS (same weights, same order; the patch covers every clamped index)
__device__ __forceinline__ float4 downsample_2to1_from_patch(const TailPatch &patch, int x, int y, int in_w, int in_h)
{
	const float wt[6] = {0.0625f, 0.1875f, 0.25f, 0.25f, 0.1875f, 0.0625f};
	float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll 1
	for (int r = 0; r < 6; r++)
	{
		const int iy = clampi(2 * y - 2 + r, 0, in_h - 1);
		float4 h = mul4(patch.fetch(clampi(2 * x - 2, 0, in_w - 1), iy), wt[0]);
#pragma unroll
		for (int c = 1; c < 6; c++)
			h = fma4(patch.fetch(clampi(2 * x - 2 + c, 0, in_w - 1), iy), wt[c], h);
		const float wr = r == 0 || r == 5 ? 0.0625f : (r == 1 || r == 4 ? 0.1875f : 0.25f);
		acc = fma4(h, wr, acc);
	}
	return acc;
}

// Two consecutive downsample levels in one launch: `lower` (level B, rows [y_first, y_end)) from `upper` (level A), which the
// workgroup first makes from `src` -- and stores -- under its tile's taps.  Instantiated for downsample-0 / downsample-1 (from the
// threshold level; row bands restrict level B) and for downsample-2 / downsample-3 (+ the temporal feedback).
// (the work of one workgroup of NT threads on tile (bx, by), s_patch: TAIL_PATCH^2 texels of LDS: the kernel below, and a phase of k_bloom_pyramid)
template <bool A_EXACT, bool B_EXACT, bool FEEDBACK, int NT>
__device__ __forceinline__ void down_pair_block(const int bx, const int by, const int thread, f16x4 *s_patch, const DevImage &src, const DevImageRW &upper,
                                                const DevImageRW &lower, const DevImage &history, const gr_push_bloom_downsample &push_a,
                                                const gr_push_bloom_downsample &push_b, uint32_t y_first, uint32_t y_end)
{
	const int tile_x0 = bx * TAIL_TILE, tile_y0 = int(y_first) + by * TAIL_TILE;
	const int tile_x1 = min(tile_x0 + TAIL_TILE, lower.w) - 1, tile_y1 = min(tile_y0 + TAIL_TILE, int(y_end)) - 1;
	int px0, px1, py0, py1;
	if (B_EXACT)
	{
		px0 = clampi(2 * tile_x0 - 2, 0, upper.w - 1), px1 = clampi(2 * tile_x1 + 3, 0, upper.w - 1);
		py0 = clampi(2 * tile_y0 - 2, 0, upper.h - 1), py1 = clampi(2 * tile_y1 + 3, 0, upper.h - 1);
	}
	else
	{
		tap_span(tile_x0, tile_x1, lower.w, upper.w, 1.75f, px0, px1);
		tap_span(tile_y0, tile_y1, lower.h, upper.h, 1.75f, py0, py1);
	}
	const int pw = px1 - px0 + 1, ph = py1 - py0 + 1; // <= TAIL_PATCH (checked by the launcher)
	const int upper_w = upper.w, upper_h = upper.h;
	for (int i = thread; i < pw * ph; i += NT)
	{
		const int ly = i / pw, lx = i - ly * pw;
		// level A exactly half of its input: the 2:1 stencil; else the nine taps of the generic kernel (odd level sizes: 1080p)
		const TentTaps ta = tent_taps(px0 + lx, py0 + ly, push_a.inv_output_size, push_a.inv_input_size, 1.75f);
		const f16x4 texel = pack_rgba16f(A_EXACT ? downsample_2to1_value<6>(src, px0 + lx, py0 + ly) : tent9(src, ta.u, ta.v, ta.ox, ta.oy));
		s_patch[i] = texel;
		*reinterpret_cast<f16x4 *>(upper.ptr + size_t(py0 + ly) * upper.pitch + size_t(px0 + lx) * 8u) = texel;
	}
	__syncthreads();
	if (thread >= TAIL_TILE * TAIL_TILE)
		return;
	const int x = tile_x0 + int(thread & (TAIL_TILE - 1)), y = tile_y0 + int(thread / TAIL_TILE);
	if (x >= lower.w || y > tile_y1)
		return;
	const TailPatch patch{s_patch, px0, py0, pw, ph};
	const TentTaps tb = tent_taps(x, y, push_b.inv_output_size, push_b.inv_input_size, 1.75f);
	float4 value;
	if (B_EXACT)
		value = downsample_2to1_from_patch(patch, x, y, upper_w, upper_h);
	else
	{
		const auto sample = [&](float su, float sv) {
			return sample_linear_with([&patch](int tx, int ty) { return patch.fetch(tx, ty); }, upper_w, upper_h, su, sv);
		};
		value = tent9_with(sample, tb.u, tb.v, tb.ox, tb.oy);
	}
	if (FEEDBACK)
		value = apply_feedback(value, history, tb.u, tb.v, push_b.lerp);
	store_rgba16f(lower, x, y, value);
}

template <bool A_EXACT, bool B_EXACT, bool FEEDBACK>
__global__ __launch_bounds__(256) void k_bloom_down_pair(DevImage src, DevImageRW upper, DevImageRW lower, DevImage history,
                                                         gr_push_bloom_downsample push_a, gr_push_bloom_downsample push_b, uint32_t y_first,
                                                         uint32_t y_end)
{
	post_wave_priority();
	__shared__ f16x4 s_patch[TAIL_PATCH * TAIL_PATCH];
	down_pair_block<A_EXACT, B_EXACT, FEEDBACK, 256>(int(blockIdx.x), int(blockIdx.y), int(threadIdx.x), s_patch, src, upper, lower, history, push_a, push_b, y_first, y_end);
}

// The head of the pyramid in one launch: threshold -> downsample-0 -> downsample-1, for frames whose chain of launches, not their
// arithmetic, sets the pace (up to 1440p: gr_bloom_down_head_supported) and whose three levels are exactly half of their inputs.  A
// workgroup makes an 8 x 8 tile of downsample-1; under it the <= 20 x 20 patch of downsample-0, under that the <= 44 x 44 patch of
// the threshold level, each texel by the function the separate kernels use (threshold_of_quad, the 2:1 stencil over a patch), rounded
// to fp16 between the levels as the stores round them, and stored: neighbouring workgroups write identical values into the overlap.
// 1.9 x the threshold level and 1.56 x downsample-0 are computed; one launch and one fill / drain of the machine less than
// gr_bloom_threshold + gr_bloom_down_mid.
constexpr int HEAD_D0_PATCH = 2 * TAIL_TILE + 4;       // 20
constexpr int HEAD_T_PATCH = 2 * HEAD_D0_PATCH + 4;    // 44
template <bool DYNAMIC_EXPOSURE, int NT>
__device__ __forceinline__ void down_head_block(const int bx, const int by, const int thread, f16x4 *s_thr, f16x4 *s_d0, const DevImage &hdr,
                                                const DevImageRW &thr, const DevImageRW &d0, const DevImageRW &d1, const gr_luminance_data *lum,
                                                float inv_thr_w, float inv_thr_h, bool hdr_b10)
{
	const int tile_x0 = bx * TAIL_TILE, tile_y0 = by * TAIL_TILE;
	const int tile_x1 = min(tile_x0 + TAIL_TILE, d1.w) - 1, tile_y1 = min(tile_y0 + TAIL_TILE, d1.h) - 1;
	const int px0 = clampi(2 * tile_x0 - 2, 0, d0.w - 1), px1 = clampi(2 * tile_x1 + 3, 0, d0.w - 1);
	const int py0 = clampi(2 * tile_y0 - 2, 0, d0.h - 1), py1 = clampi(2 * tile_y1 + 3, 0, d0.h - 1);
	const int qx0 = clampi(2 * px0 - 2, 0, thr.w - 1), qx1 = clampi(2 * px1 + 3, 0, thr.w - 1);
	const int qy0 = clampi(2 * py0 - 2, 0, thr.h - 1), qy1 = clampi(2 * py1 + 3, 0, thr.h - 1);
	const int pw = px1 - px0 + 1, ph = py1 - py0 + 1, qw = qx1 - qx0 + 1, qh = qy1 - qy0 + 1;
	const float threshold = DYNAMIC_EXPOSURE ? __fmul_rn(8.0f, lum->average_linear_luminance) : 8.0f;
	for (int i = thread; i < qw * qh; i += NT)
	{
		const int ly = i / qw, lx = i - ly * qw;
		const uint32_t x = uint32_t(qx0 + lx), y = uint32_t(qy0 + ly);
		u32x4 top, bottom; // the HDR texels (2 x, 2 y) .. (2 x + 1, 2 y + 1) as RGBA16F dwords
		if (hdr_b10)
		{
			const uint8_t *row0 = hdr.ptr + size_t(2u * y) * hdr.pitch + size_t(x) * 8u;
			const u32x2 pa = *reinterpret_cast<const u32x2 *>(row0), pb = *reinterpret_cast<const u32x2 *>(row0 + hdr.pitch);
			top = expand_b10g11r11_pair(pa.x, pa.y), bottom = expand_b10g11r11_pair(pb.x, pb.y);
		}
		else
		{
			const uint8_t *row0 = hdr.ptr + size_t(2u * y) * hdr.pitch + size_t(x) * 16u;
			top = *reinterpret_cast<const u32x4 *>(row0), bottom = *reinterpret_cast<const u32x4 *>(row0 + hdr.pitch);
		}
		const f16x4 texel = threshold_of_quad(top, bottom, threshold_fraction(x, inv_thr_w, hdr.w), threshold_fraction(y, inv_thr_h, hdr.h), threshold);
		s_thr[i] = texel;
		*reinterpret_cast<f16x4 *>(thr.ptr + size_t(y) * thr.pitch + size_t(x) * 8u) = texel;
	}
	__syncthreads();
	const TailPatch thr_patch{s_thr, qx0, qy0, qw, qh};
	for (int i = thread; i < pw * ph; i += NT)
	{
		const int ly = i / pw, lx = i - ly * pw;
		const f16x4 texel = pack_rgba16f(downsample_2to1_from_patch(thr_patch, px0 + lx, py0 + ly, thr.w, thr.h));
		s_d0[i] = texel;
		*reinterpret_cast<f16x4 *>(d0.ptr + size_t(py0 + ly) * d0.pitch + size_t(px0 + lx) * 8u) = texel;
	}
	__syncthreads();
	if (thread >= TAIL_TILE * TAIL_TILE)
		return;
	const int x = tile_x0 + int(thread & (TAIL_TILE - 1)), y = tile_y0 + int(thread / TAIL_TILE);
	if (x >= d1.w || y > tile_y1)
		return;
	const TailPatch d0_patch{s_d0, px0, py0, pw, ph};
	store_rgba16f(d1, x, y, downsample_2to1_from_patch(d0_patch, x, y, d0.w, d0.h));
}

template <bool DYNAMIC_EXPOSURE>
__global__ __launch_bounds__(256) void k_bloom_down_head(DevImage hdr, DevImageRW thr, DevImageRW d0, DevImageRW d1, const gr_luminance_data *lum,
                                                         float inv_thr_w, float inv_thr_h, bool hdr_b10)
{
	post_wave_priority();
	__shared__ f16x4 s_thr[HEAD_T_PATCH * HEAD_T_PATCH];
	__shared__ f16x4 s_d0[HEAD_D0_PATCH * HEAD_D0_PATCH];
	down_head_block<DYNAMIC_EXPOSURE, 256>(int(blockIdx.x), int(blockIdx.y), int(threadIdx.x), s_thr, s_d0, hdr, thr, d0, d1, lum, inv_thr_w, inv_thr_h, hdr_b10);
}

// A 256-thread form (16 x 16 tiles, one wave per SIMD: finds room on a CU beside the resident lighting waves, where a 1024-thread
// workgroup waits for a whole CU to drain -- 15 us instead of 60-100 us inside the 4K frame) was built for VERDICT r2 item 5 and
// measured on one box against this one: the FRAME gets slower with it, 0.2426 vs 0.2295 ms sustained (profiles/r03_up_tail_shape_ab.txt):
// the back chain then runs beside the lighting kernel for all of its length and lighting, which sets the frame period, loses more
// than the chain gains.  Nothing waits for the back chain, so its starved tail is free; the wide form stays.
constexpr int UP_TILE = 32;
constexpr int UP_PATCH = UP_TILE / 2 + 8; // upsample-2 texels under 32 outputs of upsample-1: k - 2 .. k + 2 for k = x / 2 when the level is
                                          // exactly twice its input, a few more under the generic taps of an odd-sized level
static_assert(UP_PATCH * UP_PATCH <= LUM_THREADS && UP_TILE * UP_TILE == LUM_THREADS, "one patch texel and one output per thread");
template <bool U2_EXACT, bool U1_EXACT, bool LUMINANCE>
__global__ __launch_bounds__(LUM_THREADS) POST_VGPR_BUDGET void k_bloom_up_tail(DevImage d3, DevImageRW u2, DevImageRW u1, gr_luminance_data *lum,
                                                                               gr_push_bloom_upsample push2, gr_push_bloom_upsample push1,
                                                                               gr_push_luminance push_lum)
{
	post_wave_priority();
	__shared__ f16x4 s_patch[UP_PATCH * UP_PATCH];
	__shared__ float wave_partial[LUM_THREADS / 64];
	const int thread = int(threadIdx.x);
	const int tile_x0 = blockIdx.x * UP_TILE, tile_y0 = blockIdx.y * UP_TILE;
	const int tile_x1 = min(tile_x0 + UP_TILE, u1.w) - 1, tile_y1 = min(tile_y0 + UP_TILE, u1.h) - 1;
	int px0, px1, py0, py1;
	if (U1_EXACT)
	{
		px0 = clampi((tile_x0 >> 1) - 2, 0, u2.w - 1), px1 = clampi((tile_x1 >> 1) + 2, 0, u2.w - 1);
		py0 = clampi((tile_y0 >> 1) - 2, 0, u2.h - 1), py1 = clampi((tile_y1 >> 1) + 2, 0, u2.h - 1);
	}
	else
	{
		tap_span(tile_x0, tile_x1, u1.w, u2.w, 0.875f, px0, px1);
		tap_span(tile_y0, tile_y1, u1.h, u2.h, 0.875f, py0, py1);
	}
	const int pw = px1 - px0 + 1, ph = py1 - py0 + 1; // <= UP_PATCH (the launcher checks the level sizes)
	if (thread < pw * ph)
	{
		const int ly = thread / pw, lx = thread - ly * pw;
		const int x = px0 + lx, y = py0 + ly;
		float4 value;
		if (U2_EXACT)
		{
			const auto texel = [&d3](int tx, int ty) { return *reinterpret_cast<const u32x2 *>(d3.ptr + size_t(ty) * d3.pitch + size_t(tx) * 8u); };
			value = upsample_1to2_value(texel, d3.w, d3.h, x, y);
		}
		else
		{
			const TentTaps t = tent_taps(x, y, push2.inv_output_size, push2.inv_input_size, 0.875f);
			value = tent9(d3, t.u, t.v, t.ox, t.oy);
		}
		const f16x4 texel16 = pack_rgba16f(value);
		s_patch[thread] = texel16;
		*reinterpret_cast<f16x4 *>(u2.ptr + size_t(y) * u2.pitch + size_t(x) * 8u) = texel16;
	}
	__syncthreads();
	{
		const int x = tile_x0 + (thread & (UP_TILE - 1)), y = tile_y0 + thread / UP_TILE;
		if (x < u1.w && y < u1.h)
		{
			if (U1_EXACT)
			{
				const auto texel = [&](int tx, int ty) { return __builtin_bit_cast(u32x2, s_patch[(ty - py0) * pw + (tx - px0)]); };
				store_rgba16f(u1, x, y, upsample_1to2_value(texel, u2.w, u2.h, x, y));
			}
			else
			{
				// the nine taps of the generic kernel over the patch (sample_linear_with clamps the indices to the level as the sampler does)
				const TailPatch patch{s_patch, px0, py0, pw, ph};
				const int level_w = u2.w, level_h = u2.h;
				const auto sample = [&](float su, float sv) {
					return sample_linear_with([&patch](int tx, int ty) { return patch.fetch(tx, ty); }, level_w, level_h, su, sv);
				};
				const TentTaps t = tent_taps(x, y, push1.inv_output_size, push1.inv_input_size, 0.875f);
				store_rgba16f(u1, x, y, tent9_with(sample, t.u, t.v, t.ox, t.oy));
			}
		}
	}
	// hdr.cpp:368-371 records the luminance pass between downsample-3 and upsample-2; nothing in between reads its result
	if (LUMINANCE && blockIdx.x == 0 && blockIdx.y == 0)
		luminance_block(d3, lum, push_lum, thread, wave_partial);
}

// The whole upsample chain in one launch (gr_bloom_up_all_supported: upsample-0 exactly twice upsample-1): a workgroup makes a 32 x 32 tile of upsample-0; under it the <= 20 x 20 patch of upsample-1, under that the patch of
// upsample-2 (<= 14 x 14 on the 1:2 stencil, a few more under the generic taps of an odd-sized level), which it makes from downsample-3.
// Every texel by the functions of k_bloom_up_tail / k_bloom_upsample_1to2, rounded to fp16 between the levels as the stores round them, all three
// levels stored (neighbouring workgroups write identical values into the overlap); workgroup 0 also runs the luminance reduction.
constexpr int UPALL_TILE = 32;               // one output per thread.  (64-wide tiles, a quarter of the workgroups: config 1 0.0393 against 0.0366 ms,
                                             // config 2 0.063 against 0.058 ms -- these launches are latency, not arithmetic; profiles/r04_host_lead_ab.txt)
constexpr int UPALL_P1 = UPALL_TILE / 2 + 4; // 20
constexpr int UPALL_P2 = 18;                 // 1:2 stencil: P1 / 2 + 4 = 14; generic taps: (P1 - 0.5) * 0.514 + 6.75 < 17
template <bool U2_EXACT, bool U1_EXACT, bool LUMINANCE, int NT>
__device__ __forceinline__ void up_all_block(const int bx, const int by, const int thread, f16x4 *s_p2, f16x4 *s_p1, float *wave_partial, const DevImage &d3,
                                             const DevImageRW &u2, const DevImageRW &u1, const DevImageRW &u0, gr_luminance_data *lum,
                                             const gr_push_bloom_upsample &push2, const gr_push_bloom_upsample &push1, const gr_push_luminance &push_lum)
{
	const int tile_x0 = bx * UPALL_TILE, tile_y0 = by * UPALL_TILE;
	const int tile_x1 = min(tile_x0 + UPALL_TILE, u0.w) - 1, tile_y1 = min(tile_y0 + UPALL_TILE, u0.h) - 1;
	// upsample-1 under the tile (1:2 stencil: k - 2 .. k + 2 for k = x / 2), upsample-2 under that
	const int ax0 = clampi((tile_x0 >> 1) - 2, 0, u1.w - 1), ax1 = clampi((tile_x1 >> 1) + 2, 0, u1.w - 1);
	const int ay0 = clampi((tile_y0 >> 1) - 2, 0, u1.h - 1), ay1 = clampi((tile_y1 >> 1) + 2, 0, u1.h - 1);
	int bx0, bx1, by0, by1;
	if (U1_EXACT)
	{
		bx0 = clampi((ax0 >> 1) - 2, 0, u2.w - 1), bx1 = clampi((ax1 >> 1) + 2, 0, u2.w - 1);
		by0 = clampi((ay0 >> 1) - 2, 0, u2.h - 1), by1 = clampi((ay1 >> 1) + 2, 0, u2.h - 1);
	}
	else
	{
		tap_span(ax0, ax1, u1.w, u2.w, 0.875f, bx0, bx1);
		tap_span(ay0, ay1, u1.h, u2.h, 0.875f, by0, by1);
	}
	const int aw = ax1 - ax0 + 1, ah = ay1 - ay0 + 1, bw = bx1 - bx0 + 1, bh = by1 - by0 + 1; // <= UPALL_P1, UPALL_P2 (the launcher checks the level sizes)
	for (int i = thread; i < bw * bh; i += NT)
	{
		const int ly = i / bw, lx = i - ly * bw;
		const int x = bx0 + lx, y = by0 + ly;
		float4 value;
		if (U2_EXACT)
		{
			const auto texel = [&d3](int tx, int ty) { return *reinterpret_cast<const u32x2 *>(d3.ptr + size_t(ty) * d3.pitch + size_t(tx) * 8u); };
			value = upsample_1to2_value(texel, d3.w, d3.h, x, y);
		}
		else
		{
			const TentTaps t = tent_taps(x, y, push2.inv_output_size, push2.inv_input_size, 0.875f);
			value = tent9(d3, t.u, t.v, t.ox, t.oy);
		}
		const f16x4 texel16 = pack_rgba16f(value);
		s_p2[i] = texel16;
		*reinterpret_cast<f16x4 *>(u2.ptr + size_t(y) * u2.pitch + size_t(x) * 8u) = texel16;
	}
	__syncthreads();
	for (int i = thread; i < aw * ah; i += NT)
	{
		const int ly = i / aw, lx = i - ly * aw;
		const int x = ax0 + lx, y = ay0 + ly;
		float4 value;
		if (U1_EXACT)
		{
			const auto texel = [&](int tx, int ty) { return __builtin_bit_cast(u32x2, s_p2[(ty - by0) * bw + (tx - bx0)]); };
			value = upsample_1to2_value(texel, u2.w, u2.h, x, y);
		}
		else
		{
			const TailPatch patch{s_p2, bx0, by0, bw, bh};
			const int level_w = u2.w, level_h = u2.h;
			const auto sample = [&](float su, float sv) {
				return sample_linear_with([&patch](int tx, int ty) { return patch.fetch(tx, ty); }, level_w, level_h, su, sv);
			};
			const TentTaps t = tent_taps(x, y, push1.inv_output_size, push1.inv_input_size, 0.875f);
			value = tent9_with(sample, t.u, t.v, t.ox, t.oy);
		}
		const f16x4 texel16 = pack_rgba16f(value);
		s_p1[i] = texel16;
		*reinterpret_cast<f16x4 *>(u1.ptr + size_t(y) * u1.pitch + size_t(x) * 8u) = texel16;
	}
	__syncthreads();
	const int tw = tile_x1 - tile_x0 + 1, th = tile_y1 - tile_y0 + 1;
	for (int i = thread; i < UPALL_TILE * th; i += NT)
	{
		const int ly = i / UPALL_TILE, lx = i - ly * UPALL_TILE;
		if (lx >= tw)
			continue;
		const int x = tile_x0 + lx, y = tile_y0 + ly;
		const auto texel = [&](int tx, int ty) { return __builtin_bit_cast(u32x2, s_p1[(ty - ay0) * aw + (tx - ax0)]); };
		store_rgba16f(u0, x, y, upsample_1to2_value(texel, u1.w, u1.h, x, y));
	}
	if (LUMINANCE && bx == 0 && by == 0)
		luminance_block<NT>(d3, lum, push_lum, thread, wave_partial);
}

// NT = 1024: one output per thread.  NT = 256 (the same tiles, four outputs per thread): a workgroup of four waves, one per SIMD, starts wherever
// ONE lighting workgroup has retired; the sixteen waves of the 1024-thread form need four retirements on one CU (65-88 us inside the 4K frame for
// 14 us of work: profiles/r06_back_chain_beside_lighting.txt).
template <bool U2_EXACT, bool U1_EXACT, bool LUMINANCE, int NT>
__global__ __launch_bounds__(NT) POST_VGPR_BUDGET void k_bloom_up_all(DevImage d3, DevImageRW u2, DevImageRW u1, DevImageRW u0, gr_luminance_data *lum,
                                                                     gr_push_bloom_upsample push2, gr_push_bloom_upsample push1,
                                                                     gr_push_luminance push_lum)
{
	post_wave_priority();
	__shared__ f16x4 s_p2[UPALL_P2 * UPALL_P2];
	__shared__ f16x4 s_p1[UPALL_P1 * UPALL_P1];
	__shared__ float wave_partial[LUM_THREADS / 64];
	up_all_block<U2_EXACT, U1_EXACT, LUMINANCE, NT>(int(blockIdx.x), int(blockIdx.y), int(threadIdx.x), s_p2, s_p1, wave_partial, d3, u2, u1, u0, lum, push2, push1,
	                                                push_lum);
}

// ---- the whole pyramid in ONE launch (frames up to 640 x 384) ------------------------------------------------------------------------
// For a small frame the bloom pass is a chain of launches, not arithmetic: each of down_head -> down_tail -> up_all pays a dispatch, a fill and a
// drain of the machine for a few microseconds of work, and the host pays a runtime call for each.  k_bloom_pyramid is the three of them as three
// block ranges ("phases") of one grid of 256-thread workgroups: [0, n0) the tiles of down_head_block, [n0, n0 + n1) those of down_pair_block
// (downsample-2 / -3 + feedback), the rest those of up_all_block (+ the luminance reduction in its first workgroup, in the 1024-thread order).
// A workgroup of phase k + 1 waits until ALL workgroups of phase k have published their level (one counter per phase: release add after the
// workgroup's stores, acquire spin before its loads, agent scope).  Workgroups are dispatched in linear id order (per XCD: id mod 8) and a
// workgroup only ever waits for lower ids, so the unfinished workgroup with the lowest id can always run: no co-residency requirement, other
// kernels may share the machine.  (Two such launches overlapping on different streams could hold each other's slots; the executor issues them on
// one in-order stream, and the spin is bounded: after PYRAMID_SPIN_LIMIT polls a workgroup gives up, counts itself in sync[3] and goes on --
// gr_debug_pyramid_giveups reads the count, which every test of this launch checks to be zero.)
// Values: every texel by the block functions the separate launches run => byte-identical levels (tests/test_gpu_post.py).
constexpr uint32_t PYRAMID_SPIN_LIMIT = 1u << 22; // x ~0.1 us per poll
struct PyramidArgs
{
	DevImage hdr, history;
	DevImageRW thr, d0, d1, d2, d3, u2, u1, u0;
	gr_luminance_data *lum;
	float inv_thr_w, inv_thr_h;
	gr_push_bloom_downsample push_d2, push_d3;
	gr_push_bloom_upsample push_u2, push_u1;
	gr_push_luminance push_lum;
	uint32_t n0, n1, n2;      // workgroups per phase
	uint32_t head_x, pair_x, up_x; // tiles per row of each phase
	uint32_t *sync;           // {phase 0 done, phase 1 done, phase 2 done, give-ups}: zero between launches (the last workgroup clears 0..2)
	bool hdr_b10;
};

__device__ __forceinline__ void pyramid_publish(uint32_t *counter)
{
	__syncthreads(); // every wave's stores have been issued and waited for (workgroup-scope release)
	if (threadIdx.x == 0)
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pyramid_wait(uint32_t *counter, uint32_t target, uint32_t *giveups)
{
	if (threadIdx.x == 0)
	{
		uint32_t polls = 0;
		// relaxed polls (an acquire load invalidates the caches on every poll: with a hundred workgroups waiting that is a storm the other kernels
		// on the machine pay for); the acquire is the one fence behind the barrier below
		while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
		{
			__builtin_amdgcn_s_sleep(8);
			if (++polls == PYRAMID_SPIN_LIMIT)
			{
				__hip_atomic_fetch_add(giveups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				break;
			}
		}
	}
	__syncthreads();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // every wave's loads below see what the counter's release published
}

constexpr size_t PYRAMID_LDS_HEAD = sizeof(f16x4) * (HEAD_T_PATCH * HEAD_T_PATCH + HEAD_D0_PATCH * HEAD_D0_PATCH);
constexpr size_t PYRAMID_LDS_UP = sizeof(f16x4) * (UPALL_P2 * UPALL_P2 + UPALL_P1 * UPALL_P1) + sizeof(float) * (LUM_THREADS / 64);
constexpr size_t PYRAMID_LDS = PYRAMID_LDS_HEAD > PYRAMID_LDS_UP ? PYRAMID_LDS_HEAD : PYRAMID_LDS_UP;
static_assert(PYRAMID_LDS >= sizeof(f16x4) * TAIL_PATCH * TAIL_PATCH, "the down_pair patch fits too");
constexpr int PYRAMID_THREADS = 256;

template <bool DYNAMIC_EXPOSURE, bool D2_EXACT, bool D3_EXACT, bool U2_EXACT, bool U1_EXACT>
__global__ __launch_bounds__(PYRAMID_THREADS) void k_bloom_pyramid(PyramidArgs a)
{
	post_wave_priority();
	__shared__ __attribute__((aligned(16))) uint8_t s_raw[PYRAMID_LDS];
	const int thread = int(threadIdx.x);
	const uint32_t block = blockIdx.x;
	if (block < a.n0)
	{
		f16x4 *s_thr = reinterpret_cast<f16x4 *>(s_raw), *s_d0 = s_thr + HEAD_T_PATCH * HEAD_T_PATCH;
		down_head_block<DYNAMIC_EXPOSURE, PYRAMID_THREADS>(int(block % a.head_x), int(block / a.head_x), thread, s_thr, s_d0, a.hdr, a.thr, a.d0, a.d1, a.lum,
		                                                   a.inv_thr_w, a.inv_thr_h, a.hdr_b10);
		pyramid_publish(a.sync + 0);
	}
	else if (block < a.n0 + a.n1)
	{
		const uint32_t tile = block - a.n0;
		pyramid_wait(a.sync + 0, a.n0, a.sync + 3);
		down_pair_block<D2_EXACT, D3_EXACT, true, PYRAMID_THREADS>(int(tile % a.pair_x), int(tile / a.pair_x), thread, reinterpret_cast<f16x4 *>(s_raw), DevImage{a.d1.ptr, a.d1.w, a.d1.h, a.d1.pitch},
		                                                           a.d2, a.d3, a.history, a.push_d2, a.push_d3, 0u, uint32_t(a.d3.h));
		pyramid_publish(a.sync + 1);
	}
	else
	{
		const uint32_t tile = block - a.n0 - a.n1;
		pyramid_wait(a.sync + 1, a.n1, a.sync + 3);
		f16x4 *s_p2 = reinterpret_cast<f16x4 *>(s_raw), *s_p1 = s_p2 + UPALL_P2 * UPALL_P2;
		float *wave_partial = reinterpret_cast<float *>(s_p1 + UPALL_P1 * UPALL_P1);
		up_all_block<U2_EXACT, U1_EXACT, DYNAMIC_EXPOSURE, PYRAMID_THREADS>(int(tile % a.up_x), int(tile / a.up_x), thread, s_p2, s_p1, wave_partial,
		                                                                     DevImage{a.d3.ptr, a.d3.w, a.d3.h, a.d3.pitch}, a.u2, a.u1, a.u0, a.lum, a.push_u2, a.push_u1,
		                                                                     a.push_lum);
		// the last workgroup of the launch leaves the counters at zero for the next one
		__syncthreads();
		if (thread == 0 && __hip_atomic_fetch_add(a.sync + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == a.n2 - 1u)
		{
			__hip_atomic_store(a.sync + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(a.sync + 2, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

// ---- tonemap (tonemap.frag:30-66) -----------------------------------------------------------------------------------
// Per pixel this pass is 12.5 B of traffic but three filmic curves + three sRGB encodes, i.e. VALU-heavy; the arithmetic is
// kept lean so that the kernel stays on the HBM side of its roofline: the filmic division is num * v_rcp_f32(den) (1 ulp),
// and when the bloom level is exactly 1/4 resolution (the InputRelative 0.25 level of even-sized targets) the four pixels of
// a lane share one 3x2 bloom footprint that is lerped separably instead of four independent 4-tap fetches.
// uncharted2(x) * white_scale with the scale folded into the numerator's constants (tonemap.frag:42-53).
__device__ __forceinline__ float uncharted2_scaled(float x)
{
	constexpr float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f, W = 11.2f;
	constexpr float white = ((W * (A * W + C * B) + D * E) / (W * (A * W + B) + D * F)) - E / F;
	constexpr float ws = 1.0f / white;
	const float num = fmaf(x, fmaf(A * ws, x, C * B * ws), D * E * ws);
	const float den = fmaf(x, fmaf(A, x, B), D * F);
	return fmaf(num, __builtin_amdgcn_rcpf(den), -(E / F) * ws);
}

constexpr int TONEMAP_PX = 4; // pixels per lane: 2 x 16 B loads, one 16 B store
constexpr int TONEMAP_BLOCK_X = 64;
constexpr int TONEMAP_BLOCK_Y = 4;
constexpr int TONEMAP_ROW_GROUPS = 8; // a workgroup walks 8 x TONEMAP_BLOCK_Y rows: the 8 KiB curve table is staged once per 8192 pixels

// v_fma_mix_f32: an fp32 fma whose operands may be fp16 halves of a 32-bit register, converted (exactly) on the way in.  The pass reads
// fp16 texels and computes in fp32: a conversion per operand is a half-rate VALU slot that this instruction does not spend.  HALF = 0:
// the low half of the word, 1: the high half.  Results are the ones of v_cvt_f32_f16 followed by the fp32 operation (one rounding each).
template <int HALF>
__device__ __forceinline__ float mix_sub(uint32_t a, uint32_t b) // float(half(a)) - float(half(b))
{
	float r;
	if (HALF)
		asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
	else
		asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
template <int HALF>
__device__ __forceinline__ float mix_fma(float x, float y, uint32_t c) // fmaf(x, y, float(half(c)))
{
	float r;
	if (HALF)
		asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(c));
	else
		asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(c));
	return r;
}
template <int HALF>
__device__ __forceinline__ float mix_add(uint32_t h, float b) // float(half(h)) + b
{
	float r;
	if (HALF)
		asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(b));
	else
		asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(b));
	return r;
}

// The staircase of device_common.hpp (tonemap_srgb8_lut) as this kernel stages it: 1024 buckets rotated so that the bucket of x is at
// (bits(x) >> 17) & 1023 -- no subtraction of the table's first float -- and 0xff000000 (the alpha byte) in every value word.  x is
// clamped to [2^-12, 16): the curve is 255 from the white point 11.2 on, so the last bucket below 16 answers for everything above.
constexpr uint32_t TONEMAP_TABLE_ROTATION = (TONEMAP_TABLE_MIN_BITS >> TONEMAP_TABLE_BUCKET_SHIFT) & 1023u;
static_assert(TONEMAP_TABLE_ENTRIES == 1025u && TONEMAP_TABLE_BUCKET_SHIFT == 17u, "bucket index = 10 bits of the float");
__device__ __forceinline__ uint32_t tonemap_srgb8_staged(float x, const uint2 *table)
{
	x = __builtin_amdgcn_fmed3f(x, 0x1p-12f, 0x1.fffffep+3f);
	const uint32_t offset = (__builtin_bit_cast(uint32_t, x) >> (TONEMAP_TABLE_BUCKET_SHIFT - 3u)) & (1023u << 3u);
	const uint2 e = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(table) + offset);
	return e.y + (x >= __builtin_bit_cast(float, e.x) ? 1u : 0u);
}

template <bool DYNAMIC_EXPOSURE, bool SRGB, bool QUARTER_BLOOM>
__global__ __launch_bounds__(TONEMAP_BLOCK_X *TONEMAP_BLOCK_Y) void k_tonemap(DevImage hdr, DevImage bloom, DevImageRW out,
                                                                              const gr_luminance_data *lum, const uint2 *encode_lut,
                                                                              gr_push_tonemap push, uint32_t y_first, uint32_t y_end, bool hdr_b10, int row_groups)
{
	post_wave_priority();
	// *_SRGB output: the curve and the store's encode are one table lookup per channel (tonemap_srgb8_staged), staged in LDS.
	__shared__ uint2 s_table[SRGB ? 1024 : 1];
	if (SRGB)
	{
		for (uint32_t i = threadIdx.y * TONEMAP_BLOCK_X + threadIdx.x; i < 1024u; i += TONEMAP_BLOCK_X * TONEMAP_BLOCK_Y)
		{
			const uint2 e = encode_lut[i];
			s_table[(i + TONEMAP_TABLE_ROTATION) & 1023u] = make_uint2(e.x, e.y | 0xff000000u);
		}
		__syncthreads();
	}
	const int x0 = (blockIdx.x * TONEMAP_BLOCK_X + threadIdx.x) * TONEMAP_PX;
	if (x0 >= hdr.w)
		return;
	float scale = push.dynamic_exposure;
	if (DYNAMIC_EXPOSURE)
		scale *= lum->average_inv_linear_luminance;
	const float inv_w = 1.0f / float(hdr.w), inv_h = 1.0f / float(hdr.h);

#pragma unroll 1
	for (int group = 0; group < row_groups; group++)
	{
		// a wave is one row of the block: the row's base addresses are scalars, the lane adds a 32-bit offset
		static_assert(TONEMAP_BLOCK_X == 64, "threadIdx.y is wave-uniform");
		const int y = __builtin_amdgcn_readfirstlane(int(y_first) + (blockIdx.y * row_groups + group) * TONEMAP_BLOCK_Y + int(threadIdx.y));
		if (uint32_t(y) >= y_end)
			return;
		const float v = (float(y) + 0.5f) * inv_h;

		const uint8_t *row = hdr.ptr + size_t(y) * hdr.pitch;
		uint32_t packed[TONEMAP_PX];
		const bool full = (x0 + TONEMAP_PX <= hdr.w) && ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
		u32x2 texels[TONEMAP_PX]; // RGBA16F words: {g:r, a:b}
		if (hdr_b10)
		{
			// B10G11R11 target: 16 bytes per lane; every texel expands exactly into the RGBA16F texel the code below reads
			uint32_t words[TONEMAP_PX];
			if (full)
			{
				const u32x4 t = *reinterpret_cast<const u32x4 *>(row + uint32_t(x0) * 4u);
				words[0] = t.x, words[1] = t.y, words[2] = t.z, words[3] = t.w;
			}
			else
			{
#pragma unroll
				for (int i = 0; i < TONEMAP_PX; i++)
					words[i] = *reinterpret_cast<const uint32_t *>(row + uint32_t(min(x0 + i, hdr.w - 1)) * 4u);
			}
#pragma unroll
			for (int i = 0; i < TONEMAP_PX; i++)
			{
				uint32_t rg, ba;
				expand_b10g11r11(words[i], rg, ba);
				texels[i] = u32x2{rg, ba};
			}
		}
		else if (full)
		{
			// 32 contiguous bytes per lane, 2 KiB per wave per row.
			const u32x4 lo = *reinterpret_cast<const u32x4 *>(row + uint32_t(x0) * 8u);
			const u32x4 hi = *reinterpret_cast<const u32x4 *>(row + uint32_t(x0) * 8u + 16u);
			texels[0] = u32x2{lo.x, lo.y};
			texels[1] = u32x2{lo.z, lo.w};
			texels[2] = u32x2{hi.x, hi.y};
			texels[3] = u32x2{hi.z, hi.w};
		}
		else
		{
#pragma unroll
			for (int i = 0; i < TONEMAP_PX; i++)
				texels[i] = *reinterpret_cast<const u32x2 *>(row + uint32_t(min(x0 + i, hdr.w - 1)) * 8u);
		}

		// bloom per pixel and channel
		float bloom_rgb[TONEMAP_PX][3];
		if (QUARTER_BLOOM)
		{
			// hdr = 4 x bloom in both axes and x0 = 4k: the unnormalised bloom coordinate of pixel x0 + i is
			// k + (i + 0.5)/4 - 0.5, i.e. texel pairs (k-1,k),(k-1,k),(k,k+1),(k,k+1) with weights .625,.875,.125,.375;
			// rows likewise from y.  Same StockSampler::LinearClamp result, evaluated separably (rows first).
			const int k = x0 >> 2;
			const int j = (y >> 2) - (((y & 3) < 2) ? 1 : 0);
			const float wy = 0.125f + 0.25f * float((y + 2) & 3);
			const int r0 = clampi(j, 0, bloom.h - 1), r1 = clampi(j + 1, 0, bloom.h - 1);
			const uint8_t *row0 = bloom.ptr + size_t(r0) * bloom.pitch, *row1 = bloom.ptr + size_t(r1) * bloom.pitch;
			float col[3][3];
#pragma unroll
			for (int c = 0; c < 3; c++)
			{
				const uint32_t cx = uint32_t(clampi(k - 1 + c, 0, bloom.w - 1)) * 8u;
				const u32x2 t0 = *reinterpret_cast<const u32x2 *>(row0 + cx), t1 = *reinterpret_cast<const u32x2 *>(row1 + cx);
				// fmaf(float(t1) - float(t0), wy, float(t0)) per channel
				col[c][0] = mix_fma<0>(mix_sub<0>(t1.x, t0.x), wy, t0.x);
				col[c][1] = mix_fma<1>(mix_sub<1>(t1.x, t0.x), wy, t0.x);
				col[c][2] = mix_fma<0>(mix_sub<0>(t1.y, t0.y), wy, t0.y);
			}
#pragma unroll
			for (int ch = 0; ch < 3; ch++)
			{
				bloom_rgb[0][ch] = fmaf(col[1][ch] - col[0][ch], 0.625f, col[0][ch]);
				bloom_rgb[1][ch] = fmaf(col[1][ch] - col[0][ch], 0.875f, col[0][ch]);
				bloom_rgb[2][ch] = fmaf(col[2][ch] - col[1][ch], 0.125f, col[1][ch]);
				bloom_rgb[3][ch] = fmaf(col[2][ch] - col[1][ch], 0.375f, col[1][ch]);
			}
		}
		else
		{
#pragma unroll
			for (int i = 0; i < TONEMAP_PX; i++)
			{
				const float u = (float(x0 + i) + 0.5f) * inv_w;
				const float4 b = sample_linear_rgba16f(bloom, u, v);
				bloom_rgb[i][0] = b.x;
				bloom_rgb[i][1] = b.y;
				bloom_rgb[i][2] = b.z;
			}
		}

		// (hdr + bloom) * exposure
		float x[TONEMAP_PX][3];
		uint32_t top = 0u;
#pragma unroll
		for (int i = 0; i < TONEMAP_PX; i++)
		{
			x[i][0] = mix_add<0>(texels[i].x, bloom_rgb[i][0]) * scale;
			x[i][1] = mix_add<1>(texels[i].x, bloom_rgb[i][1]) * scale;
			x[i][2] = mix_add<0>(texels[i].y, bloom_rgb[i][2]) * scale;
			if (SRGB)
				top = max(max(top, __builtin_bit_cast(uint32_t, x[i][0])), max(__builtin_bit_cast(uint32_t, x[i][1]), __builtin_bit_cast(uint32_t, x[i][2])));
		}
		// The table covers finite x >= 0.  A negative, infinite or NaN colour (as an unsigned word: >= 0x7f800000) sends the
		// wave through the formula, so that even then the bytes are the ones the shader's arithmetic produces.
		if (SRGB && !__any(top >= 0x7f800000u))
		{
			// the value words carry the alpha byte; a channel's shift pushes it out of the word
#pragma unroll
			for (int i = 0; i < TONEMAP_PX; i++)
				packed[i] = tonemap_srgb8_staged(x[i][0], s_table) | (tonemap_srgb8_staged(x[i][1], s_table) << 8) | (tonemap_srgb8_staged(x[i][2], s_table) << 16);
		}
		else
		{
#pragma unroll
			for (int i = 0; i < TONEMAP_PX; i++)
			{
				const float tr = uncharted2_scaled(x[i][0]), tg = uncharted2_scaled(x[i][1]), tb = uncharted2_scaled(x[i][2]);
				if (SRGB)
					packed[i] = encode_srgb8(tr) | (encode_srgb8(tg) << 8) | (encode_srgb8(tb) << 16) | 0xff000000u;
				else
					packed[i] = encode_unorm8(tr) | (encode_unorm8(tg) << 8) | (encode_unorm8(tb) << 16) | 0xff000000u;
			}
		}

		uint8_t *orow = out.ptr + size_t(y) * out.pitch;
		if (full && ((reinterpret_cast<uintptr_t>(orow) & 15u) == 0))
			*reinterpret_cast<u32x4 *>(orow + uint32_t(x0) * 4u) = u32x4{packed[0], packed[1], packed[2], packed[3]};
		else
		{
			for (int i = 0; i < TONEMAP_PX && x0 + i < hdr.w; i++)
				*reinterpret_cast<uint32_t *>(orow + uint32_t(x0 + i) * 4u) = packed[i];
		}
	}
}


static bool is_rgba16f(const gr_image *img)
{
	return img && img->ptr && img->format == GR_FORMAT_R16G16B16A16_SFLOAT && img->width && img->height &&
	       img->pitch_bytes >= img->width * 8u && (img->pitch_bytes & 7u) == 0;
}
static bool is_b10g11r11(const gr_image *img)
{
	return img && img->ptr && img->format == GR_FORMAT_B10G11R11_UFLOAT_PACK32 && img->width && img->height &&
	       img->pitch_bytes >= img->width * 4u && (img->pitch_bytes & 3u) == 0;
}
// an HDR colour target as the passes that only read it take it: RGBA16F, or the reference's default B10G11R11_UFLOAT_PACK32
static bool is_hdr_target(const gr_image *img) { return is_rgba16f(img) || is_b10g11r11(img); }
// What gr_bloom_downsample / gr_bloom_upsample pick for a level: the constant-weight stencil when it is exactly 2:1 / 1:2.
static bool downsample_is_exact(const gr_image *in, const gr_push_bloom_downsample *push)
{
	static const bool allow_stencil = gr_measurement_switch("GR_NO_STENCIL") == nullptr;
	return allow_stencil && in->width == 2u * push->threads[0] && in->height == 2u * push->threads[1] && (in->pitch_bytes & 15u) == 0 &&
	       (reinterpret_cast<uintptr_t>(in->ptr) & 15u) == 0 && push->inv_output_size[0] == 1.0f / float(push->threads[0]) &&
	       push->inv_output_size[1] == 1.0f / float(push->threads[1]) && push->inv_input_size[0] == 1.0f / float(in->width) &&
	       push->inv_input_size[1] == 1.0f / float(in->height);
}
static bool upsample_is_exact(const gr_image *in, const gr_push_bloom_upsample *push)
{
	static const bool allow_stencil = gr_measurement_switch("GR_NO_STENCIL") == nullptr;
	return allow_stencil && push->threads[0] == 2u * in->width && push->threads[1] == 2u * in->height &&
	       push->inv_output_size[0] == 1.0f / float(push->threads[0]) && push->inv_output_size[1] == 1.0f / float(push->threads[1]) &&
	       push->inv_input_size[0] == 1.0f / float(in->width) && push->inv_input_size[1] == 1.0f / float(in->height);
}

} // namespace

extern "C" {

int gr_bloom_threshold(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *out, const gr_luminance_data *lum,
                       const gr_push_bloom_threshold *push)
{
	return gr_bloom_threshold_rows(ctx, stream, hdr, out, lum, push, nullptr);
}

int gr_bloom_threshold_rows(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *out, const gr_luminance_data *lum,
                            const gr_push_bloom_threshold *push, const gr_rows *rows)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push != nullptr);
	GR_CHECK_ARG(ctx, is_hdr_target(hdr) && is_rgba16f(out));
	GR_CHECK_ARG(ctx, push->threads[0] <= out->width && push->threads[1] <= out->height);
	const bool b10 = hdr->format == GR_FORMAT_B10G11R11_UFLOAT_PACK32;
	if (push->threads[0] == 0 || push->threads[1] == 0)
		return GR_OK;
	const RowSpan span = resolve_rows(rows, push->threads[1]);
	if (span.count() == 0)
		return GR_OK;
	dim3 block(POST_BLOCK_X, POST_BLOCK_Y);
	dim3 grid(gr_div_up(push->threads[0], POST_BLOCK_X), gr_div_up(span.count(), POST_BLOCK_Y));
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "bloom_threshold"};
	static const bool allow_stencil = gr_measurement_switch("GR_NO_STENCIL") == nullptr; // A/B switch for measurements
	const bool exact = allow_stencil && hdr->width == 2u * push->threads[0] && hdr->height == 2u * push->threads[1] && (push->threads[0] & 1u) == 0 &&
	                   out->width == push->threads[0] && (hdr->pitch_bytes & 15u) == 0 && (out->pitch_bytes & 15u) == 0 &&
	                   (reinterpret_cast<uintptr_t>(hdr->ptr) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out->ptr) & 15u) == 0 &&
	                   push->inv_output_size[0] == 1.0f / float(push->threads[0]) && push->inv_output_size[1] == 1.0f / float(push->threads[1]);
	if (exact)
	{
		const uint32_t pairs = push->threads[0] / 2u;
		dim3 grid2(gr_div_up(pairs, POST_BLOCK_X), gr_div_up(span.count(), POST_BLOCK_Y));
		if (lum)
			hipLaunchKernelGGL(k_bloom_threshold_2to1<true>, grid2, block, 0, gr_to_stream(stream), to_dev(hdr), to_dev_rw(out), lum, pairs, push->inv_output_size[0], push->inv_output_size[1], span.first, span.end, b10);
		else
			hipLaunchKernelGGL(k_bloom_threshold_2to1<false>, grid2, block, 0, gr_to_stream(stream), to_dev(hdr), to_dev_rw(out), lum, pairs, push->inv_output_size[0], push->inv_output_size[1], span.first, span.end, b10);
	}
	else if (lum)
		hipLaunchKernelGGL(k_bloom_threshold<true>, grid, block, 0, gr_to_stream(stream), to_dev(hdr), to_dev_rw(out), lum, *push,
		                   span.first, span.end, b10);
	else
		hipLaunchKernelGGL(k_bloom_threshold<false>, grid, block, 0, gr_to_stream(stream), to_dev(hdr), to_dev_rw(out), lum, *push,
		                   span.first, span.end, b10);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_bloom_downsample(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, const gr_image *history,
                        const gr_push_bloom_downsample *push)
{
	return gr_bloom_downsample_rows(ctx, stream, in, out, history, push, nullptr);
}

int gr_bloom_downsample_rows(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, const gr_image *history,
                             const gr_push_bloom_downsample *push, const gr_rows *rows)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push != nullptr);
	GR_CHECK_ARG(ctx, is_rgba16f(in) && is_rgba16f(out));
	GR_CHECK_ARG(ctx, !history || (is_rgba16f(history) && history->ptr != out->ptr));
	GR_CHECK_ARG(ctx, push->threads[0] <= out->width && push->threads[1] <= out->height);
	if (push->threads[0] == 0 || push->threads[1] == 0)
		return GR_OK;
	const RowSpan span = resolve_rows(rows, push->threads[1]);
	if (span.count() == 0)
		return GR_OK;
	dim3 block(POST_BLOCK_X, POST_BLOCK_Y);
	dim3 grid(gr_div_up(push->threads[0], POST_BLOCK_X), gr_div_up(span.count(), POST_BLOCK_Y));
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "bloom_downsample"};
	// Exact 2:1 level (and push constants that say so): constant-weight stencil, two outputs per thread.
	const bool exact = downsample_is_exact(in, push);
	if (exact)
	{
		grid.y = gr_div_up(gr_div_up(span.count(), 2u), POST_BLOCK_Y); // two output rows per thread
		if (history)
			hipLaunchKernelGGL(k_bloom_downsample_2to1<true>, grid, block, 0, gr_to_stream(stream), to_dev(in), to_dev_rw(out),
			                   to_dev(history), *push, span.first, span.end);
		else
			hipLaunchKernelGGL(k_bloom_downsample_2to1<false>, grid, block, 0, gr_to_stream(stream), to_dev(in), to_dev_rw(out),
			                   DevImage{}, *push, span.first, span.end);
	}
	else if (history)
		hipLaunchKernelGGL(k_bloom_downsample<true>, grid, block, 0, gr_to_stream(stream), to_dev(in), to_dev_rw(out),
		                   to_dev(history), *push, span.first, span.end);
	else
		hipLaunchKernelGGL(k_bloom_downsample<false>, grid, block, 0, gr_to_stream(stream), to_dev(in), to_dev_rw(out),
		                   DevImage{}, *push, span.first, span.end);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_bloom_upsample(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, const gr_push_bloom_upsample *push)
{
	return gr_bloom_upsample_rows(ctx, stream, in, out, push, nullptr);
}

int gr_bloom_upsample_rows(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, const gr_push_bloom_upsample *push,
                           const gr_rows *rows)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push != nullptr);
	GR_CHECK_ARG(ctx, is_rgba16f(in) && is_rgba16f(out));
	GR_CHECK_ARG(ctx, push->threads[0] <= out->width && push->threads[1] <= out->height);
	if (push->threads[0] == 0 || push->threads[1] == 0)
		return GR_OK;
	const RowSpan span = resolve_rows(rows, push->threads[1]);
	if (span.count() == 0)
		return GR_OK;
	dim3 block(POST_BLOCK_X, POST_BLOCK_Y);
	dim3 grid(gr_div_up(push->threads[0], POST_BLOCK_X), gr_div_up(span.count(), POST_BLOCK_Y));
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "bloom_upsample"};
	const bool exact = upsample_is_exact(in, push);
	if (exact)
		hipLaunchKernelGGL(k_bloom_upsample_1to2, grid, block, 0, gr_to_stream(stream), to_dev(in), to_dev_rw(out), *push, span.first,
		                   span.end);
	else
		hipLaunchKernelGGL(k_bloom_upsample, grid, block, 0, gr_to_stream(stream), to_dev(in), to_dev_rw(out), *push, span.first, span.end);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_bloom_down_mid_supported(const gr_image *threshold, const gr_image *d0, const gr_image *d1, const gr_push_bloom_downsample *push_d0,
                                const gr_push_bloom_downsample *push_d1)
{
	static const bool allow_fusion = gr_measurement_switch("GR_NO_MID_FUSION") == nullptr; // A/B switch for measurements
	if (!allow_fusion || !threshold || !d0 || !d1 || !push_d0 || !push_d1 || !is_rgba16f(threshold) || !is_rgba16f(d0) || !is_rgba16f(d1))
		return 0;
	if (push_d0->threads[0] != d0->width || push_d0->threads[1] != d0->height || push_d1->threads[0] != d1->width || push_d1->threads[1] != d1->height)
		return 0;
	// the patch of downsample-0 under an 8 x 8 tile of downsample-1 must fit the kernel's LDS patch
	if (d1->width == 0 || d1->height == 0 || float(d0->width) > 2.3f * float(d1->width) || float(d0->height) > 2.3f * float(d1->height))
		return 0;
	// Worth it only where the chain of launches, not the arithmetic, sets the pace: the fused form recomputes 56 % of downsample-0
	// (overlapping patches).  Measured on one box: 1080p frame 0.0748 -> 0.0725 ms, 256 x 256 post chain 0.0622 -> 0.0587 ms, but the
	// 4K frame 0.2323 -> 0.2389 ms (the extra work runs beside the lighting kernel).  Hence a size limit: up to a 1440p frame's level.
	static const bool any_size = gr_measurement_switch("GR_MID_FUSION_ANY_SIZE") != nullptr;
	if (!any_size && uint64_t(d1->width) * d1->height > 65536u)
		return 0;
	return 1;
}

int gr_bloom_down_mid(gr_ctx *ctx, gr_stream stream, const gr_image *threshold, const gr_image *d0, const gr_image *d1,
                      const gr_push_bloom_downsample *push_d0, const gr_push_bloom_downsample *push_d1, const gr_rows *rows_d1)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, threshold && d0 && d1 && push_d0 && push_d1);
	GR_CHECK_ARG(ctx, is_rgba16f(threshold) && is_rgba16f(d0) && is_rgba16f(d1) && d0->ptr != d1->ptr && threshold->ptr != d0->ptr);
	GR_CHECK_ARG(ctx, push_d0->threads[0] == d0->width && push_d0->threads[1] == d0->height);
	GR_CHECK_ARG(ctx, push_d1->threads[0] == d1->width && push_d1->threads[1] == d1->height);
	GR_CHECK_ARG(ctx, float(d0->width) <= 2.3f * float(d1->width) && float(d0->height) <= 2.3f * float(d1->height));
	const RowSpan span = resolve_rows(rows_d1, d1->height);
	if (span.count() == 0 || d1->width == 0)
		return GR_OK;
	dim3 grid(gr_div_up(d1->width, TAIL_TILE), gr_div_up(span.count(), TAIL_TILE));
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "bloom_down_mid"};
	auto launch = [&](auto kernel) {
		hipLaunchKernelGGL(kernel, grid, dim3(256), 0, gr_to_stream(stream), to_dev(threshold), to_dev_rw(d0), to_dev_rw(d1), DevImage{}, *push_d0, *push_d1,
		                   span.first, span.end);
	};
	const bool d0_exact = downsample_is_exact(threshold, push_d0), d1_exact = downsample_is_exact(d0, push_d1);
	if (d0_exact && d1_exact) launch(k_bloom_down_pair<true, true, false>);
	else if (d0_exact) launch(k_bloom_down_pair<true, false, false>);
	else if (d1_exact) launch(k_bloom_down_pair<false, true, false>);
	else launch(k_bloom_down_pair<false, false, false>);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_bloom_down_head_supported(const gr_image *hdr, const gr_image *threshold, const gr_image *d0, const gr_image *d1,
                                 const gr_push_bloom_threshold *push_t, const gr_push_bloom_downsample *push_d0, const gr_push_bloom_downsample *push_d1)
{
	static const bool allow_fusion = gr_measurement_switch("GR_NO_HEAD_FUSION") == nullptr; // A/B switch for measurements
	if (!allow_fusion || !hdr || !push_t || !gr_bloom_down_mid_supported(threshold, d0, d1, push_d0, push_d1))
		return 0;
	if (!is_hdr_target(hdr) || !downsample_is_exact(threshold, push_d0) || !downsample_is_exact(d0, push_d1))
		return 0;
	// the threshold level as the 2:1 form of gr_bloom_threshold takes it
	return hdr->width == 2u * push_t->threads[0] && hdr->height == 2u * push_t->threads[1] && threshold->width == push_t->threads[0] &&
	       threshold->height == push_t->threads[1] && (hdr->pitch_bytes & 15u) == 0 && (reinterpret_cast<uintptr_t>(hdr->ptr) & 15u) == 0 &&
	       push_t->inv_output_size[0] == 1.0f / float(push_t->threads[0]) && push_t->inv_output_size[1] == 1.0f / float(push_t->threads[1]);
}

int gr_bloom_down_head(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *threshold, const gr_image *d0, const gr_image *d1,
                       const gr_luminance_data *lum, const gr_push_bloom_threshold *push_t, const gr_push_bloom_downsample *push_d0,
                       const gr_push_bloom_downsample *push_d1)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, hdr && threshold && d0 && d1 && push_t && push_d0 && push_d1);
	GR_CHECK_ARG(ctx, is_hdr_target(hdr) && is_rgba16f(threshold) && is_rgba16f(d0) && is_rgba16f(d1));
	GR_CHECK_ARG(ctx, hdr->ptr != threshold->ptr && threshold->ptr != d0->ptr && d0->ptr != d1->ptr);
	// every level exactly half of its input (what gr_bloom_down_head_supported answers, whatever the size)
	GR_CHECK_ARG(ctx, hdr->width == 2u * threshold->width && hdr->height == 2u * threshold->height && threshold->width == 2u * d0->width &&
	                      threshold->height == 2u * d0->height && d0->width == 2u * d1->width && d0->height == 2u * d1->height);
	GR_CHECK_ARG(ctx, push_t->threads[0] == threshold->width && push_t->threads[1] == threshold->height && downsample_is_exact(threshold, push_d0) &&
	                      downsample_is_exact(d0, push_d1) && push_d0->threads[0] == d0->width && push_d1->threads[0] == d1->width);
	GR_CHECK_ARG(ctx, (hdr->pitch_bytes & 15u) == 0 && (reinterpret_cast<uintptr_t>(hdr->ptr) & 15u) == 0);
	if (d1->width == 0 || d1->height == 0)
		return GR_OK;
	const dim3 grid(gr_div_up(d1->width, TAIL_TILE), gr_div_up(d1->height, TAIL_TILE));
	const bool b10 = hdr->format == GR_FORMAT_B10G11R11_UFLOAT_PACK32;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "bloom_down_head"};
	if (lum)
		hipLaunchKernelGGL(k_bloom_down_head<true>, grid, dim3(256), 0, gr_to_stream(stream), to_dev(hdr), to_dev_rw(threshold), to_dev_rw(d0), to_dev_rw(d1), lum,
		                   push_t->inv_output_size[0], push_t->inv_output_size[1], b10);
	else
		hipLaunchKernelGGL(k_bloom_down_head<false>, grid, dim3(256), 0, gr_to_stream(stream), to_dev(hdr), to_dev_rw(threshold), to_dev_rw(d0), to_dev_rw(d1), lum,
		                   push_t->inv_output_size[0], push_t->inv_output_size[1], b10);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_bloom_tail_supported(const gr_image *d1, const gr_image *d2, const gr_image *d3, const gr_image *u2, const gr_image *u1,
                            const gr_push_bloom_downsample *push_d2, const gr_push_bloom_downsample *push_d3,
                            const gr_push_bloom_upsample *push_u2, const gr_push_bloom_upsample *push_u1)
{
	static const bool allow_fusion = gr_measurement_switch("GR_NO_TAIL_FUSION") == nullptr; // A/B switch for measurements
	if (!allow_fusion || !is_rgba16f(d1) || !is_rgba16f(d2) || !is_rgba16f(d3) || !is_rgba16f(u2) || !is_rgba16f(u1))
		return 0;
	if (!push_d2 || !push_d3 || !push_u2 || !push_u1)
		return 0;
	// whole levels only, upsample-2 the size of downsample-2 (the 2:1 / 1:2 stencils where a level is exactly half / twice its input, the nine taps else)
	if (push_d2->threads[0] != d2->width || push_d2->threads[1] != d2->height || push_d3->threads[0] != d3->width || push_d3->threads[1] != d3->height ||
	    push_u2->threads[0] != u2->width || push_u2->threads[1] != u2->height || push_u1->threads[0] != u1->width || push_u1->threads[1] != u1->height)
		return 0;
	if (u2->width != d2->width || u2->height != d2->height)
		return 0;
	// upsample-1 at most twice upsample-2 (+ 1: a level is ceil(half) of the one above), or its patch would not fit the kernel's LDS
	if (u1->width > 2 * u2->width || u1->height > 2 * u2->height || 2 * u2->width > u1->width + 1 || 2 * u2->height > u1->height + 1)
		return 0;
	// the patch of downsample-2 under an 8 x 8 tile of downsample-3 must fit the kernel's LDS patch
	if (d3->width == 0 || d3->height == 0 || float(d2->width) > 2.3f * float(d3->width) || float(d2->height) > 2.3f * float(d3->height))
		return 0;
	return 1;
}

int gr_bloom_down_tail(gr_ctx *ctx, gr_stream stream, const gr_image *d1, const gr_image *d2, const gr_image *d3, const gr_image *history,
                       const gr_push_bloom_downsample *push_d2, const gr_push_bloom_downsample *push_d3)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push_d2 != nullptr && push_d3 != nullptr);
	GR_CHECK_ARG(ctx, is_rgba16f(d1) && is_rgba16f(d2) && is_rgba16f(d3));
	GR_CHECK_ARG(ctx, !history || (is_rgba16f(history) && history->ptr != d3->ptr));
	GR_CHECK_ARG(ctx, push_d2->threads[0] == d2->width && push_d2->threads[1] == d2->height);
	GR_CHECK_ARG(ctx, push_d3->threads[0] == d3->width && push_d3->threads[1] == d3->height);
	GR_CHECK_ARG(ctx, float(d2->width) <= 2.3f * float(d3->width) && float(d2->height) <= 2.3f * float(d3->height));
	GR_CHECK_ARG(ctx, history != nullptr); // the last level of the pyramid always carries the temporal feedback (hdr.cpp:366)
	dim3 grid(gr_div_up(d3->width, TAIL_TILE), gr_div_up(d3->height, TAIL_TILE));
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "bloom_down_tail"};
	auto launch = [&](auto kernel) {
		hipLaunchKernelGGL(kernel, grid, dim3(256), 0, gr_to_stream(stream), to_dev(d1), to_dev_rw(d2), to_dev_rw(d3), to_dev(history), *push_d2, *push_d3, 0u,
		                   d3->height);
	};
	const bool d2_exact = downsample_is_exact(d1, push_d2), d3_exact = downsample_is_exact(d2, push_d3);
	if (d2_exact && d3_exact) launch(k_bloom_down_pair<true, true, true>);
	else if (d2_exact) launch(k_bloom_down_pair<true, false, true>);
	else if (d3_exact) launch(k_bloom_down_pair<false, true, true>);
	else launch(k_bloom_down_pair<false, false, true>);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_bloom_up_tail(gr_ctx *ctx, gr_stream stream, const gr_image *d3, const gr_image *u2, const gr_image *u1, gr_luminance_data *lum,
                     const gr_push_bloom_upsample *push_u2, const gr_push_bloom_upsample *push_u1, const gr_push_luminance *push_lum)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push_u2 != nullptr && push_u1 != nullptr && (lum == nullptr) == (push_lum == nullptr));
	GR_CHECK_ARG(ctx, is_rgba16f(d3) && is_rgba16f(u2) && is_rgba16f(u1));
	GR_CHECK_ARG(ctx, push_u1->threads[0] == u1->width && push_u1->threads[1] == u1->height);
	GR_CHECK_ARG(ctx, push_u2->threads[0] == u2->width && push_u2->threads[1] == u2->height);
	GR_CHECK_ARG(ctx, u1->width <= 2 * u2->width && u1->height <= 2 * u2->height && 2 * u2->width <= u1->width + 1 && 2 * u2->height <= u1->height + 1);
	GR_CHECK_ARG(ctx, !push_lum || (push_lum->size[0] != 0 && push_lum->size[1] != 0));
	dim3 grid(gr_div_up(u1->width, UP_TILE), gr_div_up(u1->height, UP_TILE));
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "bloom_up_tail"};
	const gr_push_luminance no_lum = {};
	const bool u2_exact = upsample_is_exact(d3, push_u2), u1_exact = upsample_is_exact(u2, push_u1);
	auto launch = [&](auto kernel) {
		hipLaunchKernelGGL(kernel, grid, dim3(LUM_THREADS), 0, gr_to_stream(stream), to_dev(d3), to_dev_rw(u2), to_dev_rw(u1), lum, *push_u2, *push_u1,
		                   push_lum ? *push_lum : no_lum);
	};
	auto pick = [&](auto u2e, auto u1e) {
		constexpr bool A = decltype(u2e)::value, B = decltype(u1e)::value;
		if (lum) launch(k_bloom_up_tail<A, B, true>);
		else launch(k_bloom_up_tail<A, B, false>);
	};
	using T = std::true_type;
	using F = std::false_type;
	if (u2_exact && u1_exact) pick(T{}, T{});
	else if (u2_exact) pick(T{}, F{});
	else if (u1_exact) pick(F{}, T{});
	else pick(F{}, F{});
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_bloom_up_all_supported(const gr_image *d3, const gr_image *u2, const gr_image *u1, const gr_image *u0, const gr_push_bloom_upsample *push_u2,
                              const gr_push_bloom_upsample *push_u1, const gr_push_bloom_upsample *push_u0)
{
	static const bool allow_fusion = gr_measurement_switch("GR_NO_UP_FUSION") == nullptr && gr_measurement_switch("GR_NO_TAIL_FUSION") == nullptr;
	if (!allow_fusion || !d3 || !u2 || !u1 || !u0 || !push_u2 || !push_u1 || !push_u0)
		return 0;
	if (!is_rgba16f(d3) || !is_rgba16f(u2) || !is_rgba16f(u1) || !is_rgba16f(u0))
		return 0;
	if (push_u2->threads[0] != u2->width || push_u2->threads[1] != u2->height || push_u1->threads[0] != u1->width || push_u1->threads[1] != u1->height ||
	    push_u0->threads[0] != u0->width || push_u0->threads[1] != u0->height)
		return 0;
	// upsample-1 at most twice upsample-2 (+ 1: a level is ceil(half) of the one above), as for gr_bloom_up_tail; upsample-0 on the 1:2 stencil
	if (u1->width > 2 * u2->width || u1->height > 2 * u2->height || 2 * u2->width > u1->width + 1 || 2 * u2->height > u1->height + 1)
		return 0;
	if (!upsample_is_exact(u1, push_u0) || u1->width == 0 || u1->height == 0)
		return 0;
	// Up to the quarter level of a 4K frame.  The levels involved are a quarter of the frame and coarser, but a 32 x 32 tile of upsample-0
	// recomputes its patches of the two levels under it (upsample-2 five times over all tiles): 3.05 M wave instructions at 4K against
	// 1.17 M for the separate launches.  Measured, same box, alternating (profiles/r05_up_fusion_by_size.txt): config 3 (4K) 0.2019 / 0.2024
	// fused against 0.2042 / 0.2029 ms, config 4 0.5922 / 0.5996 against 0.5994 / 0.5987, config 5 (8K whole on one GPU) 0.7552 / 0.7515
	// against 0.7196 / 0.7189 ms: at 8K the recomputation costs more than the launch it saves.
	if (uint64_t(u0->width) * u0->height > 960ull * 540ull)
		return 0;
	return 1;
}

int gr_bloom_up_all(gr_ctx *ctx, gr_stream stream, const gr_image *d3, const gr_image *u2, const gr_image *u1, const gr_image *u0, gr_luminance_data *lum,
                    const gr_push_bloom_upsample *push_u2, const gr_push_bloom_upsample *push_u1, const gr_push_bloom_upsample *push_u0,
                    const gr_push_luminance *push_lum, uint32_t flags)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, d3 && u2 && u1 && u0 && push_u2 && push_u1 && push_u0 && (lum == nullptr) == (push_lum == nullptr));
	GR_CHECK_ARG(ctx, is_rgba16f(d3) && is_rgba16f(u2) && is_rgba16f(u1) && is_rgba16f(u0));
	GR_CHECK_ARG(ctx, push_u0->threads[0] == u0->width && push_u0->threads[1] == u0->height && upsample_is_exact(u1, push_u0));
	GR_CHECK_ARG(ctx, push_u1->threads[0] == u1->width && push_u1->threads[1] == u1->height);
	GR_CHECK_ARG(ctx, push_u2->threads[0] == u2->width && push_u2->threads[1] == u2->height);
	GR_CHECK_ARG(ctx, u1->width <= 2 * u2->width && u1->height <= 2 * u2->height && 2 * u2->width <= u1->width + 1 && 2 * u2->height <= u1->height + 1);
	GR_CHECK_ARG(ctx, u0->ptr != u1->ptr && u1->ptr != u2->ptr && u2->ptr != d3->ptr);
	GR_CHECK_ARG(ctx, !push_lum || (push_lum->size[0] != 0 && push_lum->size[1] != 0));
	if (u0->width == 0 || u0->height == 0)
		return GR_OK;
	dim3 grid(gr_div_up(u0->width, UPALL_TILE), gr_div_up(u0->height, UPALL_TILE));
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "bloom_up_all"};
	const gr_push_luminance no_lum = {};
	const bool u2_exact = upsample_is_exact(d3, push_u2), u1_exact = upsample_is_exact(u2, push_u1);
	// measurement switch: GR_UP_ALL_THREADS=256 / 1024 whatever the caller's hint
	static const int forced = []() { const char *env = gr_measurement_switch("GR_UP_ALL_THREADS"); return env ? atoi(env) : 0; }();
	const int threads = forced == 256 || forced == 1024 ? forced : ((flags & GR_BLOOM_BUSY_FRAME_BIT) ? 256 : 1024);
	auto launch = [&](auto kernel, int nt) {
		hipLaunchKernelGGL(kernel, grid, dim3(nt), 0, gr_to_stream(stream), to_dev(d3), to_dev_rw(u2), to_dev_rw(u1), to_dev_rw(u0), lum, *push_u2, *push_u1,
		                   push_lum ? *push_lum : no_lum);
	};
	auto pick = [&](auto u2e, auto u1e) {
		constexpr bool A = decltype(u2e)::value, B = decltype(u1e)::value;
		if (threads == 1024)
		{
			if (lum) launch(k_bloom_up_all<A, B, true, LUM_THREADS>, LUM_THREADS);
			else launch(k_bloom_up_all<A, B, false, LUM_THREADS>, LUM_THREADS);
		}
		else
		{
			if (lum) launch(k_bloom_up_all<A, B, true, 256>, 256);
			else launch(k_bloom_up_all<A, B, false, 256>, 256);
		}
	};
	using T = std::true_type;
	using F = std::false_type;
	if (u2_exact && u1_exact) pick(T{}, T{});
	else if (u2_exact) pick(T{}, F{});
	else if (u1_exact) pick(F{}, T{});
	else pick(F{}, F{});
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_bloom_pyramid_supported(const gr_bloom_pyramid_args *a)
{
	static const bool allow_fusion = gr_measurement_switch("GR_NO_PYRAMID_FUSION") == nullptr; // A/B switch for measurements
	if (!allow_fusion || !a || !a->history.ptr)
		return 0;
	if (!gr_bloom_down_head_supported(&a->hdr, &a->threshold, &a->d0, &a->d1, &a->push_threshold, &a->push_d0, &a->push_d1) ||
	    !gr_bloom_tail_supported(&a->d1, &a->d2, &a->d3, &a->u2, &a->u1, &a->push_d2, &a->push_d3, &a->push_u2, &a->push_u1) ||
	    !gr_bloom_up_all_supported(&a->d3, &a->u2, &a->u1, &a->u0, &a->push_u2, &a->push_u1, &a->push_u0))
		return 0;
	if (!is_rgba16f(&a->history) || a->history.width != a->d3.width || a->history.height != a->d3.height || a->history.ptr == a->d3.ptr)
		return 0;
	if (a->lum && (a->push_luminance.size[0] == 0 || a->push_luminance.size[1] == 0))
		return 0;
	// Up to a 640 x 384 frame.  A hand-over between two phases costs what agent-scope release / acquire cost on eight XCDs with an L2 each (an L2
	// write-back per publishing workgroup, an invalidate per waiting one): measured 22 us for the 21 workgroups of a 256 x 256 frame against 8.7 +
	// 6.5 + 5.5 us for the three launches inside their hipEvent brackets (the frame: 0.0376 against 0.047-0.052 ms, the host being what the three
	// launches wait for), but 67 us for the 685 workgroups of a 1080p frame against 14 + 10 + 8, with the other streams' kernels slowed by the
	// cache traffic (profiles/r06_pyramid_one_launch.txt).  (GR_PYRAMID_ANY_SIZE: wherever the three launches are offered.)
	static const bool any_size = gr_measurement_switch("GR_PYRAMID_ANY_SIZE") != nullptr;
	return any_size || uint64_t(a->hdr.width) * a->hdr.height <= 640ull * 384ull;
}

int gr_bloom_pyramid(gr_ctx *ctx, gr_stream stream, const gr_bloom_pyramid_args *a)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, a != nullptr);
	// what the three launches this one stands for check (gr_bloom_down_head, gr_bloom_down_tail, gr_bloom_up_all)
	GR_CHECK_ARG(ctx, is_hdr_target(&a->hdr) && is_rgba16f(&a->threshold) && is_rgba16f(&a->d0) && is_rgba16f(&a->d1) && is_rgba16f(&a->d2) && is_rgba16f(&a->d3) &&
	                      is_rgba16f(&a->u2) && is_rgba16f(&a->u1) && is_rgba16f(&a->u0) && is_rgba16f(&a->history));
	GR_CHECK_ARG(ctx, a->hdr.width == 2u * a->threshold.width && a->hdr.height == 2u * a->threshold.height && a->threshold.width == 2u * a->d0.width &&
	                      a->threshold.height == 2u * a->d0.height && a->d0.width == 2u * a->d1.width && a->d0.height == 2u * a->d1.height);
	GR_CHECK_ARG(ctx, a->push_threshold.threads[0] == a->threshold.width && a->push_threshold.threads[1] == a->threshold.height &&
	                      downsample_is_exact(&a->threshold, &a->push_d0) && downsample_is_exact(&a->d0, &a->push_d1) && a->push_d0.threads[0] == a->d0.width &&
	                      a->push_d1.threads[0] == a->d1.width);
	GR_CHECK_ARG(ctx, (a->hdr.pitch_bytes & 15u) == 0 && (reinterpret_cast<uintptr_t>(a->hdr.ptr) & 15u) == 0);
	GR_CHECK_ARG(ctx, a->push_d2.threads[0] == a->d2.width && a->push_d2.threads[1] == a->d2.height && a->push_d3.threads[0] == a->d3.width &&
	                      a->push_d3.threads[1] == a->d3.height);
	GR_CHECK_ARG(ctx, float(a->d2.width) <= 2.3f * float(a->d3.width) && float(a->d2.height) <= 2.3f * float(a->d3.height));
	GR_CHECK_ARG(ctx, a->history.ptr != a->d3.ptr && a->history.width == a->d3.width && a->history.height == a->d3.height);
	GR_CHECK_ARG(ctx, a->push_u0.threads[0] == a->u0.width && a->push_u0.threads[1] == a->u0.height && upsample_is_exact(&a->u1, &a->push_u0));
	GR_CHECK_ARG(ctx, a->push_u1.threads[0] == a->u1.width && a->push_u1.threads[1] == a->u1.height && a->push_u2.threads[0] == a->u2.width &&
	                      a->push_u2.threads[1] == a->u2.height && a->u2.width == a->d2.width && a->u2.height == a->d2.height);
	GR_CHECK_ARG(ctx, a->u1.width <= 2 * a->u2.width && a->u1.height <= 2 * a->u2.height && 2 * a->u2.width <= a->u1.width + 1 &&
	                      2 * a->u2.height <= a->u1.height + 1);
	{
		const void *levels[] = {a->hdr.ptr, a->threshold.ptr, a->d0.ptr, a->d1.ptr, a->d2.ptr, a->d3.ptr, a->u2.ptr, a->u1.ptr, a->u0.ptr, a->history.ptr};
		for (size_t i = 0; i < sizeof(levels) / sizeof(levels[0]); i++)
			for (size_t j = i + 1; j < sizeof(levels) / sizeof(levels[0]); j++)
				GR_CHECK_ARG(ctx, levels[i] != levels[j]);
	}
	GR_CHECK_ARG(ctx, !a->lum || (a->push_luminance.size[0] != 0 && a->push_luminance.size[1] != 0));
	if (a->d3.width == 0 || a->d3.height == 0)
		return GR_OK;
	PyramidArgs k{};
	k.hdr = to_dev(&a->hdr), k.history = to_dev(&a->history);
	k.thr = to_dev_rw(&a->threshold), k.d0 = to_dev_rw(&a->d0), k.d1 = to_dev_rw(&a->d1), k.d2 = to_dev_rw(&a->d2), k.d3 = to_dev_rw(&a->d3);
	k.u2 = to_dev_rw(&a->u2), k.u1 = to_dev_rw(&a->u1), k.u0 = to_dev_rw(&a->u0);
	k.lum = a->lum;
	k.inv_thr_w = a->push_threshold.inv_output_size[0], k.inv_thr_h = a->push_threshold.inv_output_size[1];
	k.push_d2 = a->push_d2, k.push_d3 = a->push_d3, k.push_u2 = a->push_u2, k.push_u1 = a->push_u1;
	if (a->lum)
		k.push_lum = a->push_luminance;
	k.head_x = gr_div_up(a->d1.width, TAIL_TILE), k.pair_x = gr_div_up(a->d3.width, TAIL_TILE), k.up_x = gr_div_up(a->u0.width, UPALL_TILE);
	k.n0 = k.head_x * gr_div_up(a->d1.height, TAIL_TILE);
	k.n1 = k.pair_x * gr_div_up(a->d3.height, TAIL_TILE);
	k.n2 = k.up_x * gr_div_up(a->u0.height, UPALL_TILE);
	k.sync = ctx->pyramid_sync + 4u * (ctx->pyramid_launches.fetch_add(1u) % gr_ctx::PYRAMID_SYNC_SLOTS);
	k.hdr_b10 = a->hdr.format == GR_FORMAT_B10G11R11_UFLOAT_PACK32;
	const dim3 grid(k.n0 + k.n1 + k.n2);
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "bloom_pyramid"};
	const bool d2e = downsample_is_exact(&a->d1, &a->push_d2), d3e = downsample_is_exact(&a->d2, &a->push_d3);
	const bool u2e = upsample_is_exact(&a->d3, &a->push_u2), u1e = upsample_is_exact(&a->u2, &a->push_u1);
	const unsigned variant = (a->lum ? 16u : 0u) | (d2e ? 8u : 0u) | (d3e ? 4u : 0u) | (u2e ? 2u : 0u) | (u1e ? 1u : 0u);
	switch (variant)
	{
#define GR_PYRAMID_CASE(V)                                                                                                                     \
	case V:                                                                                                                                    \
		hipLaunchKernelGGL((k_bloom_pyramid<((V) & 16) != 0, ((V) & 8) != 0, ((V) & 4) != 0, ((V) & 2) != 0, ((V) & 1) != 0>), grid, dim3(PYRAMID_THREADS), 0, \
		                   gr_to_stream(stream), k);                                                                                           \
		break;
		GR_PYRAMID_CASE(0) GR_PYRAMID_CASE(1) GR_PYRAMID_CASE(2) GR_PYRAMID_CASE(3) GR_PYRAMID_CASE(4) GR_PYRAMID_CASE(5) GR_PYRAMID_CASE(6) GR_PYRAMID_CASE(7)
		GR_PYRAMID_CASE(8) GR_PYRAMID_CASE(9) GR_PYRAMID_CASE(10) GR_PYRAMID_CASE(11) GR_PYRAMID_CASE(12) GR_PYRAMID_CASE(13) GR_PYRAMID_CASE(14) GR_PYRAMID_CASE(15)
		GR_PYRAMID_CASE(16) GR_PYRAMID_CASE(17) GR_PYRAMID_CASE(18) GR_PYRAMID_CASE(19) GR_PYRAMID_CASE(20) GR_PYRAMID_CASE(21) GR_PYRAMID_CASE(22) GR_PYRAMID_CASE(23)
		GR_PYRAMID_CASE(24) GR_PYRAMID_CASE(25) GR_PYRAMID_CASE(26) GR_PYRAMID_CASE(27) GR_PYRAMID_CASE(28) GR_PYRAMID_CASE(29) GR_PYRAMID_CASE(30) GR_PYRAMID_CASE(31)
#undef GR_PYRAMID_CASE
	}
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_debug_pyramid_giveups(gr_ctx *ctx, uint32_t *count)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, count != nullptr);
	uint32_t slots[gr_ctx::PYRAMID_SYNC_SLOTS * 4];
	GR_CHECK_HIP(ctx, hipDeviceSynchronize());
	GR_CHECK_HIP(ctx, hipMemcpy(slots, ctx->pyramid_sync, sizeof(slots), hipMemcpyDeviceToHost));
	*count = 0;
	for (unsigned i = 0; i < gr_ctx::PYRAMID_SYNC_SLOTS; i++)
		*count += slots[4 * i + 3];
	return GR_OK;
}

int gr_luminance(gr_ctx *ctx, gr_stream stream, const gr_image *in, gr_luminance_data *lum, const gr_push_luminance *push)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push != nullptr && lum != nullptr);
	GR_CHECK_ARG(ctx, is_rgba16f(in));
	GR_CHECK_ARG(ctx, push->size[0] != 0 && push->size[1] != 0);
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "luminance"};
	hipLaunchKernelGGL(k_luminance, dim3(1), dim3(LUM_THREADS), 0, gr_to_stream(stream), to_dev(in), lum, *push);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_tonemap(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *bloom, const gr_image *out,
               const gr_luminance_data *lum, const gr_push_tonemap *push)
{
	return gr_tonemap_rows(ctx, stream, hdr, bloom, out, lum, push, nullptr);
}

int gr_tonemap_rows(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *bloom, const gr_image *out,
                    const gr_luminance_data *lum, const gr_push_tonemap *push, const gr_rows *rows)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push != nullptr);
	GR_CHECK_ARG(ctx, is_hdr_target(hdr) && is_rgba16f(bloom));
	GR_CHECK_ARG(ctx, out && out->ptr && out->width == hdr->width && out->height == hdr->height &&
	                       out->pitch_bytes >= out->width * 4u);
	const bool srgb = out->format == GR_FORMAT_R8G8B8A8_SRGB;
	if (!srgb && out->format != GR_FORMAT_R8G8B8A8_UNORM)
		return ctx->fail(GR_ERR_UNSUPPORTED_FORMAT, "gr_tonemap: output format %u unsupported", out->format);
	const RowSpan span = resolve_rows(rows, hdr->height);
	if (span.count() == 0)
		return GR_OK;
	dim3 block(TONEMAP_BLOCK_X, TONEMAP_BLOCK_Y);
	// A workgroup walks TONEMAP_ROW_GROUPS x 4 rows per staging of the table -- except for a frame so small that this leaves fewer than 64
	// workgroups (256 x 256: eight, each walking its rows one after the other on one CU): there a workgroup takes four rows.
	const unsigned columns = gr_div_up(hdr->width, TONEMAP_BLOCK_X * TONEMAP_PX);
	const int row_groups = columns * gr_div_up(span.count(), TONEMAP_BLOCK_Y * TONEMAP_ROW_GROUPS) < 64u ? 1 : TONEMAP_ROW_GROUPS;
	dim3 grid(columns, gr_div_up(span.count(), unsigned(TONEMAP_BLOCK_Y * row_groups)));
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "tonemap"};
	hipStream_t s = gr_to_stream(stream);
	const bool quarter = hdr->width == 4u * bloom->width && hdr->height == 4u * bloom->height;
	auto launch = [&](auto kernel) {
		hipLaunchKernelGGL(kernel, grid, block, 0, s, to_dev(hdr), to_dev(bloom), to_dev_rw(out), lum, ctx->tonemap_srgb8_lut, *push, span.first, span.end,
		                   hdr->format == GR_FORMAT_B10G11R11_UFLOAT_PACK32, row_groups);
	};
	if (quarter)
	{
		if (lum && srgb) launch(k_tonemap<true, true, true>);
		else if (lum) launch(k_tonemap<true, false, true>);
		else if (srgb) launch(k_tonemap<false, true, true>);
		else launch(k_tonemap<false, false, true>);
	}
	else
	{
		if (lum && srgb) launch(k_tonemap<true, true, false>);
		else if (lum) launch(k_tonemap<true, false, false>);
		else if (srgb) launch(k_tonemap<false, true, false>);
		else launch(k_tonemap<false, false, false>);
	}
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}
}
