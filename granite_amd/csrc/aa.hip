// Anti-aliasing kernels for gfx950 and their C-ABI launchers (include/granite_hip.h): FXAA, SMAA 1x (edge detection,
// blend weights, neighbourhood blend) and the TAA resolve.
//
// Replaces assets/shaders/post/{fxaa.frag, smaa_*.{vert,frag} + SMAA.hlsl, taa_resolve.frag + reprojection*.h} as
// recorded by renderer/post/{fxaa,smaa,temporal}.cpp.
//
// These passes take data-dependent decisions on 8-bit data (edge thresholds, search loops, LUT addresses), so the file
// is compiled with -ffp-contract=off and every expression keeps a fixed association order; `mad` is a real fma in the
// reference (SMAA_GLSL_4) and is written fmaf here.  Images in *_SRGB formats are read through their UNORM alias: the
// stored bytes.
#include <vector>
#include "ctx.hpp"
#include "device_common.hpp"
#include "device_vec.hpp"
#include "aa_core.hpp"
#include "aa_fast_kernels.hpp"
#include "smaa_weights.hpp"

namespace
{
constexpr int AA_BLOCK_X = 32;
constexpr int AA_BLOCK_Y = 8;

// StockSampler::LinearClamp on a UNORM8 image with CH channels; texel offsets are applied after the floor.  The sampler
// model is the oracle's (aa_core.hpp: exact fp32 weights, coordinates within 2^-8 of a texel centre read that texel): for UNORM
// texels t * 1 + t' * 0 == t, so the snapped case goes through the same lerps.
template <int CH>
struct Tex8
{
	static constexpr bool HAS_RUNS = false; // smaa_weights.hpp: no bit planes behind this accessor, the searches sample
	const uint8_t *ptr;
	int w, h;
	uint32_t pitch;

	__device__ __forceinline__ v4 fetch(int x, int y) const
	{
		x = clampi(x, 0, w - 1);
		y = clampi(y, 0, h - 1);
		const uint8_t *p = ptr + (uint32_t(y) * pitch + uint32_t(x) * uint32_t(CH)); // images stay below 4 GiB
		v4 r = mk4(0.0f, 0.0f, 0.0f, 1.0f);
		if (CH == 4)
		{
			const uint32_t t = *reinterpret_cast<const uint32_t *>(p);
			r = mk4(unorm8_to_float(t & 255u), unorm8_to_float((t >> 8) & 255u), unorm8_to_float((t >> 16) & 255u), unorm8_to_float(t >> 24));
		}
		else if (CH == 2)
		{
			const uint32_t t = *reinterpret_cast<const uint16_t *>(p);
			r.x = unorm8_to_float(t & 255u);
			r.y = unorm8_to_float(t >> 8);
		}
		else
			r.x = unorm8_to_float(p[0]);
		return r;
	}

	template <bool COLUMNS = false> // smaa_weights.hpp: which staged words a bit-plane accessor reads; one image here
	__device__ __forceinline__ v4 sample(v2 uv, int ox = 0, int oy = 0) const
	{
		int x0, y0;
		float a, b;
		aa::linear_axis(uv.x * float(w) - 0.5f, x0, a);
		aa::linear_axis(uv.y * float(h) - 0.5f, y0, b);
		x0 += ox;
		y0 += oy;
		const v4 t00 = fetch(x0, y0), t10 = fetch(x0 + 1, y0), t01 = fetch(x0, y0 + 1), t11 = fetch(x0 + 1, y0 + 1);
		const v4 top = t00 * (1.0f - a) + t10 * a;
		const v4 bot = t01 * (1.0f - a) + t11 * a;
		return top * (1.0f - b) + bot * b;
	}
};

// The neighbourhood of a 32 x 8 block of an RGBA8 image, decoded to fp32 ONCE into LDS (clamp-to-edge applied while staging):
// the passes below take 5 to 9 bilinear taps per pixel within a few texels of it, i.e. 20 to 36 texel decodes per pixel from
// global memory otherwise.  sample() is Tex8::sample statement for statement -- same coordinates, same weights, same order of
// the lerps -- with the four texels read from the tile; a tap outside the tile (the passes' reach is below HALO, so this is a
// safety net, not a path) falls back to the image.
template <int HALO>
struct Tile8
{
	static constexpr int W = AA_BLOCK_X + 2 * HALO, H = AA_BLOCK_Y + 2 * HALO;
	const float4 *texels; // [H][W]
	int ox, oy;           // image coordinates of texels[0]
	Tex8<4> tex;

	template <bool COLUMNS = false>
	__device__ __forceinline__ v4 sample(v2 uv, int offx = 0, int offy = 0) const
	{
		int x0, y0;
		float a, b;
		aa::linear_axis(uv.x * float(tex.w) - 0.5f, x0, a);
		aa::linear_axis(uv.y * float(tex.h) - 0.5f, y0, b);
		x0 += offx;
		y0 += offy;
		const int tx = x0 - ox, ty = y0 - oy;
		v4 t00, t10, t01, t11;
		// one decision per wave: the tile, or (never expected) the image for all four texels of every lane
		if (__all(unsigned(tx) < unsigned(W - 1) && unsigned(ty) < unsigned(H - 1)))
		{
			const float4 *p = texels + ty * W + tx;
			const float4 q00 = p[0], q10 = p[1], q01 = p[W], q11 = p[W + 1];
			t00 = mk4(q00.x, q00.y, q00.z, q00.w);
			t10 = mk4(q10.x, q10.y, q10.z, q10.w);
			t01 = mk4(q01.x, q01.y, q01.z, q01.w);
			t11 = mk4(q11.x, q11.y, q11.z, q11.w);
		}
		else
		{
			t00 = tex.fetch(x0, y0);
			t10 = tex.fetch(x0 + 1, y0);
			t01 = tex.fetch(x0, y0 + 1);
			t11 = tex.fetch(x0 + 1, y0 + 1);
		}
		const v4 top = t00 * (1.0f - a) + t10 * a;
		const v4 bot = t01 * (1.0f - a) + t11 * a;
		return top * (1.0f - b) + bot * b;
	}
};

// Fills `lds` (Tile8<HALO>::W * H texels) for the block whose first pixel is (bx, by); every thread of the block must call it.
template <int HALO>
__device__ __forceinline__ Tile8<HALO> stage_tile(float4 *lds, const Tex8<4> &tex, int bx, int by)
{
	constexpr int W = Tile8<HALO>::W, H = Tile8<HALO>::H;
	for (int i = threadIdx.y * AA_BLOCK_X + threadIdx.x; i < W * H; i += AA_BLOCK_X * AA_BLOCK_Y)
	{
		const int ty = i / W, tx = i - ty * W;
		const v4 t = tex.fetch(bx - HALO + tx, by - HALO + ty);
		lds[i] = make_float4(t.x, t.y, t.z, t.w);
	}
	__syncthreads();
	return {lds, bx - HALO, by - HALO, tex};
}

template <int CH>
static Tex8<CH> make_tex8(const gr_image *img)
{
	return {static_cast<const uint8_t *>(img->ptr), int(img->width), int(img->height), img->pitch_bytes};
}

__device__ __forceinline__ uint32_t unorm8(float v)
{
	// UNORM store: NaN and negatives -> 0, >= 1 -> 255, otherwise floor(v * 255 + 0.5).
	if (!(v > 0.0f))
		return 0u;
	if (v >= 1.0f)
		return 255u;
	return uint32_t(int(v * 255.0f + 0.5f));
}

// decode_srgb in the shader followed by the sRGB attachment store is the identity on the gamma-space value up to the
// transcendental round trip; the kernels store the gamma-space value directly (differs from the literal path by at
// most 1 LSB at exact .5 boundaries — inside the stated RGBA8 tolerance).
__device__ __forceinline__ void store_rgba8(uint8_t *ptr, uint32_t pitch, int x, int y, v4 c)
{
	*reinterpret_cast<uint32_t *>(ptr + size_t(y) * pitch + size_t(x) * 4u) = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
}

// ---- FXAA (fxaa.frag:20-67) --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AA_BLOCK_X *AA_BLOCK_Y) void k_fxaa_generic(Tex8<4> tex_, uint8_t *out, uint32_t out_pitch, gr_push_fxaa push, RowSpan rows)
{
	// reach: the corner taps 1 texel, the edge taps at most FXAA_SPAN_MAX * 0.5 = 4 texels, + 1 for the bilinear footprint
	constexpr int HALO = 6;
	__shared__ float4 s_tile[Tile8<HALO>::W * Tile8<HALO>::H];
	const Tex8<4> image = tex_;
	const int bx = blockIdx.x * AA_BLOCK_X, by = int(rows.first) + blockIdx.y * AA_BLOCK_Y;
	const Tile8<HALO> tex = stage_tile<HALO>(s_tile, image, bx, by);
	const int x = bx + threadIdx.x, y = by + threadIdx.y;
	if (x >= image.w || y >= int(rows.end))
		return;
	const float FXAA_REDUCE_MIN = 1.0f / 128.0f, FXAA_REDUCE_MUL = 1.0f / 8.0f, FXAA_SPAN_MAX = 8.0f;
	const v2 inv_resolution = mk2(push.inv_resolution[0], push.inv_resolution[1]);
	const v2 uv = mk2((float(x) + 0.5f) * inv_resolution.x, (float(y) + 0.5f) * inv_resolution.y);
	const v3 rgbNW = xyz(tex.sample(uv, -1, -1)), rgbNE = xyz(tex.sample(uv, +1, -1));
	const v3 rgbSW = xyz(tex.sample(uv, -1, +1)), rgbSE = xyz(tex.sample(uv, +1, +1));
	const v3 texColor = xyz(tex.sample(uv));
	const v3 luma = mk3(0.299f, 0.587f, 0.114f);
	const float lumaNW = dot3(rgbNW, luma), lumaNE = dot3(rgbNE, luma), lumaSW = dot3(rgbSW, luma), lumaSE = dot3(rgbSE, luma);
	const float lumaM = dot3(texColor, luma);
	const float lumaMin = fminf(lumaM, fminf(fminf(lumaNW, lumaNE), fminf(lumaSW, lumaSE)));
	const float lumaMax = fmaxf(lumaM, fmaxf(fmaxf(lumaNW, lumaNE), fmaxf(lumaSW, lumaSE)));
	v2 dir;
	dir.x = -((lumaNW + lumaNE) - (lumaSW + lumaSE));
	dir.y = ((lumaNW + lumaSW) - (lumaNE + lumaSE));
	const float dirReduce = fmaxf((lumaNW + lumaNE + lumaSW + lumaSE) * (0.25f * FXAA_REDUCE_MUL), FXAA_REDUCE_MIN);
	const float rcpDirMin = 1.0f / (fminf(fabsf(dir.x), fabsf(dir.y)) + dirReduce);
	dir = mk2(clampfv(dir.x * rcpDirMin, -FXAA_SPAN_MAX, FXAA_SPAN_MAX), clampfv(dir.y * rcpDirMin, -FXAA_SPAN_MAX, FXAA_SPAN_MAX)) * inv_resolution;
	const v3 rgbA = 0.5f * (xyz(tex.sample(uv + dir * (1.0f / 3.0f - 0.5f))) + xyz(tex.sample(uv + dir * (2.0f / 3.0f - 0.5f))));
	const v3 rgbB = rgbA * 0.5f + 0.25f * (xyz(tex.sample(uv + dir * -0.5f)) + xyz(tex.sample(uv + dir * 0.5f)));
	const float lumaB = dot3(rgbB, luma);
	const v3 color = ((lumaB < lumaMin) || (lumaB > lumaMax)) ? rgbA : rgbB;
	store_rgba8(out, out_pitch, x, y, mk4(color.x, color.y, color.z, 1.0f));
}

// ---- SMAA ---------------------------------------------------------------------------------------------------------------
static SmaaPreset smaa_preset(int quality)
{
	switch (quality) // SMAA.hlsl:304-324
	{
	case 0: return {0.15f, 4, 8, 0.25f, 0, 0};
	case 1: return {0.1f, 8, 8, 0.25f, 0, 0};
	case 2: return {0.1f, 16, 8, 0.25f, 1, 1};
	default: return {0.05f, 32, 16, 0.25f, 1, 1};
	}
}

// SMAALumaEdgeDetectionPS (SMAA.hlsl:689-740).  Every pixel is written (0 = what the reference leaves as the clear value
// when the fragment is discarded), so no separate clear pass is needed.
__global__ __launch_bounds__(AA_BLOCK_X *AA_BLOCK_Y) void k_smaa_edges_generic(Tex8<4> image, uint8_t *edges, uint32_t edges_pitch, gr_push_smaa push,
                                                                       SmaaPreset P, RowSpan rows)
{
	constexpr int HALO = 3; // taps two texels to the left / above, one to the right / below, + 1 for the bilinear footprint
	__shared__ float4 s_tile[Tile8<HALO>::W * Tile8<HALO>::H];
	const int bx = blockIdx.x * AA_BLOCK_X, by = int(rows.first) + blockIdx.y * AA_BLOCK_Y;
	const Tile8<HALO> tex = stage_tile<HALO>(s_tile, image, bx, by);
	const int x = bx + threadIdx.x, y = by + threadIdx.y;
	if (x >= image.w || y >= int(rows.end))
		return;
	const v2 rt = mk2(push.rt_metrics[0], push.rt_metrics[1]);
	const v2 tc = mk2((float(x) + 0.5f) * rt.x, (float(y) + 0.5f) * rt.y);
	const v3 weights = mk3(0.2126f, 0.7152f, 0.0722f);
	auto luma = [&](v2 uv) { return dot3(xyz(tex.sample(uv)), weights); };
	uint32_t result = 0u;
	const float L = luma(tc);
	const float Lleft = luma(fma2(rt, mk2(-1.0f, 0.0f), tc)), Ltop = luma(fma2(rt, mk2(0.0f, -1.0f), tc));
	const v2 delta_xy = mk2(fabsf(L - Lleft), fabsf(L - Ltop));
	v2 e = mk2(stepf(P.threshold, delta_xy.x), stepf(P.threshold, delta_xy.y));
	if (e.x + e.y != 0.0f)
	{
		const float Lright = luma(fma2(rt, mk2(1.0f, 0.0f), tc)), Lbottom = luma(fma2(rt, mk2(0.0f, 1.0f), tc));
		v2 delta_zw = mk2(fabsf(L - Lright), fabsf(L - Lbottom));
		v2 maxDelta = mk2(fmaxf(delta_xy.x, delta_zw.x), fmaxf(delta_xy.y, delta_zw.y));
		const float Lleftleft = luma(fma2(rt, mk2(-2.0f, 0.0f), tc)), Ltoptop = luma(fma2(rt, mk2(0.0f, -2.0f), tc));
		delta_zw = mk2(fabsf(Lleft - Lleftleft), fabsf(Ltop - Ltoptop));
		maxDelta = mk2(fmaxf(maxDelta.x, delta_zw.x), fmaxf(maxDelta.y, delta_zw.y));
		const float finalDelta = fmaxf(maxDelta.x, maxDelta.y);
		e.x *= stepf(finalDelta, 2.0f * delta_xy.x);
		e.y *= stepf(finalDelta, 2.0f * delta_xy.y);
		result = unorm8(e.x) | (unorm8(e.y) << 8);
	}
	*reinterpret_cast<uint16_t *>(edges + size_t(y) * edges_pitch + size_t(x) * 2u) = uint16_t(result);
}

// The edge texture around a block, decoded once into LDS (as Tile8, two channels): the weight pass samples it bilinearly
// 20 to 100 times per edge pixel, mostly within a few texels of the pixel.  HALO = 16 holds the whole diagonal search and the
// first eight steps of the orthogonal ones; a longer search leaves the tile and continues on the image, decided per wave
// and per tap.
template <int HALO>
struct EdgeTile
{
	static constexpr bool HAS_RUNS = false;
	static constexpr int W = AA_BLOCK_X + 2 * HALO, H = AA_BLOCK_Y + 2 * HALO;
	const float2 *texels; // [H][W]
	int ox, oy;
	Tex8<2> tex;

	template <bool COLUMNS = false>
	__device__ __forceinline__ v4 sample(v2 uv, int offx = 0, int offy = 0) const
	{
		int x0, y0;
		float a, b;
		aa::linear_axis(uv.x * float(tex.w) - 0.5f, x0, a);
		aa::linear_axis(uv.y * float(tex.h) - 0.5f, y0, b);
		x0 += offx;
		y0 += offy;
		const int tx = x0 - ox, ty = y0 - oy;
		v4 t00, t10, t01, t11;
		if (__all(unsigned(tx) < unsigned(W - 1) && unsigned(ty) < unsigned(H - 1)))
		{
			const float2 *p = texels + ty * W + tx;
			const float2 q00 = p[0], q10 = p[1], q01 = p[W], q11 = p[W + 1];
			t00 = mk4(q00.x, q00.y, 0.0f, 1.0f);
			t10 = mk4(q10.x, q10.y, 0.0f, 1.0f);
			t01 = mk4(q01.x, q01.y, 0.0f, 1.0f);
			t11 = mk4(q11.x, q11.y, 0.0f, 1.0f);
		}
		else
		{
			t00 = tex.fetch(x0, y0);
			t10 = tex.fetch(x0 + 1, y0);
			t01 = tex.fetch(x0, y0 + 1);
			t11 = tex.fetch(x0 + 1, y0 + 1);
		}
		const v4 top = t00 * (1.0f - a) + t10 * a;
		const v4 bot = t01 * (1.0f - a) + t11 * a;
		return top * (1.0f - b) + bot * b;
	}
};
constexpr int SMAA_EDGE_HALO = 16;

struct SmaaWeightsArgs
{
	Tex8<2> edges;
	TexF<2> area;
	TexF<1> search;
	v4 rt;
	SmaaPreset P;
};

// SMAABlendingWeightCalculationPS.  The reference runs this quad under a depth mask EQUAL to the edge pass's
// non-discarded pixels (smaa.cpp:101-112,170-177); the mask is the edge texel itself here: zero edge => zero weights.
__global__ __launch_bounds__(AA_BLOCK_X *AA_BLOCK_Y) void k_smaa_weights(SmaaWeightsArgs A, uint8_t *out, uint32_t out_pitch, RowSpan rows)
{
	using Tile = EdgeTile<SMAA_EDGE_HALO>;
	__shared__ float2 s_edges[Tile::W * Tile::H];
	const int bx = blockIdx.x * AA_BLOCK_X, by = int(rows.first) + blockIdx.y * AA_BLOCK_Y;
	const int x = bx + threadIdx.x, y = by + threadIdx.y;
	const bool inside = x < A.edges.w && y < int(rows.end);
	uint16_t e = 0;
	if (inside)
		e = *reinterpret_cast<const uint16_t *>(A.edges.ptr + size_t(y) * A.edges.pitch + size_t(x) * 2u);
	// Staging the tile costs about as much as twenty taps per pixel of the block: it pays where edges are dense (a quarter of
	// the block or more), not on the few-per-cent edge density of a rendered frame.
	const int edge_pixels = __syncthreads_count(e != 0);
	uint32_t packed = 0u;
	if (edge_pixels >= AA_BLOCK_X * AA_BLOCK_Y / 4)
	{
		for (int i = threadIdx.y * AA_BLOCK_X + threadIdx.x; i < Tile::W * Tile::H; i += AA_BLOCK_X * AA_BLOCK_Y)
		{
			const int ty = i / Tile::W, tx = i - ty * Tile::W;
			const v4 t = A.edges.fetch(bx - SMAA_EDGE_HALO + tx, by - SMAA_EDGE_HALO + ty);
			s_edges[i] = make_float2(t.x, t.y);
		}
		__syncthreads();
		if (e != 0)
		{
			SmaaWeights<Tile> S = {{s_edges, bx - SMAA_EDGE_HALO, by - SMAA_EDGE_HALO, A.edges}, A.area, A.search, A.rt, A.P};
			const v4 w = S.weights_at(x, y);
			packed = unorm8(w.x) | (unorm8(w.y) << 8) | (unorm8(w.z) << 16) | (unorm8(w.w) << 24);
		}
	}
	else if (e != 0)
	{
		SmaaWeights<Tex8<2>> S = {A.edges, A.area, A.search, A.rt, A.P};
		const v4 w = S.weights_at(x, y);
		packed = unorm8(w.x) | (unorm8(w.y) << 8) | (unorm8(w.z) << 16) | (unorm8(w.w) << 24);
	}
	if (inside)
		*reinterpret_cast<uint32_t *>(out + size_t(y) * out_pitch + size_t(x) * 4u) = packed;
}

// SMAANeighborhoodBlendingPS (SMAA.hlsl:1252-1308)
__global__ __launch_bounds__(AA_BLOCK_X *AA_BLOCK_Y) void k_smaa_blend_generic(Tex8<4> cimage, Tex8<4> bimage, uint8_t *out, uint32_t out_pitch, gr_push_smaa push,
                                                                       RowSpan rows)
{
	constexpr int HALO = 2; // weights of the right / bottom neighbour, colour up to one texel away, + 1 for the bilinear footprint
	__shared__ float4 s_color[Tile8<HALO>::W * Tile8<HALO>::H], s_weights[Tile8<HALO>::W * Tile8<HALO>::H];
	const int bx = blockIdx.x * AA_BLOCK_X, by = int(rows.first) + blockIdx.y * AA_BLOCK_Y;
	const Tile8<HALO> ctex = stage_tile<HALO>(s_color, cimage, bx, by);
	const Tile8<HALO> btex = stage_tile<HALO>(s_weights, bimage, bx, by);
	const int x = bx + threadIdx.x, y = by + threadIdx.y;
	if (x >= cimage.w || y >= int(rows.end))
		return;
	const v2 rt = mk2(push.rt_metrics[0], push.rt_metrics[1]);
	const v2 texcoord = mk2((float(x) + 0.5f) * rt.x, (float(y) + 0.5f) * rt.y);
	const v4 offset = mk4(fmaf(rt.x, 1.0f, texcoord.x), fmaf(rt.y, 0.0f, texcoord.y), fmaf(rt.x, 0.0f, texcoord.x), fmaf(rt.y, 1.0f, texcoord.y));
	v4 a;
	a.x = btex.sample(mk2(offset.x, offset.y)).w;
	a.y = btex.sample(mk2(offset.z, offset.w)).y;
	const v4 c = btex.sample(texcoord);
	a.w = c.x;
	a.z = c.z;
	v4 result;
	if (a.x + a.y + a.z + a.w < 1e-5f)
		result = ctex.sample(texcoord);
	else
	{
		const bool hz = fmaxf(a.x, a.z) > fmaxf(a.y, a.w);
		v4 blendingOffset = mk4(0.0f, a.y, 0.0f, a.w);
		v2 blendingWeight = mk2(a.y, a.w);
		if (hz)
		{
			blendingOffset = mk4(a.x, 0.0f, a.z, 0.0f);
			blendingWeight = mk2(a.x, a.z);
		}
		const float sum = blendingWeight.x + blendingWeight.y;
		blendingWeight = mk2(blendingWeight.x / sum, blendingWeight.y / sum);
		const v4 bc = mk4(fmaf(blendingOffset.x, rt.x, texcoord.x), fmaf(blendingOffset.y, rt.y, texcoord.y), fmaf(blendingOffset.z, -rt.x, texcoord.x),
		                  fmaf(blendingOffset.w, -rt.y, texcoord.y));
		result = blendingWeight.x * ctex.sample(mk2(bc.x, bc.y));
		result = result + blendingWeight.y * ctex.sample(mk2(bc.z, bc.w));
	}
	store_rgba8(out, out_pitch, x, y, result);
}

// Cached per context: do pixel-centre taps at offsets -2 .. 2 resolve to texel fetches along an axis of n texels?
static bool centre_taps_exact(gr_ctx *ctx, uint32_t n, float inv)
{
	const uint64_t key = (uint64_t(n) << 32) | __builtin_bit_cast(uint32_t, inv);
	{
		std::lock_guard<std::mutex> holder{ctx->lock};
		auto it = ctx->centre_taps_exact.find(key);
		if (it != ctx->centre_taps_exact.end())
			return it->second;
	}
	static const int ks[] = {-2, -1, 0, 1, 2};
	const bool ok = aa::axis_taps_exact(int(n), inv, ks, 5);
	std::lock_guard<std::mutex> holder{ctx->lock};
	ctx->centre_taps_exact[key] = ok;
	return ok;
}
// Cached per context: do the diagonal searches' coordinate walks (up to 17 steps either way; in x also from a quarter texel beside the
// centre) land on texels along an axis of n texels?
static bool diag_walk_exact(gr_ctx *ctx, uint32_t n, float inv, bool with_quarter)
{
	const uint64_t key = (uint64_t(n) << 32) | __builtin_bit_cast(uint32_t, inv);
	auto &cache = with_quarter ? ctx->diag_walk_exact_x : ctx->diag_walk_exact_y;
	{
		std::lock_guard<std::mutex> holder{ctx->lock};
		auto it = cache.find(key);
		if (it != cache.end())
			return it->second;
	}
	const bool ok = aa::axis_walk_exact(int(n), inv, 17, false) && (!with_quarter || aa::axis_walk_exact(int(n), inv, 17, true));
	std::lock_guard<std::mutex> holder{ctx->lock};
	cache[key] = ok;
	return ok;
}
static bool use_fast_aa(gr_ctx *ctx, uint32_t w, uint32_t h, float inv_w, float inv_h)
{
	static const bool forced_generic = gr_measurement_switch("GRANITE_AA_GENERIC") != nullptr;
	return !forced_generic && centre_taps_exact(ctx, w, inv_w) && centre_taps_exact(ctx, h, inv_h);
}

// TAA resolve (taa_resolve.frag + reprojection.h): aa_core.hpp (taa_pixel) + aa_fast_kernels.hpp (k_taa_fast).

// ---- blit (shaders/blit.frag: FragColor = textureLod(uTex, vUV, 0)) -----------------------------------------------------------
// The full-screen copy Granite's tools use between targets of different size / format (tools/aa_bench.cpp:97-105,138-147).
// Texel decode by input format, LinearClamp or NearestClamp at the output pixel's centre, store by output format.
struct BlitArgs
{
	DevImage in;
	DevImageRW out;
	uint32_t in_format, out_format;
	int linear;
	const float *srgb_decode;
	const uint2 *srgb_encode;
};

__device__ __forceinline__ v4 blit_texel(const BlitArgs &a, int x, int y)
{
	x = clampi(x, 0, a.in.w - 1);
	y = clampi(y, 0, a.in.h - 1);
	if (a.in_format == GR_FORMAT_R16G16B16A16_SFLOAT)
	{
		const float4 t = load_rgba16f(a.in, x, y);
		return mk4(t.x, t.y, t.z, t.w);
	}
	const uint32_t t = *reinterpret_cast<const uint32_t *>(a.in.ptr + uint32_t(y) * a.in.pitch + uint32_t(x) * 4u);
	if (a.in_format == GR_FORMAT_R8G8B8A8_SRGB)
		return mk4(a.srgb_decode[t & 255u], a.srgb_decode[(t >> 8) & 255u], a.srgb_decode[(t >> 16) & 255u], unorm8_to_float(t >> 24));
	return mk4(unorm8_to_float(t & 255u), unorm8_to_float((t >> 8) & 255u), unorm8_to_float((t >> 16) & 255u), unorm8_to_float(t >> 24));
}

__global__ __launch_bounds__(AA_BLOCK_X *AA_BLOCK_Y) void k_blit(BlitArgs a)
{
	const int x = blockIdx.x * AA_BLOCK_X + threadIdx.x, y = blockIdx.y * AA_BLOCK_Y + threadIdx.y;
	if (x >= a.out.w || y >= a.out.h)
		return;
	const v2 uv = mk2((float(x) + 0.5f) * (1.0f / float(a.out.w)), (float(y) + 0.5f) * (1.0f / float(a.out.h)));
	v4 c;
	if (a.linear)
	{
		int x0, y0;
		float wa, wb;
		aa::linear_axis(uv.x * float(a.in.w) - 0.5f, x0, wa);
		aa::linear_axis(uv.y * float(a.in.h) - 0.5f, y0, wb);
		const v4 t00 = blit_texel(a, x0, y0), t10 = blit_texel(a, x0 + 1, y0), t01 = blit_texel(a, x0, y0 + 1), t11 = blit_texel(a, x0 + 1, y0 + 1);
		// a weight that is exactly 0 (a tap on a texel centre) does not read its texel: no 0 * inf (oracle_common.h: linear_combine)
		const v4 top = wa == 0.0f ? t00 : t00 * (1.0f - wa) + t10 * wa;
		c = top;
		if (wb != 0.0f)
		{
			const v4 bot = wa == 0.0f ? t01 : t01 * (1.0f - wa) + t11 * wa;
			c = top * (1.0f - wb) + bot * wb;
		}
	}
	else
		c = blit_texel(a, int(floorf(uv.x * float(a.in.w))), int(floorf(uv.y * float(a.in.h))));
	if (a.out_format == GR_FORMAT_R16G16B16A16_SFLOAT)
		store_rgba16f(a.out, x, y, make_float4(c.x, c.y, c.z, c.w));
	else
	{
		uint32_t packed;
		if (a.out_format == GR_FORMAT_R8G8B8A8_SRGB)
			packed = encode_srgb8_lut(c.x, a.srgb_encode) | (encode_srgb8_lut(c.y, a.srgb_encode) << 8) | (encode_srgb8_lut(c.z, a.srgb_encode) << 16) |
			         (unorm8(c.w) << 24);
		else
			packed = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
		*reinterpret_cast<uint32_t *>(a.out.ptr + uint32_t(y) * a.out.pitch + uint32_t(x) * 4u) = packed;
	}
}

static bool check_image(const gr_image *img, uint32_t bpp, uint32_t w, uint32_t h)
{
	return img && img->ptr && img->width == w && img->height == h && img->pitch_bytes >= w * bpp;
}
static bool is_rgba8(uint32_t f) { return f == GR_FORMAT_R8G8B8A8_SRGB || f == GR_FORMAT_R8G8B8A8_UNORM; }
static dim3 aa_grid(uint32_t w, uint32_t h) { return dim3(gr_div_up(w, AA_BLOCK_X), gr_div_up(h, AA_BLOCK_Y)); }
static dim3 fast_grid(uint32_t w, uint32_t h) { return dim3(gr_div_up(w, FAST_BW), gr_div_up(h, FAST_BH)); }
} // namespace

extern "C" {

int gr_smaa_set_luts(gr_ctx *ctx, const void *area_rg8, const void *search_r8)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, area_rg8 && search_r8);
	// Both tables are kept decoded (float(v) / 255.0f, the UNORM8 conversion of the fetch, evaluated once here).
	std::vector<float> area(160 * 560 * 2), search(64 * 16);
	for (size_t i = 0; i < area.size(); i++)
		area[i] = float(static_cast<const uint8_t *>(area_rg8)[i]) / 255.0f;
	for (size_t i = 0; i < search.size(); i++)
		search[i] = float(static_cast<const uint8_t *>(search_r8)[i]) / 255.0f;
	if (!ctx->smaa_area)
		GR_CHECK_HIP(ctx, hipMalloc(&ctx->smaa_area, area.size() * sizeof(float)));
	if (!ctx->smaa_search)
		GR_CHECK_HIP(ctx, hipMalloc(&ctx->smaa_search, search.size() * sizeof(float)));
	GR_CHECK_HIP(ctx, hipMemcpy(ctx->smaa_area, area.data(), area.size() * sizeof(float), hipMemcpyHostToDevice));
	GR_CHECK_HIP(ctx, hipMemcpy(ctx->smaa_search, search.data(), search.size() * sizeof(float), hipMemcpyHostToDevice));
	return GR_OK;
}

int gr_fxaa(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, const gr_push_fxaa *push)
{
	return gr_fxaa_rows(ctx, stream, in, out, push, nullptr);
}

int gr_fxaa_rows(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, const gr_push_fxaa *push, const gr_rows *rows)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push && in && out && in->width && in->height);
	GR_CHECK_ARG(ctx, check_image(in, 4, in->width, in->height) && check_image(out, 4, in->width, in->height));
	GR_CHECK_ARG(ctx, is_rgba8(in->format) && is_rgba8(out->format) && in->ptr != out->ptr);
	const RowSpan span = resolve_rows(rows, in->height);
	if (span.count() == 0)
		return GR_OK;
	const bool fast = use_fast_aa(ctx, in->width, in->height, push->inv_resolution[0], push->inv_resolution[1]);
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "fxaa"};
	if (fast)
		hipLaunchKernelGGL(k_fxaa_fast, fast_grid(in->width, span.count()), dim3(FAST_BW, FAST_BH), 0, gr_to_stream(stream),
		                   static_cast<const uint8_t *>(in->ptr), in->pitch_bytes, int(in->width), int(in->height), static_cast<uint8_t *>(out->ptr),
		                   out->pitch_bytes, push->inv_resolution[0], push->inv_resolution[1], span);
	else
		hipLaunchKernelGGL(k_fxaa_generic, aa_grid(in->width, span.count()), dim3(AA_BLOCK_X, AA_BLOCK_Y), 0, gr_to_stream(stream), make_tex8<4>(in),
		                   static_cast<uint8_t *>(out->ptr), out->pitch_bytes, *push, span);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_smaa_edge_detection(gr_ctx *ctx, gr_stream stream, const gr_image *color, const gr_image *edges, const gr_push_smaa *push, int quality)
{
	return gr_smaa_edge_detection_rows(ctx, stream, color, edges, push, quality, nullptr);
}

// The bit planes of an edge texture of this size on this stream's allocation (grown on demand; one set per launch stream).
static int smaa_planes_of(gr_ctx *ctx, gr_stream stream, uint32_t width, uint32_t height, SmaaBitPlanes &planes)
{
	planes = {};
	planes.row_words = smaa_bit_words(int(width));
	planes.col_words = smaa_bit_words(int(height));
	const size_t row_plane = size_t(planes.rows()) * planes.row_words * 8u, col_plane = size_t(planes.cols()) * planes.col_words * 8u;
	const size_t need = 2 * row_plane + 2 * col_plane;
	uint8_t *memory = nullptr;
	{
		std::lock_guard<std::mutex> holder{ctx->lock};
		gr_ctx::SmaaBits &bits = ctx->smaa_bits[stream];
		if (bits.bytes < need)
		{
			// a replaced allocation may still be read by a launch in flight on this stream: drain it first
			if (bits.memory)
			{
				(void)hipStreamSynchronize(gr_to_stream(stream));
				(void)hipFree(bits.memory);
				bits = {};
			}
			if (hipMalloc(&bits.memory, need) != hipSuccess)
				bits = {};
			else
				bits.bytes = need;
		}
		memory = static_cast<uint8_t *>(bits.memory);
	}
	if (!memory) // (outside the lock: fail() takes it)
		return ctx->fail(GR_ERR_OUT_OF_MEMORY, "SMAA: %zu bytes of edge bit planes", need);
	planes.row_r = reinterpret_cast<uint64_t *>(memory);
	planes.row_g = reinterpret_cast<uint64_t *>(memory + row_plane);
	planes.col_r = reinterpret_cast<uint64_t *>(memory + 2 * row_plane);
	planes.col_g = reinterpret_cast<uint64_t *>(memory + 2 * row_plane + col_plane);
	return GR_OK;
}

int gr_smaa_edge_detection_rows(gr_ctx *ctx, gr_stream stream, const gr_image *color, const gr_image *edges, const gr_push_smaa *push, int quality,
                                const gr_rows *rows)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push && color && edges && color->width && color->height);
	GR_CHECK_ARG(ctx, quality >= 0 && quality <= 3);
	GR_CHECK_ARG(ctx, check_image(color, 4, color->width, color->height) && is_rgba8(color->format));
	GR_CHECK_ARG(ctx, check_image(edges, 2, color->width, color->height) && edges->format == GR_FORMAT_R8G8_UNORM);
	const RowSpan span = resolve_rows(rows, color->height);
	if (span.count() == 0)
		return GR_OK;
	const bool fast = use_fast_aa(ctx, color->width, color->height, push->rt_metrics[0], push->rt_metrics[1]);
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "smaa_edge_detection"};
	if (fast)
		hipLaunchKernelGGL(k_smaa_edges_fast, fast_grid(color->width, span.count()), dim3(FAST_BW, FAST_BH), 0, gr_to_stream(stream),
		                   static_cast<const uint8_t *>(color->ptr), color->pitch_bytes, int(color->width), int(color->height),
		                   static_cast<uint8_t *>(edges->ptr), edges->pitch_bytes, smaa_preset(quality).threshold, span);
	else
		hipLaunchKernelGGL(k_smaa_edges_generic, aa_grid(color->width, span.count()), dim3(AA_BLOCK_X, AA_BLOCK_Y), 0, gr_to_stream(stream),
		                   make_tex8<4>(color), static_cast<uint8_t *>(edges->ptr), edges->pitch_bytes, *push, smaa_preset(quality), span);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_smaa_blend_weight(gr_ctx *ctx, gr_stream stream, const gr_image *edges, const gr_image *weights, const gr_push_smaa *push, int quality)
{
	return gr_smaa_blend_weight_rows(ctx, stream, edges, weights, push, quality, nullptr);
}

int gr_smaa_blend_weight_rows(gr_ctx *ctx, gr_stream stream, const gr_image *edges, const gr_image *weights, const gr_push_smaa *push, int quality,
                              const gr_rows *rows)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push && edges && weights && edges->width && edges->height);
	GR_CHECK_ARG(ctx, quality >= 0 && quality <= 3);
	GR_CHECK_ARG(ctx, check_image(edges, 2, edges->width, edges->height) && edges->format == GR_FORMAT_R8G8_UNORM);
	GR_CHECK_ARG(ctx, check_image(weights, 4, edges->width, edges->height) && weights->format == GR_FORMAT_R8G8B8A8_UNORM);
	if (!ctx->smaa_area || !ctx->smaa_search)
		return ctx->fail(GR_ERR_INVALID_ARGUMENT, "gr_smaa_blend_weight: SMAA lookup tables not set (gr_smaa_set_luts)");
	SmaaWeightsArgs S;
	S.edges = make_tex8<2>(edges);
	S.area = {static_cast<const float *>(ctx->smaa_area), 160, 560};
	S.search = {static_cast<const float *>(ctx->smaa_search), 64, 16};
	S.rt = v4{push->rt_metrics[0], push->rt_metrics[1], push->rt_metrics[2], push->rt_metrics[3]};
	S.P = smaa_preset(quality);
	const RowSpan span = resolve_rows(rows, edges->height);
	if (span.count() == 0)
		return GR_OK;
	// Bit-plane form (smaa_weights.hpp): the searches' pass conditions are read off the flags, which stands on the search
	// coordinates staying within a few hundredths of a texel of their nominal positions -- true far beyond 16K texels per axis.
	static const bool forced_generic = gr_measurement_switch("GRANITE_AA_GENERIC") != nullptr;
	if (!forced_generic && edges->width <= 16384 && edges->height <= 16384)
	{
		SmaaBitPlanes planes;
		const int code = smaa_planes_of(ctx, stream, edges->width, edges->height, planes);
		if (code != GR_OK)
			return code;
		// tiles of 64 padded rows the band's workgroups stage from: rows first - 127 .. end + 175 (+ the block rounding)
		const int tile_first = max(0, (int(span.first) - 128 + SMAA_BITS_PAD) >> 6);
		const int tile_last = min(planes.col_words - 1, (int(span.end) + FAST_BH + 192 + SMAA_BITS_PAD) >> 6);
		const int tiles = planes.row_words * (tile_last - tile_first + 1);
		SmaaWeightsBitsArgs B = {static_cast<const uint8_t *>(edges->ptr), edges->pitch_bytes, int(edges->width), int(edges->height), planes, S.area, S.search, S.rt, S.P,
		                         centre_taps_exact(ctx, edges->width, push->rt_metrics[0]) && centre_taps_exact(ctx, edges->height, push->rt_metrics[1]), 0};
		static const bool float_walks = gr_measurement_switch("GRANITE_SMAA_FLOAT_DIAG_WALKS") != nullptr;
		B.diag_walks_exact = S.P.diag && B.centres_snap && !float_walks && diag_walk_exact(ctx, edges->width, push->rt_metrics[0], true) &&
		                     diag_walk_exact(ctx, edges->height, push->rt_metrics[1], false);
		gr_scoped_timing timing{ctx, gr_to_stream(stream), "smaa_blend_weight"};
		hipLaunchKernelGGL(k_smaa_pack_edges, dim3(tiles), dim3(256), 0, gr_to_stream(stream), B.edges, B.edges_pitch, B.w, B.h, planes,
		                   tile_first, tile_last - tile_first + 1);
		hipLaunchKernelGGL(k_smaa_weights_bits, fast_grid(edges->width, span.count()), dim3(FAST_BW, FAST_BH), 0, gr_to_stream(stream), B,
		                   static_cast<uint8_t *>(weights->ptr), weights->pitch_bytes, span);
		GR_CHECK_LAUNCH(ctx);
		return GR_OK;
	}
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "smaa_blend_weight"};
	hipLaunchKernelGGL(k_smaa_weights, aa_grid(edges->width, span.count()), dim3(AA_BLOCK_X, AA_BLOCK_Y), 0, gr_to_stream(stream), S,
	                   static_cast<uint8_t *>(weights->ptr), weights->pitch_bytes, span);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_smaa_neighbor_blend(gr_ctx *ctx, gr_stream stream, const gr_image *color, const gr_image *weights, const gr_image *out,
                           const gr_push_smaa *push)
{
	return gr_smaa_neighbor_blend_rows(ctx, stream, color, weights, out, push, nullptr);
}

int gr_smaa_neighbor_blend_rows(gr_ctx *ctx, gr_stream stream, const gr_image *color, const gr_image *weights, const gr_image *out,
                                const gr_push_smaa *push, const gr_rows *rows)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push && color && weights && out && color->width && color->height);
	GR_CHECK_ARG(ctx, check_image(color, 4, color->width, color->height) && is_rgba8(color->format));
	GR_CHECK_ARG(ctx, check_image(weights, 4, color->width, color->height) && weights->format == GR_FORMAT_R8G8B8A8_UNORM);
	GR_CHECK_ARG(ctx, check_image(out, 4, color->width, color->height) && is_rgba8(out->format) && out->ptr != color->ptr);
	const RowSpan span = resolve_rows(rows, color->height);
	if (span.count() == 0)
		return GR_OK;
	const bool fast = use_fast_aa(ctx, color->width, color->height, push->rt_metrics[0], push->rt_metrics[1]);
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "smaa_neighbor_blend"};
	if (fast)
	{
		const ColorImage c = {static_cast<const uint8_t *>(color->ptr), color->pitch_bytes, int(color->width), int(color->height)};
		const ColorImage b = {static_cast<const uint8_t *>(weights->ptr), weights->pitch_bytes, int(weights->width), int(weights->height)};
		const bool wide = color->width % 4 == 0 && ((color->pitch_bytes | weights->pitch_bytes | out->pitch_bytes) & 15u) == 0 &&
		                  ((uintptr_t(color->ptr) | uintptr_t(weights->ptr) | uintptr_t(out->ptr)) & 15u) == 0;
		const dim3 block(64, 4);
		if (wide)
			hipLaunchKernelGGL(k_smaa_blend_fast<4>, dim3(gr_div_up(color->width, 256), gr_div_up(span.count(), 4)), block, 0, gr_to_stream(stream), c, b,
			                   static_cast<uint8_t *>(out->ptr), out->pitch_bytes, push->rt_metrics[0], push->rt_metrics[1], span);
		else
			hipLaunchKernelGGL(k_smaa_blend_fast<1>, dim3(gr_div_up(color->width, 64), gr_div_up(span.count(), 4)), block, 0, gr_to_stream(stream), c, b,
			                   static_cast<uint8_t *>(out->ptr), out->pitch_bytes, push->rt_metrics[0], push->rt_metrics[1], span);
	}
	else
		hipLaunchKernelGGL(k_smaa_blend_generic, aa_grid(color->width, span.count()), dim3(AA_BLOCK_X, AA_BLOCK_Y), 0, gr_to_stream(stream),
		                   make_tex8<4>(color), make_tex8<4>(weights), static_cast<uint8_t *>(out->ptr), out->pitch_bytes, *push, span);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_taa_resolve(gr_ctx *ctx, gr_stream stream, const gr_image *current, const gr_image *depth, const gr_image *mv, const gr_image *history,
                   const gr_image *out_color, const gr_image *out_history, const gr_push_taa *push, int quality)
{
	return gr_taa_resolve_rows(ctx, stream, current, depth, mv, history, out_color, out_history, push, quality, nullptr);
}

int gr_taa_resolve_rows(gr_ctx *ctx, gr_stream stream, const gr_image *current, const gr_image *depth, const gr_image *mv, const gr_image *history,
                        const gr_image *out_color, const gr_image *out_history, const gr_push_taa *push, int quality, const gr_rows *rows)
{
	return gr_taa_resolve_band(ctx, stream, current, depth, mv, history, out_color, out_history, push, quality, rows, nullptr, nullptr);
}

int gr_taa_resolve_band(gr_ctx *ctx, gr_stream stream, const gr_image *current, const gr_image *depth, const gr_image *mv, const gr_image *history,
                        const gr_image *out_color, const gr_image *out_history, const gr_push_taa *push, int quality, const gr_rows *rows,
                        const gr_rows *history_rows, uint32_t *reach_flag)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, push && current && depth && mv && out_color && out_history);
	GR_CHECK_ARG(ctx, quality >= 0 && quality <= 2);
	const uint32_t w = current->width, h = current->height;
	GR_CHECK_ARG(ctx, w && h);
	const bool current_b10 = current->format == GR_FORMAT_B10G11R11_UFLOAT_PACK32, color_b10 = out_color->format == GR_FORMAT_B10G11R11_UFLOAT_PACK32;
	GR_CHECK_ARG(ctx, check_image(current, current_b10 ? 4 : 8, w, h) && (current_b10 || current->format == GR_FORMAT_R16G16B16A16_SFLOAT));
	GR_CHECK_ARG(ctx, check_image(depth, 4, w, h) && depth->format == GR_FORMAT_D32_SFLOAT);
	GR_CHECK_ARG(ctx, check_image(mv, 4, w, h) && mv->format == GR_FORMAT_R16G16_SFLOAT);
	GR_CHECK_ARG(ctx, check_image(out_color, color_b10 ? 4 : 8, w, h) && (color_b10 || out_color->format == GR_FORMAT_R16G16B16A16_SFLOAT));
	GR_CHECK_ARG(ctx, check_image(out_history, 8, w, h) && out_history->format == GR_FORMAT_R16G16B16A16_SFLOAT);
	GR_CHECK_ARG(ctx, !history || (check_image(history, 8, w, h) && history->format == GR_FORMAT_R16G16B16A16_SFLOAT && history->ptr != out_history->ptr));
	const RowSpan span = resolve_rows(rows, h);
	if (span.count() == 0)
		return GR_OK;
	TaaImages im = {};
	im.current = static_cast<const uint8_t *>(current->ptr);
	im.depth = static_cast<const uint8_t *>(depth->ptr);
	im.mv = static_cast<const uint8_t *>(mv->ptr);
	im.history = history ? static_cast<const uint8_t *>(history->ptr) : nullptr;
	im.out_color = static_cast<uint8_t *>(out_color->ptr);
	im.out_history = static_cast<uint8_t *>(out_history->ptr);
	im.current_pitch = current->pitch_bytes;
	im.depth_pitch = depth->pitch_bytes;
	im.mv_pitch = mv->pitch_bytes;
	im.history_pitch = history ? history->pitch_bytes : 0u;
	im.out_color_pitch = out_color->pitch_bytes;
	im.out_history_pitch = out_history->pitch_bytes;
	im.w = int(w);
	im.h = int(h);
	im.current_b10 = current_b10;
	im.color_b10 = color_b10;
	GR_CHECK_ARG(ctx, (history_rows == nullptr) == (reach_flag == nullptr));
	if (history_rows)
	{
		const RowSpan held = resolve_rows(history_rows, h);
		im.hist_first = int(held.first);
		im.hist_end = int(held.end);
		im.reach_flag = reach_flag;
	}
	aa::TaaPush tp;
	for (int i = 0; i < 16; i++)
		tp.reproj[i] = push->reproj[i];
	for (int i = 0; i < 4; i++)
		tp.rt[i] = push->rt_metrics[i];
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "taa_resolve"};
	const dim3 grid = fast_grid(w, span.count()), block(FAST_BW, FAST_BH);
	hipStream_t s = gr_to_stream(stream);
	if (!history)
		hipLaunchKernelGGL((k_taa_fast<0, false>), grid, block, 0, s, im, tp, span);
	else if (quality == 0)
		hipLaunchKernelGGL((k_taa_fast<0, true>), grid, block, 0, s, im, tp, span);
	else if (quality == 1)
		hipLaunchKernelGGL((k_taa_fast<1, true>), grid, block, 0, s, im, tp, span);
	else
		hipLaunchKernelGGL((k_taa_fast<2, true>), grid, block, 0, s, im, tp, span);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_blit(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, int linear)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, in && out && in->ptr && out->ptr && in->ptr != out->ptr && in->width && in->height && out->width && out->height);
	auto supported = [](uint32_t f) { return f == GR_FORMAT_R16G16B16A16_SFLOAT || f == GR_FORMAT_R8G8B8A8_SRGB || f == GR_FORMAT_R8G8B8A8_UNORM; };
	GR_CHECK_ARG(ctx, supported(in->format) && supported(out->format));
	GR_CHECK_ARG(ctx, in->pitch_bytes >= in->width * (in->format == GR_FORMAT_R16G16B16A16_SFLOAT ? 8u : 4u));
	GR_CHECK_ARG(ctx, out->pitch_bytes >= out->width * (out->format == GR_FORMAT_R16G16B16A16_SFLOAT ? 8u : 4u));
	BlitArgs a = {};
	a.in = DevImage{static_cast<const uint8_t *>(in->ptr), int(in->width), int(in->height), in->pitch_bytes};
	a.out = DevImageRW{static_cast<uint8_t *>(out->ptr), int(out->width), int(out->height), out->pitch_bytes};
	a.in_format = in->format;
	a.out_format = out->format;
	a.linear = linear ? 1 : 0;
	a.srgb_decode = ctx->srgb_decode_lut;
	a.srgb_encode = ctx->srgb_encode_lut;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "blit"};
	hipLaunchKernelGGL(k_blit, aa_grid(out->width, out->height), dim3(AA_BLOCK_X, AA_BLOCK_Y), 0, gr_to_stream(stream), a);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}
}
