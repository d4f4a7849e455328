// Single-pass downsampler for gfx950: emit_single_pass_downsample (renderer/post/spd.cpp:56-102) +
// assets/shaders/post/ffx-spd/spd.comp over AMD FidelityFX SPD (ffx_spd.h), with the defines that function sets
// (SUBGROUP, SINGLE_INPUT_TAP, COMPONENTS, FILTER_MOD, REDUCTION_MODE).  RGBA16F in, RGBA16F mip chain out.
//
// What the shader computes (restated in oracle/oracle_spd.cpp, which the executed shader pins bit for bit):
//   level 0 = one sampler tap per texel at the centre of its 2 x 2 source footprint; levels 1..5 from the UNROUNDED fp32
//   values of the level above, ((tl + tr) + bl) + br; level 6 from level 5 AS STORED (fp16, filter_mod applied, coordinates
//   clamped), ((p(0,0) + p(0,1)) + p(1,0)) + p(1,1); levels 7..11 from unrounded values again.  Stores are bounds-checked,
//   multiplied by filter_mods[mip], chopped to the component count.
//
// Mapping (own design): a workgroup of 256 threads owns 64 x 64 source texels; thread (x, y) of the 16 x 16 grid takes the
// four taps of its 2 x 2 level-0 block, reduces them to its level-1 texel in registers and the workgroup walks levels 2..5
// through 4 KB of LDS.  The shader's last-workgroup hand-over through an atomic counter is a second, one-workgroup launch on
// the same stream here (k_spd_tail), as for the depth hierarchy (hiz.hip): with eight XCD-private L2s that is the cheaper
// way to make level 5 visible.  Compiled with -ffp-contract=off: every level is bit-identical to the oracle.
#include "ctx.hpp"
#include "device_common.hpp"

namespace
{
constexpr int SPD_MAX_MIPS = 12; // uImages[12] (spd.comp:38)

struct SpdParams
{
	const uint8_t *in;
	int iw, ih;
	uint32_t in_pitch;
	uint8_t *chain;
	int w0, h0, mips, components, depth_mode, has_mods;
	float inv_w, inv_h;
	float mods[SPD_MAX_MIPS][4];
	uint32_t offset[SPD_MAX_MIPS]; // byte offset of level l inside the chain
};

__device__ __forceinline__ float4 chop(float4 v, int components)
{
	if (components < 4)
		v.w = 0.0f;
	if (components < 3)
		v.z = 0.0f;
	if (components < 2)
		v.y = 0.0f;
	return v;
}

__device__ __forceinline__ float4 fetch_clamped(const SpdParams &p, int x, int y)
{
	x = clampi(x, 0, p.iw - 1);
	y = clampi(y, 0, p.ih - 1);
	const f16x4 h = *reinterpret_cast<const f16x4 *>(p.in + size_t(y) * p.in_pitch + size_t(x) * 8u);
	return make_float4(float(h.x), float(h.y), float(h.z), float(h.w));
}

__device__ __forceinline__ float lerp2(float a, float b, float wa, float wb) { return a * wa + b * wb; }

// SpdLoadSourceImage (spd.comp:74-89) at source coordinate (px, py) = 2 x the level-0 texel.
__device__ __forceinline__ float4 load_source(const SpdParams &p, int px, int py)
{
	const float fx = float(px), fy = float(py);
	if (p.depth_mode)
	{
		const float u = 0.5f * (fx * p.inv_w + p.inv_w), v = 0.5f * (fy * p.inv_h + p.inv_h);
		return chop(fetch_clamped(p, int(floorf(u * float(p.iw))), int(floorf(v * float(p.ih)))), p.components);
	}
	// StockSampler::LinearClamp, LOD 0: unnormalised coordinate uv * size - 0.5, weights (1 - a, a), rows first.
	const float u = (fx * p.inv_w + p.inv_w) * float(p.iw) - 0.5f;
	const float v = (fy * p.inv_h + p.inv_h) * float(p.ih) - 0.5f;
	const float fu = floorf(u), fv = floorf(v);
	const float a = u - fu, b = v - fv;
	const int x0 = int(fu), y0 = int(fv);
	const float4 t00 = fetch_clamped(p, x0, y0), t10 = fetch_clamped(p, x0 + 1, y0);
	const float4 t01 = fetch_clamped(p, x0, y0 + 1), t11 = fetch_clamped(p, x0 + 1, y0 + 1);
	const float na = 1.0f - a, nb = 1.0f - b;
	const float4 top = make_float4(lerp2(t00.x, t10.x, na, a), lerp2(t00.y, t10.y, na, a), lerp2(t00.z, t10.z, na, a), lerp2(t00.w, t10.w, na, a));
	const float4 bot = make_float4(lerp2(t01.x, t11.x, na, a), lerp2(t01.y, t11.y, na, a), lerp2(t01.z, t11.z, na, a), lerp2(t01.w, t11.w, na, a));
	return chop(make_float4(lerp2(top.x, bot.x, nb, b), lerp2(top.y, bot.y, nb, b), lerp2(top.z, bot.z, nb, b), lerp2(top.w, bot.w, nb, b)),
	            p.components);
}

// SpdReduce4 (spd.comp:177-187)
__device__ __forceinline__ float4 reduce4(const SpdParams &p, float4 v0, float4 v1, float4 v2, float4 v3)
{
	if (p.depth_mode)
	{
		const float m0 = (v1.x < v0.x) ? v1.x : v0.x, m1 = (v3.x < v2.x) ? v3.x : v2.x;
		return make_float4((m1 < m0) ? m1 : m0, 0.0f, 0.0f, 0.0f);
	}
	return chop(make_float4((((v0.x + v1.x) + v2.x) + v3.x) * 0.25f, (((v0.y + v1.y) + v2.y) + v3.y) * 0.25f,
	                        (((v0.z + v1.z) + v2.z) + v3.z) * 0.25f, (((v0.w + v1.w) + v2.w) + v3.w) * 0.25f),
	            p.components);
}

// SpdStore (spd.comp:91-102)
__device__ __forceinline__ void store_level(const SpdParams &p, int x, int y, float4 v, int mip)
{
	const int mw = max(p.w0 >> mip, 1), mh = max(p.h0 >> mip, 1);
	if (x >= mw || y >= mh)
		return;
	if (p.has_mods)
		v = make_float4(v.x * p.mods[mip][0], v.y * p.mods[mip][1], v.z * p.mods[mip][2], v.w * p.mods[mip][3]);
	*reinterpret_cast<f16x4 *>(p.chain + p.offset[mip] + (size_t(y) * mw + x) * 8u) = pack_rgba16f(chop(v, p.components));
}

// The workgroup holds a 16 x 16 block of level `level` in s (row-major), top-left texel (16 gx, 16 gy) of that level: levels
// level + 1 .. level + 4 (SpdDownsampleNextFour, ffx_spd.h:796-813).
__device__ __forceinline__ void reduce_ladder(const SpdParams &p, float4 *s, int t, int gx, int gy, int level)
{
	int n = 16;
	for (int mip = level + 1; n > 1 && mip < p.mips; mip++)
	{
		const int half = n >> 1;
		__syncthreads();
		float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		const bool mine = t < half * half;
		const int x = t & (half - 1), y = t / half;
		if (mine)
		{
			v = reduce4(p, s[(2 * y) * n + 2 * x], s[(2 * y) * n + 2 * x + 1], s[(2 * y + 1) * n + 2 * x], s[(2 * y + 1) * n + 2 * x + 1]);
			store_level(p, half * gx + x, half * gy + y, v, mip);
		}
		__syncthreads();
		if (mine)
			s[y * half + x] = v;
		n = half;
	}
}

__global__ __launch_bounds__(256) void k_spd_tiles(SpdParams p)
{
	__shared__ float4 s[256];
	const int t = threadIdx.x, lx = t & 15, ly = t >> 4;
	const int gx = blockIdx.x, gy = blockIdx.y;
	// level-0 block of this thread: texels (32 gx + 2 lx + i, 32 gy + 2 ly + j), source coordinate = 2 x that
	const int x0 = 32 * gx + 2 * lx, y0 = 32 * gy + 2 * ly;
	const float4 v00 = load_source(p, 2 * x0, 2 * y0), v10 = load_source(p, 2 * x0 + 2, 2 * y0);
	const float4 v01 = load_source(p, 2 * x0, 2 * y0 + 2), v11 = load_source(p, 2 * x0 + 2, 2 * y0 + 2);
	store_level(p, x0, y0, v00, 0);
	store_level(p, x0 + 1, y0, v10, 0);
	store_level(p, x0, y0 + 1, v01, 0);
	store_level(p, x0 + 1, y0 + 1, v11, 0);
	if (p.mips <= 1)
		return;
	const float4 v = reduce4(p, v00, v10, v01, v11);
	store_level(p, 16 * gx + lx, 16 * gy + ly, v, 1);
	s[ly * 16 + lx] = v;
	reduce_ladder(p, s, t, gx, gy, 1);
}

// Levels 6.. (the shader's last workgroup, ffx_spd.h:826-837): level 6 from the stored level 5.
__global__ __launch_bounds__(256) void k_spd_tail(SpdParams p)
{
	__shared__ float4 s[256];
	const int t = threadIdx.x, lx = t & 15, ly = t >> 4;
	const int w5 = max(p.w0 >> 5, 1), h5 = max(p.h0 >> 5, 1);
	const uint8_t *level5 = p.chain + p.offset[5];
	auto load5 = [&](int x, int y) {
		x = clampi(x, 0, w5 - 1);
		y = clampi(y, 0, h5 - 1);
		const f16x4 h = *reinterpret_cast<const f16x4 *>(level5 + (size_t(y) * w5 + x) * 8u);
		return chop(make_float4(float(h.x), float(h.y), float(h.z), float(h.w)), p.components);
	};
	float4 v6[2][2];
#pragma unroll
	for (int j = 0; j < 2; j++)
#pragma unroll
		for (int i = 0; i < 2; i++)
		{
			const int x = 2 * lx + i, y = 2 * ly + j;
			// SpdReduceLoad4(base): (0,0), (0,1), (1,0), (1,1) -- column first (ffx_spd.h:472-480)
			v6[j][i] = reduce4(p, load5(2 * x, 2 * y), load5(2 * x, 2 * y + 1), load5(2 * x + 1, 2 * y), load5(2 * x + 1, 2 * y + 1));
			store_level(p, x, y, v6[j][i], 6);
		}
	if (p.mips <= 7)
		return;
	const float4 v = reduce4(p, v6[0][0], v6[0][1], v6[1][0], v6[1][1]);
	store_level(p, lx, ly, v, 7);
	s[ly * 16 + lx] = v;
	reduce_ladder(p, s, t, 0, 0, 7);
}
} // namespace

extern "C" {

int gr_spd_downsample(gr_ctx *ctx, gr_stream stream, const gr_spd_args *args)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, args && args->input.ptr && args->chain);
	GR_CHECK_ARG(ctx, args->input.format == GR_FORMAT_R16G16B16A16_SFLOAT && args->input.width > 0 && args->input.height > 0 &&
	                      args->input.pitch_bytes >= args->input.width * 8u && (args->input.pitch_bytes & 7u) == 0);
	GR_CHECK_ARG(ctx, args->width > 0 && args->height > 0);
	GR_CHECK_ARG(ctx, args->mips >= 1 && args->mips <= uint32_t(SPD_MAX_MIPS));
	GR_CHECK_ARG(ctx, args->components >= 1 && args->components <= 4);
	GR_CHECK_ARG(ctx, args->reduction_mode == GR_SPD_REDUCTION_COLOR || args->reduction_mode == GR_SPD_REDUCTION_DEPTH);
	// One workgroup reduces what is left after level 5, at most 64 x 64 texels of it (ffx_spd.h:833): 2048 x 2048 at level 0.
	if (args->mips > 6 && (args->width > 2048u || args->height > 2048u))
		return ctx->fail(GR_ERR_INVALID_ARGUMENT, "gr_spd_downsample: %u x %u with %u mips is beyond SPD's single tail workgroup",
		                 args->width, args->height, args->mips);

	SpdParams p{};
	p.in = static_cast<const uint8_t *>(args->input.ptr);
	p.iw = int(args->input.width);
	p.ih = int(args->input.height);
	p.in_pitch = args->input.pitch_bytes;
	p.chain = static_cast<uint8_t *>(args->chain);
	p.w0 = int(args->width);
	p.h0 = int(args->height);
	p.mips = int(args->mips);
	p.components = int(args->components);
	p.depth_mode = args->reduction_mode == GR_SPD_REDUCTION_DEPTH;
	p.has_mods = args->filter_mods != nullptr;
	p.inv_w = 1.0f / float(args->input.width); // spd.cpp:87-88
	p.inv_h = 1.0f / float(args->input.height);
	for (uint32_t m = 0; m < args->mips; m++)
	{
		for (int c = 0; c < 4; c++)
			p.mods[m][c] = args->filter_mods ? args->filter_mods[4 * m + c] : 1.0f;
		p.offset[m] = uint32_t(gr_mip_chain_offset(args->width, args->height, 8, m));
	}

	gr_scoped_timing timing(ctx, gr_to_stream(stream), "spd");
	hipLaunchKernelGGL(k_spd_tiles, dim3((args->width + 31u) / 32u, (args->height + 31u) / 32u), dim3(256), 0, gr_to_stream(stream), p);
	GR_CHECK_LAUNCH(ctx);
	if (args->mips > 6)
	{
		hipLaunchKernelGGL(k_spd_tail, dim3(1), dim3(256), 0, gr_to_stream(stream), p);
		GR_CHECK_LAUNCH(ctx);
	}
	return GR_OK;
}
}
