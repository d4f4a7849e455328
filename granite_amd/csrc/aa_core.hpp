// Per-pixel arithmetic of the anti-aliasing passes, shared by the gfx950 kernels (aa.hip) and by a host build that the CPU
// tests run against the oracle (tests/cpp/aa_core_host.cpp): the kernels add tiling, LDS staging and launch geometry around
// these functions, nothing else, so what a GPU run still has to prove is the plumbing.
//
// Everything here keeps the association order of the shaders (assets/shaders/post/fxaa.frag, SMAA.hlsl) as the oracle restates
// it (oracle/oracle_aa.cpp); compile without FMA contraction.  `mad` of SMAA is fmaf.
//
// Sampler model (oracle_common.h "Sub-texel resolution"): LinearClamp with exact fp32 weights, except that a coordinate within
// 2^-8 of a texel centre selects that texel alone.  A pass whose taps sit on pixel centres therefore reads texels, and the
// fast kernels below are built on that; `axis_taps_exact` is the host-side proof, per image size, that every such tap of a
// launch does snap (the launchers fall back to the generic sampler kernels where it does not -- beyond ~16K pixels per axis).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define AA_HD __host__ __device__ __forceinline__
#else
#define AA_HD inline
#endif

namespace aa
{
constexpr float SAMPLER_SNAP = 1.0f / 256.0f;

AA_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// f = unnormalised coordinate - 0.5: index of the first texel and weight of the second (oracle_common.h: linear_axis).
AA_HD void linear_axis(float f, int &i0, float &weight)
{
	const float fl = floorf(f + SAMPLER_SNAP);
	float a = f - fl;
	if (a < SAMPLER_SNAP)
		a = 0.0f;
	i0 = int(fl);
	weight = a;
}

// UNORM8 -> float, bit-identical to float(v) / 255.0f for v = 0..255 (device_common.hpp: unorm8_to_float).
AA_HD float unorm8_decode(uint32_t v)
{
	const float f = float(v);
	return fmaf(f, 0x1.010102p-8f, f * -0x1.fdfdfep-33f);
}

// UNORM8 store: NaN and negatives -> 0, >= 1 -> 255, otherwise floor(v * 255 + 0.5).
AA_HD uint32_t unorm8_encode(float v)
{
	if (!(v > 0.0f))
		return 0u;
	if (v >= 1.0f)
		return 255u;
	return uint32_t(int(v * 255.0f + 0.5f));
}

struct f3
{
	float x, y, z;
};
struct f4
{
	float x, y, z, w;
};

AA_HD float luma_of(float r, float g, float b, float wr, float wg, float wb) { return r * wr + g * wg + b * wb; }

// ---- host-side proof that a launch's pixel-centre taps are texel fetches ---------------------------------------------------
// For every pixel index p of an axis of n texels, with tc = (p + 0.5) * inv: the tap fma(inv, k, tc) (k = 0: tc itself) must
// resolve to texel p + k with weight 0.  `ks` are the integer tap offsets the pass uses along that axis.
inline bool axis_taps_exact(int n, float inv, const int *ks, int nk)
{
	for (int p = 0; p < n; p++)
	{
		const float tc = (float(p) + 0.5f) * inv;
		for (int j = 0; j < nk; j++)
		{
			const float c = ks[j] == 0 ? tc : fmaf(inv, float(ks[j]), tc);
			int i0;
			float a;
			linear_axis(c * float(n) - 0.5f, i0, a);
			if (i0 != p + ks[j] || a != 0.0f)
				return false;
		}
	}
	return true;
}

// ---- FXAA (fxaa.frag:20-67) ------------------------------------------------------------------------------------------------
constexpr float FXAA_LUMA_R = 0.299f, FXAA_LUMA_G = 0.587f, FXAA_LUMA_B = 0.114f;

// Tile: f4 texel(int x, int y) = (r, g, b, luma) of the clamped texel, decoded once.  The pixel's own taps (centre and
// corners) are texel fetches; the four taps along the edge direction are bilinear.
template <typename Tile>
AA_HD f3 fxaa_sample(const Tile &t, float u, float v, float fw, float fh)
{
	int ix, iy;
	float a, b;
	linear_axis(u * fw - 0.5f, ix, a);
	linear_axis(v * fh - 0.5f, iy, b);
	const f4 t00 = t.texel(ix, iy), t10 = t.texel(ix + 1, iy), t01 = t.texel(ix, iy + 1), t11 = t.texel(ix + 1, iy + 1);
	// t * (1 - 0) + t' * 0 == t for UNORM values, so the snapped case needs no branch
	const float oma = 1.0f - a, omb = 1.0f - b;
	f3 top = {t00.x * oma + t10.x * a, t00.y * oma + t10.y * a, t00.z * oma + t10.z * a};
	f3 bot = {t01.x * oma + t11.x * a, t01.y * oma + t11.y * a, t01.z * oma + t11.z * a};
	return {top.x * omb + bot.x * b, top.y * omb + bot.y * b, top.z * omb + bot.z * b};
}

// Returns the packed RGBA8 (gamma-space bytes, alpha 255) of pixel (x, y).
template <typename Tile>
AA_HD uint32_t fxaa_pixel(const Tile &t, int x, int y, float inv_w, float inv_h, float fw, float fh)
{
	const float FXAA_REDUCE_MIN = 1.0f / 128.0f, FXAA_REDUCE_MUL = 1.0f / 8.0f, FXAA_SPAN_MAX = 8.0f;
	const float u = (float(x) + 0.5f) * inv_w, v = (float(y) + 0.5f) * inv_h;
	const float lumaNW = t.texel(x - 1, y - 1).w, lumaNE = t.texel(x + 1, y - 1).w;
	const float lumaSW = t.texel(x - 1, y + 1).w, lumaSE = t.texel(x + 1, y + 1).w;
	const float lumaM = t.texel(x, y).w;
	const float lumaMin = fminf(lumaM, fminf(fminf(lumaNW, lumaNE), fminf(lumaSW, lumaSE)));
	const float lumaMax = fmaxf(lumaM, fmaxf(fmaxf(lumaNW, lumaNE), fmaxf(lumaSW, lumaSE)));
	float dx = -((lumaNW + lumaNE) - (lumaSW + lumaSE));
	float dy = ((lumaNW + lumaSW) - (lumaNE + lumaSE));
	const float dirReduce = fmaxf((lumaNW + lumaNE + lumaSW + lumaSE) * (0.25f * FXAA_REDUCE_MUL), FXAA_REDUCE_MIN);
	const float rcpDirMin = 1.0f / (fminf(fabsf(dx), fabsf(dy)) + dirReduce);
	dx = fminf(fmaxf(dx * rcpDirMin, -FXAA_SPAN_MAX), FXAA_SPAN_MAX) * inv_w;
	dy = fminf(fmaxf(dy * rcpDirMin, -FXAA_SPAN_MAX), FXAA_SPAN_MAX) * inv_h;
	const float k1 = 1.0f / 3.0f - 0.5f, k2 = 2.0f / 3.0f - 0.5f;
	const f3 s1 = fxaa_sample(t, u + dx * k1, v + dy * k1, fw, fh);
	const f3 s2 = fxaa_sample(t, u + dx * k2, v + dy * k2, fw, fh);
	const f3 rgbA = {0.5f * (s1.x + s2.x), 0.5f * (s1.y + s2.y), 0.5f * (s1.z + s2.z)};
	const f3 s0 = fxaa_sample(t, u + dx * -0.5f, v + dy * -0.5f, fw, fh);
	const f3 s3 = fxaa_sample(t, u + dx * 0.5f, v + dy * 0.5f, fw, fh);
	const f3 rgbB = {rgbA.x * 0.5f + 0.25f * (s0.x + s3.x), rgbA.y * 0.5f + 0.25f * (s0.y + s3.y), rgbA.z * 0.5f + 0.25f * (s0.z + s3.z)};
	const float lumaB = luma_of(rgbB.x, rgbB.y, rgbB.z, FXAA_LUMA_R, FXAA_LUMA_G, FXAA_LUMA_B);
	const bool useA = (lumaB < lumaMin) || (lumaB > lumaMax);
	const f3 c = useA ? rgbA : rgbB;
	return unorm8_encode(c.x) | (unorm8_encode(c.y) << 8) | (unorm8_encode(c.z) << 16) | 0xff000000u;
}

// A pixel whose four corner texels carry the same RGB bytes has dir = 0 exactly: all four edge taps land on the pixel centre,
// rgbA = rgbB = the centre texel, and the pass copies it (alpha 255).  The kernels test this on the raw bytes before anything
// is decoded.
AA_HD bool fxaa_corners_equal(uint32_t nw, uint32_t ne, uint32_t sw, uint32_t se)
{
	return (((nw ^ ne) | (nw ^ sw) | (nw ^ se)) & 0x00ffffffu) == 0u;
}

// ---- SMAA luma edge detection (SMAA.hlsl:689-740) ----------------------------------------------------------------------------
constexpr float SMAA_LUMA_R = 0.2126f, SMAA_LUMA_G = 0.7152f, SMAA_LUMA_B = 0.0722f;

// Luma: float luma(int x, int y) of the clamped texel.  Returns the RG8 edge texel (0 where the shader discards).
template <typename Luma>
AA_HD uint32_t smaa_edges_pixel(const Luma &t, int x, int y, float threshold)
{
	const float L = t.luma(x, y), Lleft = t.luma(x - 1, y), Ltop = t.luma(x, y - 1);
	const float dxy_x = fabsf(L - Lleft), dxy_y = fabsf(L - Ltop);
	float ex = dxy_x >= threshold ? 1.0f : 0.0f, ey = dxy_y >= threshold ? 1.0f : 0.0f;
	if (ex + ey == 0.0f)
		return 0u;
	const float Lright = t.luma(x + 1, y), Lbottom = t.luma(x, y + 1);
	float dzw_x = fabsf(L - Lright), dzw_y = fabsf(L - Lbottom);
	float max_x = fmaxf(dxy_x, dzw_x), max_y = fmaxf(dxy_y, dzw_y);
	const float Lleftleft = t.luma(x - 2, y), Ltoptop = t.luma(x, y - 2);
	dzw_x = fabsf(Lleft - Lleftleft);
	dzw_y = fabsf(Ltop - Ltoptop);
	max_x = fmaxf(max_x, dzw_x);
	max_y = fmaxf(max_y, dzw_y);
	const float finalDelta = fmaxf(max_x, max_y);
	ex *= (2.0f * dxy_x >= finalDelta) ? 1.0f : 0.0f;
	ey *= (2.0f * dxy_y >= finalDelta) ? 1.0f : 0.0f;
	return unorm8_encode(ex) | (unorm8_encode(ey) << 8);
}

// ---- SMAA neighbourhood blending (SMAA.hlsl:1252-1308) -----------------------------------------------------------------------
// Weights of the pixel and of its right / bottom neighbours are texel fetches; a pixel without weights copies its colour
// texel.  Colour: uint32_t raw(int x, int y) of the clamped texel.  Only pixels with weights reach the bilinear taps.
template <typename Color>
AA_HD f4 smaa_blend_sample(const Color &c, float u, float v, float fw, float fh)
{
	int ix, iy;
	float a, b;
	linear_axis(u * fw - 0.5f, ix, a);
	linear_axis(v * fh - 0.5f, iy, b);
	auto dec = [&](int tx, int ty) {
		const uint32_t t = c.raw(tx, ty);
		return f4{unorm8_decode(t & 255u), unorm8_decode((t >> 8) & 255u), unorm8_decode((t >> 16) & 255u), unorm8_decode(t >> 24)};
	};
	const f4 t00 = dec(ix, iy);
	f4 top = t00;
	if (a != 0.0f)
	{
		const f4 t10 = dec(ix + 1, iy);
		const float oma = 1.0f - a;
		top = {t00.x * oma + t10.x * a, t00.y * oma + t10.y * a, t00.z * oma + t10.z * a, t00.w * oma + t10.w * a};
	}
	if (b == 0.0f)
		return top;
	const f4 t01 = dec(ix, iy + 1);
	f4 bot = t01;
	if (a != 0.0f)
	{
		const f4 t11 = dec(ix + 1, iy + 1);
		const float oma = 1.0f - a;
		bot = {t01.x * oma + t11.x * a, t01.y * oma + t11.y * a, t01.z * oma + t11.z * a, t01.w * oma + t11.w * a};
	}
	const float omb = 1.0f - b;
	return {top.x * omb + bot.x * b, top.y * omb + bot.y * b, top.z * omb + bot.z * b, top.w * omb + bot.w * b};
}

// w_c / w_r / w_b: raw weight texels of the pixel, its right and its bottom neighbour (clamped).
template <typename Color>
AA_HD uint32_t smaa_blend_pixel(const Color &c, uint32_t w_c, uint32_t w_r, uint32_t w_b, int x, int y, float rt_x, float rt_y, float fw, float fh)
{
	// a = (right.a, bottom.g, this.b, this.r) -- SMAA.hlsl:1262-1265
	if (((w_r >> 24) | ((w_b >> 8) & 255u) | ((w_c >> 16) & 255u) | (w_c & 255u)) == 0u)
		return c.raw(x, y);
	const float ax = unorm8_decode(w_r >> 24), ay = unorm8_decode((w_b >> 8) & 255u);
	const float az = unorm8_decode((w_c >> 16) & 255u), aw = unorm8_decode(w_c & 255u);
	const float tx = (float(x) + 0.5f) * rt_x, ty = (float(y) + 0.5f) * rt_y;
	const bool hz = fmaxf(ax, az) > fmaxf(ay, aw);
	float ox = 0.0f, oy = ay, oz = 0.0f, ow = aw;
	float bwx = ay, bwy = aw;
	if (hz)
	{
		ox = ax;
		oy = 0.0f;
		oz = az;
		ow = 0.0f;
		bwx = ax;
		bwy = az;
	}
	const float sum = bwx + bwy;
	bwx = bwx / sum;
	bwy = bwy / sum;
	const f4 s0 = smaa_blend_sample(c, fmaf(ox, rt_x, tx), fmaf(oy, rt_y, ty), fw, fh);
	const f4 s1 = smaa_blend_sample(c, fmaf(oz, -rt_x, tx), fmaf(ow, -rt_y, ty), fw, fh);
	f4 r = {bwx * s0.x, bwx * s0.y, bwx * s0.z, bwx * s0.w};
	r = {r.x + bwy * s1.x, r.y + bwy * s1.y, r.z + bwy * s1.z, r.w + bwy * s1.w};
	return unorm8_encode(r.x) | (unorm8_encode(r.y) << 8) | (unorm8_encode(r.z) << 16) | (unorm8_encode(r.w) << 24);
}
} // namespace aa
