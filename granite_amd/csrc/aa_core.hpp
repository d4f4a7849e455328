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
#include <string.h>

#if defined(__HIPCC__)
#define AA_HD __host__ __device__ __forceinline__
#else
#define AA_HD inline
#endif
// A value every lane of the wave holds alike (a wave index, a workgroup-wide count read from LDS): on the device it is moved to a
// scalar register, so what is computed from it runs on the scalar unit.
#if defined(__HIP_DEVICE_COMPILE__)
#define AA_WAVE_UNIFORM(v) uint32_t(__builtin_amdgcn_readfirstlane(int(v)))
#else
#define AA_WAVE_UNIFORM(v) uint32_t(v)
#endif
// True when the condition holds on every lane of the wave (device: one branch for the wave; host emulation: the lane's own answer --
// both sides of such a branch compute the same values, the condition only selects the cheaper form).
#if defined(__HIP_DEVICE_COMPILE__)
#define AA_WAVE_ALL(c) (__builtin_amdgcn_ballot_w64(!(c)) == 0ull)
#else
#define AA_WAVE_ALL(c) (c)
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

// The diagonal searches of SMAA (SMAA.hlsl:831-868) walk a coordinate by one texel per step from a pixel centre -- in x from the
// centre itself or from a quarter of a texel beside it (SMAASearchDiag2: coord.x += 0.25 * rt.x) -- with one fma per step.  For an
// axis of n texels: does step k = 1 .. steps of either direction, from every pixel p, resolve to texel p +- k with the weight the
// search's decode expects (0: the texel itself; about 0.25: SMAADecodeDiagBilinearAccess rounds the pair to (R of the next texel,
// G of this one) for any weight in [0.2, 0.3])?  Then the walk can be done on integer texel positions.
inline bool axis_walk_exact(int n, float inv, int steps, bool quarter)
{
	for (int p = 0; p < n; p++)
		for (int dir = -1; dir <= 1; dir += 2)
		{
			float c = (float(p) + 0.5f) * inv;
			if (quarter)
				c += 0.25f * inv;
			for (int k = 1; k <= steps; k++)
			{
				c = fmaf(inv, float(dir), c);
				int i0;
				float a;
				linear_axis(c * float(n) - 0.5f, i0, a);
				if (i0 != p + dir * k || (quarter ? fabsf(a - 0.25f) > 0.05f : a != 0.0f))
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

// ---- fp16 storage helpers ------------------------------------------------------------------------------------------------------
// Two halves in a dword, as RGBA16F / RG16F texels hold them.  On the device the conversion rides in v_fma_mix_f32 (exact
// conversion + one fused multiply-add); the host form states the same arithmetic.
#if defined(__HIP_DEVICE_COMPILE__)
AA_HD float half_lo(uint32_t p) { return float(__builtin_bit_cast(_Float16, uint16_t(p & 0xffffu))); }
AA_HD float half_hi(uint32_t p) { return float(__builtin_bit_cast(_Float16, uint16_t(p >> 16))); }
AA_HD float mad_half_lo(uint32_t p, float w, float acc)
{
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(w), "v"(acc));
	return r;
}
AA_HD float mad_half_hi(uint32_t p, float w, float acc)
{
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(w), "v"(acc));
	return r;
}
AA_HD uint32_t pack_half2_rne(float lo, float hi)
{
	return uint32_t(__builtin_bit_cast(uint16_t, _Float16(lo))) | (uint32_t(__builtin_bit_cast(uint16_t, _Float16(hi))) << 16);
}
AA_HD float approx_rcp(float v) { return __builtin_amdgcn_rcpf(v); }   // 1 ulp; used where the result is stored as fp16
AA_HD float approx_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); } // 1 ulp
#else
inline float half_bits_to_float(uint32_t h)
{
	const uint32_t s = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
	uint32_t u;
	if (e == 0)
	{
		const float v = float(m) * 5.9604644775390625e-08f;
		return s ? -v : v;
	}
	u = e == 31 ? (s | 0x7f800000u | (m << 13)) : (s | ((e + 112u) << 23) | (m << 13));
	float f;
	memcpy(&f, &u, 4);
	return f;
}
inline uint32_t float_to_half_bits_rne(float f)
{
	uint32_t u;
	memcpy(&u, &f, 4);
	const uint32_t s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
	if (a >= 0x7f800000u)
		return s | 0x7c00u | ((a > 0x7f800000u) ? (0x200u | ((a >> 13) & 0x3ffu)) : 0u);
	if (a >= 0x477ff000u)
		return s | 0x7c00u;
	if (a < 0x38800000u)
	{
		if (a < 0x33000000u)
			return s;
		const uint32_t e = a >> 23, m = (a & 0x7fffffu) | 0x800000u, shift = 126u - e;
		uint32_t q = m >> shift;
		const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1u);
		if (rem > half || (rem == half && (q & 1u)))
			q++;
		return s | q;
	}
	uint32_t q = (((a >> 23) - 112u) << 10) | ((a & 0x7fffffu) >> 13);
	const uint32_t rem = a & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (q & 1u)))
		q++;
	return s | q;
}
inline float half_lo(uint32_t p) { return half_bits_to_float(p & 0xffffu); }
inline float half_hi(uint32_t p) { return half_bits_to_float(p >> 16); }
inline float mad_half_lo(uint32_t p, float w, float acc) { return fmaf(half_lo(p), w, acc); }
inline float mad_half_hi(uint32_t p, float w, float acc) { return fmaf(half_hi(p), w, acc); }
inline uint32_t pack_half2_rne(float lo, float hi) { return float_to_half_bits_rne(lo) | (float_to_half_bits_rne(hi) << 16); }
inline float approx_rcp(float v) { return 1.0f / v; }
inline float approx_sqrt(float v) { return sqrtf(v); }
#endif

struct u2
{
	uint32_t x, y;
};

// ---- TAA resolve (taa_resolve.frag, reprojection.h, reprojection_color_space.h) ---------------------------------------------------
// The resolve's result is stored as fp16 and compared at 3 ulp fp16: divisions and square roots on continuous paths use the
// hardware reciprocal / root (1 ulp fp32), products of tap weights are formed first (w_x * w_y) and folded into one fma per
// channel -- a few fp32 ulps.  What takes decisions (texel selection, the neighbourhood's nearest depth) is exact.
AA_HD f3 taa_from_hdr(float r, float g, float b)
{
	r *= 8.0f;
	g *= 8.0f;
	b *= 8.0f;
	const float s = approx_rcp(fmaxf(r, fmaxf(g, b)) + 1.0f);
	r *= s;
	g *= s;
	b *= s;
	return {0.25f * r + 0.5f * g + 0.25f * b, 0.5f * g - 0.25f * r - 0.25f * b, 0.5f * r - 0.5f * b};
}
AA_HD f3 taa_to_hdr(f3 c)
{
	const float tmp = c.x - c.y;
	const float r = fminf(fmaxf(tmp + c.z, 0.0f), 0.999f), g = fminf(fmaxf(c.x + c.y, 0.0f), 0.999f), b = fminf(fmaxf(tmp - c.z, 0.0f), 0.999f);
	const float s = approx_rcp(1.0f - fmaxf(r, fmaxf(g, b)));
	return {(0.125f * r) * s, (0.125f * g) * s, (0.125f * b) * s};
}

struct TaaPush
{
	float reproj[16];
	float rt[4]; // 1/w, 1/h, w, h
};

// Tile: f4 cur(int ox, int oy) = (Y, Cg, Co, depth) of the clamped neighbour (x + ox, y + oy), |o| <= 1.
// Mv:   uint32_t mv(int x, int y), the RG16F texel of the clamped pixel.
// Hist: u2 texel(int x, int y), the RGBA16F history texel; coordinates arrive clamped to the image.  row4(x, y, out): texels (x .. x + 3, y), all
// inside the image (QUALITY 2: the footprint of a wave that touches no border).
// QUALITY 0 / 1 / 2 = TAAQuality Low / Medium / High.  Writes the resolved colour and the new history as RGBA16F texels, and the
// colour in fp32 as well; hist_row_first / hist_row_last = the first and last history row the pixel fetched (row bands: a rank
// only holds the history rows around its band).
template <int QUALITY, typename Tile, typename Mv, typename Hist>
AA_HD void taa_pixel(const Tile &t, const Mv &mvs, const Hist &hist, int x, int y, int w, int h, const TaaPush &P, u2 &out_color, u2 &out_history,
                     f3 &out_color_f32, int &hist_row_first, int &hist_row_last)
{
	const float u = (float(x) + 0.5f) * P.rt[0], v = (float(y) + 0.5f) * P.rt[1];
	const f4 c11 = t.cur(0, 0);
	// sample_nearest_velocity (reprojection.h:213-283): the neighbour nearest to the camera, first wins on ties
	int mx, my;
	float d;
#define TAA_CONSIDER(OX, OY)                  \
	{                                         \
		const float dd = t.cur(OX, OY).w;     \
		if (dd > d)                           \
		{                                     \
			mx = OX;                          \
			my = OY;                          \
			d = dd;                           \
		}                                     \
	}
	if (QUALITY <= 1)
	{
		mx = -1, my = 0, d = t.cur(-1, 0).w;
		if (c11.w > d)
			mx = 0, my = 0, d = c11.w;
		TAA_CONSIDER(0, -1)
		TAA_CONSIDER(0, 1)
		TAA_CONSIDER(1, 0)
	}
	else
	{
		mx = 1, my = 1, d = t.cur(1, 1).w;
		TAA_CONSIDER(-1, 0)
		if (c11.w > d)
			mx = 0, my = 0, d = c11.w;
		TAA_CONSIDER(0, -1)
		TAA_CONSIDER(-1, -1)
		TAA_CONSIDER(1, 0)
		TAA_CONSIDER(1, -1)
		TAA_CONSIDER(-1, 1)
		TAA_CONSIDER(0, 1)
	}
#undef TAA_CONSIDER
	const uint32_t m = mvs.mv(x + mx, y + my);
	float mvx = half_lo(m), mvy = half_hi(m);
	float ou, ov;
	if (mvx == 0.0f && mvy == 0.0f)
	{
		const float cx = 2.0f * u - 1.0f, cy = 2.0f * v - 1.0f;
		// rows x, y, w of reproj * (cx, cy, d, 1), summed column by column like the shader's mat4 * vec4
		float rx = P.reproj[0] * cx, ry = P.reproj[1] * cx, rw = P.reproj[3] * cx;
		rx = rx + P.reproj[4] * cy, ry = ry + P.reproj[5] * cy, rw = rw + P.reproj[7] * cy;
		rx = rx + P.reproj[8] * d, ry = ry + P.reproj[9] * d, rw = rw + P.reproj[11] * d;
		rx = rx + P.reproj[12] * 1.0f, ry = ry + P.reproj[13] * 1.0f, rw = rw + P.reproj[15] * 1.0f;
		ou = rx / rw; // exact: the history position selects texels
		ov = ry / rw;
		mvx = u - ou;
		mvy = v - ov;
	}
	else
	{
		ou = u - mvx;
		ov = v - mvy;
	}

	f3 hc;
	if (QUALITY == 2)
	{
		// sample_catmull_rom (reprojection.h:286-334): nine LinearClamp taps on a 3 x 3 grid of positions.  The outer positions
		// sit on texel centres (texels k - 1, k + 2), the middle one between k and k + 1: sixteen texels.
		const float spx = ou * P.rt[2], spy = ov * P.rt[3];
		const float kx = floorf(spx - 0.5f), ky = floorf(spy - 0.5f);
		const float fx = spx - (kx + 0.5f), fy = spy - (ky + 0.5f);
		float wx[3], wy[3], ax, ay;
		int ix, iy;
		{
			const float f = fx;
			const float w0 = f * (-0.5f + f * (1.0f - 0.5f * f)), w1 = 1.0f + f * f * (-2.5f + 1.5f * f);
			const float w2 = f * (0.5f + f * (2.0f - 1.5f * f)), w3 = f * f * (-0.5f + 0.5f * f);
			wx[0] = w0, wx[1] = w1 + w2, wx[2] = w3;
			const float off = w2 / (w1 + w2); // exact: the tap position takes the sampler's snap decision
			linear_axis((((kx + 0.5f) + off) * P.rt[0]) * float(w) - 0.5f, ix, ax);
		}
		{
			const float f = fy;
			const float w0 = f * (-0.5f + f * (1.0f - 0.5f * f)), w1 = 1.0f + f * f * (-2.5f + 1.5f * f);
			const float w2 = f * (0.5f + f * (2.0f - 1.5f * f)), w3 = f * f * (-0.5f + 0.5f * f);
			wy[0] = w0, wy[1] = w1 + w2, wy[2] = w3;
			const float off = w2 / (w1 + w2); // exact: the tap position takes the sampler's snap decision
			linear_axis((((ky + 0.5f) + off) * P.rt[1]) * float(h) - 0.5f, iy, ay);
		}
		const int k = int(kx), j = int(ky);
		// the middle tap's pair is (k, k + 1); a coordinate that resolved onto k + 1 reads it with weight 1
		if (ix != k)
			ax = 1.0f;
		if (iy != j)
			ay = 1.0f;
		int col[4], row[4];
#pragma unroll
		for (int i = 0; i < 4; i++)
		{
			col[i] = clampi(k - 1 + i, 0, w - 1);
			row[i] = clampi(j - 1 + i, 0, h - 1);
		}
		// weight of texel column i / row i: its tap's Catmull-Rom weight times its share of the tap's lerp
		const float cwx[4] = {wx[0], wx[1] * (1.0f - ax), wx[1] * ax, wx[2]}, cwy[4] = {wy[0], wy[1] * (1.0f - ay), wy[1] * ay, wy[2]};
		float r = 0.0f, g = 0.0f, b = 0.0f;
		// Where the 4 x 4 footprint of every lane of the wave lies inside the image -- everywhere but a frame of pixels whose reprojection
		// touches the border -- nothing clamps and the four texels of a row are one run of 32 bytes from one address: Hist::row4.  Same
		// texels, same sums.
		const bool inside = k >= 1 && k + 2 <= w - 1 && j >= 1 && j + 2 <= h - 1;
		if (AA_WAVE_ALL(inside))
		{
#pragma unroll
			for (int jj = 0; jj < 4; jj++)
			{
				u2 run[4];
				hist.row4(k - 1, j - 1 + jj, run);
#pragma unroll
				for (int ii = 0; ii < 4; ii++)
				{
					const float wgt = cwx[ii] * cwy[jj];
					r = mad_half_lo(run[ii].x, wgt, r);
					g = mad_half_hi(run[ii].x, wgt, g);
					b = mad_half_lo(run[ii].y, wgt, b);
				}
			}
		}
		else
		{
#pragma unroll
			for (int jj = 0; jj < 4; jj++)
#pragma unroll
				for (int ii = 0; ii < 4; ii++)
				{
					const u2 tx = hist.texel(col[ii], row[jj]);
					const float wgt = cwx[ii] * cwy[jj];
					r = mad_half_lo(tx.x, wgt, r);
					g = mad_half_hi(tx.x, wgt, g);
					b = mad_half_lo(tx.y, wgt, b);
				}
		}
		hc = {r, g, b};
		hist_row_first = row[0];
		hist_row_last = row[3];
	}
	else
	{
		int ix, iy;
		float a, b;
		linear_axis(ou * float(w) - 0.5f, ix, a);
		linear_axis(ov * float(h) - 0.5f, iy, b);
		const int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1), y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
		const u2 t00 = hist.texel(x0, y0), t10 = hist.texel(x1, y0), t01 = hist.texel(x0, y1), t11 = hist.texel(x1, y1);
		const float oma = 1.0f - a, omb = 1.0f - b;
		const float top_r = mad_half_lo(t10.x, a, mad_half_lo(t00.x, oma, 0.0f)), bot_r = mad_half_lo(t11.x, a, mad_half_lo(t01.x, oma, 0.0f));
		const float top_g = mad_half_hi(t10.x, a, mad_half_hi(t00.x, oma, 0.0f)), bot_g = mad_half_hi(t11.x, a, mad_half_hi(t01.x, oma, 0.0f));
		const float top_b = mad_half_lo(t10.y, a, mad_half_lo(t00.y, oma, 0.0f)), bot_b = mad_half_lo(t11.y, a, mad_half_lo(t01.y, oma, 0.0f));
		hc = {fmaf(bot_r, b, top_r * omb), fmaf(bot_g, b, top_g * omb), fmaf(bot_b, b, top_b * omb)};
		hist_row_first = y0;
		hist_row_last = y1;
	}

	const float mv_length = approx_sqrt(mvx * mvx + mvy * mvy);
	const float mv_fast = fminf(mv_length * 50.0f, 1.0f);
	const float gamma = 1.5f * (1.0f - mv_fast) + 0.5f * mv_fast;
	hc = {fminf(fmaxf(hc.x, 0.0f), 1.0f), fminf(fmaxf(hc.y, -1.0f), 1.0f), fminf(fmaxf(hc.z, -1.0f), 1.0f)};
	const float lerp_factor = (1.0f + 2.0f * mv_fast) * (1.0f / 16.0f);

	// clamp_history_box (reprojection.h:107-183)
	const f4 c01 = t.cur(-1, 0), c21 = t.cur(1, 0), c10 = t.cur(0, -1), c12 = t.cur(0, 1);
	f3 lo, hi;
#define TAA_MIN5(C) fminf(fminf(fminf(fminf(c11.C, c01.C), c21.C), c10.C), c12.C)
#define TAA_MAX5(C) fmaxf(fmaxf(fmaxf(fmaxf(c11.C, c01.C), c21.C), c10.C), c12.C)
	if (QUALITY == 0)
	{
		lo = {TAA_MIN5(x), TAA_MIN5(y), TAA_MIN5(z)};
		hi = {TAA_MAX5(x), TAA_MAX5(y), TAA_MAX5(z)};
	}
	else
	{
		const f4 c00 = t.cur(-1, -1), c22 = t.cur(1, 1), c02 = t.cur(-1, 1), c20 = t.cur(1, -1);
		if (QUALITY == 1)
		{
			const f3 clo = {TAA_MIN5(x), TAA_MIN5(y), TAA_MIN5(z)}, chi = {TAA_MAX5(x), TAA_MAX5(y), TAA_MAX5(z)};
#define TAA_MIN4(C) fminf(fminf(fminf(fminf(clo.C, c00.C), c22.C), c02.C), c20.C)
#define TAA_MAX4(C) fmaxf(fmaxf(fmaxf(fmaxf(chi.C, c00.C), c22.C), c02.C), c20.C)
			lo = {0.5f * (clo.x + TAA_MIN4(x)), 0.5f * (clo.y + TAA_MIN4(y)), 0.5f * (clo.z + TAA_MIN4(z))};
			hi = {0.5f * (chi.x + TAA_MAX4(x)), 0.5f * (chi.y + TAA_MAX4(y)), 0.5f * (chi.z + TAA_MAX4(z))};
#undef TAA_MIN4
#undef TAA_MAX4
		}
		else
		{
			// m1 = (c00 + 2 c01 + c02 + 2 c10 + 4 c11 + 2 c12 + c20 + 2 c21 + c22) / 16 and m2 = the same sum over the squares, in the
			// shader's order.  A product by 2 or 4 is exact, so acc + 2 c is ONE rounding either way: fma(2, c, acc) is the shader's
			// mul + add bit for bit, and 2 * c * c == fma-free 2 * (c * c) likewise (scaling by a power of two commutes with rounding).
#define TAA_M1(C) \
	(fmaf(2.0f, c21.C, fmaf(2.0f, c12.C, fmaf(4.0f, c11.C, fmaf(2.0f, c10.C, fmaf(2.0f, c01.C, c00.C) + c02.C))) + c20.C) + c22.C) * (1.0f / 16.0f)
#define TAA_M2(C)                                                                                                                                      \
	(fmaf(2.0f, c21.C * c21.C, fmaf(2.0f, c12.C * c12.C, fmaf(4.0f, c11.C * c11.C, fmaf(2.0f, c10.C * c10.C, fmaf(2.0f, c01.C * c01.C, c00.C * c00.C) + c02.C * c02.C))) + \
	      c20.C * c20.C) +                                                                                                                             \
	 c22.C * c22.C)
			const f3 m1 = {TAA_M1(x), TAA_M1(y), TAA_M1(z)};
			const f3 m2 = {TAA_M2(x), TAA_M2(y), TAA_M2(z)};
#undef TAA_M1
#undef TAA_M2
			const f3 sigma = {approx_sqrt(fmaxf(m2.x * (1.0f / 16.0f) - m1.x * m1.x, 0.0f)), approx_sqrt(fmaxf(m2.y * (1.0f / 16.0f) - m1.y * m1.y, 0.0f)),
			                  approx_sqrt(fmaxf(m2.z * (1.0f / 16.0f) - m1.z * m1.z, 0.0f))};
			lo = {m1.x - gamma * sigma.x, m1.y - gamma * sigma.y, m1.z - gamma * sigma.z};
			hi = {m1.x + gamma * sigma.x, m1.y + gamma * sigma.y, m1.z + gamma * sigma.z};
		}
	}
#undef TAA_MIN5
#undef TAA_MAX5
	if (QUALITY == 0)
		hc = {fminf(fmaxf(hc.x, lo.x), hi.x), fminf(fmaxf(hc.y, lo.y), hi.y), fminf(fmaxf(hc.z, lo.z), hi.z)};
	else
	{
		const f3 center = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
		const f3 radius = {fmaxf(0.5f * (hi.x - lo.x), 0.0001f), fmaxf(0.5f * (hi.y - lo.y), 0.0001f), fmaxf(0.5f * (hi.z - lo.z), 0.0001f)};
		const f3 dv = {hc.x - center.x, hc.y - center.y, hc.z - center.z};
		const float max_unit = fmaxf(fmaxf(fabsf(dv.x * approx_rcp(radius.x)), fabsf(dv.y * approx_rcp(radius.y))), fabsf(dv.z * approx_rcp(radius.z)));
		if (max_unit > 1.0f)
		{
			const float s = approx_rcp(max_unit);
			hc = {center.x + dv.x * s, center.y + dv.y * s, center.z + dv.z * s};
		}
	}
	const float oml = 1.0f - lerp_factor;
	const f3 mixed = {hc.x * oml + c11.x * lerp_factor, hc.y * oml + c11.y * lerp_factor, hc.z * oml + c11.z * lerp_factor};
	const f3 o = taa_to_hdr(mixed);
	out_color_f32 = o; // for a colour target that is not RGBA16F (B10G11R11: rounded once, from fp32)
	out_color = {pack_half2_rne(o.x, o.y), pack_half2_rne(o.z, 1.0f)};
	out_history = {pack_half2_rne(mixed.x, mixed.y), pack_half2_rne(mixed.z, 1.0f)};
}
} // namespace aa
