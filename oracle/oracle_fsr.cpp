// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  The reference has no test or golden data for these passes; the
// EASU paths (fp32 and fp16) and RCAS are pinned by executing the reference's own upscale.frag / sharpen.frag / FsrEasuH with
// the vendored FidelityFX headers on the CPU (oracle/ref_build/ref_fsr.cpp, tests/test_reference_shaders_cpu.py, bit for bit).
//
// Spatial upscaling after the post chain: setup_after_post_chain_upscaling (renderer/post/aa.cpp:75-174) with
// assets/shaders/post/ffx-fsr/{upscale,sharpen}.{vert,frag}.  The arithmetic is AMD FidelityFX FSR 1.0 (EASU + RCAS), a
// third-party header vendored by the reference (assets/shaders/post/ffx-fsr/ffx_fsr1.h, helpers in ffx-a/ffx_a.h); what
// follows restates the published algorithm, it is not that header.
//
//   * upscale.frag gathers with NearestClamp from the UNORM alias of the input: the 12 taps are the stored bytes / 255 of
//     texels at fixed offsets from floor(pp), clamped to the image (textureGather at (fp + 1, fp - 1) / size etc. lands on
//     texel corners: ffx_fsr1.h:156-203,347-367).
//   * FP16 = 1 is what the reference selects on a device with shaderFloat16 (aa.cpp:118-119): two taps per packed half2
//     operation.  Every half operation is restated as "compute in fp32, round to nearest even": exact for + - * and, with
//     24 >= 2 * 11 + 2 significand bits, for division.  clamp() of a NaN (0 * inf on flat areas, where the half path divides
//     by a zero gradient) is taken as 0: IEEE maxNum(NaN, 0).
//   * min / max / clamp are IEEE minNum / maxNum throughout (fminf / fmaxf): a number wins over a NaN, as on the hardware
//     the reference targets; this decides black pixels in RCAS (0 * inf) and flat areas in the half EASU path.
//   * Constants written as expressions in the header fold in double precision (glslang) before conversion.
#include "oracle_common.h"

using namespace orc;

namespace
{
// ---- fp32 / fp16 bit tricks the algorithm is built on (ffx_a.h:1808-1821,1843-1845) --------------------------------------
inline float fast_rcp32(float a) { return u2f(0x7ef07ebbu - f2u(a)); }
inline float fast_rsq32(float a) { return u2f(0x5f347d74u - (f2u(a) >> 1)); }
inline float medium_rcp32(float a)
{
	const float b = u2f(0x7ef19fffu - f2u(a));
	return b * (-b * a + 2.0f);
}
inline float sat32(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// A half value: stored as the fp32 number it denotes; every operation rounds its fp32 result back to half.
struct h16
{
	float v;
};
inline h16 H(float f) { return {half_to_float(float_to_half_rne(f))}; }
inline h16 operator+(h16 a, h16 b) { return H(a.v + b.v); }
inline h16 operator-(h16 a, h16 b) { return H(a.v - b.v); }
inline h16 operator*(h16 a, h16 b) { return H(a.v * b.v); }
inline h16 operator/(h16 a, h16 b) { return H(a.v / b.v); }
inline h16 operator-(h16 a) { return {-a.v}; }
inline h16 habs(h16 a) { return {fabsf(a.v)}; }
// maxNum / minNum: the operand that is a number wins over a NaN
inline h16 hmax(h16 a, h16 b) { return {fmaxf(a.v, b.v)}; }
inline h16 hmin(h16 a, h16 b) { return {fminf(a.v, b.v)}; }
inline h16 hsat(h16 a) { return hmin(hmax(a, {0.0f}), {1.0f}); }
inline uint16_t hbits(h16 a) { return float_to_half_rne(a.v); }
inline h16 hfrom(uint16_t b) { return {half_to_float(b)}; }
inline h16 fast_rcp16(h16 a) { return hfrom(uint16_t(0x7784u - hbits(a))); }
inline h16 fast_rsq16(h16 a) { return hfrom(uint16_t(0x59a3u - (hbits(a) >> 1))); }

struct Rgba8
{
	const uint8_t *data;
	int w, h;
	const uint8_t *at(int x, int y) const { return data + (size_t(clampi(y, 0, h - 1)) * w + clampi(x, 0, w - 1)) * 4; }
};

// The 12-tap footprint around f = floor(pp):      b c
//                                               e f g h
//                                               i j k l
//                                                 n o
enum Tap { B, C, E, F, G, HH, I, J, K, L, N, O, TapCount };
const int tap_dx[TapCount] = {0, 1, -1, 0, 1, 2, -1, 0, 1, 2, 0, 1};
const int tap_dy[TapCount] = {-1, -1, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2};

struct EasuConstants
{
	float scale_x, scale_y, offset_x, offset_y; // output pixel -> input position of the resolve point
};

EasuConstants easu_constants(int iw, int ih, int ow, int oh)
{
	// FsrEasuCon (ffx_fsr1.h:156-175) as aa.cpp:106-108 calls it: viewport = input size.
	EasuConstants c;
	c.scale_x = float(iw) * (1.0f / float(ow));
	c.scale_y = float(ih) * (1.0f / float(oh));
	c.offset_x = 0.5f * float(iw) * (1.0f / float(ow)) - 0.5f;
	c.offset_y = 0.5f * float(ih) * (1.0f / float(oh)) - 0.5f;
	return c;
}

void store_pixel(uint8_t *out, int ow, int x, int y, const float rgb[3], bool target_srgb)
{
	uint8_t *p = out + (size_t(y) * ow + x) * 4;
	for (int c = 0; c < 3; c++)
		p[c] = target_srgb ? float_to_srgb8(srgb_decode(rgb[c])) : float_to_unorm8(rgb[c]); // upscale.frag:43-45 + attachment store
	p[3] = 255;
}

// ---- EASU, fp32 (ffx_fsr1.h:239-437) ---------------------------------------------------------------------------------------
void easu_pixel_f32(const Rgba8 &in, const EasuConstants &con, int x, int y, float rgb[3])
{
	float px = float(x) * con.scale_x + con.offset_x;
	float py = float(y) * con.scale_y + con.offset_y;
	const float fx = floorf(px), fy = floorf(py);
	px -= fx;
	py -= fy;

	float col[TapCount][3], luma[TapCount];
	for (int t = 0; t < TapCount; t++)
	{
		const uint8_t *p = in.at(int(fx) + tap_dx[t], int(fy) + tap_dy[t]);
		for (int c = 0; c < 3; c++)
			col[t][c] = float(p[c]) / 255.0f;
		luma[t] = col[t][2] * 0.5f + (col[t][0] * 0.5f + col[t][1]); // twice an approximate luma
	}

	// Edge direction and "edginess" at the four texels around the resolve point, blended bilinearly.
	float dir_x = 0.0f, dir_y = 0.0f, len = 0.0f;
	auto analyse = [&](float weight, Tap up, Tap left, Tap centre, Tap right, Tap down) {
		const float dc = luma[right] - luma[centre], cb = luma[centre] - luma[left];
		float lx = fast_rcp32(fmaxf(fabsf(dc), fabsf(cb)));
		const float gx = luma[right] - luma[left];
		dir_x += gx * weight;
		lx = sat32(fabsf(gx) * lx);
		lx *= lx;
		len += lx * weight;
		const float ec = luma[down] - luma[centre], ca = luma[centre] - luma[up];
		float ly = fast_rcp32(fmaxf(fabsf(ec), fabsf(ca)));
		const float gy = luma[down] - luma[up];
		dir_y += gy * weight;
		ly = sat32(fabsf(gy) * ly);
		ly *= ly;
		len += ly * weight;
	};
	analyse((1.0f - px) * (1.0f - py), B, E, F, G, J);
	analyse(px * (1.0f - py), C, F, G, HH, K);
	analyse((1.0f - px) * py, F, I, J, K, N);
	analyse(px * py, G, J, K, L, O);

	float norm = dir_x * dir_x + dir_y * dir_y;
	const bool flat = norm < float(1.0 / 32768.0);
	norm = fast_rsq32(norm);
	if (flat)
	{
		norm = 1.0f;
		dir_x = 1.0f;
	}
	dir_x *= norm;
	dir_y *= norm;
	len = len * 0.5f;
	len *= len;
	const float stretch = (dir_x * dir_x + dir_y * dir_y) * fast_rcp32(fmaxf(fabsf(dir_x), fabsf(dir_y)));
	const float len_x = 1.0f + (stretch - 1.0f) * len, len_y = 1.0f + -0.5f * len;
	const float lobe = 0.5f + float((1.0 / 4.0 - 0.04) - 0.5) * len;
	const float clip = fast_rcp32(lobe);

	float acc[3] = {0.0f, 0.0f, 0.0f}, acc_w = 0.0f;
	static const Tap order[TapCount] = {B, C, I, J, F, E, K, L, HH, G, O, N};
	for (Tap t : order)
	{
		const float ox = float(tap_dx[t]) - px, oy = float(tap_dy[t]) - py;
		float vx = ox * dir_x + oy * dir_y;
		float vy = ox * (-dir_y) + oy * dir_x;
		vx *= len_x;
		vy *= len_y;
		const float d2 = fminf(vx * vx + vy * vy, clip);
		float base = float(2.0 / 5.0) * d2 + -1.0f;
		float window = lobe * d2 + -1.0f;
		base *= base;
		window *= window;
		base = float(25.0 / 16.0) * base + float(-(25.0 / 16.0 - 1.0));
		const float w = base * window;
		for (int c = 0; c < 3; c++)
			acc[c] += col[t][c] * w;
		acc_w += w;
	}
	const float inv_w = 1.0f / acc_w;
	for (int c = 0; c < 3; c++)
	{
		const float lo = fminf(fminf(col[F][c], fminf(col[G][c], col[J][c])), col[K][c]);
		const float hi = fmaxf(fmaxf(col[F][c], fmaxf(col[G][c], col[J][c])), col[K][c]);
		rgb[c] = fminf(hi, fmaxf(lo, acc[c] * inv_w)); // de-ring against the 2x2 around the resolve point
	}
}

// ---- EASU, packed fp16 (ffx_fsr1.h:445-590): same algorithm, two taps per operation, true half division for the
// gradient reciprocal, pairwise partial sums folded at the end -------------------------------------------------------------
struct h16x2
{
	h16 a, b;
};

void easu_pixel_f16(const Rgba8 &in, const EasuConstants &con, int x, int y, float rgb[3])
{
	float fpx = float(x) * con.scale_x + con.offset_x;
	float fpy = float(y) * con.scale_y + con.offset_y;
	const float fx = floorf(fpx), fy = floorf(fpy);
	fpx -= fx;
	fpy -= fy;
	const h16 px = H(fpx), py = H(fpy);
	const h16 one = {1.0f}, zero = {0.0f}, half = {0.5f};

	h16 col[TapCount][3], luma[TapCount];
	for (int t = 0; t < TapCount; t++)
	{
		const uint8_t *p = in.at(int(fx) + tap_dx[t], int(fy) + tap_dy[t]);
		for (int c = 0; c < 3; c++)
			col[t][c] = H(float(p[c]) / 255.0f);
		luma[t] = col[t][2] * half + (col[t][0] * half + col[t][1]);
	}

	h16x2 dir_x = {zero, zero}, dir_y = {zero, zero}, len2 = {zero, zero};
	auto analyse = [&](h16 row_weight, const Tap up[2], const Tap left[2], const Tap centre[2], const Tap right[2], const Tap down[2]) {
		const h16 w[2] = {(one + (-px)) * row_weight, (zero + px) * row_weight};
		h16 *dx[2] = {&dir_x.a, &dir_x.b}, *dy[2] = {&dir_y.a, &dir_y.b}, *ln[2] = {&len2.a, &len2.b};
		for (int k = 0; k < 2; k++)
		{
			const h16 dc = luma[right[k]] - luma[centre[k]], cb = luma[centre[k]] - luma[left[k]];
			h16 lx = one / hmax(habs(dc), habs(cb));
			const h16 gx = luma[right[k]] - luma[left[k]];
			*dx[k] = *dx[k] + gx * w[k];
			lx = hsat(habs(gx) * lx);
			lx = lx * lx;
			*ln[k] = *ln[k] + lx * w[k];
			const h16 ec = luma[down[k]] - luma[centre[k]], ca = luma[centre[k]] - luma[up[k]];
			h16 ly = one / hmax(habs(ec), habs(ca));
			const h16 gy = luma[down[k]] - luma[up[k]];
			*dy[k] = *dy[k] + gy * w[k];
			ly = hsat(habs(gy) * ly);
			ly = ly * ly;
			*ln[k] = *ln[k] + ly * w[k];
		}
	};
	{
		const Tap up[2] = {B, C}, left[2] = {E, F}, centre[2] = {F, G}, right[2] = {G, HH}, down[2] = {J, K};
		analyse(one - py, up, left, centre, right, down);
	}
	{
		const Tap up[2] = {F, G}, left[2] = {I, J}, centre[2] = {J, K}, right[2] = {K, L}, down[2] = {N, O};
		analyse(py, up, left, centre, right, down);
	}
	h16 dx = dir_x.a + dir_x.b, dy = dir_y.a + dir_y.b;
	h16 len = len2.a + len2.b;

	h16 norm = dx * dx + dy * dy;
	const bool flat = norm.v < float(1.0 / 32768.0);
	norm = fast_rsq16(norm);
	if (flat)
	{
		norm = one;
		dx = one;
	}
	dx = dx * norm;
	dy = dy * norm;
	len = len * half;
	len = len * len;
	const h16 stretch = (dx * dx + dy * dy) * fast_rcp16(hmax(habs(dx), habs(dy)));
	const h16 len_x = one + (stretch - one) * len, len_y = one + H(-0.5f) * len;
	const h16 lobe = half + H(float((1.0 / 4.0 - 0.04) - 0.5)) * len;
	const h16 clip = fast_rcp16(lobe);

	h16x2 acc[3] = {{zero, zero}, {zero, zero}, {zero, zero}}, acc_w = {zero, zero};
	static const Tap pairs[6][2] = {{B, C}, {I, J}, {F, E}, {K, L}, {HH, G}, {O, N}};
	const h16 k_base = H(float(2.0 / 5.0)), k_a = H(float(25.0 / 16.0)), k_b = H(float(-(25.0 / 16.0 - 1.0))), minus_one = {-1.0f};
	for (auto &pair : pairs)
	{
		h16 *aw[2] = {&acc_w.a, &acc_w.b};
		for (int k = 0; k < 2; k++)
		{
			const Tap t = pair[k];
			const h16 ox = H(float(tap_dx[t])) - px, oy = H(float(tap_dy[t])) - py;
			h16 vx = ox * dx + oy * dy;
			h16 vy = ox * (-dy) + oy * dx;
			vx = vx * len_x;
			vy = vy * len_y;
			const h16 d2 = hmin(vx * vx + vy * vy, clip);
			h16 base = k_base * d2 + minus_one;
			h16 window = lobe * d2 + minus_one;
			base = base * base;
			window = window * window;
			base = k_a * base + k_b;
			const h16 w = base * window;
			for (int c = 0; c < 3; c++)
			{
				h16 &slot = k ? acc[c].b : acc[c].a;
				slot = slot + col[t][c] * w;
			}
			*aw[k] = *aw[k] + w;
		}
	}
	const h16 total_w = acc_w.a + acc_w.b;
	const h16 inv_w = one / total_w;
	for (int c = 0; c < 3; c++)
	{
		const h16 sum = acc[c].a + acc[c].b;
		const h16 lo = hmin(hmin(col[F][c], col[G][c]), hmin(col[J][c], col[K][c]));
		const h16 hi = hmax(hmax(col[F][c], col[G][c]), hmax(col[J][c], col[K][c]));
		rgb[c] = hmin(hi, hmax(lo, sum * inv_w)).v;
	}
}
} // namespace

extern "C" {

// {scale_x, scale_y, offset_x, offset_y} as the upscale pass uses them (checked against the vendored FsrEasuCon in CPU mode,
// tests/test_reference_math_cpu.py).
void orc_fsr_easu_constants(int iw, int ih, int ow, int oh, float *out4)
{
	const EasuConstants c = easu_constants(iw, ih, ow, oh);
	out4[0] = c.scale_x, out4[1] = c.scale_y, out4[2] = c.offset_x, out4[3] = c.offset_y;
}

// upscale pass.  in: RGBA8 bytes (gamma space, iw x ih); out: RGBA8 ow x oh.  fp16: the FP16 shader variant.
// target_srgb: the output attachment is *_SRGB (upscale.frag:43-45 decodes, the store re-encodes).
void orc_fsr_easu(const uint8_t *in, int iw, int ih, uint8_t *out, int ow, int oh, int fp16, int target_srgb)
{
	const Rgba8 src{in, iw, ih};
	const EasuConstants con = easu_constants(iw, ih, ow, oh);
#pragma omp parallel for schedule(static)
	for (int y = 0; y < oh; y++)
		for (int x = 0; x < ow; x++)
		{
			float rgb[3];
			if (fp16)
				easu_pixel_f16(src, con, x, y, rgb);
			else
				easu_pixel_f32(src, con, x, y, rgb);
			store_pixel(out, ow, x, y, rgb, target_srgb != 0);
		}
}

// sharpen pass (sharpen.frag + FsrRcasF, ffx_fsr1.h:682-760; FSR_RCAS_DENOISE off, no alpha pass-through).
// sharpness = the linear value FsrRcasCon stores, exp2(-stops) (aa.cpp:64-74,157 with 0.5 stops).  srgb: the output is *_SRGB, so the
// input is read through its sRGB view (decode on load, aa.cpp:147-151) and the store encodes.
void orc_fsr_rcas(const uint8_t *in, int w, int h, uint8_t *out, float sharpness, int srgb)
{
	const Rgba8 src{in, w, h};
#pragma omp parallel for schedule(static)
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			float tap[5][3]; // up, left, centre, right, down; coordinates clamp to the image (sharpen.frag:17)
			static const int dx[5] = {0, -1, 0, 1, 0}, dy[5] = {-1, 0, 0, 0, 1};
			for (int t = 0; t < 5; t++)
			{
				const uint8_t *p = src.at(x + dx[t], y + dy[t]);
				for (int c = 0; c < 3; c++)
					tap[t][c] = srgb ? srgb8_to_float(p[c]) : float(p[c]) / 255.0f;
			}
			// Per channel, the strongest negative lobe that keeps the result inside [0, 1] given the ring's extremes.
			float lobe_c[3];
			for (int c = 0; c < 3; c++)
			{
				const float ring_min = fminf(fminf(tap[0][c], fminf(tap[1][c], tap[3][c])), tap[4][c]);
				const float ring_max = fmaxf(fmaxf(tap[0][c], fmaxf(tap[1][c], tap[3][c])), tap[4][c]);
				const float hit_min = ring_min * (1.0f / (4.0f * ring_max));
				const float hit_max = (1.0f - ring_max) * (1.0f / (4.0f * ring_min + -4.0f));
				lobe_c[c] = fmaxf(-hit_min, hit_max);
			}
			const float limit = float(0.25 - (1.0 / 16.0));
			const float widest = fmaxf(fmaxf(lobe_c[0], lobe_c[1]), lobe_c[2]);
			const float lobe = fmaxf(-limit, fminf(widest, 0.0f)) * sharpness;
			const float inv = medium_rcp32(4.0f * lobe + 1.0f);
			float rgb[3];
			for (int c = 0; c < 3; c++)
				rgb[c] = (lobe * tap[0][c] + lobe * tap[1][c] + lobe * tap[4][c] + lobe * tap[3][c] + tap[2][c]) * inv;
			uint8_t *p = out + (size_t(y) * w + x) * 4;
			for (int c = 0; c < 3; c++)
				p[c] = srgb ? float_to_srgb8(rgb[c]) : float_to_unorm8(rgb[c]);
			p[3] = 255;
		}
}

} // extern "C"
