// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED by the reference's own tests.
//
// Anti-aliasing passes restated from:
//   assets/shaders/post/fxaa.frag (+ renderer/post/fxaa.cpp:28-55)
//   assets/shaders/post/SMAA.hlsl:304-324,517-523,579-640,689-740,831-1308 with smaa_{edge_detection,blend_weight,
//       neighbor_blend}.{vert,frag}, smaa_common.h (+ renderer/post/smaa.cpp:32-208)
//   assets/shaders/post/taa_resolve.frag, reprojection.h, reprojection_color_space.h (+ renderer/post/temporal.cpp:199-266)
//
// All AA inputs that live in *_SRGB images are read through their UNORM alias (cmd.set_unorm_texture): the stored bytes.
// `mad` is `fma` under SMAA_GLSL_4 (SMAA.hlsl:573), restated with fmaf.
#include "oracle_common.h"

using namespace orc;

namespace
{
// StockSampler::LinearClamp on a UNORM8 image with C channels.  Texel offsets (textureLodOffset) are added to the integer
// texel coordinate after the floor, as the hardware does.
struct Tex8
{
	const uint8_t *data;
	int w, h, ch;
	vec4 fetch(int x, int y) const
	{
		x = clampi(x, 0, w - 1);
		y = clampi(y, 0, h - 1);
		const uint8_t *p = data + (size_t(y) * w + x) * ch;
		vec4 r = V4(0.0f, 0.0f, 0.0f, 1.0f);
		r.x = float(p[0]) / 255.0f;
		if (ch > 1) r.y = float(p[1]) / 255.0f;
		if (ch > 2) r.z = float(p[2]) / 255.0f;
		if (ch > 3) r.w = float(p[3]) / 255.0f;
		return r;
	}
	vec4 sample(vec2 uv, int ox = 0, int oy = 0) const
	{
		float a, b;
		int x0, y0;
		linear_axis(uv.x * float(w) - 0.5f, x0, a);
		linear_axis(uv.y * float(h) - 0.5f, y0, b);
		x0 += ox;
		y0 += oy;
		return linear_combine(fetch(x0, y0), fetch(x0 + 1, y0), fetch(x0, y0 + 1), fetch(x0 + 1, y0 + 1), a, b);
	}
};

static inline vec2 fma2(vec2 a, vec2 b, vec2 c) { return V2(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)); }
static inline float stepf(float edge, float x) { return x >= edge ? 1.0f : 0.0f; }
static inline void store_rgba8(uint8_t *img, int w, int x, int y, uint8_t r, uint8_t g, uint8_t b, uint8_t a)
{
	uint8_t *p = img + (size_t(y) * w + x) * 4;
	p[0] = r; p[1] = g; p[2] = b; p[3] = a;
}
} // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------------------------------
// FXAA (fxaa.frag:20-67).  in: RGBA8 bytes (gamma space).  out: RGBA8; when target_srgb the shader decodes to linear
// and the sRGB attachment store re-encodes.
// ---------------------------------------------------------------------------------------------------------------------
void orc_fxaa(const uint8_t *in, int w, int h, uint8_t *out, int target_srgb)
{
	Tex8 tex{in, w, h, 4};
	const vec2 inv_resolution = V2(1.0f / float(w), 1.0f / float(h));
	const float FXAA_REDUCE_MIN = 1.0f / 128.0f, FXAA_REDUCE_MUL = 1.0f / 8.0f, FXAA_SPAN_MAX = 8.0f;
#pragma omp parallel for schedule(static)
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			vec2 uv = V2((float(x) + 0.5f) * inv_resolution.x, (float(y) + 0.5f) * inv_resolution.y);
			auto rgb = [](vec4 v) { return V3(v.x, v.y, v.z); };
			vec3 rgbNW = rgb(tex.sample(uv, -1, -1)), rgbNE = rgb(tex.sample(uv, +1, -1));
			vec3 rgbSW = rgb(tex.sample(uv, -1, +1)), rgbSE = rgb(tex.sample(uv, +1, +1));
			vec3 texColor = rgb(tex.sample(uv));
			const vec3 luma = V3(0.299f, 0.587f, 0.114f);
			float lumaNW = dot(rgbNW, luma), lumaNE = dot(rgbNE, luma), lumaSW = dot(rgbSW, luma), lumaSE = dot(rgbSE, luma);
			float lumaM = dot(texColor, luma);
			float lumaMin = std::min(lumaM, std::min(std::min(lumaNW, lumaNE), std::min(lumaSW, lumaSE)));
			float lumaMax = std::max(lumaM, std::max(std::max(lumaNW, lumaNE), std::max(lumaSW, lumaSE)));
			vec2 dir;
			dir.x = -((lumaNW + lumaNE) - (lumaSW + lumaSE));
			dir.y = ((lumaNW + lumaSW) - (lumaNE + lumaSE));
			float dirReduce = std::max((lumaNW + lumaNE + lumaSW + lumaSE) * (0.25f * FXAA_REDUCE_MUL), FXAA_REDUCE_MIN);
			float rcpDirMin = 1.0f / (std::min(fabsf(dir.x), fabsf(dir.y)) + dirReduce);
			dir = V2(clampf(dir.x * rcpDirMin, -FXAA_SPAN_MAX, FXAA_SPAN_MAX), clampf(dir.y * rcpDirMin, -FXAA_SPAN_MAX, FXAA_SPAN_MAX)) *
			      inv_resolution;
			vec3 rgbA = 0.5f * (rgb(tex.sample(uv + dir * (1.0f / 3.0f - 0.5f))) + rgb(tex.sample(uv + dir * (2.0f / 3.0f - 0.5f))));
			vec3 rgbB = rgbA * 0.5f + 0.25f * (rgb(tex.sample(uv + dir * -0.5f)) + rgb(tex.sample(uv + dir * 0.5f)));
			float lumaB = dot(rgbB, luma);
			vec3 color = ((lumaB < lumaMin) || (lumaB > lumaMax)) ? rgbA : rgbB;
			if (target_srgb)
				store_rgba8(out, w, x, y, float_to_srgb8(srgb_decode(color.x)), float_to_srgb8(srgb_decode(color.y)),
				            float_to_srgb8(srgb_decode(color.z)), 255);
			else
				store_rgba8(out, w, x, y, float_to_unorm8(color.x), float_to_unorm8(color.y), float_to_unorm8(color.z), 255);
		}
}

// ---------------------------------------------------------------------------------------------------------------------
// SMAA 1x
// ---------------------------------------------------------------------------------------------------------------------
struct SmaaPreset { float threshold; int max_search_steps; int max_search_steps_diag; int corner_rounding; bool diag; bool corner; };
static SmaaPreset smaa_preset(int quality)
{
	switch (quality) // SMAA.hlsl:304-324
	{
	case 0: return {0.15f, 4, 8, 25, false, false};
	case 1: return {0.1f, 8, 8, 25, false, false};
	case 2: return {0.1f, 16, 8, 25, true, true};
	default: return {0.05f, 32, 16, 25, true, true};
	}
}

// smaa_edge_detection.{vert,frag} + SMAALumaEdgeDetectionPS (SMAA.hlsl:689-740).  out: RG8, cleared to 0, discarded
// pixels untouched.
void orc_smaa_edges(const uint8_t *color, int w, int h, uint8_t *edges_rg8, int quality)
{
	Tex8 tex{color, w, h, 4};
	const SmaaPreset P = smaa_preset(quality);
	const vec4 rt = V4(1.0f / float(w), 1.0f / float(h), float(w), float(h));
	memset(edges_rg8, 0, size_t(w) * h * 2);
#pragma omp parallel for schedule(static)
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			vec2 tc = V2((float(x) + 0.5f) * rt.x, (float(y) + 0.5f) * rt.y);
			vec2 rtxy = V2(rt.x, rt.y);
			vec2 o0a = fma2(rtxy, V2(-1.0f, 0.0f), tc), o0b = fma2(rtxy, V2(0.0f, -1.0f), tc);
			vec2 o1a = fma2(rtxy, V2(1.0f, 0.0f), tc), o1b = fma2(rtxy, V2(0.0f, 1.0f), tc);
			vec2 o2a = fma2(rtxy, V2(-2.0f, 0.0f), tc), o2b = fma2(rtxy, V2(0.0f, -2.0f), tc);
			const vec3 weights = V3(0.2126f, 0.7152f, 0.0722f);
			auto luma = [&](vec2 uv) { vec4 c = tex.sample(uv); return dot(V3(c.x, c.y, c.z), weights); };
			float L = luma(tc), Lleft = luma(o0a), Ltop = luma(o0b);
			vec2 delta_xy = V2(fabsf(L - Lleft), fabsf(L - Ltop));
			vec2 edges = V2(stepf(P.threshold, delta_xy.x), stepf(P.threshold, delta_xy.y));
			if (edges.x + edges.y == 0.0f)
				continue; // discard
			float Lright = luma(o1a), Lbottom = luma(o1b);
			vec2 delta_zw = V2(fabsf(L - Lright), fabsf(L - Lbottom));
			vec2 maxDelta = V2(std::max(delta_xy.x, delta_zw.x), std::max(delta_xy.y, delta_zw.y));
			float Lleftleft = luma(o2a), Ltoptop = luma(o2b);
			delta_zw = V2(fabsf(Lleft - Lleftleft), fabsf(Ltop - Ltoptop));
			maxDelta = V2(std::max(maxDelta.x, delta_zw.x), std::max(maxDelta.y, delta_zw.y));
			float finalDelta = std::max(maxDelta.x, maxDelta.y);
			edges.x *= stepf(finalDelta, 2.0f * delta_xy.x); // SMAA_LOCAL_CONTRAST_ADAPTATION_FACTOR = 2
			edges.y *= stepf(finalDelta, 2.0f * delta_xy.y);
			uint8_t *p = edges_rg8 + (size_t(y) * w + x) * 2;
			p[0] = float_to_unorm8(edges.x);
			p[1] = float_to_unorm8(edges.y);
		}
}

namespace
{
struct SmaaWeights
{
	Tex8 edges, area, search;
	vec4 rt;
	SmaaPreset P;

	static vec2 rg(vec4 v) { return V2(v.x, v.y); }
	vec2 rtxy() const { return V2(rt.x, rt.y); }

	// ---- diagonal search (SMAA.hlsl:831-975) ----
	static vec2 decode_diag2(vec2 e)
	{
		e.x = e.x * fabsf(5.0f * e.x - 5.0f * 0.75f);
		return V2(roundf(e.x), roundf(e.y));
	}
	static vec4 decode_diag4(vec4 e)
	{
		e.x = e.x * fabsf(5.0f * e.x - 5.0f * 0.75f);
		e.z = e.z * fabsf(5.0f * e.z - 5.0f * 0.75f);
		return V4(roundf(e.x), roundf(e.y), roundf(e.z), roundf(e.w));
	}
	vec2 search_diag1(vec2 texcoord, vec2 dir, vec2 &e) const
	{
		vec4 coord = V4(texcoord.x, texcoord.y, -1.0f, 1.0f);
		vec3 t = V3(rt.x, rt.y, 1.0f);
		while (coord.z < float(P.max_search_steps_diag - 1) && coord.w > 0.9f)
		{
			coord.x = fmaf(t.x, dir.x, coord.x);
			coord.y = fmaf(t.y, dir.y, coord.y);
			coord.z = fmaf(t.z, 1.0f, coord.z);
			e = rg(edges.sample(V2(coord.x, coord.y)));
			coord.w = dot(e, V2(0.5f, 0.5f));
		}
		return V2(coord.z, coord.w);
	}
	vec2 search_diag2(vec2 texcoord, vec2 dir, vec2 &e) const
	{
		vec4 coord = V4(texcoord.x, texcoord.y, -1.0f, 1.0f);
		coord.x += 0.25f * rt.x;
		vec3 t = V3(rt.x, rt.y, 1.0f);
		while (coord.z < float(P.max_search_steps_diag - 1) && coord.w > 0.9f)
		{
			coord.x = fmaf(t.x, dir.x, coord.x);
			coord.y = fmaf(t.y, dir.y, coord.y);
			coord.z = fmaf(t.z, 1.0f, coord.z);
			e = rg(edges.sample(V2(coord.x, coord.y)));
			e = decode_diag2(e);
			coord.w = dot(e, V2(0.5f, 0.5f));
		}
		return V2(coord.z, coord.w);
	}
	vec2 area_diag(vec2 dist, vec2 e, float offset) const
	{
		vec2 texcoord = fma2(V2(20.0f, 20.0f), e, dist); // SMAA_AREATEX_MAX_DISTANCE_DIAG
		const vec2 px = V2(1.0f / 160.0f, 1.0f / 560.0f);
		texcoord = fma2(px, texcoord, 0.5f * px);
		texcoord.x += 0.5f;
		texcoord.y += (1.0f / 7.0f) * offset;
		return rg(area.sample(texcoord));
	}
	vec2 diag_weights(vec2 texcoord, vec2 e, vec4 sub) const
	{
		vec2 weights = V2(0.0f, 0.0f);
		vec4 d;
		vec2 end = V2(0.0f, 0.0f);
		if (e.x > 0.0f)
		{
			vec2 r = search_diag1(texcoord, V2(-1.0f, 1.0f), end);
			d.x = r.x; d.z = r.y;
			d.x += float(end.y > 0.9f);
		}
		else
		{
			d.x = 0.0f; d.z = 0.0f;
		}
		{
			vec2 r = search_diag1(texcoord, V2(1.0f, -1.0f), end);
			d.y = r.x; d.w = r.y;
		}
		if (d.x + d.y > 2.0f)
		{
			vec4 coords = V4(fmaf(-d.x + 0.25f, rt.x, texcoord.x), fmaf(d.x, rt.y, texcoord.y), fmaf(d.y, rt.x, texcoord.x),
			                 fmaf(-d.y - 0.25f, rt.y, texcoord.y));
			vec4 c;
			vec2 a = rg(edges.sample(V2(coords.x, coords.y), -1, 0));
			vec2 b = rg(edges.sample(V2(coords.z, coords.w), 1, 0));
			c = V4(a.x, a.y, b.x, b.y);
			vec4 dec = decode_diag4(c);
			c = V4(dec.y, dec.x, dec.w, dec.z); // c.yxwz = decode(c.xyzw)
			vec2 cc = fma2(V2(2.0f, 2.0f), V2(c.x, c.z), V2(c.y, c.w));
			if (stepf(0.9f, d.z) != 0.0f) cc.x = 0.0f;
			if (stepf(0.9f, d.w) != 0.0f) cc.y = 0.0f;
			weights = weights + area_diag(V2(d.x, d.y), cc, sub.z);
		}

		{
			vec2 r = search_diag2(texcoord, V2(-1.0f, -1.0f), end);
			d.x = r.x; d.z = r.y;
		}
		if (edges.sample(texcoord, 1, 0).x > 0.0f)
		{
			vec2 r = search_diag2(texcoord, V2(1.0f, 1.0f), end);
			d.y = r.x; d.w = r.y;
			d.y += float(end.y > 0.9f);
		}
		else
		{
			d.y = 0.0f; d.w = 0.0f;
		}
		if (d.x + d.y > 2.0f)
		{
			vec4 coords = V4(fmaf(-d.x, rt.x, texcoord.x), fmaf(-d.x, rt.y, texcoord.y), fmaf(d.y, rt.x, texcoord.x),
			                 fmaf(d.y, rt.y, texcoord.y));
			vec4 c;
			c.x = edges.sample(V2(coords.x, coords.y), -1, 0).y;
			c.y = edges.sample(V2(coords.x, coords.y), 0, -1).x;
			vec4 zw = edges.sample(V2(coords.z, coords.w), 1, 0);
			c.z = zw.y; c.w = zw.x; // .gr
			vec2 cc = fma2(V2(2.0f, 2.0f), V2(c.x, c.z), V2(c.y, c.w));
			if (stepf(0.9f, d.z) != 0.0f) cc.x = 0.0f;
			if (stepf(0.9f, d.w) != 0.0f) cc.y = 0.0f;
			vec2 ar = area_diag(V2(d.x, d.y), cc, sub.w);
			weights = weights + V2(ar.y, ar.x); // .gr
		}
		return weights;
	}

	// ---- orthogonal search (SMAA.hlsl:977-1093) ----
	float search_length(vec2 e, float offset) const
	{
		vec2 scale = V2(66.0f * 0.5f, 33.0f * -1.0f);
		vec2 bias = V2(66.0f * offset, 33.0f * 1.0f);
		scale = scale + V2(-1.0f, 1.0f);
		bias = bias + V2(0.5f, -0.5f);
		scale = scale * V2(1.0f / 64.0f, 1.0f / 16.0f);
		bias = bias * V2(1.0f / 64.0f, 1.0f / 16.0f);
		return search.sample(fma2(scale, e, bias)).x;
	}
	float search_x_left(vec2 texcoord, float end) const
	{
		vec2 e = V2(0.0f, 1.0f);
		while (texcoord.x > end && e.y > 0.8281f && e.x == 0.0f)
		{
			e = rg(edges.sample(texcoord));
			texcoord = fma2(V2(-2.0f, -0.0f), rtxy(), texcoord);
		}
		float offset = fmaf(-(255.0f / 127.0f), search_length(e, 0.0f), 3.25f);
		return fmaf(rt.x, offset, texcoord.x);
	}
	float search_x_right(vec2 texcoord, float end) const
	{
		vec2 e = V2(0.0f, 1.0f);
		while (texcoord.x < end && e.y > 0.8281f && e.x == 0.0f)
		{
			e = rg(edges.sample(texcoord));
			texcoord = fma2(V2(2.0f, 0.0f), rtxy(), texcoord);
		}
		float offset = fmaf(-(255.0f / 127.0f), search_length(e, 0.5f), 3.25f);
		return fmaf(-rt.x, offset, texcoord.x);
	}
	float search_y_up(vec2 texcoord, float end) const
	{
		vec2 e = V2(1.0f, 0.0f);
		while (texcoord.y > end && e.x > 0.8281f && e.y == 0.0f)
		{
			e = rg(edges.sample(texcoord));
			texcoord = fma2(V2(-0.0f, -2.0f), rtxy(), texcoord);
		}
		float offset = fmaf(-(255.0f / 127.0f), search_length(V2(e.y, e.x), 0.0f), 3.25f);
		return fmaf(rt.y, offset, texcoord.y);
	}
	float search_y_down(vec2 texcoord, float end) const
	{
		vec2 e = V2(1.0f, 0.0f);
		while (texcoord.y < end && e.x > 0.8281f && e.y == 0.0f)
		{
			e = rg(edges.sample(texcoord));
			texcoord = fma2(V2(0.0f, 2.0f), rtxy(), texcoord);
		}
		float offset = fmaf(-(255.0f / 127.0f), search_length(V2(e.y, e.x), 0.5f), 3.25f);
		return fmaf(-rt.y, offset, texcoord.y);
	}
	vec2 area_lookup(vec2 dist, float e1, float e2, float offset) const
	{
		vec2 texcoord = fma2(V2(16.0f, 16.0f), V2(roundf(4.0f * e1), roundf(4.0f * e2)), dist); // SMAA_AREATEX_MAX_DISTANCE
		const vec2 px = V2(1.0f / 160.0f, 1.0f / 560.0f);
		texcoord = fma2(px, texcoord, 0.5f * px);
		texcoord.y = fmaf(1.0f / 7.0f, offset, texcoord.y);
		return rg(area.sample(texcoord));
	}
	void corner_horizontal(vec2 &weights, vec4 texcoord, vec2 d) const
	{
		if (!P.corner)
			return;
		vec2 leftRight = V2(stepf(d.x, d.y), stepf(d.y, d.x));
		vec2 rounding = (1.0f - float(P.corner_rounding) / 100.0f) * leftRight;
		float sum = leftRight.x + leftRight.y;
		rounding = V2(rounding.x / sum, rounding.y / sum);
		vec2 factor = V2(1.0f, 1.0f);
		factor.x -= rounding.x * edges.sample(V2(texcoord.x, texcoord.y), 0, 1).x;
		factor.x -= rounding.y * edges.sample(V2(texcoord.z, texcoord.w), 1, 1).x;
		factor.y -= rounding.x * edges.sample(V2(texcoord.x, texcoord.y), 0, -2).x;
		factor.y -= rounding.y * edges.sample(V2(texcoord.z, texcoord.w), 1, -2).x;
		weights = weights * V2(saturate(factor.x), saturate(factor.y));
	}
	void corner_vertical(vec2 &weights, vec4 texcoord, vec2 d) const
	{
		if (!P.corner)
			return;
		vec2 leftRight = V2(stepf(d.x, d.y), stepf(d.y, d.x));
		vec2 rounding = (1.0f - float(P.corner_rounding) / 100.0f) * leftRight;
		float sum = leftRight.x + leftRight.y;
		rounding = V2(rounding.x / sum, rounding.y / sum);
		vec2 factor = V2(1.0f, 1.0f);
		factor.x -= rounding.x * edges.sample(V2(texcoord.x, texcoord.y), 1, 0).y;
		factor.x -= rounding.y * edges.sample(V2(texcoord.z, texcoord.w), 1, 1).y;
		factor.y -= rounding.x * edges.sample(V2(texcoord.x, texcoord.y), -2, 0).y;
		factor.y -= rounding.y * edges.sample(V2(texcoord.z, texcoord.w), -2, 1).y;
		weights = weights * V2(saturate(factor.x), saturate(factor.y));
	}

	// SMAABlendingWeightCalculationVS + PS (SMAA.hlsl:594-609,1141-1250)
	vec4 weights_at(int x, int y) const
	{
		vec2 texcoord = V2((float(x) + 0.5f) * rt.x, (float(y) + 0.5f) * rt.y);
		vec2 pixcoord = V2(texcoord.x * rt.z, texcoord.y * rt.w);
		vec4 off0 = V4(fmaf(rt.x, -0.25f, texcoord.x), fmaf(rt.y, -0.125f, texcoord.y), fmaf(rt.x, 1.25f, texcoord.x),
		               fmaf(rt.y, -0.125f, texcoord.y));
		vec4 off1 = V4(fmaf(rt.x, -0.125f, texcoord.x), fmaf(rt.y, -0.25f, texcoord.y), fmaf(rt.x, -0.125f, texcoord.x),
		               fmaf(rt.y, 1.25f, texcoord.y));
		float steps = float(P.max_search_steps);
		vec4 off2 = V4(fmaf(rt.x, -2.0f * steps, off0.x), fmaf(rt.x, 2.0f * steps, off0.z), fmaf(rt.y, -2.0f * steps, off1.y),
		               fmaf(rt.y, 2.0f * steps, off1.w));
		const vec4 sub = V4(0.0f); // SMAA 1x

		vec4 weights = V4(0.0f);
		vec2 e = rg(edges.sample(texcoord));
		if (e.y > 0.0f) // edge at north
		{
			bool orthogonal = true;
			if (P.diag)
			{
				vec2 dw = diag_weights(texcoord, e, sub);
				weights.x = dw.x; weights.y = dw.y;
				orthogonal = (weights.x == -weights.y);
			}
			if (orthogonal)
			{
				vec2 d;
				vec3 coords;
				coords.x = search_x_left(V2(off0.x, off0.y), off2.x);
				coords.y = off1.y;
				d.x = coords.x;
				float e1 = edges.sample(V2(coords.x, coords.y)).x;
				coords.z = search_x_right(V2(off0.z, off0.w), off2.y);
				d.y = coords.z;
				d = V2(fabsf(roundf(fmaf(rt.z, d.x, -pixcoord.x))), fabsf(roundf(fmaf(rt.z, d.y, -pixcoord.x))));
				vec2 sqrt_d = V2(sqrtf(d.x), sqrtf(d.y));
				float e2 = edges.sample(V2(coords.z, coords.y), 1, 0).x;
				vec2 wrg = area_lookup(sqrt_d, e1, e2, sub.y);
				coords.y = texcoord.y;
				corner_horizontal(wrg, V4(coords.x, coords.y, coords.z, coords.y), d);
				weights.x = wrg.x; weights.y = wrg.y;
			}
			else
				e.x = 0.0f; // skip vertical processing
		}
		if (e.x > 0.0f) // edge at west
		{
			vec2 d;
			vec3 coords;
			coords.y = search_y_up(V2(off1.x, off1.y), off2.z);
			coords.x = off0.x;
			d.x = coords.y;
			float e1 = edges.sample(V2(coords.x, coords.y)).y;
			coords.z = search_y_down(V2(off1.z, off1.w), off2.w);
			d.y = coords.z;
			d = V2(fabsf(roundf(fmaf(rt.w, d.x, -pixcoord.y))), fabsf(roundf(fmaf(rt.w, d.y, -pixcoord.y))));
			vec2 sqrt_d = V2(sqrtf(d.x), sqrtf(d.y));
			float e2 = edges.sample(V2(coords.x, coords.z), 0, 1).y;
			vec2 wba = area_lookup(sqrt_d, e1, e2, sub.x);
			coords.x = texcoord.x;
			corner_vertical(wba, V4(coords.x, coords.y, coords.x, coords.z), d);
			weights.z = wba.x; weights.w = wba.y;
		}
		return weights;
	}
};
} // namespace

// smaa_blend_weight.{vert,frag}; depth-mask EQUAL == "edge pass did not discard" == edge texel non-zero (smaa.cpp:101-112,170-177).
void orc_smaa_weights(const uint8_t *edges_rg8, int w, int h, const uint8_t *area_rg8, const uint8_t *search_r8, uint8_t *weights_rgba8,
                      int quality)
{
	SmaaWeights S{{edges_rg8, w, h, 2}, {area_rg8, 160, 560, 2}, {search_r8, 64, 16, 1}, V4(1.0f / float(w), 1.0f / float(h), float(w), float(h)),
	              smaa_preset(quality)};
	memset(weights_rgba8, 0, size_t(w) * h * 4);
#pragma omp parallel for schedule(dynamic, 4)
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			const uint8_t *e = edges_rg8 + (size_t(y) * w + x) * 2;
			if (!(e[0] | e[1]))
				continue; // masked out
			vec4 wt = S.weights_at(x, y);
			store_rgba8(weights_rgba8, w, x, y, float_to_unorm8(wt.x), float_to_unorm8(wt.y), float_to_unorm8(wt.z), float_to_unorm8(wt.w));
		}
}

// smaa_neighbor_blend.{vert,frag} + SMAANeighborhoodBlendingPS (SMAA.hlsl:1252-1308).
void orc_smaa_blend(const uint8_t *color, const uint8_t *weights_rgba8, int w, int h, uint8_t *out, int target_srgb)
{
	Tex8 ctex{color, w, h, 4}, btex{weights_rgba8, w, h, 4};
	const vec4 rt = V4(1.0f / float(w), 1.0f / float(h), float(w), float(h));
#pragma omp parallel for schedule(static)
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			vec2 texcoord = V2((float(x) + 0.5f) * rt.x, (float(y) + 0.5f) * rt.y);
			vec4 offset = V4(fmaf(rt.x, 1.0f, texcoord.x), fmaf(rt.y, 0.0f, texcoord.y), fmaf(rt.x, 0.0f, texcoord.x), fmaf(rt.y, 1.0f, texcoord.y));
			vec4 a;
			a.x = btex.sample(V2(offset.x, offset.y)).w; // right
			a.y = btex.sample(V2(offset.z, offset.w)).y; // top
			vec4 c = btex.sample(texcoord);
			a.w = c.x; // bottom
			a.z = c.z; // left
			vec4 result;
			if (a.x + a.y + a.z + a.w < 1e-5f)
				result = ctex.sample(texcoord);
			else
			{
				bool hz = std::max(a.x, a.z) > std::max(a.y, a.w);
				vec4 blendingOffset = V4(0.0f, a.y, 0.0f, a.w);
				vec2 blendingWeight = V2(a.y, a.w);
				if (hz)
				{
					blendingOffset = V4(a.x, 0.0f, a.z, 0.0f);
					blendingWeight = V2(a.x, a.z);
				}
				float sum = blendingWeight.x + blendingWeight.y;
				blendingWeight = V2(blendingWeight.x / sum, blendingWeight.y / sum);
				vec4 bc = V4(fmaf(blendingOffset.x, rt.x, texcoord.x), fmaf(blendingOffset.y, rt.y, texcoord.y),
				             fmaf(blendingOffset.z, -rt.x, texcoord.x), fmaf(blendingOffset.w, -rt.y, texcoord.y));
				result = blendingWeight.x * ctex.sample(V2(bc.x, bc.y));
				result = result + blendingWeight.y * ctex.sample(V2(bc.z, bc.w));
			}
			if (target_srgb)
				store_rgba8(out, w, x, y, float_to_srgb8(srgb_decode(result.x)), float_to_srgb8(srgb_decode(result.y)),
				            float_to_srgb8(srgb_decode(result.z)), float_to_unorm8(result.w));
			else
				store_rgba8(out, w, x, y, float_to_unorm8(result.x), float_to_unorm8(result.y), float_to_unorm8(result.z), float_to_unorm8(result.w));
		}
}

// ---------------------------------------------------------------------------------------------------------------------
// TAA resolve (taa_resolve.frag + reprojection.h).  Current / depth / MVs use NearestClamp, history LinearClamp.
// quality 0/1/2 = Low/Medium/High.  history_in == NULL => REPROJECTION_HISTORY = 0 (first frame).
// Outputs: color (HDR space) and history (YCgCo of tonemapped), both RGBA16F with alpha = 1 (shader outputs are vec3).
// ---------------------------------------------------------------------------------------------------------------------
namespace
{
static inline vec3 taa_tonemap(vec3 c)
{
	c = c * 8.0f;
	return c * (1.0f / (std::max(c.x, std::max(c.y, c.z)) + 1.0f));
}
static inline vec3 taa_tonemap_invert(vec3 c)
{
	return (1.0f / 8.0f) * c * (1.0f / (1.0f - std::max(c.x, std::max(c.y, c.z))));
}
static inline vec3 rgb_to_ycgco(vec3 c)
{
	return V3(0.25f * c.x + 0.5f * c.y + 0.25f * c.z, 0.5f * c.y - 0.25f * c.x - 0.25f * c.z, 0.5f * c.x - 0.5f * c.z);
}
static inline vec3 ycgco_to_rgb(vec3 c)
{
	float tmp = c.x - c.y;
	return V3(tmp + c.z, c.x + c.y, tmp - c.z);
}
static inline vec3 hdr_to_taa(vec3 c) { return rgb_to_ycgco(taa_tonemap(c)); }
static inline vec3 taa_to_hdr(vec3 c)
{
	vec3 r = ycgco_to_rgb(c);
	return taa_tonemap_invert(V3(clampf(r.x, 0.0f, 0.999f), clampf(r.y, 0.0f, 0.999f), clampf(r.z, 0.0f, 0.999f)));
}
static inline vec3 clamp_box(vec3 color, vec3 lo, vec3 hi, bool aabb)
{
	if (!aabb)
		return V3(clampf(color.x, lo.x, hi.x), clampf(color.y, lo.y, hi.y), clampf(color.z, lo.z, hi.z));
	vec3 center = 0.5f * (lo + hi);
	vec3 radius = max3(0.5f * (hi - lo), V3(0.0001f));
	vec3 v = color - center;
	vec3 units = v / radius;
	float max_unit = std::max(std::max(fabsf(units.x), fabsf(units.y)), fabsf(units.z));
	return max_unit > 1.0f ? center + v / max_unit : color;
}
} // namespace

// color_b10g11r11: the resolved colour goes to a B10G11R11_UFLOAT_PACK32 attachment (temporal.cpp:211-213: the reference's
// choice wherever the format is renderable); out_color then holds the packed values as their exact RGBA16F texels.  The history
// stays RGBA16F (temporal.cpp:216).
void orc_taa_resolve_fmt(const uint16_t *current, const float *depth, const uint16_t *mv_rg16f, const uint16_t *history_in, int w, int h,
                         const float *reproj16, int quality, uint16_t *out_color, uint16_t *out_history, int color_b10g11r11);
void orc_taa_resolve(const uint16_t *current, const float *depth, const uint16_t *mv_rg16f, const uint16_t *history_in, int w, int h,
                     const float *reproj16, int quality, uint16_t *out_color, uint16_t *out_history)
{
	orc_taa_resolve_fmt(current, depth, mv_rg16f, history_in, w, h, reproj16, quality, out_color, out_history, 0);
}
void orc_taa_resolve_fmt(const uint16_t *current, const float *depth, const uint16_t *mv_rg16f, const uint16_t *history_in, int w, int h,
                         const float *reproj16, int quality, uint16_t *out_color, uint16_t *out_history, int color_b10g11r11)
{
	Tex16F cur{current, w, h};
	Tex16F hist{history_in, w, h};
	const bool cubic = quality == 2;
	const bool aabb = quality >= 1;
	const vec4 rt = V4(1.0f / float(w), 1.0f / float(h), float(w), float(h));
	mat4 reproj;
	for (int c = 0; c < 4; c++)
		reproj.c[c] = V4(reproj16[4 * c], reproj16[4 * c + 1], reproj16[4 * c + 2], reproj16[4 * c + 3]);

	auto cur_at = [&](int x, int y) { vec4 t = cur.fetch(x, y); return hdr_to_taa(V3(t.x, t.y, t.z)); };
	auto depth_at = [&](int x, int y) { return depth[size_t(clampi(y, 0, h - 1)) * w + clampi(x, 0, w - 1)]; };
	auto mv_at = [&](int x, int y) {
		const uint16_t *p = mv_rg16f + (size_t(clampi(y, 0, h - 1)) * w + clampi(x, 0, w - 1)) * 2;
		return V2(half_to_float(p[0]), half_to_float(p[1]));
	};

#pragma omp parallel for schedule(static)
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			vec2 uv = V2((float(x) + 0.5f) * rt.x, (float(y) + 0.5f) * rt.y);
			vec3 current_c = cur_at(x, y);
			vec3 out_c, hist_c;
			if (!history_in)
			{
				out_c = taa_to_hdr(current_c);
				hist_c = current_c;
			}
			else
			{
				// sample_nearest_velocity (reprojection.h:213-283): candidates in the reference's comparison order.
				vec2 mv;
				float d;
				auto consider = [&](int ox, int oy) {
					float dd = depth_at(x + ox, y + oy);
					if (dd > d) { mv = mv_at(x + ox, y + oy); d = dd; }
				};
				if (quality <= 1) // NEAREST_METHOD_5TAP_CROSS
				{
					mv = mv_at(x - 1, y); d = depth_at(x - 1, y);
					consider(0, 0); consider(0, -1); consider(0, 1); consider(1, 0);
				}
				else // NEAREST_METHOD_3x3
				{
					mv = mv_at(x + 1, y + 1); d = depth_at(x + 1, y + 1);
					consider(-1, 0); consider(0, 0); consider(0, -1); consider(-1, -1);
					consider(1, 0); consider(1, -1); consider(-1, 1); consider(0, 1);
				}

				vec2 old_uv;
				if (mv.x == 0.0f && mv.y == 0.0f)
				{
					vec4 clip = V4(2.0f * uv.x - 1.0f, 2.0f * uv.y - 1.0f, d, 1.0f);
					vec4 rp = mul(reproj, clip);
					old_uv = V2(rp.x / rp.w, rp.y / rp.w);
					mv = uv - old_uv;
				}
				else
					old_uv = uv - mv;

				vec3 history_color;
				if (cubic)
				{
					// sample_catmull_rom (reprojection.h:286-334)
					vec2 samplePos = V2(old_uv.x * rt.z, old_uv.y * rt.w);
					vec2 texPos1 = V2(floorf(samplePos.x - 0.5f) + 0.5f, floorf(samplePos.y - 0.5f) + 0.5f);
					vec2 f = samplePos - texPos1;
					auto W0 = [](float f) { return f * (-0.5f + f * (1.0f - 0.5f * f)); };
					auto W1 = [](float f) { return 1.0f + f * f * (-2.5f + 1.5f * f); };
					auto W2 = [](float f) { return f * (0.5f + f * (2.0f - 1.5f * f)); };
					auto W3 = [](float f) { return f * f * (-0.5f + 0.5f * f); };
					vec2 w0 = V2(W0(f.x), W0(f.y)), w1 = V2(W1(f.x), W1(f.y)), w2 = V2(W2(f.x), W2(f.y)), w3 = V2(W3(f.x), W3(f.y));
					vec2 w12 = w1 + w2;
					vec2 offset12 = w2 / (w1 + w2);
					vec2 texPos0 = (texPos1 - V2(1.0f, 1.0f)) * V2(rt.x, rt.y);
					vec2 texPos3 = (texPos1 + V2(2.0f, 2.0f)) * V2(rt.x, rt.y);
					vec2 texPos12 = (texPos1 + offset12) * V2(rt.x, rt.y);
					auto S = [&](float u, float v) { vec4 t = hist.sample_linear(V2(u, v)); return V3(t.x, t.y, t.z); };
					vec3 r = V3(0.0f);
					r += S(texPos0.x, texPos0.y) * w0.x * w0.y;
					r += S(texPos12.x, texPos0.y) * w12.x * w0.y;
					r += S(texPos3.x, texPos0.y) * w3.x * w0.y;
					r += S(texPos0.x, texPos12.y) * w0.x * w12.y;
					r += S(texPos12.x, texPos12.y) * w12.x * w12.y;
					r += S(texPos3.x, texPos12.y) * w3.x * w12.y;
					r += S(texPos0.x, texPos3.y) * w0.x * w3.y;
					r += S(texPos12.x, texPos3.y) * w12.x * w3.y;
					r += S(texPos3.x, texPos3.y) * w3.x * w3.y;
					history_color = r;
				}
				else
				{
					vec4 t = hist.sample_linear(old_uv);
					history_color = V3(t.x, t.y, t.z);
				}

				float mv_length = length(mv);
				float mv_fast = std::min(mv_length * 50.0f, 1.0f);
				float gamma = mixf(1.5f, 0.5f, mv_fast);
				history_color = V3(clampf(history_color.x, 0.0f, 1.0f), clampf(history_color.y, -1.0f, 1.0f), clampf(history_color.z, -1.0f, 1.0f));
				float lerp_factor = (1.0f + 2.0f * mv_fast) / 16.0f;

				// clamp_history_box (reprojection.h:107-183)
				vec3 c11 = current_c;
				vec3 c01 = cur_at(x - 1, y), c21 = cur_at(x + 1, y), c10 = cur_at(x, y - 1), c12 = cur_at(x, y + 1);
				vec3 lo, hi;
				if (quality == 0) // 5TAP_CROSS, min/max
				{
					lo = min3(min3(min3(min3(c11, c01), c21), c10), c12);
					hi = max3(max3(max3(max3(c11, c01), c21), c10), c12);
				}
				else
				{
					vec3 c00 = cur_at(x - 1, y - 1), c22 = cur_at(x + 1, y + 1), c02 = cur_at(x - 1, y + 1), c20 = cur_at(x + 1, y - 1);
					if (quality == 1) // ROUNDED_CORNER
					{
						vec3 clo = min3(min3(min3(min3(c11, c01), c21), c10), c12);
						vec3 chi = max3(max3(max3(max3(c11, c01), c21), c10), c12);
						lo = min3(min3(min3(min3(clo, c00), c22), c02), c20);
						hi = max3(max3(max3(max3(chi, c00), c22), c02), c20);
						lo = 0.5f * (clo + lo);
						hi = 0.5f * (chi + hi);
					}
					else // VARIANCE
					{
						vec3 m1 = (c00 + 2.0f * c01 + c02 + 2.0f * c10 + 4.0f * c11 + 2.0f * c12 + c20 + 2.0f * c21 + c22) / 16.0f;
						vec3 m2 = c00 * c00 + 2.0f * c01 * c01 + c02 * c02 + 2.0f * c10 * c10 + 4.0f * c11 * c11 + 2.0f * c12 * c12 + c20 * c20 +
						          2.0f * c21 * c21 + c22 * c22;
						vec3 variance = max3(m2 / 16.0f - m1 * m1, V3(0.0f));
						vec3 sigma = V3(sqrtf(variance.x), sqrtf(variance.y), sqrtf(variance.z));
						lo = m1 - gamma * sigma;
						hi = m1 + gamma * sigma;
					}
				}
				history_color = clamp_box(history_color, lo, hi, aabb);
				vec3 mixed = mix(history_color, current_c, lerp_factor);
				hist_c = mixed;
				out_c = taa_to_hdr(mixed);
			}
			if (color_b10g11r11)
				store_rgba16f_as_b10g11r11(out_color, w, x, y, V4(out_c, 1.0f));
			else
				store_rgba16f(out_color, w, x, y, V4(out_c, 1.0f));
			store_rgba16f(out_history, w, x, y, V4(hist_c, 1.0f));
		}
}
}
