// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED by the reference's own tests.
//
// HDR post chain restated from the reference GLSL + the host code that fills its push constants:
//   bloom_threshold.comp, bloom_downsample.comp, bloom_upsample.comp, luminance.comp, tonemap.frag
//   (assets/shaders/post/*) and renderer/post/hdr.cpp:68-216,283-306.
#include "oracle_common.h"

using namespace orc;

extern "C" {

// assets/shaders/post/bloom_threshold.comp:23-44; push constants hdr.cpp:133-142.
// lum3 = LuminanceData {log, linear, inv_linear} or NULL (DYNAMIC_EXPOSURE=0).
void orc_bloom_threshold(const uint16_t *hdr, int iw, int ih, uint16_t *out, int ow, int oh, const float *lum3)
{
	Tex16F tex{hdr, iw, ih};
	vec2 inv_output_size = V2(1.0f / float(ow), 1.0f / float(oh));
#pragma omp parallel for schedule(static)
	for (int y = 0; y < oh; y++)
	{
		for (int x = 0; x < ow; x++)
		{
			vec2 uv = (V2(float(x), float(y)) + V2(0.5f, 0.5f)) * inv_output_size;
			vec4 t = tex.sample_linear(uv);
			vec3 color = V3(t.x, t.y, t.z);
			float luminance = std::max(std::max(color.x, color.y), color.z) + 0.0001f;
			float loglum = log2f(luminance);
			color = color / luminance;
			if (lum3)
				luminance -= 8.0f * lum3[1];
			else
				luminance -= 8.0f;
			vec3 thres = max3(color * luminance, V3(0.0f));
			store_rgba16f(out, ow, x, y, V4(thres, loglum));
		}
	}
}

static inline vec4 tent9(const Tex16F &tex, vec2 uv, vec2 inv_in, float off)
{
	// Tap order exactly as bloom_downsample.comp:30-38 / bloom_upsample.comp:25-33.
	vec4 value = 0.25f * tex.sample_linear(uv);
	value += 0.0625f * tex.sample_linear(uv + V2(-off, +off) * inv_in);
	value += 0.125f * tex.sample_linear(uv + V2(+0.0f, +off) * inv_in);
	value += 0.0625f * tex.sample_linear(uv + V2(+off, +off) * inv_in);
	value += 0.125f * tex.sample_linear(uv + V2(-off, +0.0f) * inv_in);
	value += 0.125f * tex.sample_linear(uv + V2(+off, +0.0f) * inv_in);
	value += 0.0625f * tex.sample_linear(uv + V2(-off, -off) * inv_in);
	value += 0.125f * tex.sample_linear(uv + V2(+0.0f, -off) * inv_in);
	value += 0.0625f * tex.sample_linear(uv + V2(+off, -off) * inv_in);
	return value;
}

// bloom_downsample.comp:22-44; hdr.cpp:146-187.  history (same size as out, NearestClamp) or NULL.
void orc_bloom_downsample(const uint16_t *in, int iw, int ih, uint16_t *out, int ow, int oh, const uint16_t *history,
                          float lerp)
{
	Tex16F tex{in, iw, ih};
	Tex16F hist{history, ow, oh};
	vec2 inv_out = V2(1.0f / float(ow), 1.0f / float(oh));
	vec2 inv_in = V2(1.0f / float(iw), 1.0f / float(ih));
#pragma omp parallel for schedule(static)
	for (int y = 0; y < oh; y++)
	{
		for (int x = 0; x < ow; x++)
		{
			vec2 uv = (V2(float(x), float(y)) + V2(0.5f, 0.5f)) * inv_out;
			vec4 value = tent9(tex, uv, inv_in, 1.75f);
			if (history)
				value = mix(hist.sample_nearest(uv), value, V4(lerp, lerp, lerp, 1.0f));
			store_rgba16f(out, ow, x, y, value);
		}
	}
}

// bloom_upsample.comp:17-35; hdr.cpp:189-216.
void orc_bloom_upsample(const uint16_t *in, int iw, int ih, uint16_t *out, int ow, int oh)
{
	Tex16F tex{in, iw, ih};
	vec2 inv_out = V2(1.0f / float(ow), 1.0f / float(oh));
	vec2 inv_in = V2(1.0f / float(iw), 1.0f / float(ih));
#pragma omp parallel for schedule(static)
	for (int y = 0; y < oh; y++)
	{
		for (int x = 0; x < ow; x++)
		{
			vec2 uv = (V2(float(x), float(y)) + V2(0.5f, 0.5f)) * inv_out;
			store_rgba16f(out, ow, x, y, tent9(tex, uv, inv_in, 0.875f));
		}
	}
}

// luminance.comp:25-67 — one 8x8 workgroup, per-thread strided partial sums, 64->1 shared-memory tree in the
// reference's exact association order; hdr.cpp:68-98 gives size = d3 dims / 2, lerp, min=-3, max=2.
void orc_luminance(const uint16_t *d3, int w, int h, float *lum3, float lerp, float min_loglum, float max_loglum)
{
	Tex16F tex{d3, w, h};
	int size_x = w / 2, size_y = h / 2;
	int iter_y = (size_y + 7) >> 3;
	int iter_x = (size_x + 7) >> 3;
	vec2 inv_size = V2(1.0f / float(size_x), 1.0f / float(size_y));
	float shared_loglum[64];
	for (int ly = 0; ly < 8; ly++)
	{
		for (int lx = 0; lx < 8; lx++)
		{
			float total = 0.0f;
			for (int y = 0; y < iter_y; y++)
			{
				for (int x = 0; x < iter_x; x++)
				{
					int sx = x * 8 + lx;
					int sy = y * 8 + ly;
					if (sx < size_x && sy < size_y)
						total += tex.sample_linear((V2(float(sx), float(sy)) + V2(0.5f, 0.5f)) * inv_size).w;
				}
			}
			shared_loglum[ly * 8 + lx] = total;
		}
	}
	for (int step = 32; step >= 2; step >>= 1)
		for (int i = 0; i < step; i++)
			shared_loglum[i] += shared_loglum[i + step];
	float loglum = shared_loglum[0] + shared_loglum[1];
	loglum *= inv_size.x * inv_size.y;
	loglum = clampf(loglum, min_loglum, max_loglum);
	float new_log_luma = mixf(lum3[0], loglum, lerp);
	lum3[0] = new_log_luma;
	lum3[1] = exp2f(new_log_luma);
	lum3[2] = exp2f(-new_log_luma);
}

static inline float uncharted2(float x)
{
	const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
	return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}

// tonemap.frag:30-66 + hdr.cpp:283-306.  Output is the RGBA8_SRGB backbuffer (headless swapchain format,
// application_headless.cpp:207-229): the shader writes linear RGB, the attachment store encodes sRGB.
// The shader's output is vec3, alpha is written as 1.0 (undefined in the reference; fixed here).
void orc_tonemap(const uint16_t *hdr, int w, int h, const uint16_t *bloom, int bw, int bh, const float *lum3,
                 float dynamic_exposure, uint8_t *out_srgb8)
{
	Tex16F thdr{hdr, w, h};
	Tex16F tbloom{bloom, bw, bh};
	vec2 inv = V2(1.0f / float(w), 1.0f / float(h));
	const float W = 11.2f;
	float white_scale = 1.0f / uncharted2(W);
	float scale = lum3 ? (lum3[2] * dynamic_exposure) : dynamic_exposure;
#pragma omp parallel for schedule(static)
	for (int y = 0; y < h; y++)
	{
		for (int x = 0; x < w; x++)
		{
			vec2 uv = (V2(float(x), float(y)) + V2(0.5f, 0.5f)) * inv;
			vec4 c = thdr.sample_linear(uv);
			vec4 b = tbloom.sample_linear(uv);
			vec3 color = V3(c.x + b.x, c.y + b.y, c.z + b.z) * scale;
			vec3 r = V3(uncharted2(color.x) * white_scale, uncharted2(color.y) * white_scale,
			            uncharted2(color.z) * white_scale);
			uint8_t *p = out_srgb8 + (size_t(y) * w + x) * 4;
			p[0] = float_to_srgb8(r.x);
			p[1] = float_to_srgb8(r.y);
			p[2] = float_to_srgb8(r.z);
			p[3] = 255;
		}
	}
}

// ---- format helpers exported for the tests ---------------------------------------------------
uint16_t orc_float_to_half(float f) { return float_to_half_rne(f); }
float orc_half_to_float(uint16_t h) { return half_to_float(h); }
uint16_t orc_float_to_half_muglm(float f) { return float_to_half_muglm(f); }
void orc_float_to_half_muglm_array(const float *in, uint16_t *out, uint64_t count)
{
	for (uint64_t i = 0; i < count; i++)
		out[i] = float_to_half_muglm(in[i]);
}
uint8_t orc_float_to_srgb8(float f) { return float_to_srgb8(f); }
float orc_srgb8_to_float(uint8_t v) { return srgb8_to_float(v); }
void orc_sample_linear_rgba16f(const uint16_t *img, int w, int h, float u, float v, float *out4)
{
	Tex16F t{img, w, h};
	vec4 r = t.sample_linear(V2(u, v));
	out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
}
