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
// B10G11R11_UFLOAT_PACK32 <-> its exact RGBA16F image (oracle_common.h); `count` texels.
void orc_pack_b10g11r11_from_f32(const float *rgb, uint32_t *out, uint64_t count)
{
	for (uint64_t i = 0; i < count; i++)
		out[i] = pack_b10g11r11(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
}
void orc_pack_b10g11r11_from_rgba16f(const uint16_t *rgba16f, uint32_t *out, uint64_t count)
{
	for (uint64_t i = 0; i < count; i++)
		out[i] = pack_b10g11r11(half_to_float(rgba16f[4 * i]), half_to_float(rgba16f[4 * i + 1]), half_to_float(rgba16f[4 * i + 2]));
}
void orc_unpack_b10g11r11_to_rgba16f(const uint32_t *in, uint16_t *rgba16f, uint64_t count)
{
	for (uint64_t i = 0; i < count; i++)
	{
		const vec4 v = unpack_b10g11r11(in[i]);
		rgba16f[4 * i] = float_to_half_rne(v.x);
		rgba16f[4 * i + 1] = float_to_half_rne(v.y);
		rgba16f[4 * i + 2] = float_to_half_rne(v.z);
		rgba16f[4 * i + 3] = 0x3c00u;
	}
}

uint16_t orc_float_to_half(float f) { return float_to_half_rne(f); }
float orc_half_to_float(uint16_t h) { return half_to_float(h); }
uint16_t orc_float_to_half_muglm(float f) { return float_to_half_muglm(f); }
void orc_float_to_half_muglm_array(const float *in, uint16_t *out, uint64_t count)
{
	for (uint64_t i = 0; i < count; i++)
		out[i] = float_to_half_muglm(in[i]);
}
uint8_t orc_float_to_srgb8(float f) { return float_to_srgb8(f); }
// the same over an array (checks the product's encode table value by value)
void orc_float_to_srgb8_array(const float *in, size_t n, uint8_t *out)
{
#pragma omp parallel for
	for (long i = 0; i < long(n); i++)
		out[i] = float_to_srgb8(in[i]);
}
float orc_srgb8_to_float(uint8_t v) { return srgb8_to_float(v); }

// builtin://shaders/blit.frag over a full-screen quad: FragColor = Scale * textureLod(uImage, vUV, 0.0) with Scale = 1, vUV = the
// pixel centre (pinned: ref_build/ref_shaders.cpp ref_blit executes the shader, test_blit_shader_bit_for_bit); the
// copy Granite's tools record between targets of different size / format (tools/aa_bench.cpp:97-105 LinearClamp into the HDR
// target, :138-147 NearestClamp into the swapchain).  Formats: 0 = R16G16B16A16_SFLOAT, 1 = R8G8B8A8_UNORM, 2 = R8G8B8A8_SRGB.
void orc_blit(const void *in, int iw, int ih, int in_format, void *out, int ow, int oh, int out_format, int linear)
{
	auto texel = [&](int x, int y) {
		x = clampi(x, 0, iw - 1);
		y = clampi(y, 0, ih - 1);
		if (in_format == 0)
			return load_rgba16f(static_cast<const uint16_t *>(in), iw, x, y);
		const uint8_t *p = static_cast<const uint8_t *>(in) + (size_t(y) * iw + x) * 4;
		if (in_format == 2)
			return V4(srgb8_to_float(p[0]), srgb8_to_float(p[1]), srgb8_to_float(p[2]), unorm8_to_float(p[3]));
		return V4(unorm8_to_float(p[0]), unorm8_to_float(p[1]), unorm8_to_float(p[2]), unorm8_to_float(p[3]));
	};
#pragma omp parallel for
	for (int y = 0; y < oh; y++)
		for (int x = 0; x < ow; x++)
		{
			const vec2 uv = V2((float(x) + 0.5f) * (1.0f / float(ow)), (float(y) + 0.5f) * (1.0f / float(oh)));
			vec4 c;
			if (linear)
			{
				float a, b;
				int x0, y0;
				linear_axis(uv.x * float(iw) - 0.5f, x0, a);
				linear_axis(uv.y * float(ih) - 0.5f, y0, b);
				c = linear_combine(texel(x0, y0), texel(x0 + 1, y0), texel(x0, y0 + 1), texel(x0 + 1, y0 + 1), a, b);
			}
			else
				c = texel(int(floorf(uv.x * float(iw))), int(floorf(uv.y * float(ih))));
			if (out_format == 0)
				store_rgba16f(static_cast<uint16_t *>(out), ow, x, y, c);
			else
			{
				uint8_t *p = static_cast<uint8_t *>(out) + (size_t(y) * ow + x) * 4;
				const bool srgb = out_format == 2;
				p[0] = srgb ? float_to_srgb8(c.x) : float_to_unorm8(c.x);
				p[1] = srgb ? float_to_srgb8(c.y) : float_to_unorm8(c.y);
				p[2] = srgb ? float_to_srgb8(c.z) : float_to_unorm8(c.z);
				p[3] = float_to_unorm8(c.w);
			}
		}
}
void orc_sample_linear_rgba16f(const uint16_t *img, int w, int h, float u, float v, float *out4)
{
	Tex16F t{img, w, h};
	vec4 r = t.sample_linear(V2(u, v));
	out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
}

// ---- HDR10 output: assets/shaders/post/pq10_encode.frag + setup_hdr10_pq_encoding (renderer/post/hdr.cpp:595-658) --------------
// hdr: RGBA16F (texelFetch), ui: RGBA8 read through its sRGB view (rgb decoded, alpha linear), both w x h.
// primary_conversion: column-major mat3 (rec.709 -> display primaries, compute_rec709_to_st2020 hdr.cpp:580-593).
// out: A2B10G10R10_UNORM_PACK32 (the HDR10 swapchain format), alpha = 1.
extern "C" void orc_pq10_encode(const uint16_t *hdr, const uint8_t *ui_srgb8, int w, int h, const float *primary_conversion9, float hdr_pre_exposure,
                                float ui_pre_exposure, float max_light_level, uint32_t *out)
{
	using namespace orc;
	const float inv_max_light_level = 1.0f / max_light_level; // hdr.cpp:644
	auto encode_pq = [](float nits) {
		const float y = nits / 10000.0f;
		const float c1 = 0.8359375f, c2 = 18.8515625f, c3 = 18.6875f, m1 = 0.1593017578125f, m2 = 78.84375f;
		const float num = c1 + c2 * powf(y, m1);
		const float den = 1.0f + c3 * powf(y, m1);
		return powf(num / den, m2);
	};
	auto unorm10 = [](float v) -> uint32_t {
		if (!(v > 0.0f))
			return 0u;
		if (v >= 1.0f)
			return 1023u;
		return uint32_t(int(v * 1023.0f + 0.5f));
	};
#pragma omp parallel for schedule(static)
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			const size_t i = size_t(y) * w + x;
			const vec3 c_hdr = V3(half_to_float(hdr[4 * i]), half_to_float(hdr[4 * i + 1]), half_to_float(hdr[4 * i + 2]));
			const uint8_t *u = ui_srgb8 + 4 * i;
			const vec3 ui_rgb = V3(srgb8_to_float(u[0]), srgb8_to_float(u[1]), srgb8_to_float(u[2]));
			const float ui_a = float(u[3]) / 255.0f;
			vec3 col = c_hdr * (hdr_pre_exposure * ui_a) + ui_rgb * ui_pre_exposure;
			// mat3(config.primary_conversion) * col
			const float *m = primary_conversion9;
			col = V3(m[0], m[1], m[2]) * col.x + V3(m[3], m[4], m[5]) * col.y + V3(m[6], m[7], m[8]) * col.z;
			col = col * inv_max_light_level;
			const float K = 4.0f;
			const vec3 col_k = col * K;
			const vec3 saturated = col_k / (V3(1.0f) + col_k);
			// mix(col, saturated, greaterThan(col, vec3(0.75)))
			col = V3(col.x > 0.75f ? saturated.x : col.x, col.y > 0.75f ? saturated.y : col.y, col.z > 0.75f ? saturated.z : col.z);
			const vec3 scaled = col * max_light_level;
			out[i] = unorm10(encode_pq(scaled.x)) | (unorm10(encode_pq(scaled.y)) << 10) | (unorm10(encode_pq(scaled.z)) << 20) | (3u << 30);
		}
}
