// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the arithmetic in Granite's image-space chain (GLSL under
// /root/reference/assets/shaders/{post,lights,inc} + host code that fills the push constants).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (granite_amd/) never links, imports or calls it.
//
// PARITY: the reference holds no golden vectors / known-answer tests for this path and its
// Vulkan / GLSL toolchain cannot be built here.  The oracle is pinned instead by EXECUTING THE REFERENCE'S OWN
// SHADER SOURCES on the CPU: oracle/ref_build re-spells them at build time (scratch files, removed again) and compiles
// them as C++ against a small GLSL environment; tests/test_reference_shaders_cpu.py requires bit-for-bit
// equality for the whole post chain, deferred lighting, the cluster build, FXAA, TAA and SMAA.  The host-side
// math the reference keeps in buildable C++ (math/muglm) is checked against the real code the same way
// (tests/test_reference_math_cpu.py).  The depth hierarchy, EASU (fp32 and fp16) and RCAS are pinned the same way.
//
// Conventions
//   * Images are tightly packed row-major linear buffers, origin top-left (Vulkan framebuffer
//     convention: pixel (x,y) centre has uv = ((x,y)+0.5)/size; assets/shaders/quad.vert:1-12 and
//     vulkan/command_buffer.cpp:4605-4622).
//   * "mediump" is fp32 on desktop/lavapipe, so every computation is fp32; values are rounded to the
//     storage format only where the reference stores them (RGBA16F RNE, UNORM8 round-half-up of
//     v*255+0.5, sRGB8 after encode).
//   * No FMA contraction (compile with -ffp-contract=off) so decisions are reproducible.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

namespace orc
{
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

static inline vec2 V2(float x, float y) { return {x, y}; }
static inline vec3 V3(float x, float y, float z) { return {x, y, z}; }
static inline vec3 V3(float s) { return {s, s, s}; }
static inline vec4 V4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline vec4 V4(float s) { return {s, s, s, s}; }
static inline vec4 V4(vec3 v, float w) { return {v.x, v.y, v.z, w}; }

static inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
static inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
static inline vec2 operator*(vec2 a, vec2 b) { return {a.x * b.x, a.y * b.y}; }
static inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
static inline vec2 operator*(float s, vec2 a) { return {a.x * s, a.y * s}; }
static inline vec2 operator/(vec2 a, vec2 b) { return {a.x / b.x, a.y / b.y}; }
static inline vec2 operator-(vec2 a) { return {-a.x, -a.y}; }

static inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline vec3 operator*(float s, vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
static inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline vec3 &operator+=(vec3 &a, vec3 b) { a = a + b; return a; }

static inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
static inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline vec4 operator*(float s, vec4 a) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline vec4 &operator+=(vec4 &a, vec4 b) { a = a + b; return a; }

static inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline float length(vec2 a) { return sqrtf(dot(a, a)); }
static inline float length(vec3 a) { return sqrtf(dot(a, a)); }
static inline float distance(vec2 a, vec2 b) { return length(a - b); }
// GLSL normalize(v) = v * inversesqrt(dot(v,v)); expressed as v / length so that it is IEEE-exact per op.
static inline vec3 normalize(vec3 a) { float l = length(a); return {a.x / l, a.y / l, a.z / l}; }
static inline vec2 normalize(vec2 a) { float l = length(a); return {a.x / l, a.y / l}; }

static inline float clampf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }
static inline int clampi(int v, int lo, int hi) { return std::min(std::max(v, lo), hi); }
static inline float saturate(float v) { return clampf(v, 0.0f, 1.0f); }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; } // GLSL mix: x*(1-a)+y*a
static inline vec3 mix(vec3 a, vec3 b, float t) { return {mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)}; }
static inline vec4 mix(vec4 a, vec4 b, vec4 t) { return {mixf(a.x, b.x, t.x), mixf(a.y, b.y, t.y), mixf(a.z, b.z, t.z), mixf(a.w, b.w, t.w)}; }
static inline vec4 mix(vec4 a, vec4 b, float t) { return mix(a, b, V4(t)); }
static inline float smoothstep(float e0, float e1, float x)
{
	float t = saturate((x - e0) / (e1 - e0));
	return t * t * (3.0f - 2.0f * t);
}
static inline vec3 max3(vec3 a, vec3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
static inline vec3 min3(vec3 a, vec3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
static inline vec4 max4(vec4 a, vec4 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z), std::max(a.w, b.w)}; }
static inline vec4 min4(vec4 a, vec4 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z), std::min(a.w, b.w)}; }

// Column-major 4x4 like muglm/GLSL: m.c[col] is a column.
struct mat4 { vec4 c[4]; };
static inline vec4 mul(const mat4 &m, vec4 v)
{
	// GLSL M * v = sum_i column_i * v_i, evaluated left to right.
	vec4 r = m.c[0] * v.x;
	r = r + m.c[1] * v.y;
	r = r + m.c[2] * v.z;
	r = r + m.c[3] * v.w;
	return r;
}

// ---------------------------------------------------------------------------------------------
// fp16 <-> fp32.  Stores to RGBA16F attachments round to nearest even (Vulkan float16 conversion,
// what every GPU does); muglm::floatToHalf used for CPU-side light packing rounds ties *up*
// (math/muglm/muglm_impl.hpp:860-907) and is restated separately.
// ---------------------------------------------------------------------------------------------
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float half_to_float(uint16_t h)
{
	uint32_t s = uint32_t(h & 0x8000u) << 16;
	uint32_t e = (h >> 10) & 0x1fu;
	uint32_t m = h & 0x3ffu;
	if (e == 0)
	{
		if (m == 0)
			return u2f(s);
		// subnormal: value = m * 2^-24
		float v = float(m) * 5.9604644775390625e-08f;
		return (s ? -v : v);
	}
	if (e == 31)
		return u2f(s | 0x7f800000u | (m << 13));
	return u2f(s | ((e + 112u) << 23) | (m << 13));
}

static inline uint16_t float_to_half_rne(float f)
{
	uint32_t u = f2u(f);
	uint32_t s = (u >> 16) & 0x8000u;
	uint32_t a = u & 0x7fffffffu;
	if (a >= 0x7f800000u) // inf / nan
		return uint16_t(s | 0x7c00u | ((a > 0x7f800000u) ? (0x200u | ((a >> 13) & 0x3ffu)) : 0u));
	if (a >= 0x477ff000u) // >= 65520 rounds to inf
		return uint16_t(s | 0x7c00u);
	if (a < 0x38800000u) // < 2^-14: subnormal half (or zero)
	{
		if (a < 0x33000000u) // < 2^-25 -> 0 (2^-25 itself ties to even = 0)
			return uint16_t(s);
		uint32_t e = a >> 23;
		uint32_t m = (a & 0x7fffffu) | 0x800000u;
		uint32_t shift = 126u - e; // 14..24
		uint32_t q = m >> shift;
		uint32_t rem = m & ((1u << shift) - 1u);
		uint32_t half = 1u << (shift - 1u);
		if (rem > half || (rem == half && (q & 1u)))
			q++;
		return uint16_t(s | q);
	}
	uint32_t e = (a >> 23) - 112u;
	uint32_t m = a & 0x7fffffu;
	uint32_t q = (e << 10) | (m >> 13);
	uint32_t rem = m & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (q & 1u)))
		q++;
	return uint16_t(s | q);
}

// muglm::floatToHalf restated (math/muglm/muglm_impl.hpp:860-907): adds 0x2000 whenever bit 12 is set.
static inline uint16_t float_to_half_muglm(float v)
{
	int i = int(f2u(v));
	int s = (i >> 16) & 0x00008000;
	int e = ((i >> 23) & 0x000000ff) - (127 - 15);
	int m = i & 0x007fffff;
	if (e <= 0)
	{
		if (e < -10)
			return uint16_t(s);
		m = (m | 0x00800000) >> (1 - e);
		if (m & 0x00001000)
			m += 0x00002000;
		return uint16_t(s | (m >> 13));
	}
	else if (e == 0xff - (127 - 15))
	{
		if (m == 0)
			return uint16_t(s | 0x7c00);
		m >>= 13;
		return uint16_t(s | 0x7c00 | m | (m == 0));
	}
	else
	{
		if (m & 0x00001000)
		{
			m += 0x00002000;
			if (m & 0x00800000)
			{
				m = 0;
				e += 1;
			}
		}
		if (e > 30)
			return uint16_t(s | 0x7c00);
		return uint16_t(s | (e << 10) | (m >> 13));
	}
}

// ---------------------------------------------------------------------------------------------
// Storage formats.
// ---------------------------------------------------------------------------------------------
static inline vec4 load_rgba16f(const uint16_t *img, int w, int x, int y)
{
	const uint16_t *p = img + (size_t(y) * w + x) * 4;
	return {half_to_float(p[0]), half_to_float(p[1]), half_to_float(p[2]), half_to_float(p[3])};
}
static inline void store_rgba16f(uint16_t *img, int w, int x, int y, vec4 v)
{
	uint16_t *p = img + (size_t(y) * w + x) * 4;
	p[0] = float_to_half_rne(v.x);
	p[1] = float_to_half_rne(v.y);
	p[2] = float_to_half_rne(v.z);
	p[3] = float_to_half_rne(v.w);
}

// UNORM8: Vulkan float->unorm is round-to-nearest of clamp(v,0,1)*255.
static inline uint8_t float_to_unorm8(float v)
{
	if (!(v > 0.0f)) // also NaN -> 0
		return 0;
	if (v >= 1.0f)
		return 255;
	return uint8_t(int(v * 255.0f + 0.5f));
}
static inline float unorm8_to_float(uint8_t v) { return float(v) / 255.0f; }

// assets/shaders/inc/srgb.h:4-19 semantics (also the hardware sRGB transfer functions).
static inline float srgb_decode(float c)
{
	float r = (c <= 0.0404482362771082f) ? (c / 12.92f) : powf((c + 0.055f) / 1.055f, 2.4f);
	return saturate(r);
}
static inline float srgb_encode(float c)
{
	float r = (c <= 0.0031308f) ? (c * 12.92f) : (1.055f * powf(c, 1.0f / 2.4f) - 0.055f);
	return saturate(r);
}
static inline uint8_t float_to_srgb8(float linear)
{
	if (!(linear > 0.0f))
		return 0;
	return float_to_unorm8(srgb_encode(std::min(linear, 1.0f)));
}
static inline float srgb8_to_float(uint8_t v) { return srgb_decode(float(v) / 255.0f); }

// B10G11R11_UFLOAT_PACK32 (the reference's default HDR target, scene_viewer_application.cpp:881-883 with renderTargetFp16 =
// false, and its TAA output, temporal.cpp:211-213): R = bits 0..10 and G = bits 11..21 as unsigned 11-bit floats (5 exponent
// bits, bias 15, 6 mantissa bits), B = bits 22..31 as an unsigned 10-bit float (5 mantissa bits); no alpha (reads give 1).
// Every value is exactly a half float.  Conversion from fp32, as the Vulkan / OpenGL packed-float rules have it: finite values
// round to the closest representable FINITE value (ties to even here; > max -> max: 65024 / 64512), negative values and -inf
// -> 0, +inf -> +inf, NaN -> NaN.  `mant_bits` = 6 (R, G) or 5 (B).
static inline uint32_t float_to_ufloat(float f, int mant_bits)
{
	const uint32_t u = f2u(f);
	const uint32_t a = u & 0x7fffffffu;
	const uint32_t inf = 31u << mant_bits, max_finite = inf - 1u;
	if (a > 0x7f800000u) // NaN (either sign)
		return inf | 1u;
	if (u & 0x80000000u) // negative, -0, -inf
		return 0u;
	if (a == 0x7f800000u)
		return inf;
	const int shift = 23 - mant_bits;
	if (a < 0x38800000u) // < 2^-14: denormal of the packed format (or zero)
	{
		const uint32_t e = a >> 23;
		if (e < uint32_t(127 - 14 - mant_bits - 1))
			return 0u; // below half of the smallest denormal
		const uint32_t m = (a & 0x7fffffu) | 0x800000u;
		const uint32_t sh = uint32_t(shift) + (113u - e); // value = m * 2^(e - 150); unit = 2^(-14 - mant_bits)
		uint32_t q = m >> sh;
		const uint32_t rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1u);
		if (rem > half || (rem == half && (q & 1u)))
			q++;
		return q; // a carry into 1 << mant_bits is the smallest normal: the right encoding
	}
	uint32_t q = (((a >> 23) - 112u) << mant_bits) | ((a & 0x7fffffu) >> shift);
	const uint32_t rem = a & ((1u << shift) - 1u), half = 1u << (shift - 1);
	if (rem > half || (rem == half && (q & 1u)))
		q++;
	return q > max_finite ? max_finite : q;
}
static inline float ufloat_to_float(uint32_t v, int mant_bits)
{
	const uint32_t e = v >> mant_bits, m = v & ((1u << mant_bits) - 1u);
	if (e == 0)
		return float(m) * (mant_bits == 6 ? 0x1p-20f : 0x1p-19f);
	if (e == 31)
		return u2f(0x7f800000u | (m << (23 - mant_bits)));
	return u2f(((e + 112u) << 23) | (m << (23 - mant_bits)));
}
static inline uint32_t pack_b10g11r11(float r, float g, float b)
{
	return float_to_ufloat(r, 6) | (float_to_ufloat(g, 6) << 11) | (float_to_ufloat(b, 5) << 22);
}
static inline vec4 unpack_b10g11r11(uint32_t p)
{
	return {ufloat_to_float(p & 0x7ffu, 6), ufloat_to_float((p >> 11) & 0x7ffu, 6), ufloat_to_float(p >> 22, 5), 1.0f};
}
// What a B10G11R11 attachment holds after a store of (r, g, b): the same texel seen as RGBA16F (exact), alpha 1.
static inline void store_rgba16f_as_b10g11r11(uint16_t *img, int w, int x, int y, vec4 v)
{
	const vec4 q = unpack_b10g11r11(pack_b10g11r11(v.x, v.y, v.z));
	uint16_t *p = img + (size_t(y) * w + x) * 4;
	p[0] = float_to_half_rne(q.x);
	p[1] = float_to_half_rne(q.y);
	p[2] = float_to_half_rne(q.z);
	p[3] = 0x3c00u;
}

// A2B10G10R10_UNORM_PACK32: R bits 0..9, G 10..19, B 20..29, A 30..31.
static inline vec4 unpack_a2b10g10r10(uint32_t p)
{
	return {float(p & 1023u) / 1023.0f, float((p >> 10) & 1023u) / 1023.0f, float((p >> 20) & 1023u) / 1023.0f,
	        float(p >> 30) / 3.0f};
}

// ---------------------------------------------------------------------------------------------
// Software sampler over an RGBA16F linear image: StockSampler::LinearClamp / NearestClamp
// (vulkan/device.cpp:1142-1149), explicit LOD 0, unnormalised coord = uv*size-0.5, exact fp32
// weights outside the sub-texel snap below (Vulkan leaves weight precision implementation-defined;
// tolerance is stated in tests).
// ---------------------------------------------------------------------------------------------
// Sub-texel resolution of the sampler.  Vulkan leaves the precision of bilinear weights to the implementation
// (subTexelPrecisionBits; every desktop implementation, lavapipe included, reports 8).  The model here: weights are exact
// fp32, EXCEPT that a coordinate within 2^-8 of a texel centre selects that texel alone -- no implementation with <= 8
// fractional coordinate bits can resolve it from the centre, and the fp32 round trip (x + 0.5) / w * w - 0.5 of a pixel-centre
// tap (error <= 3 x 2^-24 ~ 1.5e-3 texels at x = 8192) then reads the texel itself at any image width, as a hardware sampler
// does.  A weight of exactly zero means the neighbouring texel is not part of the result (no 0 * inf).
// linear_axis: f = unnormalised coordinate - 0.5; returns the index of the first texel and the weight of the second.
constexpr float SAMPLER_SNAP = 1.0f / 256.0f;
static inline void linear_axis(float f, int &i0, float &weight)
{
	const float fl = floorf(f + SAMPLER_SNAP);
	float a = f - fl;
	if (a < SAMPLER_SNAP)
		a = 0.0f;
	i0 = int(fl);
	weight = a;
}
// The two lerps of a bilinear fetch in the order every sampler of the oracle uses; T has +, * float.
template <typename T>
static inline T linear_combine(const T &t00, const T &t10, const T &t01, const T &t11, float a, float b)
{
	const T top = (a == 0.0f) ? t00 : t00 * (1.0f - a) + t10 * a;
	if (b == 0.0f)
		return top;
	const T bot = (a == 0.0f) ? t01 : t01 * (1.0f - a) + t11 * a;
	return top * (1.0f - b) + bot * b;
}

struct Tex16F
{
	const uint16_t *data;
	int w, h;
	vec4 fetch(int x, int y) const
	{
		x = clampi(x, 0, w - 1);
		y = clampi(y, 0, h - 1);
		return load_rgba16f(data, w, x, y);
	}
	vec4 sample_linear(vec2 uv) const
	{
		float a, b;
		int x0, y0;
		linear_axis(uv.x * float(w) - 0.5f, x0, a);
		linear_axis(uv.y * float(h) - 0.5f, y0, b);
		return linear_combine(fetch(x0, y0), fetch(x0 + 1, y0), fetch(x0, y0 + 1), fetch(x0 + 1, y0 + 1), a, b);
	}
	vec4 sample_nearest(vec2 uv) const
	{
		int x = int(floorf(uv.x * float(w)));
		int y = int(floorf(uv.y * float(h)));
		return fetch(x, y);
	}
};
} // namespace orc
