// Device-side helpers shared by the gfx950 kernels: storage-format codecs and the software sampler over linear
// HBM images.  Wavefront = 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "packed_float.hpp"

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct DevImage
{
	const uint8_t *ptr;
	int w, h;
	uint32_t pitch;
};

struct DevImageRW
{
	uint8_t *ptr;
	int w, h;
	uint32_t pitch;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// UNORM8 -> float: v / 255 for an integer v in [0, 255], bit-identical to the IEEE quotient for all 256 inputs
// (tests/test_oracle_kat.py) at two VALU operations instead of a full division: 1 / 255 split into its fp32 rounding and
// the remainder, v * hi + fl(v * lo) with one rounding at the end.
__device__ __forceinline__ float unorm8_to_float(uint32_t v)
{
	const float f = float(v);
	const float hi = 0x1.010102p-8f, lo = -0x1.fdfdfep-33f;
	return fmaf(f, hi, f * lo);
}

// RGBA16F texel fetch: one 8-byte load, hardware cvt to fp32.
__device__ __forceinline__ float4 load_rgba16f(const DevImage &img, int x, int y)
{
	const f16x4 h = *reinterpret_cast<const f16x4 *>(img.ptr + size_t(y) * img.pitch + size_t(x) * 8u);
	return make_float4(float(h.x), float(h.y), float(h.z), float(h.w));
}

__device__ __forceinline__ float4 load_rgba16f_clamped(const DevImage &img, int x, int y)
{
	return load_rgba16f(img, clampi(x, 0, img.w - 1), clampi(y, 0, img.h - 1));
}

// fp32 -> fp16 with round-to-nearest-even (v_cvt_f16_f32 under the default rounding mode), as an RGBA16F attachment
// store does.
__device__ __forceinline__ f16x4 pack_rgba16f(float4 v)
{
	f16x4 h;
	h.x = _Float16(v.x);
	h.y = _Float16(v.y);
	h.z = _Float16(v.z);
	h.w = _Float16(v.w);
	return h;
}

__device__ __forceinline__ void store_rgba16f(const DevImageRW &img, int x, int y, float4 v)
{
	*reinterpret_cast<f16x4 *>(img.ptr + size_t(y) * img.pitch + size_t(x) * 8u) = pack_rgba16f(v);
}

// the value a channel holds after such a store (what the next blend reads back)
template <int MB>
__device__ __forceinline__ float round_to_ufloat(float f)
{
	return float(__builtin_bit_cast(_Float16, uint16_t(float_to_ufloat<MB>(f) << (10 - MB))));
}

// two packed texels -> the four dwords of two RGBA16F texels
__device__ __forceinline__ u32x4 expand_b10g11r11_pair(uint32_t p0, uint32_t p1)
{
	uint32_t a, b, c, d;
	expand_b10g11r11(p0, a, b);
	expand_b10g11r11(p1, c, d);
	return u32x4{a, b, c, d};
}
__device__ __forceinline__ float4 load_b10g11r11(const DevImage &img, int x, int y)
{
	uint32_t rg, ba;
	expand_b10g11r11(*reinterpret_cast<const uint32_t *>(img.ptr + size_t(y) * img.pitch + size_t(x) * 4u), rg, ba);
	const f16x4 h = __builtin_bit_cast(f16x4, u32x2{rg, ba});
	return make_float4(float(h.x), float(h.y), float(h.z), float(h.w));
}

// acc + float(half) * w in ONE instruction (v_fma_mix_f32: the fp16 -> fp32 conversion is exact and folded into the
// fused multiply-add, so the result is bit-identical to cvt + fma at 3.6 instead of 5.9 issue cycles).  `packed` holds
// two halves; _lo / _hi pick one.
__device__ __forceinline__ float fma_mix_lo(uint32_t packed, float w, float acc)
{
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(w), "v"(acc));
	return r;
}
__device__ __forceinline__ float fma_mix_hi(uint32_t packed, float w, float acc)
{
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(w), "v"(acc));
	return r;
}
// acc += texel * w for an RGBA16F texel given as two dwords (r|g, b|a)
__device__ __forceinline__ float4 fma_mix_texel(uint32_t rg, uint32_t ba, float w, float4 acc)
{
	return make_float4(fma_mix_lo(rg, w, acc.x), fma_mix_hi(rg, w, acc.y), fma_mix_lo(ba, w, acc.z), fma_mix_hi(ba, w, acc.w));
}

__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 c)
{
	return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}

// StockSampler::LinearClamp, LOD 0, unnormalised coordinate uv*size - 0.5, exact fp32 weights (+ the snap above), over any
// source of texels: fetch(x, y) takes coordinates already clamped to the image and returns the
// four channels as fp32 (sample_linear_rgba16f is this with a global-memory fetch; the fused pyramid kernels fetch a staged
// tile from LDS).  One definition, so that every path weighs and sums in the same order.
// Sub-texel resolution of the sampler (the oracle's model, oracle_common.h / aa_core.hpp): f = unnormalised coordinate - 0.5;
// a coordinate within 2^-8 of a texel centre reads that texel alone (weight 0 for its neighbour).
constexpr float SAMPLER_SNAP = 1.0f / 256.0f;
__device__ __forceinline__ void linear_axis(float f, int &i0, float &weight)
{
	const float fl = floorf(f + SAMPLER_SNAP);
	float a = f - fl;
	if (a < SAMPLER_SNAP)
		a = 0.0f;
	i0 = int(fl);
	weight = a;
}

template <typename Fetch>
__device__ __forceinline__ float4 sample_linear_with(Fetch fetch, int w, int h, float u, float v)
{
	// no contraction: a * b - c stays a product and a difference in every kernel this is inlined into, so that the same tap gives the
	// same bits whichever kernel takes it (the fused pyramid tails must equal the separate launches byte for byte)
#pragma clang fp contract(off)
	int ix, iy;
	float a, b;
	linear_axis(u * float(w) - 0.5f, ix, a);
	linear_axis(v * float(h) - 0.5f, iy, b);
	const int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
	const int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
	const float4 t00 = fetch(x0, y0), t10 = fetch(x1, y0);
	const float4 t01 = fetch(x0, y1), t11 = fetch(x1, y1);
	const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
	float4 r = t00 * w00;
	// A tap on a texel centre (weight exactly 0 after the snap) does not read its neighbour: 0 * inf would turn an overflowed fp16
	// texel next to the tap into NaN, where the oracle's linear_combine (oracle_common.h) and a hardware sampler return the texel.
	// For finite texels the sums below are bit-identical with or without the skipped terms (fma(t, 0, r) == r).
	if (a == 0.0f || b == 0.0f)
	{
		if (a != 0.0f)
			r = fma4(t10, w10, r);
		if (b != 0.0f)
			r = fma4(t01, w01, r);
		return r;
	}
	r = fma4(t10, w10, r);
	r = fma4(t01, w01, r);
	r = fma4(t11, w11, r);
	return r;
}

// StockSampler::LinearClamp, LOD 0, on an RGBA16F image: unnormalised coordinate uv*size - 0.5, exact fp32 weights.
__device__ __forceinline__ float4 sample_linear_rgba16f(const DevImage &img, float u, float v)
{
	return sample_linear_with([&img](int x, int y) { return load_rgba16f(img, x, y); }, img.w, img.h, u, v);
}

__device__ __forceinline__ float saturatef(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// Linear -> sRGB8 as an *_SRGB attachment store: encode (assets/shaders/inc/srgb.h:12-18 formula), then UNORM8 rounding.
__device__ __forceinline__ uint32_t encode_srgb8(float c)
{
	c = saturatef(c);
	const float lo = c * 12.92f;
	const float hi = fmaf(1.055f, __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(fmaxf(c, 1e-30f)) * (1.0f / 2.4f)), -0.055f);
	const float e = saturatef(c <= 0.0031308f ? lo : hi);
	return uint32_t(e * 255.0f + 0.5f);
}

// Table form of the same store (gr_ctx::srgb_encode_lut, built on the host from the formula above, staged in LDS by the
// kernels that use it).  The encode is a monotone staircase of 255 steps: the linear value's exponent and top 7 mantissa
// bits select a bucket [2^e (1 + i/128), 2^e (1 + (i+1)/128)) that holds at most one step (the densest stretch, c = 0.5,
// has 0.66 steps per bucket; with 64 buckets per octave some would hold two), so one entry {threshold, value below it} decides the byte exactly: no log / exp / pow, and
// bit-identical to the formula evaluated in fp32 on the host.
constexpr uint32_t SRGB_ENCODE_MIN_BITS = 0x39000000u;                 // 2^-13: 255 * 12.92 * 2^-13 = 0.40 -> byte 0 below
constexpr uint32_t SRGB_ENCODE_BUCKET_SHIFT = 16;                      // 23 - 7 mantissa bits
constexpr uint32_t SRGB_ENCODE_ENTRIES = 13u * 128u + 1u;              // 13 octaves x 128 buckets, + the entry of 1.0
__device__ __forceinline__ uint32_t encode_srgb8_lut(float c, const uint2 *lut)
{
	c = __builtin_amdgcn_fmed3f(c, 0x1p-13f, 1.0f); // NaN -> 2^-13 -> 0, like the formula's !(c > 0) -> 0
	const uint32_t index = (__builtin_bit_cast(uint32_t, c) - SRGB_ENCODE_MIN_BITS) >> SRGB_ENCODE_BUCKET_SHIFT;
	const uint2 e = lut[index];
	return e.y + (c >= __builtin_bit_cast(float, e.x) ? 1u : 0u);
}

// The whole tonemap of one channel as a staircase: byte = srgb8(uncharted2(x) * white_scale) is a monotone function of the
// exposed colour x alone (tonemap.frag:42-53 into an *_SRGB store), 0 below 2^-11.3 and 255 from the white point 11.2 on.
// Bucketed like encode_srgb8_lut on x itself (16 octaves from 2^-12, 64 buckets each: at most 0.81 steps per bucket), it
// replaces three fmas, a division, a pow and the rounding by one compare.  Valid for finite x >= 0 (callers send anything
// else through the formula); built on the host from the formula in fp32 (gr_tonemap_srgb8_table).
constexpr uint32_t TONEMAP_TABLE_MIN_BITS = 0x39800000u; // 2^-12
constexpr uint32_t TONEMAP_TABLE_BUCKET_SHIFT = 17;      // 23 - 6 mantissa bits
constexpr uint32_t TONEMAP_TABLE_ENTRIES = 16u * 64u + 1u;
__device__ __forceinline__ uint32_t tonemap_srgb8_lut(float x, const uint2 *lut)
{
	x = __builtin_amdgcn_fmed3f(x, 0x1p-12f, 16.0f);
	const uint32_t index = (__builtin_bit_cast(uint32_t, x) - TONEMAP_TABLE_MIN_BITS) >> TONEMAP_TABLE_BUCKET_SHIFT;
	const uint2 e = lut[index];
	return e.y + (x >= __builtin_bit_cast(float, e.x) ? 1u : 0u);
}

__device__ __forceinline__ uint32_t encode_unorm8(float c)
{
	return uint32_t(saturatef(c) * 255.0f + 0.5f);
}

// Wave64 reductions by cross-lane shuffles (no LDS).
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v += __shfl_xor(v, off, 64);
	return v;
}

__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v |= uint32_t(__shfl_xor(int(v), off, 64));
	return v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v = min(v, uint32_t(__shfl_xor(int(v), off, 64)));
	return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v = max(v, uint32_t(__shfl_xor(int(v), off, 64)));
	return v;
}
