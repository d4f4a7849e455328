// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// A small GLSL execution environment for the CPU: enough of the language's vector types (with swizzles), built-in functions
// and resource types that the REFERENCE's own shader sources -- lightly re-spelled by glsl2cpp.py into a scratch directory at
// build time, never committed -- compile as C++ and run one invocation per call.  Everything here is this repository's own
// code; it defines what the hardware defines for a shader (texture filtering, format conversion on image stores, fp32
// arithmetic with no contraction), in exactly the terms the oracle states them, so that what is compared against the oracle
// is the shader TEXT: its expressions, association order and constants.
//
//   * float is IEEE fp32; compile with -ffp-contract=off.  "mediump" is fp32 (desktop / lavapipe behaviour).
//   * min / max / clamp: IEEE minNum / maxNum.
//   * textureLod: LinearClamp or NearestClamp on a tightly packed image; texel weights (1 - a, a) from
//     a = fract(u * w - 0.5), lerp horizontally then vertically -- the oracle's sampler.
//   * imageStore / fragment outputs convert to the attachment format: RGBA16F round-to-nearest-even, UNORM8
//     floor(v * 255 + 0.5) after clamping, sRGB8 encode then UNORM8.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include "../oracle_common.h"

namespace glsl
{
using uint = uint32_t;

// ---- vectors ------------------------------------------------------------------------------------------------------------
template <typename T> struct tvec2;
template <typename T> struct tvec3;
template <typename T> struct tvec4;

// Swizzle proxies: plain arrays that sit in a union with the components and convert to the selected vector.  A swizzle that
// names consecutive components in order (.xy, .zw, .rgb ...) IS such a vector in memory and converts to a reference, so it
// can be passed to an inout parameter or compound-assigned; the others convert by value and support plain assignment.
template <typename T, int N, int A, int B, bool Contiguous = (B == A + 1)>
struct swz2;
template <typename T, int N, int A, int B>
struct swz2<T, N, A, B, false>
{
	T d[N];
	operator tvec2<T>() const;
	swz2 &operator=(const tvec2<T> &v);
};
template <typename T, int N, int A, int B>
struct swz2<T, N, A, B, true>
{
	T d[N];
	operator tvec2<T> &();
	operator const tvec2<T> &() const;
	swz2 &operator=(const tvec2<T> &v);
};
template <typename T, int N, int A, int B, int C, bool Contiguous = (B == A + 1 && C == A + 2)>
struct swz3;
template <typename T, int N, int A, int B, int C>
struct swz3<T, N, A, B, C, false>
{
	T d[N];
	operator tvec3<T>() const;
	swz3 &operator=(const tvec3<T> &v);
};
template <typename T, int N, int A, int B, int C>
struct swz3<T, N, A, B, C, true>
{
	T d[N];
	operator tvec3<T> &();
	operator const tvec3<T> &() const;
	swz3 &operator=(const tvec3<T> &v);
};
template <typename T, int N, int A, int B, int C, int D>
struct swz4
{
	T d[N];
	operator tvec4<T>() const;
	swz4 &operator=(const tvec4<T> &v);
};

template <typename T>
struct tvec2
{
	using scalar = T;
	union
	{
		struct { T x, y; };
		struct { T r, g; };
		T d[2];
#include "gen/swizzles_2.inc"
	};
	tvec2() : x(0), y(0) {}
	explicit tvec2(T s) : x(s), y(s) {}
	tvec2(T x_, T y_) : x(x_), y(y_) {}
	// GLSL converts int to uint implicitly (ffx_spd.h passes ivec2 coordinates to uvec2 parameters); everything else is spelled out
	template <typename U>
	explicit(!(std::is_same_v<U, int> && std::is_same_v<T, unsigned>)) tvec2(const tvec2<U> &o) : x(T(o.x)), y(T(o.y)) {}
	template <typename U, int N, int A, int B, bool K> explicit tvec2(const swz2<U, N, A, B, K> &o) : x(T(o.d[A])), y(T(o.d[B])) {}
	tvec2(const tvec2 &o) : x(o.x), y(o.y) {}
	tvec2 &operator=(const tvec2 &o) { x = o.x; y = o.y; return *this; }
	T &operator[](int i) { return d[i]; }
	const T &operator[](int i) const { return d[i]; }
};

template <typename T>
struct tvec3
{
	using scalar = T;
	union
	{
		struct { T x, y, z; };
		struct { T r, g, b; };
		T d[3];
#include "gen/swizzles_3.inc"
	};
	tvec3() : x(0), y(0), z(0) {}
	explicit tvec3(T s) : x(s), y(s), z(s) {}
	tvec3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
	tvec3(const tvec2<T> &v, T z_) : x(v.x), y(v.y), z(z_) {}
	tvec3(T x_, const tvec2<T> &v) : x(x_), y(v.x), z(v.y) {}
	template <typename U> explicit tvec3(const tvec3<U> &o) : x(T(o.x)), y(T(o.y)), z(T(o.z)) {}
	explicit tvec3(const tvec4<T> &o);
	tvec3(const tvec3 &o) : x(o.x), y(o.y), z(o.z) {}
	tvec3 &operator=(const tvec3 &o) { x = o.x; y = o.y; z = o.z; return *this; }
	T &operator[](int i) { return d[i]; }
	const T &operator[](int i) const { return d[i]; }
};

template <typename T>
struct tvec4
{
	using scalar = T;
	union
	{
		struct { T x, y, z, w; };
		struct { T r, g, b, a; };
		T d[4];
#include "gen/swizzles_4.inc"
	};
	tvec4() : x(0), y(0), z(0), w(0) {}
	explicit tvec4(T s) : x(s), y(s), z(s), w(s) {}
	tvec4(T x_, T y_, T z_, T w_) : x(x_), y(y_), z(z_), w(w_) {}
	tvec4(const tvec3<T> &v, T w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
	tvec4(const tvec2<T> &v, T z_, T w_) : x(v.x), y(v.y), z(z_), w(w_) {}
	tvec4(const tvec2<T> &a_, const tvec2<T> &b_) : x(a_.x), y(a_.y), z(b_.x), w(b_.y) {}
	tvec4(T x_, const tvec3<T> &v) : x(x_), y(v.x), z(v.y), w(v.z) {}
	tvec4(T x_, const tvec2<T> &v, T w_) : x(x_), y(v.x), z(v.y), w(w_) {}
	tvec4(T x_, T y_, const tvec2<T> &v) : x(x_), y(y_), z(v.x), w(v.y) {}
	template <typename U> explicit tvec4(const tvec4<U> &o) : x(T(o.x)), y(T(o.y)), z(T(o.z)), w(T(o.w)) {}
	// GLSL's implicit conversion int -> uint, component-wise (an ivec4 passed where a uvec4 parameter is declared)
	operator tvec4<uint32_t>() const requires std::is_same_v<T, int> { return tvec4<uint32_t>(uint32_t(x), uint32_t(y), uint32_t(z), uint32_t(w)); }
	tvec4(const tvec4 &o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
	tvec4 &operator=(const tvec4 &o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
	T &operator[](int i) { return d[i]; }
	const T &operator[](int i) const { return d[i]; }
};

template <typename T> tvec3<T>::tvec3(const tvec4<T> &o) : x(o.x), y(o.y), z(o.z) {}

template <typename T, int N, int A, int B> swz2<T, N, A, B, false>::operator tvec2<T>() const { return tvec2<T>(d[A], d[B]); }
template <typename T, int N, int A, int B> swz2<T, N, A, B, false> &swz2<T, N, A, B, false>::operator=(const tvec2<T> &v) { d[A] = v.x; d[B] = v.y; return *this; }
template <typename T, int N, int A, int B> swz2<T, N, A, B, true>::operator tvec2<T> &() { return *reinterpret_cast<tvec2<T> *>(&d[A]); }
template <typename T, int N, int A, int B> swz2<T, N, A, B, true>::operator const tvec2<T> &() const { return *reinterpret_cast<const tvec2<T> *>(&d[A]); }
template <typename T, int N, int A, int B> swz2<T, N, A, B, true> &swz2<T, N, A, B, true>::operator=(const tvec2<T> &v) { d[A] = v.x; d[B] = v.y; return *this; }
template <typename T, int N, int A, int B, int C> swz3<T, N, A, B, C, false>::operator tvec3<T>() const { return tvec3<T>(d[A], d[B], d[C]); }
template <typename T, int N, int A, int B, int C> swz3<T, N, A, B, C, false> &swz3<T, N, A, B, C, false>::operator=(const tvec3<T> &v) { d[A] = v.x; d[B] = v.y; d[C] = v.z; return *this; }
template <typename T, int N, int A, int B, int C> swz3<T, N, A, B, C, true>::operator tvec3<T> &() { return *reinterpret_cast<tvec3<T> *>(&d[A]); }
template <typename T, int N, int A, int B, int C> swz3<T, N, A, B, C, true>::operator const tvec3<T> &() const { return *reinterpret_cast<const tvec3<T> *>(&d[A]); }
template <typename T, int N, int A, int B, int C> swz3<T, N, A, B, C, true> &swz3<T, N, A, B, C, true>::operator=(const tvec3<T> &v) { d[A] = v.x; d[B] = v.y; d[C] = v.z; return *this; }
template <typename T, int N, int A, int B, int C, int D> swz4<T, N, A, B, C, D>::operator tvec4<T>() const { return tvec4<T>(d[A], d[B], d[C], d[D]); }
template <typename T, int N, int A, int B, int C, int D> swz4<T, N, A, B, C, D> &swz4<T, N, A, B, C, D>::operator=(const tvec4<T> &v)
{
	const tvec4<T> t = v; // the right-hand side may alias this vector
	d[A] = t.x; d[B] = t.y; d[C] = t.z; d[D] = t.w;
	return *this;
}

// GLSL float16_t (GL_EXT_shader_explicit_arithmetic_types_float16): holds the fp32 value of a half; every operation computes
// in fp32 and rounds to the nearest half -- exact for + - * and, with 24 >= 2 * 11 + 2 significand bits, for division.
// Conversion back to float is explicit, so an accidental fp32 operation on halves does not compile.
struct float16_t
{
	float v;
	float16_t() : v(0.0f) {}
	float16_t(float f) : v(orc::half_to_float(orc::float_to_half_rne(f))) {}
	explicit operator float() const { return v; }
};
inline float16_t operator+(float16_t a, float16_t b) { return float16_t(a.v + b.v); }
inline float16_t operator-(float16_t a, float16_t b) { return float16_t(a.v - b.v); }
inline float16_t operator*(float16_t a, float16_t b) { return float16_t(a.v * b.v); }
inline float16_t operator/(float16_t a, float16_t b) { return float16_t(a.v / b.v); }
inline float16_t operator-(float16_t a) { float16_t r; r.v = -a.v; return r; }
inline float16_t &operator+=(float16_t &a, float16_t b) { a = a + b; return a; }
inline float16_t &operator-=(float16_t &a, float16_t b) { a = a - b; return a; }
inline float16_t &operator*=(float16_t &a, float16_t b) { a = a * b; return a; }
inline bool operator<(float16_t a, float16_t b) { return a.v < b.v; }
inline bool operator>(float16_t a, float16_t b) { return a.v > b.v; }
inline bool operator<=(float16_t a, float16_t b) { return a.v <= b.v; }
inline bool operator>=(float16_t a, float16_t b) { return a.v >= b.v; }
inline bool operator==(float16_t a, float16_t b) { return a.v == b.v; }
inline bool operator!=(float16_t a, float16_t b) { return a.v != b.v; }

using vec2 = tvec2<float>;
using f16vec2 = tvec2<float16_t>;
using f16vec3 = tvec3<float16_t>;
using f16vec4 = tvec4<float16_t>;
using u16vec2 = tvec2<uint16_t>;
using u16vec3 = tvec3<uint16_t>;
using u16vec4 = tvec4<uint16_t>;
using i16vec2 = tvec2<int16_t>;
using i16vec3 = tvec3<int16_t>;
using i16vec4 = tvec4<int16_t>;
using vec3 = tvec3<float>;
using vec4 = tvec4<float>;
using ivec2 = tvec2<int>;
using ivec3 = tvec3<int>;
using ivec4 = tvec4<int>;
using uvec2 = tvec2<uint>;
using uvec3 = tvec3<uint>;
using uvec4 = tvec4<uint>;
using bvec2 = tvec2<bool>;
using bvec3 = tvec3<bool>;
using bvec4 = tvec4<bool>;

// ---- scalar built-ins (fp32, IEEE minNum / maxNum) -----------------------------------------------------------------------
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline uint min(uint a, uint b) { return a < b ? a : b; }
inline uint max(uint a, uint b) { return a > b ? a : b; }
inline uint16_t min(uint16_t a, uint16_t b) { return a < b ? a : b; }
inline uint16_t max(uint16_t a, uint16_t b) { return a > b ? a : b; }
inline int16_t min(int16_t a, int16_t b) { return a < b ? a : b; }
inline int16_t max(int16_t a, int16_t b) { return a > b ? a : b; }
inline int16_t abs(int16_t v) { return int16_t(v < 0 ? -v : v); }
inline float clamp(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
inline int clamp(int v, int lo, int hi) { return min(max(v, lo), hi); }
inline uint clamp(uint v, uint lo, uint hi) { return min(max(v, lo), hi); }
inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; } // GLSL: x * (1 - a) + y * a
inline float abs(float v) { return fabsf(v); }
inline int abs(int v) { return v < 0 ? -v : v; }
inline float floor(float v) { return floorf(v); }
inline float ceil(float v) { return ceilf(v); }
inline float fract(float v) { return v - floorf(v); }
inline float sqrt(float v) { return sqrtf(v); }
inline float inversesqrt(float v) { return 1.0f / sqrtf(v); }
inline float exp2(float v) { return exp2f(v); }
inline float log2(float v) { return log2f(v); }
inline float pow(float a, float b) { return powf(a, b); }
inline float exp(float v) { return expf(v); }
inline float log(float v) { return logf(v); }
inline float sin(float v) { return sinf(v); }
inline float cos(float v) { return cosf(v); }
inline float sign(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
inline float step(float edge, float v) { return v < edge ? 0.0f : 1.0f; }
inline float smoothstep(float e0, float e1, float v)
{
	const float t = clamp((v - e0) / (e1 - e0), 0.0f, 1.0f);
	return t * t * (3.0f - 2.0f * t);
}
inline tvec2<float> smoothstep(const tvec2<float> &e0, const tvec2<float> &e1, const tvec2<float> &v)
{
	return tvec2<float>(smoothstep(e0.x, e1.x, v.x), smoothstep(e0.y, e1.y, v.y));
}
inline float fma(float a, float b, float c) { return fmaf(a, b, c); }
inline uint floatBitsToUint(float v) { return orc::f2u(v); }
inline int floatBitsToInt(float v) { return int(orc::f2u(v)); }
inline float uintBitsToFloat(uint v) { return orc::u2f(v); }
inline float intBitsToFloat(int v) { return orc::u2f(uint(v)); }
inline bool isnan(float v) { return v != v; }

inline float16_t min(float16_t a, float16_t b) { float16_t r; r.v = fminf(a.v, b.v); return r; }
inline float16_t max(float16_t a, float16_t b) { float16_t r; r.v = fmaxf(a.v, b.v); return r; }
inline float16_t abs(float16_t a) { float16_t r; r.v = fabsf(a.v); return r; }
inline float16_t clamp(float16_t v, float16_t lo, float16_t hi) { return min(max(v, lo), hi); }
inline float16_t floor(float16_t a) { return float16_t(floorf(a.v)); }
inline float16_t fract(float16_t a) { return a - floor(a); }
inline float16_t sqrt(float16_t a) { return float16_t(sqrtf(a.v)); }
inline float16_t inversesqrt(float16_t a) { return float16_t(1.0f / sqrtf(a.v)); }
inline float16_t exp2(float16_t a) { return float16_t(exp2f(a.v)); }
inline float16_t log2(float16_t a) { return float16_t(log2f(a.v)); }
inline float16_t pow(float16_t a, float16_t b) { return float16_t(powf(a.v, b.v)); }
inline float16_t sign(float16_t a) { return float16_t(sign(a.v)); }
inline float16_t sin(float16_t a) { return float16_t(sinf(a.v)); }
inline float16_t cos(float16_t a) { return float16_t(cosf(a.v)); }
inline float16_t mix(float16_t a, float16_t b, float16_t t) { return a * (float16_t(1.0f) - t) + b * t; }
inline uint16_t halfBitsToUint16(float16_t a) { return orc::float_to_half_rne(a.v); }
inline float16_t uint16BitsToHalf(uint16_t b) { float16_t r; r.v = orc::half_to_float(b); return r; }
inline int16_t halfBitsToInt16(float16_t a) { return int16_t(orc::float_to_half_rne(a.v)); }
inline float16_t int16BitsToHalf(int16_t b) { return uint16BitsToHalf(uint16_t(b)); }

// ---- component-wise operators and functions -------------------------------------------------------------------------------
#define GLSL_VEC_BINOP(V, N, op)                                                                                                    \
	inline V operator op(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] op b.d[i]; return r; }       \
	inline V operator op(const V &a, V::scalar s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] op s; return r; }     \
	inline V operator op(V::scalar s, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = s op b.d[i]; return r; }     \
	inline V &operator op##=(V &a, const V &b) { for (int i = 0; i < N; i++) a.d[i] = a.d[i] op b.d[i]; return a; }               \
	inline V &operator op##=(V &a, V::scalar s) { for (int i = 0; i < N; i++) a.d[i] = a.d[i] op s; return a; }

#define GLSL_VEC_ARITH(V, N)                                                                                  \
	GLSL_VEC_BINOP(V, N, +)                                                                                     \
	GLSL_VEC_BINOP(V, N, -)                                                                                     \
	GLSL_VEC_BINOP(V, N, *)                                                                                     \
	GLSL_VEC_BINOP(V, N, /)                                                                                     \
	inline V operator-(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = -a.d[i]; return r; }

GLSL_VEC_ARITH(vec2, 2)
GLSL_VEC_ARITH(vec3, 3)
GLSL_VEC_ARITH(vec4, 4)
GLSL_VEC_ARITH(ivec2, 2)
GLSL_VEC_ARITH(ivec3, 3)
GLSL_VEC_ARITH(ivec4, 4)
GLSL_VEC_ARITH(uvec2, 2)
GLSL_VEC_ARITH(uvec3, 3)
GLSL_VEC_ARITH(uvec4, 4)
GLSL_VEC_ARITH(f16vec2, 2)
GLSL_VEC_ARITH(f16vec3, 3)
GLSL_VEC_ARITH(f16vec4, 4)
GLSL_VEC_ARITH(u16vec2, 2)
GLSL_VEC_ARITH(u16vec3, 3)
GLSL_VEC_ARITH(u16vec4, 4)

#define GLSL_HALF_FUNCS(V, N)                                                                                                       \
	inline V min(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = min(a.d[i], b.d[i]); return r; }              \
	inline V max(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = max(a.d[i], b.d[i]); return r; }              \
	inline V abs(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = abs(a.d[i]); return r; }                                  \
	inline V clamp(const V &v, const V &lo, const V &hi) { return min(max(v, lo), hi); }                                           \
	inline V floor(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = floor(a.d[i]); return r; }                              \
	inline V fract(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = fract(a.d[i]); return r; }                              \
	inline V sqrt(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = sqrt(a.d[i]); return r; }                                \
	inline V inversesqrt(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = inversesqrt(a.d[i]); return r; }                  \
	inline V exp2(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = exp2(a.d[i]); return r; }                                \
	inline V log2(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = log2(a.d[i]); return r; }                                \
	inline V sign(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = sign(a.d[i]); return r; }                                \
	inline V sin(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = sin(a.d[i]); return r; }                                  \
	inline V cos(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = cos(a.d[i]); return r; }                                  \
	inline V pow(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = pow(a.d[i], b.d[i]); return r; }              \
	inline V mix(const V &a, const V &b, const V &t) { V r; for (int i = 0; i < N; i++) r.d[i] = mix(a.d[i], b.d[i], t.d[i]); return r; }
GLSL_HALF_FUNCS(f16vec2, 2)
GLSL_HALF_FUNCS(f16vec3, 3)
GLSL_HALF_FUNCS(f16vec4, 4)
inline uint packFloat2x16(const f16vec2 &v) { return uint(halfBitsToUint16(v.x)) | (uint(halfBitsToUint16(v.y)) << 16); }
inline f16vec2 unpackFloat2x16(uint v) { return f16vec2(uint16BitsToHalf(uint16_t(v & 0xffffu)), uint16BitsToHalf(uint16_t(v >> 16))); }
inline u16vec2 halfBitsToUint16(const f16vec2 &a) { return u16vec2(halfBitsToUint16(a.x), halfBitsToUint16(a.y)); }
inline u16vec3 halfBitsToUint16(const f16vec3 &a) { return u16vec3(halfBitsToUint16(a.x), halfBitsToUint16(a.y), halfBitsToUint16(a.z)); }
inline u16vec4 halfBitsToUint16(const f16vec4 &a) { return u16vec4(halfBitsToUint16(a.x), halfBitsToUint16(a.y), halfBitsToUint16(a.z), halfBitsToUint16(a.w)); }
inline f16vec2 uint16BitsToHalf(const u16vec2 &a) { return f16vec2(uint16BitsToHalf(a.x), uint16BitsToHalf(a.y)); }
inline f16vec3 uint16BitsToHalf(const u16vec3 &a) { return f16vec3(uint16BitsToHalf(a.x), uint16BitsToHalf(a.y), uint16BitsToHalf(a.z)); }
inline f16vec4 uint16BitsToHalf(const u16vec4 &a) { return f16vec4(uint16BitsToHalf(a.x), uint16BitsToHalf(a.y), uint16BitsToHalf(a.z), uint16BitsToHalf(a.w)); }

#define GLSL_INT_OPS(V, N)                                                                                                         \
	inline V operator^(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] ^ b.d[i]; return r; }          \
	inline V operator>>(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] >> b.d[i]; return r; }        \
	inline V operator<<(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] << b.d[i]; return r; }        \
	inline V operator~(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = ~a.d[i]; return r; }                               \
	inline bool operator==(const V &a, const V &b) { for (int i = 0; i < N; i++) if (a.d[i] != b.d[i]) return false; return true; } \
	inline bool operator!=(const V &a, const V &b) { return !(a == b); }                                                          \
	inline V operator>>(const V &a, int s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] >> s; return r; }                   \
	inline V operator<<(const V &a, int s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] << s; return r; }                   \
	inline V operator&(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] & b.d[i]; return r; }          \
	inline V operator|(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] | b.d[i]; return r; }          \
	inline V operator&(const V &a, V::scalar s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] & s; return r; }               \
	inline V operator|(const V &a, V::scalar s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] | s; return r; }               \
	inline V min(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = min(a.d[i], b.d[i]); return r; }             \
	inline V max(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = max(a.d[i], b.d[i]); return r; }             \
	inline V clamp(const V &v, const V &lo, const V &hi) { return min(max(v, lo), hi); }
GLSL_INT_OPS(ivec2, 2)
inline ivec2 abs(const ivec2 &a) { return ivec2(abs(a.x), abs(a.y)); }
inline ivec3 abs(const ivec3 &a) { return ivec3(abs(a.x), abs(a.y), abs(a.z)); }
inline ivec4 abs(const ivec4 &a) { return ivec4(abs(a.x), abs(a.y), abs(a.z), abs(a.w)); }
GLSL_INT_OPS(ivec3, 3)
GLSL_INT_OPS(ivec4, 4)
GLSL_INT_OPS(uvec2, 2)
GLSL_INT_OPS(uvec3, 3)
GLSL_INT_OPS(uvec4, 4)
GLSL_VEC_ARITH(i16vec2, 2)
GLSL_VEC_ARITH(i16vec3, 3)
GLSL_VEC_ARITH(i16vec4, 4)
GLSL_INT_OPS(u16vec2, 2)
GLSL_INT_OPS(u16vec3, 3)
GLSL_INT_OPS(u16vec4, 4)
GLSL_INT_OPS(i16vec2, 2)
GLSL_INT_OPS(i16vec3, 3)
GLSL_INT_OPS(i16vec4, 4)
inline i16vec2 abs(const i16vec2 &a) { return i16vec2(abs(a.x), abs(a.y)); }
inline i16vec3 abs(const i16vec3 &a) { return i16vec3(abs(a.x), abs(a.y), abs(a.z)); }
inline i16vec4 abs(const i16vec4 &a) { return i16vec4(abs(a.x), abs(a.y), abs(a.z), abs(a.w)); }
inline uint packUint2x16(const u16vec2 &v) { return uint(v.x) | (uint(v.y) << 16); }
inline u16vec2 unpackUint2x16(uint v) { return u16vec2(uint16_t(v & 0xffffu), uint16_t(v >> 16)); }

#define GLSL_MAP1(V, N, fn) \
	inline V fn(const V &a) { V r; for (int i = 0; i < N; i++) r.d[i] = fn(a.d[i]); return r; }
#define GLSL_MAP2(V, N, fn)                                                                               \
	inline V fn(const V &a, const V &b) { V r; for (int i = 0; i < N; i++) r.d[i] = fn(a.d[i], b.d[i]); return r; } \
	inline V fn(const V &a, float s) { V r; for (int i = 0; i < N; i++) r.d[i] = fn(a.d[i], s); return r; }

#define GLSL_FLOAT_FUNCS(V, N)                                                                                                     \
	GLSL_MAP1(V, N, abs) GLSL_MAP1(V, N, floor) GLSL_MAP1(V, N, ceil) GLSL_MAP1(V, N, fract) GLSL_MAP1(V, N, sqrt)                 \
	GLSL_MAP1(V, N, inversesqrt) GLSL_MAP1(V, N, exp2) GLSL_MAP1(V, N, log2) GLSL_MAP1(V, N, sign) GLSL_MAP1(V, N, exp)            \
	GLSL_MAP1(V, N, log) GLSL_MAP1(V, N, sin) GLSL_MAP1(V, N, cos)                                                                \
	GLSL_MAP2(V, N, min) GLSL_MAP2(V, N, max) GLSL_MAP2(V, N, pow)                                                                \
	inline V clamp(const V &v, const V &lo, const V &hi) { return min(max(v, lo), hi); }                                          \
	inline V clamp(const V &v, float lo, float hi) { return min(max(v, lo), hi); }                                                \
	inline V mix(const V &a, const V &b, const V &t) { V r; for (int i = 0; i < N; i++) r.d[i] = mix(a.d[i], b.d[i], t.d[i]); return r; } \
	inline V mix(const V &a, const V &b, float t) { V r; for (int i = 0; i < N; i++) r.d[i] = mix(a.d[i], b.d[i], t); return r; } \
	inline V step(const V &e, const V &v) { V r; for (int i = 0; i < N; i++) r.d[i] = step(e.d[i], v.d[i]); return r; }          \
	inline V step(float e, const V &v) { V r; for (int i = 0; i < N; i++) r.d[i] = step(e, v.d[i]); return r; }                  \
	inline float dot(const V &a, const V &b) { float s = a.d[0] * b.d[0]; for (int i = 1; i < N; i++) s = s + a.d[i] * b.d[i]; return s; } \
	inline float length(const V &a) { return sqrtf(dot(a, a)); }                                                                  \
	inline float distance(const V &a, const V &b) { return length(a - b); }                                                       \
	inline V normalize(const V &a) { return a / length(a); } /* as the oracle states it: every step an IEEE operation */
GLSL_FLOAT_FUNCS(vec2, 2)
GLSL_FLOAT_FUNCS(vec3, 3)
GLSL_FLOAT_FUNCS(vec4, 4)

inline vec3 cross(const vec3 &a, const vec3 &b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }

// fma on vectors (SMAA's mad under SMAA_GLSL_4): one fused operation per component.
#define GLSL_FMA(V, N)                                                                                                          \
	inline V fma(const V &a, const V &b, const V &c) { V r; for (int i = 0; i < N; i++) r.d[i] = fmaf(a.d[i], b.d[i], c.d[i]); return r; } \
	inline V fma(const V &a, float b, const V &c) { V r; for (int i = 0; i < N; i++) r.d[i] = fmaf(a.d[i], b, c.d[i]); return r; }      \
	inline V fma(float a, const V &b, const V &c) { V r; for (int i = 0; i < N; i++) r.d[i] = fmaf(a, b.d[i], c.d[i]); return r; }      \
	inline V fma(const V &a, const V &b, float c) { V r; for (int i = 0; i < N; i++) r.d[i] = fmaf(a.d[i], b.d[i], c); return r; }      \
	inline V fma(const V &a, float b, float c) { V r; for (int i = 0; i < N; i++) r.d[i] = fmaf(a.d[i], b, c); return r; }
GLSL_FMA(vec2, 2)
GLSL_FMA(vec3, 3)
GLSL_FMA(vec4, 4)
inline float round(float v) { return roundf(v); }
GLSL_MAP1(vec2, 2, round)
GLSL_MAP1(vec4, 4, round)

// vec4[](a, b, c): an array temporary that decays to a pointer for the call it is written in.
struct vec4_array
{
	vec4 v[4];
	operator vec4 *() { return v; }
};
inline vec4_array array_of_vec4(const vec4 &a, const vec4 &b, const vec4 &c) { vec4_array r; r.v[0] = a; r.v[1] = b; r.v[2] = c; return r; }

// discard: leaves the invocation; the runner keeps the attachment's previous contents.
struct Discard {};

// mix with a boolean selector picks a side (GLSL 4.5 mix(x, y, bvec)).
inline vec3 mix(const vec3 &a, const vec3 &b, const bvec3 &t) { return vec3(t.x ? b.x : a.x, t.y ? b.y : a.y, t.z ? b.z : a.z); }
inline vec2 mix(const vec2 &a, const vec2 &b, const bvec2 &t) { return vec2(t.x ? b.x : a.x, t.y ? b.y : a.y); }

#define GLSL_COMPARE(V, B, N)                                                                                                        \
	inline B lessThan(const V &a, const V &b) { B r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] < b.d[i]; return r; }             \
	inline B lessThanEqual(const V &a, const V &b) { B r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] <= b.d[i]; return r; }       \
	inline B greaterThan(const V &a, const V &b) { B r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] > b.d[i]; return r; }          \
	inline B greaterThanEqual(const V &a, const V &b) { B r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] >= b.d[i]; return r; }    \
	inline B equal(const V &a, const V &b) { B r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] == b.d[i]; return r; }               \
	inline B notEqual(const V &a, const V &b) { B r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] != b.d[i]; return r; }
GLSL_COMPARE(vec2, bvec2, 2)
GLSL_COMPARE(vec3, bvec3, 3)
GLSL_COMPARE(vec4, bvec4, 4)
GLSL_COMPARE(ivec2, bvec2, 2)
GLSL_COMPARE(ivec3, bvec3, 3)
GLSL_COMPARE(uvec2, bvec2, 2)
GLSL_COMPARE(uvec3, bvec3, 3)
GLSL_COMPARE(ivec4, bvec4, 4)
GLSL_COMPARE(uvec4, bvec4, 4)
inline ivec4 mix(const ivec4 &a, const ivec4 &b, const bvec4 &t) { return ivec4(t.x ? b.x : a.x, t.y ? b.y : a.y, t.z ? b.z : a.z, t.w ? b.w : a.w); }
inline uvec4 mix(const uvec4 &a, const uvec4 &b, const bvec4 &t) { return uvec4(t.x ? b.x : a.x, t.y ? b.y : a.y, t.z ? b.z : a.z, t.w ? b.w : a.w); }
// Shift of an unsigned vector by a signed per-component count.  GLSL leaves counts >= 32 undefined; the hardware the reference
// runs on uses the low five bits, and so does this (clusterer_bindless_z_range_opt.comp replaces those lanes itself).
inline uvec4 operator<<(const uvec4 &a, const ivec4 &s) { return uvec4(a.x << (s.x & 31), a.y << (s.y & 31), a.z << (s.z & 31), a.w << (s.w & 31)); }
inline uvec4 operator>>(const uvec4 &a, const ivec4 &s) { return uvec4(a.x >> (s.x & 31), a.y >> (s.y & 31), a.z >> (s.z & 31), a.w >> (s.w & 31)); }
inline bool any(const bvec2 &v) { return v.x || v.y; }
inline bool any(const bvec3 &v) { return v.x || v.y || v.z; }
inline bool any(const bvec4 &v) { return v.x || v.y || v.z || v.w; }
inline bool all(const bvec2 &v) { return v.x && v.y; }
inline bool all(const bvec3 &v) { return v.x && v.y && v.z; }
inline bool all(const bvec4 &v) { return v.x && v.y && v.z && v.w; }
inline bvec2 operator!(const bvec2 &v) { bvec2 r; r.x = !v.x; r.y = !v.y; return r; }
inline bvec4 operator!(const bvec4 &v) { bvec4 r; for (int i = 0; i < 4; i++) r.d[i] = !v.d[i]; return r; } // GLSL not(): `not` is C++'s !
inline bool isinf(float v) { return std::isinf(v); }
inline bvec4 isinf(const vec4 &v) { bvec4 r; for (int i = 0; i < 4; i++) r.d[i] = std::isinf(v.d[i]); return r; }
inline bvec2 isinf(const vec2 &v) { bvec2 r; r.x = std::isinf(v.x); r.y = std::isinf(v.y); return r; }

// ---- matrices (column major) ----------------------------------------------------------------------------------------------
struct mat2
{
	vec2 c[2];
	mat2() {}
	explicit mat2(float d) { c[0] = vec2(d, 0.0f); c[1] = vec2(0.0f, d); }
	mat2(float a, float b, float cc, float d) { c[0] = vec2(a, b); c[1] = vec2(cc, d); } // column major
	mat2(const vec2 &a, const vec2 &b) { c[0] = a; c[1] = b; }
	vec2 &operator[](int i) { return c[i]; }
	const vec2 &operator[](int i) const { return c[i]; }
};
inline vec2 operator*(const mat2 &m, const vec2 &v) { return m.c[0] * v.x + m.c[1] * v.y; }
inline mat2 operator*(const mat2 &m, float s) { return mat2(m.c[0] * s, m.c[1] * s); }
struct mat3x2 // 3 columns of 2 components
{
	vec2 c[3];
	mat3x2() {}
	mat3x2(const vec2 &a, const vec2 &b, const vec2 &cc) { c[0] = a; c[1] = b; c[2] = cc; }
	vec2 &operator[](int i) { return c[i]; }
	const vec2 &operator[](int i) const { return c[i]; }
};
struct mat4
{
	vec4 c[4];
	mat4() {}
	mat4(const vec4 &a, const vec4 &b, const vec4 &cc, const vec4 &d) { c[0] = a; c[1] = b; c[2] = cc; c[3] = d; }
	vec4 &operator[](int i) { return c[i]; }
	const vec4 &operator[](int i) const { return c[i]; }
};
inline vec4 operator*(const mat4 &m, const vec4 &v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3] * v.w; }
struct mat3
{
	vec3 c[3];
	mat3() {}
	mat3(const vec3 &a, const vec3 &b, const vec3 &cc) { c[0] = a; c[1] = b; c[2] = cc; }
	explicit mat3(const mat4 &m) { for (int i = 0; i < 3; i++) c[i] = vec3(m.c[i]); }
	vec3 &operator[](int i) { return c[i]; }
	const vec3 &operator[](int i) const { return c[i]; }
};
inline vec3 operator*(const mat3 &m, const vec3 &v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }

// mat3x4: 3 columns of 4 components; vec4 * mat3x4 = the three column dot products.
struct mat3x4
{
	vec4 c[3];
	mat3x4() {}
	mat3x4(const vec4 &a, const vec4 &b, const vec4 &cc) { c[0] = a; c[1] = b; c[2] = cc; }
};
inline vec3 operator*(const vec4 &v, const mat3x4 &m) { return vec3(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2])); }
inline vec3 operator*(const vec3 &v, const mat3 &m) { return vec3(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2])); }

// ---- integer / packing built-ins -------------------------------------------------------------------------------------------
inline int findLSB(uint v) { return v ? __builtin_ctz(v) : -1; }
inline int findLSB(int v) { return findLSB(uint(v)); }
inline int findMSB(uint v) { return v ? 31 - __builtin_clz(v) : -1; }
inline ivec4 findLSB(const uvec4 &v) { return ivec4(findLSB(v.x), findLSB(v.y), findLSB(v.z), findLSB(v.w)); }
inline ivec4 findMSB(const uvec4 &v) { return ivec4(findMSB(v.x), findMSB(v.y), findMSB(v.z), findMSB(v.w)); }
inline int bitCount(uint v) { return __builtin_popcount(v); }
inline vec2 unpackHalf2x16(uint v) { return vec2(orc::half_to_float(uint16_t(v & 0xffffu)), orc::half_to_float(uint16_t(v >> 16))); }
inline uvec2 floatBitsToUint(const vec2 &v) { return uvec2(floatBitsToUint(v.x), floatBitsToUint(v.y)); }
inline uvec3 floatBitsToUint(const vec3 &v) { return uvec3(floatBitsToUint(v.x), floatBitsToUint(v.y), floatBitsToUint(v.z)); }
inline uvec4 floatBitsToUint(const vec4 &v) { return uvec4(floatBitsToUint(v.x), floatBitsToUint(v.y), floatBitsToUint(v.z), floatBitsToUint(v.w)); }
inline vec2 uintBitsToFloat(const uvec2 &v) { return vec2(uintBitsToFloat(v.x), uintBitsToFloat(v.y)); }
inline vec3 uintBitsToFloat(const uvec3 &v) { return vec3(uintBitsToFloat(v.x), uintBitsToFloat(v.y), uintBitsToFloat(v.z)); }
inline vec4 uintBitsToFloat(const uvec4 &v) { return vec4(uintBitsToFloat(v.x), uintBitsToFloat(v.y), uintBitsToFloat(v.z), uintBitsToFloat(v.w)); }
inline uint packHalf2x16(const vec2 &v) { return uint(orc::float_to_half_rne(v.x)) | (uint(orc::float_to_half_rne(v.y)) << 16); }
inline uint bitfieldInsert(uint base, uint insert, int offset, int bits)
{
	if (bits == 0)
		return base;
	const uint mask = (bits == 32 ? 0xffffffffu : ((1u << bits) - 1u)) << offset;
	return (base & ~mask) | ((insert << offset) & mask);
}
inline uint bitfieldExtract(uint v, int offset, int bits) { return bits == 0 ? 0u : (v >> offset) & (bits == 32 ? 0xffffffffu : ((1u << bits) - 1u)); }

// ---- subgroup operations of a one-invocation subgroup (each invocation runs alone: the exact per-pixel form) -----------------
template <typename T> inline T subgroupMin(const T &v) { return v; }
template <typename T> inline T subgroupMax(const T &v) { return v; }
template <typename T> inline T subgroupOr(const T &v) { return v; }
template <typename T> inline T subgroupBroadcastFirst(const T &v) { return v; }
template <typename T> inline const T &nonuniformEXT(const T &v) { return v; }

// ---- resources ------------------------------------------------------------------------------------------------------------
enum class Format { RGBA16F, RGBA8_UNORM, RGBA8_SRGB, R32F, RG16F, RG8_UNORM, R8_UNORM, A2B10G10R10_UNORM, R16F, B10G11R11_UFLOAT };
enum class Filter { Linear, Nearest };

struct Texture
{
	const void *data = nullptr;
	int w = 0, h = 0;
	Format format = Format::RGBA16F;
	Filter filter = Filter::Linear;

	vec4 texel(int x, int y) const
	{
		x = orc::clampi(x, 0, w - 1);
		y = orc::clampi(y, 0, h - 1);
		const size_t i = size_t(y) * w + x;
		switch (format)
		{
		case Format::RGBA16F:
		{
			const uint16_t *p = static_cast<const uint16_t *>(data) + i * 4;
			return vec4(orc::half_to_float(p[0]), orc::half_to_float(p[1]), orc::half_to_float(p[2]), orc::half_to_float(p[3]));
		}
		case Format::RGBA8_UNORM:
		{
			const uint8_t *p = static_cast<const uint8_t *>(data) + i * 4;
			return vec4(float(p[0]) / 255.0f, float(p[1]) / 255.0f, float(p[2]) / 255.0f, float(p[3]) / 255.0f);
		}
		case Format::RGBA8_SRGB:
		{
			const uint8_t *p = static_cast<const uint8_t *>(data) + i * 4;
			return vec4(orc::srgb8_to_float(p[0]), orc::srgb8_to_float(p[1]), orc::srgb8_to_float(p[2]), float(p[3]) / 255.0f);
		}
		case Format::R32F:
			return vec4(static_cast<const float *>(data)[i], 0.0f, 0.0f, 1.0f);
		case Format::RG16F:
		{
			const uint16_t *p = static_cast<const uint16_t *>(data) + i * 2;
			return vec4(orc::half_to_float(p[0]), orc::half_to_float(p[1]), 0.0f, 1.0f);
		}
		case Format::RG8_UNORM:
		{
			const uint8_t *p = static_cast<const uint8_t *>(data) + i * 2;
			return vec4(float(p[0]) / 255.0f, float(p[1]) / 255.0f, 0.0f, 1.0f);
		}
		case Format::R8_UNORM:
			return vec4(float(static_cast<const uint8_t *>(data)[i]) / 255.0f, 0.0f, 0.0f, 1.0f);
		case Format::A2B10G10R10_UNORM:
		{
			const orc::vec4 v = orc::unpack_a2b10g10r10(static_cast<const uint32_t *>(data)[i]);
			return vec4(v.x, v.y, v.z, v.w);
		}
		case Format::R16F:
			return vec4(orc::half_to_float(static_cast<const uint16_t *>(data)[i]), 0.0f, 0.0f, 1.0f);
		case Format::B10G11R11_UFLOAT:
		{
			const orc::vec4 v = orc::unpack_b10g11r11(static_cast<const uint32_t *>(data)[i]);
			return vec4(v.x, v.y, v.z, 1.0f);
		}
		}
		return vec4();
	}

	vec4 sample(const vec2 &uv, int ox = 0, int oy = 0) const
	{
		if (filter == Filter::Nearest)
			return texel(int(floorf(uv.x * float(w))) + ox, int(floorf(uv.y * float(h))) + oy);
		// the sampler model stated once in oracle_common.h: exact fp32 weights + the sub-texel snap onto texel centres
		float a, b;
		int x0, y0;
		orc::linear_axis(uv.x * float(w) - 0.5f, x0, a);
		orc::linear_axis(uv.y * float(h) - 0.5f, y0, b);
		x0 += ox;
		y0 += oy;
		return orc::linear_combine(texel(x0, y0), texel(x0 + 1, y0), texel(x0, y0 + 1), texel(x0 + 1, y0 + 1), a, b);
	}
};
using sampler2D = Texture;
using texture2D = Texture;
using subpassInput = Texture; // an input attachment: the texel under the fragment

inline vec4 textureLod(const Texture &t, const vec2 &uv, float) { return t.sample(uv); }
inline vec4 texture(const Texture &t, const vec2 &uv) { return t.sample(uv); }
inline vec4 textureLodOffset(const Texture &t, const vec2 &uv, float, const ivec2 &o) { return t.sample(uv, o.x, o.y); }
inline vec4 texelFetch(const Texture &t, const ivec2 &p, int) { return t.texel(p.x, p.y); }
// textureGather: the 2 x 2 footprint a bilinear fetch at uv would read, one component: (x, y, z, w) = texels
// (i0, j0 + 1), (i0 + 1, j0 + 1), (i0 + 1, j0), (i0, j0) with (i0, j0) = floor(uv * size - 0.5); coordinates clamp per texel.
inline vec4 textureGatherOffset(const Texture &t, const vec2 &uv, const ivec2 &o, int comp = 0)
{
	const int i0 = int(floorf(uv.x * float(t.w) - 0.5f)) + o.x, j0 = int(floorf(uv.y * float(t.h) - 0.5f)) + o.y;
	return vec4(t.texel(i0, j0 + 1).d[comp], t.texel(i0 + 1, j0 + 1).d[comp], t.texel(i0 + 1, j0).d[comp], t.texel(i0, j0).d[comp]);
}
inline vec4 textureGather(const Texture &t, const vec2 &uv, int comp = 0) { return textureGatherOffset(t, uv, ivec2(0, 0), comp); }
inline ivec2 textureSize(const Texture &t, int) { return ivec2(t.w, t.h); }

struct Image
{
	void *data = nullptr;
	int w = 0, h = 0;
	Format format = Format::RGBA16F;
};
using image2D = Image;

inline void imageStore(Image &img, const ivec2 &p, const vec4 &v)
{
	if (p.x < 0 || p.y < 0 || p.x >= img.w || p.y >= img.h)
		return;
	const size_t i = size_t(p.y) * img.w + p.x;
	switch (img.format)
	{
	case Format::RGBA16F:
	{
		uint16_t *o = static_cast<uint16_t *>(img.data) + i * 4;
		for (int c = 0; c < 4; c++)
			o[c] = orc::float_to_half_rne(v.d[c]);
		break;
	}
	case Format::RGBA8_UNORM:
	{
		uint8_t *o = static_cast<uint8_t *>(img.data) + i * 4;
		for (int c = 0; c < 4; c++)
			o[c] = orc::float_to_unorm8(v.d[c]);
		break;
	}
	case Format::RGBA8_SRGB:
	{
		uint8_t *o = static_cast<uint8_t *>(img.data) + i * 4;
		for (int c = 0; c < 3; c++)
			o[c] = orc::float_to_srgb8(v.d[c]);
		o[3] = orc::float_to_unorm8(v.d[3]);
		break;
	}
	case Format::R32F:
		static_cast<float *>(img.data)[i] = v.x;
		break;
	case Format::R16F:
		static_cast<uint16_t *>(img.data)[i] = orc::float_to_half_rne(v.x);
		break;
	case Format::B10G11R11_UFLOAT: // the packed-float store conversion, stated once (oracle_common.h: float_to_ufloat)
		static_cast<uint32_t *>(img.data)[i] = orc::pack_b10g11r11(v.x, v.y, v.z);
		break;
	case Format::R8_UNORM:
		static_cast<uint8_t *>(img.data)[i] = orc::float_to_unorm8(v.x);
		break;
	default:
		break;
	}
}

// imageLoad: texels outside the image read as zero (robust access).
inline vec4 imageLoad(const Image &img, const ivec2 &p)
{
	if (p.x < 0 || p.y < 0 || p.x >= img.w || p.y >= img.h)
		return vec4(0.0f);
	const size_t i = size_t(p.y) * img.w + p.x;
	if (img.format == Format::R32F)
		return vec4(static_cast<const float *>(img.data)[i], 0.0f, 0.0f, 1.0f);
	Texture t;
	t.data = img.data;
	t.w = img.w;
	t.h = img.h;
	t.format = img.format;
	return t.texel(p.x, p.y);
}
inline void memoryBarrierImage() {}
inline uint atomicAdd(uint &mem, uint v) { return __atomic_fetch_add(&mem, v, __ATOMIC_SEQ_CST); }

// ---- per-invocation built-in variables --------------------------------------------------------------------------------------
inline thread_local uvec3 gl_GlobalInvocationID, gl_LocalInvocationID, gl_WorkGroupID;
inline thread_local uint gl_LocalInvocationIndex;
inline thread_local vec4 gl_FragCoord, gl_Position;
inline thread_local uint gl_SubgroupSize = 1, gl_SubgroupInvocationID = 0, gl_NumSubgroups = 1, gl_SubgroupID = 0;
inline uint atomicOr(uint &mem, uint v) { return __atomic_fetch_or(&mem, v, __ATOMIC_SEQ_CST); }
inline vec4 subpassLoad(const Texture &t) { return t.texel(int(gl_FragCoord.x), int(gl_FragCoord.y)); }
} // namespace glsl

// Qualifiers that mean nothing on the CPU.
#define mediump
#define highp
#define lowp
#define uniform
#define writeonly
#define readonly
#define coherent
#define restrict
#define shared
#define buffer
#define precise
#define discard throw glsl::Discard()
