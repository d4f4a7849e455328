// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// Minimal column-major vector/matrix types with muglm's conventions (math/muglm/muglm.hpp): mat4 m[col][row],
// mat_affine = 3 row vec4.  Only what the image-space host code needs.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace Granite
{
struct vec2
{
	float x = 0, y = 0;
	vec2() = default;
	vec2(float x_, float y_) : x(x_), y(y_) {}
	explicit vec2(float s) : x(s), y(s) {}
};
struct uvec2
{
	uint32_t x = 0, y = 0;
	uvec2() = default;
	uvec2(uint32_t x_, uint32_t y_) : x(x_), y(y_) {}
};
struct ivec2
{
	int32_t x = 0, y = 0;
	ivec2() = default;
	ivec2(int32_t x_, int32_t y_) : x(x_), y(y_) {}
};
struct vec3
{
	float x = 0, y = 0, z = 0;
	vec3() = default;
	vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
	explicit vec3(float s) : x(s), y(s), z(s) {}
	float &operator[](int i) { return (&x)[i]; }
	float operator[](int i) const { return (&x)[i]; }
};
struct vec4
{
	float x = 0, y = 0, z = 0, w = 0;
	vec4() = default;
	vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
	vec4(const vec3 &v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
	explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
	float &operator[](int i) { return (&x)[i]; }
	float operator[](int i) const { return (&x)[i]; }
	vec3 xyz() const { return {x, y, z}; }
};

inline vec3 operator+(const vec3 &a, const vec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(const vec3 &a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, const vec3 &a) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(const vec3 &a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator-(const vec3 &a) { return {-a.x, -a.y, -a.z}; }
inline vec4 operator+(const vec4 &a, const vec4 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator*(const vec4 &a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline float dot(const vec3 &a, const vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(const vec3 &a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(const vec3 &a) { float l = length(a); return {a.x / l, a.y / l, a.z / l}; }
inline vec3 cross(const vec3 &a, const vec3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct mat4
{
	vec4 c[4];
	mat4() : mat4(1.0f) {}
	explicit mat4(float d) { c[0] = {d, 0, 0, 0}; c[1] = {0, d, 0, 0}; c[2] = {0, 0, d, 0}; c[3] = {0, 0, 0, d}; }
	vec4 &operator[](int i) { return c[i]; }
	const vec4 &operator[](int i) const { return c[i]; }
	const float *data() const { return &c[0].x; }
	float *data() { return &c[0].x; }
};

inline vec4 operator*(const mat4 &m, const vec4 &v)
{
	vec4 r = m[0] * v.x;
	r = r + m[1] * v.y;
	r = r + m[2] * v.z;
	r = r + m[3] * v.w;
	return r;
}
inline mat4 operator*(const mat4 &a, const mat4 &b)
{
	mat4 r(0.0f);
	for (int i = 0; i < 4; i++)
		r[i] = a * b[i];
	return r;
}

mat4 inverse(const mat4 &m);
mat4 translate(const vec3 &v);
mat4 scale(const vec3 &v);
// muglm::perspective (math/muglm/muglm.cpp:319-337): reverse-Z, Vulkan clip space with Y flip.
mat4 perspective(float fovy, float aspect, float z_near, float z_far);
mat4 look_at(const vec3 &eye, const vec3 &center, const vec3 &up);

struct mat_affine
{
	vec4 rows[3];
	mat_affine() { rows[0] = {1, 0, 0, 0}; rows[1] = {0, 1, 0, 0}; rows[2] = {0, 0, 1, 0}; }
	vec4 &operator[](int i) { return rows[i]; }
	const vec4 &operator[](int i) const { return rows[i]; }
	float get_uniform_scale() const { return length(rows[0].xyz()); }
	vec3 get_translation() const { return {rows[0].w, rows[1].w, rows[2].w}; }
	vec3 get_forward() const { return {-rows[0].z, -rows[1].z, -rows[2].z}; }
	vec3 get_right() const { return {rows[0].x, rows[1].x, rows[2].x}; }
	vec3 get_up() const { return {rows[0].y, rows[1].y, rows[2].y}; }
};

// muglm::floatToHalf (math/muglm/muglm_impl.hpp:860-907): rounds ties upward.
uint16_t floatToHalf(float v);
inline uint32_t floatToHalf2(float a, float b) { return uint32_t(floatToHalf(a)) | (uint32_t(floatToHalf(b)) << 16); }
} // namespace Granite
