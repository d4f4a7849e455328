// Tiny fp32 vector types for kernels that must evaluate in a fixed association order (compiled with
// -ffp-contract=off): plain structs, every operator is exactly one IEEE op per component.
#pragma once
#include <hip/hip_runtime.h>

struct v2 { float x, y; };
struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };

__device__ __forceinline__ v2 mk2(float x, float y) { return {x, y}; }
__device__ __forceinline__ v3 mk3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ v4 mk4(float x, float y, float z, float w) { return {x, y, z, w}; }
__device__ __forceinline__ v2 operator+(v2 a, v2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ v2 operator-(v2 a, v2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2 operator*(v2 a, v2 b) { return {a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ v2 operator*(v2 a, float s) { return {a.x * s, a.y * s}; }
__device__ __forceinline__ v2 operator-(v2 a) { return {-a.x, -a.y}; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ v4 operator+(v4 a, v4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ v4 operator*(v4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
__device__ __forceinline__ float dot2(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len2(v2 a) { return sqrtf(dot2(a, a)); }
__device__ __forceinline__ float dist2(v2 a, v2 b) { return len2(a - b); }
__device__ __forceinline__ float cross_2d(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ v3 mix3(v3 a, v3 b, float t) { return {mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)}; }
__device__ __forceinline__ v4 mix4(v4 a, v4 b, float t)
{
	return {mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t), mixf(a.w, b.w, t)};
}
__device__ __forceinline__ v3 xyz(v4 v) { return {v.x, v.y, v.z}; }
__device__ __forceinline__ v2 xy(v3 v) { return {v.x, v.y}; }
__device__ __forceinline__ float signf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

__device__ __forceinline__ v2 operator*(float s, v2 a) { return {a.x * s, a.y * s}; }
__device__ __forceinline__ v2 operator/(v2 a, v2 b) { return {a.x / b.x, a.y / b.y}; }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ v3 operator/(v3 a, v3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
__device__ __forceinline__ v3 operator*(float s, v3 a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ v4 operator*(float s, v4 a) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
__device__ __forceinline__ v2 fma2(v2 a, v2 b, v2 c) { return {fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ v3 min3v(v3 a, v3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
__device__ __forceinline__ v3 max3v(v3 a, v3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
__device__ __forceinline__ float clampfv(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float stepf(float edge, float x) { return x >= edge ? 1.0f : 0.0f; }
