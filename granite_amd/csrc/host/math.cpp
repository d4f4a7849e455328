// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
#include "math.hpp"

namespace Granite
{
mat4 inverse(const mat4 &m)
{
	// Cofactor expansion in double precision, rounded once to fp32.
	double a[16], inv[16];
	const float *src = m.data();
	for (int i = 0; i < 16; i++)
		a[i] = src[i];

	inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
	inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
	inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
	inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
	inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
	inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
	inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
	inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
	inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
	inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
	inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
	inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
	inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
	inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
	inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
	inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];

	double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
	double inv_det = 1.0 / det;
	mat4 r(0.0f);
	float *dst = r.data();
	for (int i = 0; i < 16; i++)
		dst[i] = float(inv[i] * inv_det);
	return r;
}

mat4 translate(const vec3 &v)
{
	mat4 m(1.0f);
	m[3] = vec4(v, 1.0f);
	return m;
}

mat4 scale(const vec3 &v)
{
	mat4 m(1.0f);
	m[0].x = v.x;
	m[1].y = v.y;
	m[2].z = v.z;
	return m;
}

mat4 perspective(float fovy, float aspect, float z_near, float z_far)
{
	float t = std::tan(fovy / 2.0f);
	mat4 r(0.0f);
	r[0][0] = 1.0f / (aspect * t);
	r[1][1] = 1.0f / t;
	r[2][2] = -1.0f - z_far / (z_near - z_far);
	r[3][2] = -(z_far * z_near) / (z_near - z_far);
	r[2][3] = -1.0f;
	for (int i = 0; i < 4; i++)
		r[i].y *= -1.0f;
	return r;
}

mat4 look_at(const vec3 &eye, const vec3 &center, const vec3 &up)
{
	vec3 f = normalize(center - eye);
	vec3 s = normalize(cross(f, up));
	vec3 u = cross(s, f);
	mat4 m(1.0f);
	m[0] = {s.x, u.x, -f.x, 0.0f};
	m[1] = {s.y, u.y, -f.y, 0.0f};
	m[2] = {s.z, u.z, -f.z, 0.0f};
	m[3] = {-dot(s, eye), -dot(u, eye), dot(f, eye), 1.0f};
	return m;
}

uint16_t floatToHalf(float v)
{
	uint32_t bits;
	memcpy(&bits, &v, sizeof(bits));
	const uint32_t sign = (bits >> 16) & 0x8000u;
	int exponent = int((bits >> 23) & 0xffu) - 112; // rebias 127 -> 15
	uint32_t mantissa = bits & 0x7fffffu;

	if (exponent >= 143) // source was inf / nan
	{
		if (mantissa == 0)
			return uint16_t(sign | 0x7c00u);
		mantissa >>= 13;
		return uint16_t(sign | 0x7c00u | mantissa | (mantissa == 0 ? 1u : 0u));
	}
	if (exponent <= 0)
	{
		if (exponent < -10)
			return uint16_t(sign);
		mantissa = (mantissa | 0x800000u) >> (1 - exponent);
		if (mantissa & 0x1000u)
			mantissa += 0x2000u; // ties (and everything above half) go up
		return uint16_t(sign | (mantissa >> 13));
	}
	if (mantissa & 0x1000u)
	{
		mantissa += 0x2000u;
		if (mantissa & 0x800000u)
		{
			mantissa = 0;
			exponent++;
		}
	}
	if (exponent > 30)
		return uint16_t(sign | 0x7c00u);
	return uint16_t(sign | (uint32_t(exponent) << 10) | (mantissa >> 13));
}
} // namespace Granite
