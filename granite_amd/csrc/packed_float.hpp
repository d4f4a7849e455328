// B10G11R11_UFLOAT_PACK32 texels for the gfx950 kernels and for the host emulation of the CPU tests (tests/cpp/hip_emu.hpp):
// plain integer code, the same on both sides.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define GR_PF_HD __host__ __device__ __forceinline__
#else
#define GR_PF_HD inline
#endif

GR_PF_HD uint32_t gr_pf_bits(float f)
{
	uint32_t u;
	memcpy(&u, &f, 4);
	return u;
}
GR_PF_HD float gr_pf_float(uint32_t u)
{
	float f;
	memcpy(&f, &u, 4);
	return f;
}

// B10G11R11_UFLOAT_PACK32 (the reference's default HDR target and its TAA output): R, G unsigned 11-bit floats (6 mantissa bits),
// B an unsigned 10-bit float (5), exponent as in a half float -- every packed value IS a half float, so a texel expands exactly
// into the two dwords (r | g, b | 1.0) of the RGBA16F texel the kernels already consume.
GR_PF_HD void expand_b10g11r11(uint32_t p, uint32_t &rg, uint32_t &ba)
{
	rg = ((p & 0x7ffu) << 4) | ((p & 0x3ff800u) << 9); // r: bits 0..10 -> 4..14; g: bits 11..21 -> 20..30
	ba = ((p >> 22) << 5) | 0x3c000000u;
}
// fp32 -> unsigned MB-mantissa-bit float, the attachment store conversion as the oracle states it (oracle_common.h: float_to_ufloat):
// the closest representable finite value, ties to even; above the largest finite value -> it; negative, -0, -inf -> 0; +inf -> +inf;
// NaN -> NaN.  Straight-line integer code.
template <int MB>
GR_PF_HD uint32_t float_to_ufloat(float f)
{
	constexpr int SHIFT = 23 - MB;
	constexpr uint32_t INF = 31u << MB, MAX_FINITE = INF - 1u;
	const uint32_t u = gr_pf_bits(f), a = u & 0x7fffffffu;
	// normal range: drop SHIFT mantissa bits with round-to-nearest-even, re-bias the exponent (127 -> 15)
	const uint32_t qn = ((a + ((1u << (SHIFT - 1)) - 1u) + ((a >> SHIFT) & 1u)) >> SHIFT) - (112u << MB);
	// below 2^-14: a multiple of the format's denormal unit 2^(-14 - MB); the scaling is exact, v_rndne rounds to even
	const bool small = a < 0x38800000u;
	const uint32_t qd = uint32_t(__builtin_rintf(gr_pf_float(small ? a : 0u) * float(1u << (14 + MB))));
	uint32_t q = small ? qd : qn;
	q = q < MAX_FINITE ? q : MAX_FINITE;
	q = a == 0x7f800000u ? INF : q;
	q = (u >> 31) ? 0u : q;
	return a > 0x7f800000u ? (INF | 1u) : q;
}
GR_PF_HD uint32_t pack_b10g11r11(float r, float g, float b)
{
	return float_to_ufloat<6>(r) | (float_to_ufloat<6>(g) << 11) | (float_to_ufloat<5>(b) << 22);
}
