// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  The reference has no test or golden data for this pass; pinned
// by executing the reference's own spd.comp + ffx_spd.h on the CPU (oracle/ref_build/ref_spd.cpp,
// tests/test_reference_shaders_cpu.py, bit for bit) and by analytic known-answer cases (tests/test_oracle_spd_cpu.py).
//
// Single-pass downsampler: emit_single_pass_downsample (renderer/post/spd.cpp:56-102) + assets/shaders/post/ffx-spd/spd.comp
// with the defines that function sets (SUBGROUP, SINGLE_INPUT_TAP -> SPD_LINEAR_SAMPLER, COMPONENTS, FILTER_MOD,
// REDUCTION_MODE) over AMD FidelityFX SPD (ffx_spd.h, vendored by the reference).  What the arithmetic comes to:
//   * one workgroup per 64 x 64 source texels.  Output level 0 = one sampler tap per texel at the centre of its 2 x 2 source
//     footprint (spd.comp:74-89, ffx_spd.h:491-494, :505-530).  Levels 1..5 are reduced from the UNROUNDED fp32 values of
//     the level above inside the workgroup -- quad swaps / LDS, ffx_spd.h:411-417, :538-560, :646-763 -- always as
//     ((tl + tr) + bl) + br (quad lane 0 = top left, ARmpRed8x8).
//   * the last workgroup reads level 5 back FROM THE IMAGE (rounded to the storage format, filter_mod applied, coordinates
//     clamped, spd.comp:150-156) and reduces it as ((p(0,0) + p(0,1)) + p(1,0)) + p(1,1) -- column first, ffx_spd.h:472-480 --
//     into level 6; levels 7..11 again from unrounded values in the ((tl + tr) + bl) + br order (:766-795, :796-813).
//   * stores are bounds-checked against max(base >> mip, 1), multiplied by filter_mods[mip] and chopped to COMPONENTS
//     (spd.comp:91-102); texels of a level that no workgroup reaches keep their previous contents.
#include "oracle_common.h"
#include <vector>

using namespace orc;

namespace
{
struct Params
{
	int w0, h0, mips, components, depth_mode;
	const float *filter_mods;
	uint16_t *chain;
};

size_t level_offset(const Params &p, int level)
{
	size_t texels = 0;
	for (int l = 0; l < level; l++)
		texels += size_t(std::max(p.w0 >> l, 1)) * size_t(std::max(p.h0 >> l, 1));
	return texels * 4;
}

vec4 chop(vec4 v, int components)
{
	if (components < 4)
		v.w = 0.0f;
	if (components < 3)
		v.z = 0.0f;
	if (components < 2)
		v.y = 0.0f;
	return v;
}

// SpdReduce4 (spd.comp:177-187)
vec4 reduce4(const Params &p, vec4 v0, vec4 v1, vec4 v2, vec4 v3)
{
	if (p.depth_mode)
		return {std::min(std::min(v0.x, v1.x), std::min(v2.x, v3.x)), 0.0f, 0.0f, 0.0f};
	return chop((v0 + v1 + v2 + v3) * 0.25f, p.components);
}

// SpdStore (spd.comp:91-102)
void store(const Params &p, int x, int y, vec4 value, int mip)
{
	const int mw = std::max(p.w0 >> mip, 1), mh = std::max(p.h0 >> mip, 1);
	if (x >= mw || y >= mh)
		return;
	if (p.filter_mods)
	{
		const float *m = p.filter_mods + 4 * mip;
		value = {value.x * m[0], value.y * m[1], value.z * m[2], value.w * m[3]};
	}
	store_rgba16f(p.chain + level_offset(p, mip), mw, x, y, chop(value, p.components));
}

// Levels `first_mip + 1 ...` from an n x n block of values of level `first_mip` whose top-left texel is (n * gx, n * gy) of
// that level: SpdDownsampleNextFour and the quad half of SpdDownsampleMips_0_1.
void reduce_block(const Params &p, std::vector<vec4> &block, int n, int gx, int gy, int first_mip)
{
	int mip = first_mip + 1;
	while (n > 1 && mip < p.mips)
	{
		const int half = n / 2;
		std::vector<vec4> next(size_t(half) * half);
		for (int y = 0; y < half; y++)
			for (int x = 0; x < half; x++)
			{
				const vec4 v = reduce4(p, block[size_t(2 * y) * n + 2 * x], block[size_t(2 * y) * n + 2 * x + 1],
				                       block[size_t(2 * y + 1) * n + 2 * x], block[size_t(2 * y + 1) * n + 2 * x + 1]);
				next[size_t(y) * half + x] = v;
				store(p, half * gx + x, half * gy + y, v, mip);
			}
		block.swap(next);
		n = half;
		mip++;
	}
}
} // namespace

extern "C" {

// input: iw x ih RGBA16F.  chain: RGBA16F levels, level l = max(w0 >> l, 1) x max(h0 >> l, 1) texels, tightly packed one after the
// other (w0 x h0 = base_image_resolution = the size of output_mips[0]).  mips <= 12, components 1..4,
// filter_mods = mips x vec4 or NULL.
void orc_spd(const uint16_t *input, int iw, int ih, int w0, int h0, int mips, int components, int depth_mode,
             const float *filter_mods, uint16_t *chain)
{
	const Params p = {w0, h0, mips, components, depth_mode, filter_mods, chain};
	const Tex16F tex = {input, iw, ih};
	const vec2 inv = {1.0f / float(iw), 1.0f / float(ih)}; // spd.cpp:87-88
	const int groups_x = (w0 + 31) / 32, groups_y = (h0 + 31) / 32;

	for (int gy = 0; gy < groups_y; gy++)
		for (int gx = 0; gx < groups_x; gx++)
		{
			std::vector<vec4> level0(32 * 32);
			for (int y = 0; y < 32; y++)
				for (int x = 0; x < 32; x++)
				{
					// SpdLoadSourceImage at p = 2 * output texel (ffx_spd.h:509-510)
					const vec2 pf = {float(64 * gx + 2 * x), float(64 * gy + 2 * y)};
					vec4 v;
					if (depth_mode)
						v = tex.sample_nearest({0.5f * (pf.x * inv.x + inv.x), 0.5f * (pf.y * inv.y + inv.y)});
					else
						v = tex.sample_linear({pf.x * inv.x + inv.x, pf.y * inv.y + inv.y});
					v = chop(v, components);
					level0[size_t(y) * 32 + x] = v;
					store(p, 32 * gx + x, 32 * gy + y, v, 0);
				}
			reduce_block(p, level0, 32, gx, gy, 0); // levels 1..5: mip < min(mips, 6) because n reaches 1 at level 5
		}

	if (mips <= 6)
		return;

	// The last workgroup (ffx_spd.h:826-837): level 6 from the stored level 5.
	const int w5 = std::max(w0 >> 5, 1), h5 = std::max(h0 >> 5, 1);
	const uint16_t *level5 = chain + level_offset(p, 5);
	auto load5 = [&](int x, int y) {
		return chop(load_rgba16f(level5, w5, clampi(x, 0, w5 - 1), clampi(y, 0, h5 - 1)), components);
	};
	std::vector<vec4> level6(32 * 32);
	for (int y = 0; y < 32; y++)
		for (int x = 0; x < 32; x++)
		{
			const vec4 v = reduce4(p, load5(2 * x, 2 * y), load5(2 * x, 2 * y + 1), load5(2 * x + 1, 2 * y), load5(2 * x + 1, 2 * y + 1));
			level6[size_t(y) * 32 + x] = v;
			store(p, x, y, v, 6);
		}
	reduce_block(p, level6, 32, 0, 0, 6);
}
}
