// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  The reference has no test or golden data for this pass; pinned
// by executing the reference's own hiz.comp on the CPU (oracle/ref_build/ref_hiz.cpp, tests/test_reference_shaders_cpu.py, bit
// for bit) and by analytic known-answer cases (tests/test_oracle_hiz_cpu.py).
//
// Depth hierarchy: assets/shaders/post/hiz.comp + HiZPassState::build_render_pass (renderer/post/spd.cpp:141-194).
#include "oracle_common.h"
#include <vector>

namespace
{
struct Level
{
	float *data;
	int w, h;
};
} // namespace

extern "C" {

// Bytes offset (in floats) of chain level `level` for a chain whose level 0 is w x h: levels are tightly packed one after
// the other, level l being max(w >> l, 1) x max(h >> l, 1).
size_t orc_mip_chain_offset(int w, int h, int level)
{
	size_t offset = 0;
	for (int l = 0; l < level; l++)
		offset += size_t(std::max(w >> l, 1)) * size_t(std::max(h >> l, 1));
	return offset;
}

// depth: iw x ih.  res_w x res_h = push.resolution (spd.cpp:166-168: the output size, rounded up to multiples of 64 by
// setup_depth_hierarchy_pass :214-215, times two when output_downsample).  mips = push.mips.  z_transform = the mat2 of
// spd.cpp:164-165, column-major.  write_top_level = !output_downsample (WRITE_TOP_LEVEL, spd.cpp:149).
// out: the chain; with write_top_level its level k is mip k, otherwise its level k is mip k + 1.
void orc_hiz(const float *depth, int iw, int ih, int res_w, int res_h, int mips, const float *z_transform,
             int write_top_level, float *out)
{
	// mip 0: fetch_2x2_texture (hiz.comp:113-121, NearestClamp: texels past the input edge repeat it) + transform_z :65-72.
	std::vector<float> top(size_t(res_w) * res_h);
	for (int y = 0; y < res_h; y++)
		for (int x = 0; x < res_w; x++)
		{
			const float z = depth[size_t(std::min(y, ih - 1)) * iw + std::min(x, iw - 1)];
			const float num = z_transform[0] * z + z_transform[2] * 1.0f;
			const float den = z_transform[1] * z + z_transform[3] * 1.0f;
			const float q = num / den;
			top[size_t(y) * res_w + x] = (1e30f < q) ? 1e30f : q; // GLSL min(x, y) = y < x ? y : x
		}

	const int chain_w = write_top_level ? res_w : res_w / 2, chain_h = write_top_level ? res_h : res_h / 2;
	auto chain_level = [&](int mip) -> float * {
		const int k = write_top_level ? mip : mip - 1;
		return k < 0 ? nullptr : out + orc_mip_chain_offset(chain_w, chain_h, k);
	};
	if (write_top_level)
		memcpy(chain_level(0), top.data(), top.size() * sizeof(float));

	// mip m from mip m - 1 (hiz.comp:163-176 up to mip 6, :215-247 mip 7, :178-213 above): max over the 2x2 footprint;
	// where the finer level has an odd size the last texel also folds in the row / column that halving would drop
	// (mip_resolution = max(resolution >> mip, 1), :38-41).  Coordinates clamp to the finer level (:124-132).
	std::vector<float> prev = std::move(top), cur;
	int pw = res_w, ph = res_h;
	for (int mip = 1; mip < mips; mip++)
	{
		const int w = std::max(res_w >> mip, 1), h = std::max(res_h >> mip, 1);
		cur.assign(size_t(w) * h, 0.0f);
		for (int y = 0; y < h; y++)
			for (int x = 0; x < w; x++)
			{
				const int nx = 2 + ((x + 1 == w && (pw & 1)) ? 1 : 0);
				const int ny = 2 + ((y + 1 == h && (ph & 1)) ? 1 : 0);
				float r = prev[size_t(std::min(2 * y, ph - 1)) * pw + std::min(2 * x, pw - 1)];
				for (int j = 0; j < ny; j++)
					for (int i = 0; i < nx; i++)
						r = std::max(r, prev[size_t(std::min(2 * y + j, ph - 1)) * pw + std::min(2 * x + i, pw - 1)]);
				cur[size_t(y) * w + x] = r;
			}
		memcpy(chain_level(mip), cur.data(), cur.size() * sizeof(float));
		prev.swap(cur);
		pw = w;
		ph = h;
	}
}

} // extern "C"
