// TEST INFRASTRUCTURE.  Runs the fast anti-aliasing kernels of granite_amd/csrc/aa_fast_kernels.hpp -- the text the GPU build
// compiles -- on the CPU through tests/cpp/hip_emu.hpp, with the launch geometry of the launchers in aa.hip.  Built by
// tests/test_aa_fast_kernels_cpu.py with g++ -ffp-contract=off and compared with the oracle bit for bit.
#include "hip_emu.hpp"
#include <cstdio>
#include <cstdlib>
// smaa_weights.hpp under AA_EMU_CHECK_REACH: a texel fetch outside the staged bit words is a defect of the reach analysis
#define AA_EMU_CHECK_REACH 1
static void aa_emu_reach_violation(bool columns, int x, int y, int x0, int y0)
{
	fprintf(stderr, "EdgeBitTiles: %s fetch of texel (%d, %d) outside the words staged for the block at (%d, %d)\n", columns ? "column" : "row", x, y, x0, y0);
	abort();
}
#include "../../granite_amd/csrc/aa_fast_kernels.hpp"
#include "../../granite_amd/csrc/smaa_weights.hpp"

static unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }
static RowSpan span_of(int h, int first, int count)
{
	if (count <= 0)
		return {0u, uint32_t(h)};
	const uint32_t f = uint32_t(std::min(first, h)), e = uint32_t(std::min(first + count, h));
	return {f, e};
}

extern "C" {
int aah_diag_walk_exact(int n, float inv, int quarter) { return aa::axis_walk_exact(n, inv, 17, quarter != 0) ? 1 : 0; }
int aah_centre_taps_exact(int n, float inv)
{
	static const int ks[] = {-2, -1, 0, 1, 2};
	return aa::axis_taps_exact(n, inv, ks, 5) ? 1 : 0;
}

void aah_fxaa(const uint8_t *in, int w, int h, uint8_t *out, int row_first, int row_count)
{
	const RowSpan rows = span_of(h, row_first, row_count);
	emu::launch(k_fxaa_fast, dim3(div_up(w, FAST_BW), div_up(rows.count(), FAST_BH)), dim3(FAST_BW, FAST_BH), in, uint32_t(w * 4), w, h, out, uint32_t(w * 4),
	            1.0f / float(w), 1.0f / float(h), rows);
}

void aah_smaa_edges(const uint8_t *in, int w, int h, uint8_t *edges, float threshold, int row_first, int row_count)
{
	const RowSpan rows = span_of(h, row_first, row_count);
	emu::launch(k_smaa_edges_fast, dim3(div_up(w, FAST_BW), div_up(rows.count(), FAST_BH)), dim3(FAST_BW, FAST_BH), in, uint32_t(w * 4), w, h, edges,
	            uint32_t(w * 2), threshold, rows);
}

void aah_smaa_blend(const uint8_t *color, const uint8_t *weights, int w, int h, uint8_t *out, int wide, int row_first, int row_count)
{
	const RowSpan rows = span_of(h, row_first, row_count);
	const ColorImage c = {color, uint32_t(w * 4), w, h}, b = {weights, uint32_t(w * 4), w, h};
	if (wide)
		emu::launch(k_smaa_blend_fast<4>, dim3(div_up(w, 256), div_up(rows.count(), 4)), dim3(64, 4), c, b, out, uint32_t(w * 4), 1.0f / float(w),
		            1.0f / float(h), rows);
	else
		emu::launch(k_smaa_blend_fast<1>, dim3(div_up(w, 64), div_up(rows.count(), 4)), dim3(64, 4), c, b, out, uint32_t(w * 4), 1.0f / float(w),
		            1.0f / float(h), rows);
}

void aah_taa_fmt(const uint8_t *current, const uint8_t *depth, const uint8_t *mv, const uint8_t *history, int w, int h, const float *reproj16, int quality,
                 uint8_t *out_color, uint8_t *out_history, int row_first, int row_count, int current_b10, int color_b10);
void aah_taa(const uint8_t *current, const uint8_t *depth, const uint8_t *mv, const uint8_t *history, int w, int h, const float *reproj16, int quality,
             uint8_t *out_color, uint8_t *out_history, int row_first, int row_count)
{
	aah_taa_fmt(current, depth, mv, history, w, h, reproj16, quality, out_color, out_history, row_first, row_count, 0, 0);
}
// current_b10 / color_b10: the current frame / the resolved colour are B10G11R11_UFLOAT_PACK32 words (4 bytes per texel)
void aah_taa_band(const uint8_t *current, const uint8_t *depth, const uint8_t *mv, const uint8_t *history, int w, int h, const float *reproj16, int quality,
                  uint8_t *out_color, uint8_t *out_history, int row_first, int row_count, int current_b10, int color_b10, int hist_first, int hist_count,
                  uint32_t *reach_flag);
void aah_taa_fmt(const uint8_t *current, const uint8_t *depth, const uint8_t *mv, const uint8_t *history, int w, int h, const float *reproj16, int quality,
                 uint8_t *out_color, uint8_t *out_history, int row_first, int row_count, int current_b10, int color_b10)
{
	aah_taa_band(current, depth, mv, history, w, h, reproj16, quality, out_color, out_history, row_first, row_count, current_b10, color_b10, 0, 0, nullptr);
}
// history rows [hist_first, hist_first + hist_count) are the ones a band holds; *reach_flag is set when a pixel fetches another row
void aah_taa_band(const uint8_t *current, const uint8_t *depth, const uint8_t *mv, const uint8_t *history, int w, int h, const float *reproj16, int quality,
                  uint8_t *out_color, uint8_t *out_history, int row_first, int row_count, int current_b10, int color_b10, int hist_first, int hist_count,
                  uint32_t *reach_flag)
{
	const RowSpan rows = span_of(h, row_first, row_count);
	TaaImages im = {};
	im.hist_first = hist_first, im.hist_end = hist_first + hist_count, im.reach_flag = reach_flag;
	im.current = current, im.depth = depth, im.mv = mv, im.history = history;
	im.out_color = out_color, im.out_history = out_history;
	im.history_pitch = im.out_history_pitch = uint32_t(w * 8);
	im.current_pitch = uint32_t(w * (current_b10 ? 4 : 8));
	im.out_color_pitch = uint32_t(w * (color_b10 ? 4 : 8));
	im.current_b10 = current_b10, im.color_b10 = color_b10;
	im.depth_pitch = im.mv_pitch = uint32_t(w * 4);
	im.w = w, im.h = h;
	aa::TaaPush push;
	for (int i = 0; i < 16; i++)
		push.reproj[i] = reproj16[i];
	push.rt[0] = 1.0f / float(w), push.rt[1] = 1.0f / float(h), push.rt[2] = float(w), push.rt[3] = float(h);
	const dim3 grid(div_up(w, FAST_BW), div_up(rows.count(), FAST_BH)), block(FAST_BW, FAST_BH);
	if (!history)
		emu::launch(k_taa_fast<0, false>, grid, block, im, push, rows);
	else if (quality == 0)
		emu::launch(k_taa_fast<0, true>, grid, block, im, push, rows);
	else if (quality == 1)
		emu::launch(k_taa_fast<1, true>, grid, block, im, push, rows);
	else
		emu::launch(k_taa_fast<2, true>, grid, block, im, push, rows);
}

// SMAA.hlsl:304-324
static SmaaPreset preset_of(int quality)
{
	switch (quality)
	{
	case 0: return {0.15f, 4, 8, 0.25f, 0, 0};
	case 1: return {0.1f, 8, 8, 0.25f, 0, 0};
	case 2: return {0.1f, 16, 8, 0.25f, 1, 1};
	default: return {0.05f, 32, 16, 0.25f, 1, 1};
	}
}

void aah_smaa_weights(const uint8_t *edges, int w, int h, const uint8_t *area_rg8, const uint8_t *search_r8, int quality, uint8_t *out, int row_first,
                      int row_count)
{
	const RowSpan rows = span_of(h, row_first, row_count);
	std::vector<float> area(160 * 560 * 2), search(64 * 16);
	for (size_t i = 0; i < area.size(); i++)
		area[i] = float(area_rg8[i]) / 255.0f;
	for (size_t i = 0; i < search.size(); i++)
		search[i] = float(search_r8[i]) / 255.0f;
	SmaaBitPlanes planes = {};
	planes.row_words = smaa_bit_words(w);
	planes.col_words = smaa_bit_words(h);
	// poisoned: a word the pack kernel did not write for this band must not matter
	std::vector<uint64_t> row_r(size_t(planes.rows()) * planes.row_words, 0xA5A5A5A5A5A5A5A5ull), row_g(row_r), col_r(size_t(planes.cols()) * planes.col_words, 0x5A5A5A5A5A5A5A5Aull), col_g(col_r);
	planes.row_r = row_r.data(), planes.row_g = row_g.data(), planes.col_r = col_r.data(), planes.col_g = col_g.data();
	const int tile_first = std::max(0, (int(rows.first) - 128 + SMAA_BITS_PAD) >> 6);
	const int tile_last = std::min(planes.col_words - 1, (int(rows.end) + FAST_BH + 192 + SMAA_BITS_PAD) >> 6);
	const int tiles = planes.row_words * (tile_last - tile_first + 1);
	emu::launch(k_smaa_pack_edges, dim3(tiles), dim3(256), edges, uint32_t(w * 2), w, h, planes, tile_first, tile_last - tile_first + 1);
	SmaaWeightsBitsArgs B = {edges, uint32_t(w * 2), w, h, planes, {area.data(), 160, 560}, {search.data(), 64, 16},
	                         v4{1.0f / float(w), 1.0f / float(h), float(w), float(h)}, preset_of(quality),
	                         aah_centre_taps_exact(w, 1.0f / float(w)) && aah_centre_taps_exact(h, 1.0f / float(h)) && !getenv("AAH_NO_CENTRE_SNAP"), 0};
	B.diag_walks_exact = B.centres_snap && !getenv("AAH_FLOAT_DIAG_WALKS") && aa::axis_walk_exact(w, 1.0f / float(w), 17, false) &&
	                     aa::axis_walk_exact(w, 1.0f / float(w), 17, true) && aa::axis_walk_exact(h, 1.0f / float(h), 17, false);
	emu::launch(k_smaa_weights_bits, dim3(div_up(w, FAST_BW), div_up(rows.count(), FAST_BH)), dim3(FAST_BW, FAST_BH), B, out, uint32_t(w * 4), rows);
}
}
