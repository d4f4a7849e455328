// Row-band tiling of one frame across N executors.  No reference analogue: Granite renders a frame on one
// device.  Rank g owns a band of full-resolution rows; per-pixel passes (lighting, threshold, tonemap) and the fine bloom
// levels run on the band plus the halo their consumers reach into, the 1/8-resolution level is all-gathered, the coarse
// levels (1/16, 1/32, luminance, the first two upsamples) are computed redundantly and identically on every rank, and the
// tonemapped bands are all-gathered into the swapchain image.  Every kernel keeps full-image coordinates, so a band holds
// bit-identical values to the same rows of a single-device frame.
#pragma once
#include <cstdint>
#include <functional>
#include "../../../include/granite_hip.h"

namespace HIP
{
class CommandBuffer;
class Image;
}

namespace Granite
{
// Rows [first, first + count) of one render target.  whole == true: no restriction (count is ignored).
struct RowRange
{
	bool whole = true;
	uint32_t first = 0;
	uint32_t count = 0;
	bool empty() const { return !whole && count == 0; }
};

// RowRange -> the C ABI's render area.  Returns false when the band is empty (nothing to launch on this rank).
inline bool to_rows(const RowRange *range, gr_rows &rows)
{
	rows = {0, 0xffffffffu}; // the whole image (a launcher clamps the end to the image height; count == 0 would be "no rows")
	if (!range || range->whole)
		return true;
	if (range->count == 0)
		return false;
	rows.first = range->first;
	rows.count = range->count;
	return true;
}

// The anti-aliasing passes around the post chain, as far as the band arithmetic cares.
struct StripAA
{
	enum class Post { None, FXAA, SMAA };
	Post post = Post::None;          // after the tonemap: reads the tonemapped image above and below its band
	unsigned smaa_search_steps = 0;  // SMAA_MAX_SEARCH_STEPS of the preset (4 / 8 / 16 / 32)
	bool temporal = false;           // TAA resolve between lighting and the post chain
	// Rows a resolved pixel's history fetch may lie away from the pixel (motion + the 4 x 4 filter footprint).  > 0: the history
	// bands meet their neighbours' boundary rows only; 0: whole bands are all-gathered (any motion).
	unsigned taa_history_reach = 0;
};

struct StripPlan
{
	unsigned index = 0, count = 1;
	uint32_t width = 0, height = 0;
	// heights of the InputRelative levels (render_graph.cpp:3158-3170: ceil(input * scale))
	uint32_t h_threshold = 0, h_d0 = 0, h_d1 = 0, h_u0 = 0;

	RowRange lighting;  // HDR rows: own output band + what the own 1/8 chunk needs through threshold / d0 / d1
	RowRange threshold; // 1/2 level
	RowRange d0;        // 1/4 level
	RowRange d1;        // 1/8 level: exactly this rank's all-gather chunk
	RowRange u0;        // 1/4 level rows the tonemap band samples
	RowRange tonemap;   // full-res rows: this rank's all-gather chunk, plus the rows a post-tonemap AA pass reads around it
	uint32_t d1_chunk_rows = 0;  // rows per rank in the 1/8-level all-gather (last ranks may own fewer real rows)
	uint32_t out_chunk_rows = 0; // rows per rank in the final all-gather (and in the TAA history all-gather)

	// Anti-aliasing under row bands.  The halo rows are RECOMPUTED (lighting / resolve / tonemap run on a taller band), not
	// exchanged: values are those of the whole-frame launch, no further meeting point is needed for FXAA / SMAA.  The TAA
	// history is the exception -- reprojection may reach any row, so the history bands are all-gathered every frame.
	StripAA aa;
	RowRange taa;          // rows the temporal resolve writes (= the HDR rows the post chain reads on this rank)
	// TAA history under a bounded reach: every rank contributes the first and the last taa_exchange_rows rows of its chunk to one
	// all-gather and takes its neighbours' blocks; afterwards it holds taa_history_held, which covers every row the resolve of
	// `taa` can fetch within the reach.  taa_exchange_rows == 0: the bands are all-gathered whole (reach 0, or bands too thin).
	uint32_t taa_exchange_rows = 0;
	RowRange taa_history_held;
	uint32_t *taa_reach_flag = nullptr; // host-visible word the resolve sets when a fetch leaves taa_history_held (owned by the app)
	RowRange smaa_edges;   // rows of "smaa-edge" the weight pass searches through
	RowRange smaa_weights; // rows of "smaa-weights" the blend pass reads
	RowRange aa_out;       // rows of the post-AA output: exactly this rank's all-gather chunk
	bool post_aa() const { return aa.post != StripAA::Post::None; }

	bool active() const { return count > 1; }
	static StripPlan build(unsigned index, unsigned count, uint32_t width, uint32_t height, const StripAA &aa = {});

	// All-gather of `chunk_rows` rows per rank inside `image` (rank r's rows start at r * chunk_rows; the image's memory
	// is padded to count * chunk_rows rows).  Installed by the application; runs on the command buffer's stream.
	using ExchangeHook = std::function<void(HIP::CommandBuffer &cmd, HIP::Image &image, uint32_t chunk_rows, const char *tag)>;
	ExchangeHook exchange;
	// Optional: the all-gather of the tonemapped bands, issued beside the frame instead of inside it (its own stream and
	// communicator), so that it overlaps the next frame; `exchange` is then only used for the 1/8 bloom level.  Together
	// with acquire_output, which the pass that writes the output image calls first: it makes the writing stream wait for the
	// previous gather that still reads or writes that image.
	ExchangeHook exchange_output;
	std::function<void(HIP::CommandBuffer &cmd, HIP::Image &image)> acquire_output;
};
} // namespace Granite
