#include "strip_plan.hpp"
#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace Granite
{
namespace
{
uint32_t level_size(uint32_t full, float scale)
{
	return uint32_t(std::ceil(float(full) * scale)); // get_resource_dimensions: ceil(input * size)
}

struct Span
{
	int64_t lo, hi; // inclusive
};

// Input rows a band of output rows reads through a LinearClamp fetch at v = (r + 0.5) / out_h displaced by up to
// `reach` input texels (tent taps: 1.75 down, 0.875 up; 0 for a plain bilinear fetch), one row of safety either side
// for the fp32 coordinate arithmetic of the kernels.
Span footprint(Span out_rows, uint32_t out_h, uint32_t in_h, double reach)
{
	const double scale = double(in_h) / double(out_h);
	const double lo = (double(out_rows.lo) + 0.5) * scale - 0.5 - reach;
	const double hi = (double(out_rows.hi) + 0.5) * scale - 0.5 + reach;
	Span s;
	s.lo = std::max<int64_t>(int64_t(std::floor(lo)) - 1, 0);
	s.hi = std::min<int64_t>(int64_t(std::floor(hi)) + 2, int64_t(in_h) - 1);
	return s;
}

RowRange to_range(Span s)
{
	RowRange r;
	r.whole = false;
	if (s.hi < s.lo)
	{
		r.first = 0;
		r.count = 0;
	}
	else
	{
		r.first = uint32_t(s.lo);
		r.count = uint32_t(s.hi - s.lo + 1);
	}
	return r;
}

// `rows` grown by `reach` rows either side, inside the image.
Span grown(Span rows, int64_t reach, uint32_t height)
{
	if (rows.hi < rows.lo)
		return rows;
	return {std::max<int64_t>(rows.lo - reach, 0), std::min<int64_t>(rows.hi + reach, int64_t(height) - 1)};
}

Span chunk_of(unsigned index, uint32_t chunk, uint32_t height)
{
	const int64_t lo = int64_t(index) * chunk;
	const int64_t hi = std::min<int64_t>(lo + chunk, height) - 1;
	return {lo, hi};
}
} // namespace

StripPlan StripPlan::build(unsigned index, unsigned count, uint32_t width, uint32_t height, const StripAA &aa)
{
	if (count == 0 || index >= count)
		throw std::logic_error("StripPlan: rank index out of range.");
	StripPlan plan;
	plan.index = index;
	plan.count = count;
	plan.width = width;
	plan.height = height;
	plan.h_threshold = level_size(height, 0.5f);
	plan.h_d0 = level_size(height, 0.25f);
	plan.h_d1 = level_size(height, 0.125f);
	plan.h_u0 = plan.h_d0;
	plan.out_chunk_rows = (height + count - 1) / count;
	plan.d1_chunk_rows = (plan.h_d1 + count - 1) / count;
	plan.aa = aa;
	if (count == 1)
		return plan; // every range stays "whole"; one chunk = the whole level

	const Span chunk = chunk_of(index, plan.out_chunk_rows, height);
	const Span d1 = chunk_of(index, plan.d1_chunk_rows, plan.h_d1);
	plan.d1 = to_range(d1);

	// Tonemapped rows this rank needs: its chunk, or what the post-tonemap AA reads to produce its chunk.
	Span out = chunk;
	if (aa.post == StripAA::Post::FXAA)
	{
		// fxaa.frag: the four corner taps (1 row) and two pairs of bilinear taps along the edge direction, which is clamped
		// to FXAA_SPAN_MAX = 8 texels and scaled by at most 0.5: 4 rows + 1 for the bilinear footprint, + 1 of safety.
		plan.aa_out = to_range(chunk);
		out = grown(chunk, 6, height);
	}
	else if (aa.post == StripAA::Post::SMAA)
	{
		// SMAA.hlsl, back to front.  Neighbourhood blending reads the weights of its own pixel, of the one to the right and
		// of the one below, and the colour up to one texel away.  The weight pass walks the edge texture up and down by
		// at most 2 * SMAA_MAX_SEARCH_STEPS texels, + 3.25 for the search-texture correction of the last step, + 1.5 for the
		// crossing-edge and corner fetches at the ends (the diagonal search, 16 + 4 texels at most, stays inside that):
		// 2 * steps + 8 rows are kept.  Edge detection compares with the pixels above (2 rows up for the local contrast
		// adaptation) and below (1 row); one row of safety.
		plan.aa_out = to_range(chunk);
		const Span weights = grown(chunk, 1 + 1, height);
		const Span edges = grown(weights, int64_t(2 * aa.smaa_search_steps) + 8, height);
		plan.smaa_weights = to_range(weights);
		plan.smaa_edges = to_range(edges);
		out = grown(edges, 2 + 1, height);
	}
	plan.tonemap = to_range(out);

	Span hdr_rows = out;
	if (d1.hi >= d1.lo)
	{
		const Span d0 = footprint(d1, plan.h_d1, plan.h_d0, 1.75);
		const Span thr = footprint(d0, plan.h_d0, plan.h_threshold, 1.75);
		const Span hdr_for_threshold = footprint(thr, plan.h_threshold, height, 0.0);
		plan.d0 = to_range(d0);
		plan.threshold = to_range(thr);
		if (out.hi >= out.lo)
			hdr_rows = {std::min(out.lo, hdr_for_threshold.lo), std::max(out.hi, hdr_for_threshold.hi)};
		else
			hdr_rows = hdr_for_threshold;
	}
	else
	{
		plan.d0 = to_range({0, -1});
		plan.threshold = to_range({0, -1});
	}
	if (aa.temporal)
	{
		// taa_resolve.frag reads the lit image, depth and motion vectors in the 3 x 3 neighbourhood of its pixel.
		plan.taa = to_range(hdr_rows);
		hdr_rows = grown(hdr_rows, 1 + 1, height);
	}
	if (aa.temporal && aa.taa_history_reach > 0 && plan.taa.count > 0)
	{
		// One exchange depth for all ranks (the all-gather's chunks are uniform): the deepest halo any rank resolves beyond its
		// chunk, plus the reach.  Every rank must own that many rows, or the neighbour alone could not supply them.
		StripAA whole_gather = aa;
		whole_gather.taa_history_reach = 0;
		int64_t depth = 0, thinnest = int64_t(height);
		for (unsigned g = 0; g < count; g++)
		{
			const StripPlan other = g == index ? plan : build(g, count, width, height, whole_gather);
			const Span c = chunk_of(g, plan.out_chunk_rows, height);
			thinnest = std::min(thinnest, c.hi - c.lo + 1);
			if (other.taa.count == 0 || c.hi < c.lo)
				continue;
			depth = std::max(depth, std::max<int64_t>(c.lo - int64_t(other.taa.first), int64_t(other.taa.first + other.taa.count) - 1 - c.hi));
		}
		depth += int64_t(aa.taa_history_reach);
		if (depth <= thinnest)
		{
			plan.taa_exchange_rows = uint32_t(depth);
			plan.taa_history_held = to_range(grown(chunk, depth, height));
		}
	}
	plan.lighting = to_range(hdr_rows);
	plan.u0 = out.hi >= out.lo ? to_range(footprint(out, height, plan.h_u0, 0.0)) : to_range({0, -1});
	return plan;
}
} // namespace Granite
