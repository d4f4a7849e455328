// SMAA blending-weight calculation (SMAABlendingWeightCalculationPS, SMAA.hlsl:831-1250) for gfx950: the shader's arithmetic as
// a template over the source of edge texels, the bit-plane form of the edge texture that turns the orthogonal searches into bit
// scans, and the kernels around them.  Included by aa.hip and, unchanged, by the host emulation of the CPU tests
// (tests/cpp/hip_emu.hpp).  `mad` is fmaf; compile without FMA contraction.
#pragma once
#include "aa_core.hpp"
#include "aa_fast_kernels.hpp"
#include "device_vec.hpp"

// The same sampler over a texture that was decoded to fp32 when it was uploaded (SMAA's area and search tables: constant data,
// v / 255 evaluated once on the host instead of at every fetch -- the same float either way).
template <int CH>
struct TexF
{
	const float *data;
	int w, h;

	__device__ __forceinline__ v4 fetch(int x, int y) const
	{
		x = aa::clampi(x, 0, w - 1);
		y = aa::clampi(y, 0, h - 1);
		const float *p = data + (uint32_t(y) * uint32_t(w) + uint32_t(x)) * uint32_t(CH);
		v4 r = mk4(0.0f, 0.0f, 0.0f, 1.0f);
		r.x = p[0];
		if (CH >= 2)
			r.y = p[1];
		return r;
	}

	__device__ __forceinline__ v4 sample(v2 uv, int ox = 0, int oy = 0) const
	{
		int x0, y0;
		float a, b;
		aa::linear_axis(uv.x * float(w) - 0.5f, x0, a);
		aa::linear_axis(uv.y * float(h) - 0.5f, y0, b);
		x0 += ox;
		y0 += oy;
		const v4 t00 = fetch(x0, y0), t10 = fetch(x0 + 1, y0), t01 = fetch(x0, y0 + 1), t11 = fetch(x0 + 1, y0 + 1);
		const v4 top = t00 * (1.0f - a) + t10 * a;
		const v4 bot = t01 * (1.0f - a) + t11 * a;
		return top * (1.0f - b) + bot * b;
	}
};


struct SmaaPreset
{
	float threshold;
	int max_search_steps;
	int max_search_steps_diag;
	float corner_rounding_norm;
	int diag;
	int corner;
};


// Edges = Tex8<2> (the image) or EdgeTile (its LDS copy around the block): the same sample() either way.
template <typename Edges>
struct SmaaWeights
{
	Edges edges;
	TexF<2> area;
	TexF<1> search;
	v4 rt;
	SmaaPreset P;
	int px = 0, py = 0; // the pixel weights_at() works on (for the accessor's run queries)

	__device__ __forceinline__ static v2 rg(v4 v) { return mk2(v.x, v.y); }
	__device__ __forceinline__ v2 rtxy() const { return mk2(rt.x, rt.y); }

	__device__ static v2 decode_diag2(v2 e)
	{
		e.x = e.x * fabsf(5.0f * e.x - 5.0f * 0.75f);
		return mk2(roundf(e.x), roundf(e.y));
	}
	__device__ static v4 decode_diag4(v4 e)
	{
		e.x = e.x * fabsf(5.0f * e.x - 5.0f * 0.75f);
		e.z = e.z * fabsf(5.0f * e.z - 5.0f * 0.75f);
		return mk4(roundf(e.x), roundf(e.y), roundf(e.z), roundf(e.w));
	}
	// The same walks on integer texel positions, for an accessor that holds the edge flags as bits and an image whose walks are
	// proven to land on texels (aa_core.hpp: axis_walk_exact).  SECOND: the tap sits a quarter of a texel to the right of a
	// centre, and the decoded sample is (R of the texel to the right, G of this one).
	template <bool SECOND>
	__device__ __forceinline__ v2 search_diag_bits(v2 dir, v2 &e) const
	{
		int x = px, y = py;
		const int dx = int(dir.x), dy = int(dir.y);
		float z = -1.0f, wsum = 1.0f;
		while (z < float(P.max_search_steps_diag - 1) && wsum > 0.9f)
		{
			x += dx;
			y += dy;
			z += 1.0f;
			e = edges.template texel<false>(x, y);
			if (SECOND)
				e.x = edges.template texel<false>(x + 1, y).x;
			wsum = dot2(e, mk2(0.5f, 0.5f));
		}
		return mk2(z, wsum);
	}
	__device__ v2 search_diag1(v2 texcoord, v2 dir, v2 &e) const
	{
		if constexpr (Edges::HAS_RUNS)
			if (edges.diag_exact)
				return search_diag_bits<false>(dir, e);
		v4 coord = mk4(texcoord.x, texcoord.y, -1.0f, 1.0f);
		while (coord.z < float(P.max_search_steps_diag - 1) && coord.w > 0.9f)
		{
			coord.x = fmaf(rt.x, dir.x, coord.x);
			coord.y = fmaf(rt.y, dir.y, coord.y);
			coord.z = fmaf(1.0f, 1.0f, coord.z);
			e = rg(edges.sample(mk2(coord.x, coord.y)));
			coord.w = dot2(e, mk2(0.5f, 0.5f));
		}
		return mk2(coord.z, coord.w);
	}
	__device__ v2 search_diag2(v2 texcoord, v2 dir, v2 &e) const
	{
		if constexpr (Edges::HAS_RUNS)
			if (edges.diag_exact)
				return search_diag_bits<true>(dir, e);
		v4 coord = mk4(texcoord.x, texcoord.y, -1.0f, 1.0f);
		coord.x += 0.25f * rt.x;
		while (coord.z < float(P.max_search_steps_diag - 1) && coord.w > 0.9f)
		{
			coord.x = fmaf(rt.x, dir.x, coord.x);
			coord.y = fmaf(rt.y, dir.y, coord.y);
			coord.z = fmaf(1.0f, 1.0f, coord.z);
			e = decode_diag2(rg(edges.sample(mk2(coord.x, coord.y))));
			coord.w = dot2(e, mk2(0.5f, 0.5f));
		}
		return mk2(coord.z, coord.w);
	}
	__device__ v2 area_diag(v2 dist, v2 e, float offset) const
	{
		v2 texcoord = fma2(mk2(20.0f, 20.0f), e, dist);
		const v2 px = mk2(1.0f / 160.0f, 1.0f / 560.0f);
		texcoord = fma2(px, texcoord, 0.5f * px);
		texcoord.x += 0.5f;
		texcoord.y += (1.0f / 7.0f) * offset;
		return rg(area.sample(texcoord));
	}
	__device__ v2 diag_weights(v2 texcoord, v2 e) const
	{
		v2 weights = mk2(0.0f, 0.0f);
		v4 d;
		v2 end = mk2(0.0f, 0.0f);
		if (e.x > 0.0f)
		{
			const v2 r = search_diag1(texcoord, mk2(-1.0f, 1.0f), end);
			d.x = r.x;
			d.z = r.y;
			d.x += float(end.y > 0.9f);
		}
		else
		{
			d.x = 0.0f;
			d.z = 0.0f;
		}
		{
			const v2 r = search_diag1(texcoord, mk2(1.0f, -1.0f), end);
			d.y = r.x;
			d.w = r.y;
		}
		if (d.x + d.y > 2.0f)
		{
			const v4 coords = mk4(fmaf(-d.x + 0.25f, rt.x, texcoord.x), fmaf(d.x, rt.y, texcoord.y), fmaf(d.y, rt.x, texcoord.x),
			                      fmaf(-d.y - 0.25f, rt.y, texcoord.y));
			const v2 a = rg(edges.sample(mk2(coords.x, coords.y), -1, 0));
			const v2 b = rg(edges.sample(mk2(coords.z, coords.w), 1, 0));
			const v4 dec = decode_diag4(mk4(a.x, a.y, b.x, b.y));
			const v4 c = mk4(dec.y, dec.x, dec.w, dec.z);
			v2 cc = fma2(mk2(2.0f, 2.0f), mk2(c.x, c.z), mk2(c.y, c.w));
			if (d.z >= 0.9f)
				cc.x = 0.0f;
			if (d.w >= 0.9f)
				cc.y = 0.0f;
			weights = weights + area_diag(mk2(d.x, d.y), cc, 0.0f);
		}

		{
			const v2 r = search_diag2(texcoord, mk2(-1.0f, -1.0f), end);
			d.x = r.x;
			d.z = r.y;
		}
		if (edges.sample(texcoord, 1, 0).x > 0.0f)
		{
			const v2 r = search_diag2(texcoord, mk2(1.0f, 1.0f), end);
			d.y = r.x;
			d.w = r.y;
			d.y += float(end.y > 0.9f);
		}
		else
		{
			d.y = 0.0f;
			d.w = 0.0f;
		}
		if (d.x + d.y > 2.0f)
		{
			const v4 coords = mk4(fmaf(-d.x, rt.x, texcoord.x), fmaf(-d.x, rt.y, texcoord.y), fmaf(d.y, rt.x, texcoord.x), fmaf(d.y, rt.y, texcoord.y));
			v4 c;
			c.x = edges.sample(mk2(coords.x, coords.y), -1, 0).y;
			c.y = edges.sample(mk2(coords.x, coords.y), 0, -1).x;
			const v4 zw = edges.sample(mk2(coords.z, coords.w), 1, 0);
			c.z = zw.y;
			c.w = zw.x;
			v2 cc = fma2(mk2(2.0f, 2.0f), mk2(c.x, c.z), mk2(c.y, c.w));
			if (d.z >= 0.9f)
				cc.x = 0.0f;
			if (d.w >= 0.9f)
				cc.y = 0.0f;
			const v2 ar = area_diag(mk2(d.x, d.y), cc, 0.0f);
			weights = weights + mk2(ar.y, ar.x);
		}
		return weights;
	}

	__device__ float search_length(v2 e, float offset) const
	{
		v2 scale = mk2(66.0f * 0.5f, 33.0f * -1.0f);
		v2 bias = mk2(66.0f * offset, 33.0f * 1.0f);
		scale = scale + mk2(-1.0f, 1.0f);
		bias = bias + mk2(0.5f, -0.5f);
		scale = scale * mk2(1.0f / 64.0f, 1.0f / 16.0f);
		bias = bias * mk2(1.0f / 64.0f, 1.0f / 16.0f);
		return search.sample(fma2(scale, e, bias)).x;
	}
	// The four orthogonal searches step two texels at a time while the bilinear sample between the rows (columns) of an edge says
	// "edge continues, no crossing edge".  Over bit planes (Edges::HAS_RUNS) that is a count of leading set pairs in a 128-bit
	// window -- edges.run_*() -- and the loop only replays the shader's coordinate recurrence (one fma and two compares per step
	// instead of a four-texel sample), then takes the ONE sample the search-texture lookup needs: the last one.  Same trip count,
	// same final coordinate, same final sample as the stepping form (checked against the oracle on the CPU: tests/test_aa_fast_kernels_cpu.py).
	__device__ float search_x_left(v2 texcoord, float end) const
	{
		v2 e = mk2(0.0f, 1.0f);
		if constexpr (Edges::HAS_RUNS)
		{
			const int run = edges.run_left(px, py);
			float prev = texcoord.x;
			int n = 0;
			bool pass = true;
			while (texcoord.x > end && pass)
			{
				pass = n < run;
				prev = texcoord.x;
				texcoord.x = fmaf(-2.0f, rt.x, texcoord.x);
				n++;
			}
			if (n > 0)
				e = rg(edges.sample(mk2(prev, texcoord.y)));
		}
		else
			while (texcoord.x > end && e.y > 0.8281f && e.x == 0.0f)
			{
				e = rg(edges.sample(texcoord));
				texcoord = fma2(mk2(-2.0f, -0.0f), rtxy(), texcoord);
			}
		const float offset = fmaf(-(255.0f / 127.0f), search_length(e, 0.0f), 3.25f);
		return fmaf(rt.x, offset, texcoord.x);
	}
	__device__ float search_x_right(v2 texcoord, float end) const
	{
		v2 e = mk2(0.0f, 1.0f);
		if constexpr (Edges::HAS_RUNS)
		{
			const int run = edges.run_right(px, py);
			float prev = texcoord.x;
			int n = 0;
			bool pass = true;
			while (texcoord.x < end && pass)
			{
				pass = n < run;
				prev = texcoord.x;
				texcoord.x = fmaf(2.0f, rt.x, texcoord.x);
				n++;
			}
			if (n > 0)
				e = rg(edges.sample(mk2(prev, texcoord.y)));
		}
		else
			while (texcoord.x < end && e.y > 0.8281f && e.x == 0.0f)
			{
				e = rg(edges.sample(texcoord));
				texcoord = fma2(mk2(2.0f, 0.0f), rtxy(), texcoord);
			}
		const float offset = fmaf(-(255.0f / 127.0f), search_length(e, 0.5f), 3.25f);
		return fmaf(-rt.x, offset, texcoord.x);
	}
	__device__ float search_y_up(v2 texcoord, float end) const
	{
		v2 e = mk2(1.0f, 0.0f);
		if constexpr (Edges::HAS_RUNS)
		{
			const int run = edges.run_up(px, py);
			float prev = texcoord.y;
			int n = 0;
			bool pass = true;
			while (texcoord.y > end && pass)
			{
				pass = n < run;
				prev = texcoord.y;
				texcoord.y = fmaf(-2.0f, rt.y, texcoord.y);
				n++;
			}
			if (n > 0)
				e = rg(edges.template sample<true>(mk2(texcoord.x, prev)));
		}
		else
			while (texcoord.y > end && e.x > 0.8281f && e.y == 0.0f)
			{
				e = rg(edges.sample(texcoord));
				texcoord = fma2(mk2(-0.0f, -2.0f), rtxy(), texcoord);
			}
		const float offset = fmaf(-(255.0f / 127.0f), search_length(mk2(e.y, e.x), 0.0f), 3.25f);
		return fmaf(rt.y, offset, texcoord.y);
	}
	__device__ float search_y_down(v2 texcoord, float end) const
	{
		v2 e = mk2(1.0f, 0.0f);
		if constexpr (Edges::HAS_RUNS)
		{
			const int run = edges.run_down(px, py);
			float prev = texcoord.y;
			int n = 0;
			bool pass = true;
			while (texcoord.y < end && pass)
			{
				pass = n < run;
				prev = texcoord.y;
				texcoord.y = fmaf(2.0f, rt.y, texcoord.y);
				n++;
			}
			if (n > 0)
				e = rg(edges.template sample<true>(mk2(texcoord.x, prev)));
		}
		else
			while (texcoord.y < end && e.x > 0.8281f && e.y == 0.0f)
			{
				e = rg(edges.sample(texcoord));
				texcoord = fma2(mk2(0.0f, 2.0f), rtxy(), texcoord);
			}
		const float offset = fmaf(-(255.0f / 127.0f), search_length(mk2(e.y, e.x), 0.5f), 3.25f);
		return fmaf(-rt.y, offset, texcoord.y);
	}
	__device__ v2 area_lookup(v2 dist, float e1, float e2) const
	{
		v2 texcoord = fma2(mk2(16.0f, 16.0f), mk2(roundf(4.0f * e1), roundf(4.0f * e2)), dist);
		const v2 px = mk2(1.0f / 160.0f, 1.0f / 560.0f);
		texcoord = fma2(px, texcoord, 0.5f * px);
		texcoord.y = fmaf(1.0f / 7.0f, 0.0f, texcoord.y);
		return rg(area.sample(texcoord));
	}
	__device__ void corner(v2 &weights, v4 texcoord, v2 d, bool horizontal) const
	{
		if (!P.corner)
			return;
		const v2 leftRight = mk2(stepf(d.x, d.y), stepf(d.y, d.x));
		v2 rounding = leftRight * (1.0f - P.corner_rounding_norm);
		const float sum = leftRight.x + leftRight.y;
		rounding = mk2(rounding.x / sum, rounding.y / sum);
		v2 factor = mk2(1.0f, 1.0f);
		if constexpr (Edges::HAS_RUNS)
		{
			// The crossing-edge taps of a horizontal (vertical) edge sit on the pixel's own row (column): where pixel-centre taps are
			// texel fetches (edges.centres_snap, aa_core.hpp: axis_taps_exact) they read one line of texels, lerped along the edge only.
			if (edges.centres_snap)
			{
				if (horizontal)
				{
					factor.x -= rounding.x * edges.template sample_along<false>(texcoord.x, px, py, 0, 1).x;
					factor.x -= rounding.y * edges.template sample_along<false>(texcoord.z, px, py, 1, 1).x;
					factor.y -= rounding.x * edges.template sample_along<false>(texcoord.x, px, py, 0, -2).x;
					factor.y -= rounding.y * edges.template sample_along<false>(texcoord.z, px, py, 1, -2).x;
				}
				else
				{
					factor.x -= rounding.x * edges.template sample_along<true>(texcoord.y, px, py, 1, 0).y;
					factor.x -= rounding.y * edges.template sample_along<true>(texcoord.w, px, py, 1, 1).y;
					factor.y -= rounding.x * edges.template sample_along<true>(texcoord.y, px, py, -2, 0).y;
					factor.y -= rounding.y * edges.template sample_along<true>(texcoord.w, px, py, -2, 1).y;
				}
				weights = weights * mk2(clampfv(factor.x, 0.0f, 1.0f), clampfv(factor.y, 0.0f, 1.0f));
				return;
			}
		}
		if (horizontal)
		{
			factor.x -= rounding.x * edges.sample(mk2(texcoord.x, texcoord.y), 0, 1).x;
			factor.x -= rounding.y * edges.sample(mk2(texcoord.z, texcoord.w), 1, 1).x;
			factor.y -= rounding.x * edges.sample(mk2(texcoord.x, texcoord.y), 0, -2).x;
			factor.y -= rounding.y * edges.sample(mk2(texcoord.z, texcoord.w), 1, -2).x;
		}
		else
		{
			factor.x -= rounding.x * edges.template sample<true>(mk2(texcoord.x, texcoord.y), 1, 0).y;
			factor.x -= rounding.y * edges.template sample<true>(mk2(texcoord.z, texcoord.w), 1, 1).y;
			factor.y -= rounding.x * edges.template sample<true>(mk2(texcoord.x, texcoord.y), -2, 0).y;
			factor.y -= rounding.y * edges.template sample<true>(mk2(texcoord.z, texcoord.w), -2, 1).y;
		}
		weights = weights * mk2(clampfv(factor.x, 0.0f, 1.0f), clampfv(factor.y, 0.0f, 1.0f));
	}

	__device__ v4 weights_at(int x, int y)
	{
		px = x;
		py = y;
		const v2 texcoord = mk2((float(x) + 0.5f) * rt.x, (float(y) + 0.5f) * rt.y);
		const v2 pixcoord = mk2(texcoord.x * rt.z, texcoord.y * rt.w);
		const v4 off0 = mk4(fmaf(rt.x, -0.25f, texcoord.x), fmaf(rt.y, -0.125f, texcoord.y), fmaf(rt.x, 1.25f, texcoord.x), fmaf(rt.y, -0.125f, texcoord.y));
		const v4 off1 = mk4(fmaf(rt.x, -0.125f, texcoord.x), fmaf(rt.y, -0.25f, texcoord.y), fmaf(rt.x, -0.125f, texcoord.x), fmaf(rt.y, 1.25f, texcoord.y));
		const float steps = float(P.max_search_steps);
		const v4 off2 = mk4(fmaf(rt.x, -2.0f * steps, off0.x), fmaf(rt.x, 2.0f * steps, off0.z), fmaf(rt.y, -2.0f * steps, off1.y),
		                    fmaf(rt.y, 2.0f * steps, off1.w));

		v4 weights = mk4(0.0f, 0.0f, 0.0f, 0.0f);
		v2 e = rg(edges.sample(texcoord));
		if (e.y > 0.0f)
		{
			bool orthogonal = true;
			if (P.diag)
			{
				const v2 dw = diag_weights(texcoord, e);
				weights.x = dw.x;
				weights.y = dw.y;
				orthogonal = (weights.x == -weights.y);
			}
			if (orthogonal)
			{
				v2 d;
				v3 coords;
				coords.x = search_x_left(mk2(off0.x, off0.y), off2.x);
				coords.y = off1.y;
				d.x = coords.x;
				const float e1 = edges.sample(mk2(coords.x, coords.y)).x;
				coords.z = search_x_right(mk2(off0.z, off0.w), off2.y);
				d.y = coords.z;
				d = mk2(fabsf(roundf(fmaf(rt.z, d.x, -pixcoord.x))), fabsf(roundf(fmaf(rt.z, d.y, -pixcoord.x))));
				const v2 sqrt_d = mk2(sqrtf(d.x), sqrtf(d.y));
				const float e2 = edges.sample(mk2(coords.z, coords.y), 1, 0).x;
				v2 wrg = area_lookup(sqrt_d, e1, e2);
				coords.y = texcoord.y;
				corner(wrg, mk4(coords.x, coords.y, coords.z, coords.y), d, true);
				weights.x = wrg.x;
				weights.y = wrg.y;
			}
			else
				e.x = 0.0f;
		}
		if (e.x > 0.0f)
		{
			v2 d;
			v3 coords;
			coords.y = search_y_up(mk2(off1.x, off1.y), off2.z);
			coords.x = off0.x;
			d.x = coords.y;
			const float e1 = edges.template sample<true>(mk2(coords.x, coords.y)).y;
			coords.z = search_y_down(mk2(off1.z, off1.w), off2.w);
			d.y = coords.z;
			d = mk2(fabsf(roundf(fmaf(rt.w, d.x, -pixcoord.y))), fabsf(roundf(fmaf(rt.w, d.y, -pixcoord.y))));
			const v2 sqrt_d = mk2(sqrtf(d.x), sqrtf(d.y));
			const float e2 = edges.template sample<true>(mk2(coords.x, coords.z), 0, 1).y;
			v2 wba = area_lookup(sqrt_d, e1, e2);
			coords.x = texcoord.x;
			corner(wba, mk4(coords.x, coords.y, coords.x, coords.z), d, false);
			weights.z = wba.x;
			weights.w = wba.y;
		}
		return weights;
	}
};


// ---- the edge texture as bit planes ------------------------------------------------------------------------------------------------
// An edge texel is two flags (R: edge at the left, G: edge at the top; the bytes are 0 or 255).  k_smaa_pack_edges writes them as
// four planes of bits -- R and G along rows, R and G along columns -- padded by SMAA_BITS_PAD texels of clamp-to-edge replicas on
// every side, so that nothing downstream clamps.  A workgroup of the weight pass stages the 32-bit words around its 32 x 16 pixels
// into LDS: rows y0 - 18 .. y0 + 33 over x0 - 64 .. x0 + 95 (the horizontal searches, the diagonal searches, every near fetch of
// a horizontal edge) and columns x0 - 2 .. x0 + 33 over 192 rows from the 32-aligned row at or below y0 - 64 (the vertical
// searches and what follows them): 3.7 KB.  A search step of the shader -- "edge continues on both texels of this pair, no
// crossing edge on the four" -- is then a bit pair of  C = G(y) & ~R(y) & ~R(y - 1)  (columns:  R(x) & ~G(x) & ~G(x - 1)),  and
// the run length a count of trailing ones in a 64-bit window: 32 steps, the longest search of any preset.
constexpr int SMAA_BITS_PAD = 192;

struct SmaaBitPlanes
{
	uint64_t *row_r, *row_g; // [rows()][row_words]   bit (x + PAD) of row (y + PAD)
	uint64_t *col_r, *col_g; // [cols()][col_words]   bit (y + PAD) of column (x + PAD)
	int row_words, col_words;
	__host__ __device__ int rows() const { return col_words * 64; }
	__host__ __device__ int cols() const { return row_words * 64; }
};
inline int smaa_bit_words(int n) { return (2 * SMAA_BITS_PAD + ((n + 63) & ~63)) / 64; }

// One workgroup of four waves per 64 x 64 tile of the padded domain; wave v takes rows 16 v .. 16 v + 15: sixteen coalesced row
// reads issued together, the row words by ballot, the column words collected per lane and joined through LDS.
// tile_y_first / tile_y_count: only the tiles a row band needs are written.
__global__ __launch_bounds__(256) void k_smaa_pack_edges(const uint8_t *edges, uint32_t pitch, int w, int h, SmaaBitPlanes planes, int tile_y_first,
                                                          int tile_y_count)
{
	__shared__ uint32_t s_col[2][4][64]; // [plane][wave][column]: 16 bits each
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int ty = tile_y_first + int(blockIdx.x) / planes.row_words, tx = int(blockIdx.x) % planes.row_words;
	const int x = aa::clampi(tx * 64 - SMAA_BITS_PAD + lane, 0, w - 1);
	uint32_t t[16];
#pragma unroll
	for (int r = 0; r < 16; r++)
	{
		const int y = aa::clampi(ty * 64 - SMAA_BITS_PAD + wave * 16 + r, 0, h - 1);
		t[r] = *reinterpret_cast<const uint16_t *>(edges + (uint32_t(y) * pitch + uint32_t(x) * 2u));
	}
	uint32_t col_r = 0, col_g = 0;
	uint64_t my_r = 0, my_g = 0;
#pragma unroll
	for (int r = 0; r < 16; r++)
	{
		const bool er = (t[r] & 255u) != 0u, eg = (t[r] >> 8) != 0u;
		const uint64_t word_r = __ballot(er), word_g = __ballot(eg);
		if (lane == r)
		{
			my_r = word_r;
			my_g = word_g;
		}
		col_r |= uint32_t(er) << r;
		col_g |= uint32_t(eg) << r;
	}
	if (lane < 16)
	{
		planes.row_r[size_t(ty * 64 + wave * 16 + lane) * planes.row_words + tx] = my_r;
		planes.row_g[size_t(ty * 64 + wave * 16 + lane) * planes.row_words + tx] = my_g;
	}
	s_col[0][wave][lane] = col_r;
	s_col[1][wave][lane] = col_g;
	__syncthreads();
	if (wave < 2)
	{
		const uint64_t word = uint64_t(s_col[wave][0][lane]) | (uint64_t(s_col[wave][1][lane]) << 16) | (uint64_t(s_col[wave][2][lane]) << 32) |
		                      (uint64_t(s_col[wave][3][lane]) << 48);
		(wave == 0 ? planes.col_r : planes.col_g)[size_t(tx * 64 + lane) * planes.col_words + ty] = word;
	}
}

struct EdgeBitTiles
{
	static constexpr bool HAS_RUNS = true;
	// Everything the pass reads for a pixel lies within 2 x (32 + 1) + 2 = 68 texels of it along a search axis and within 18
	// across (the 16-step diagonal searches); the staged words cover 96 / 18 on every side of the block, so no fetch leaves them.
	static constexpr int HALO = 96;
	static constexpr int ROWS = FAST_BH + 36, ROW_DWORDS = (FAST_BW + 2 * HALO) / 32;       // x0 - 96 .. x0 + 127: 7 words
	static constexpr int COLS = FAST_BW + 4, COL_DWORDS = (FAST_BH + 2 * HALO + 31) / 32 + 1; // from col_y0 <= y0 - 96: 8 words
	const uint32_t *row_r, *row_g; // [ROWS][ROW_DWORDS]: bit b of row i is texel (x0 - 96 + b, y0 - 18 + i)
	const uint32_t *col_r, *col_g; // [COLS][COL_DWORDS]: bit b of column i is texel (x0 - 2 + i, col_y0 + b)
	int x0, y0, col_y0;
	int w, h;
	bool centres_snap; // pixel-centre coordinates of this image resolve to texel fetches (aa_core.hpp: axis_taps_exact)
	bool diag_exact;   // ... and so does every step of the diagonal searches' coordinate walks (aa_core.hpp: axis_walk_exact)

	// 64 bits starting at bit p of a staged row / column
	__device__ __forceinline__ static uint64_t window(const uint32_t *words, int p)
	{
		const int k = p >> 5, sh = p & 31;
		const uint64_t ab = uint64_t(words[k]) | (uint64_t(words[k + 1]) << 32), bc = uint64_t(words[k + 1]) | (uint64_t(words[k + 2]) << 32);
		return uint64_t(uint32_t(ab >> sh)) | (uint64_t(uint32_t(bc >> sh)) << 32);
	}
	__device__ __forceinline__ static uint64_t reverse64(uint64_t v)
	{
#if defined(__HIP_DEVICE_COMPILE__)
		return __brevll(v);
#else
		uint64_t r = 0;
		for (int i = 0; i < 64; i++)
			r |= ((v >> i) & 1ull) << (63 - i);
		return r;
#endif
	}
	// number of leading steps j = 0, 1, .. whose texel pair (bits 2j, 2j + 1) is set on both: 0 .. 32
	__device__ __forceinline__ static int leading_pairs(uint64_t c)
	{
		const uint64_t miss = ~(c & (c >> 1)) & 0x5555555555555555ull;
		return miss ? (__builtin_ctzll(miss) >> 1) : 32;
	}
	__device__ __forceinline__ uint64_t row_condition(int x_first, int y) const
	{
		const int i = y - (y0 - 18), p = x_first - (x0 - HALO);
		return window(row_g + i * ROW_DWORDS, p) & ~window(row_r + i * ROW_DWORDS, p) & ~window(row_r + (i - 1) * ROW_DWORDS, p);
	}
	__device__ __forceinline__ uint64_t column_condition(int x, int y_first) const
	{
		const int i = x - (x0 - 2), p = y_first - col_y0;
		return window(col_r + i * COL_DWORDS, p) & ~window(col_g + i * COL_DWORDS, p) & ~window(col_g + (i - 1) * COL_DWORDS, p);
	}
	// The search's condition on a pair beyond the 64-bit window (the 33rd step: only the rounding of the shader's own
	// end-of-search comparison lets it happen).  Horizontal: G on both texels of row y, R clear on both of rows y, y - 1;
	// vertical: the transposed statement.
	template <bool COLUMNS>
	__device__ __forceinline__ bool pair_continues(int xa, int ya, int xb, int yb) const
	{
		const v2 a = texel<COLUMNS>(xa, ya), b = texel<COLUMNS>(xb, yb);
		const v2 a1 = COLUMNS ? texel<COLUMNS>(xa - 1, ya) : texel<COLUMNS>(xa, ya - 1), b1 = COLUMNS ? texel<COLUMNS>(xb - 1, yb) : texel<COLUMNS>(xb, yb - 1);
		if (COLUMNS)
			return a.x * b.x != 0.0f && a.y + b.y + a1.y + b1.y == 0.0f;
		return a.y * b.y != 0.0f && a.x + b.x + a1.x + b1.x == 0.0f;
	}
	// step j of the left search samples texels (x - 1 - 2j, x - 2j) of rows y - 1, y: the window [x - 63, x], mirrored
	__device__ __forceinline__ int run_left(int x, int y) const
	{
		int run = leading_pairs(reverse64(row_condition(x - 63, y)));
		if (run == 32)
			run += pair_continues<false>(x - 64, y, x - 65, y) ? 1 : 0;
		return run;
	}
	// step j of the right search samples texels (x + 1 + 2j, x + 2 + 2j)
	__device__ __forceinline__ int run_right(int x, int y) const
	{
		int run = leading_pairs(row_condition(x + 1, y));
		if (run == 32)
			run += pair_continues<false>(x + 65, y, x + 66, y) ? 1 : 0;
		return run;
	}
	__device__ __forceinline__ int run_up(int x, int y) const
	{
		int run = leading_pairs(reverse64(column_condition(x, y - 63)));
		if (run == 32)
			run += pair_continues<true>(x, y - 64, x, y - 65) ? 1 : 0;
		return run;
	}
	__device__ __forceinline__ int run_down(int x, int y) const
	{
		int run = leading_pairs(column_condition(x, y + 1));
		if (run == 32)
			run += pair_continues<true>(x, y + 65, x, y + 66) ? 1 : 0;
		return run;
	}

	// (R, G) of one texel as 0.0 / 1.0.  COLUMNS = false: from the staged rows (everything a horizontal edge and the diagonal
	// searches touch); true: from the staged columns (a vertical edge's searches and what follows them).  No branch: the reach of
	// the pass stays inside the staged words (above); the index is clamped into them all the same.
	template <bool COLUMNS>
	__device__ __forceinline__ v2 texel(int x, int y) const
	{
		int line = COLUMNS ? x - (x0 - 2) : y - (y0 - 18), bit = COLUMNS ? y - col_y0 : x - (x0 - HALO);
#if defined(AA_EMU_CHECK_REACH)
		if (unsigned(line) >= unsigned(COLUMNS ? COLS : ROWS) || unsigned(bit) >= unsigned((COLUMNS ? COL_DWORDS : ROW_DWORDS) * 32))
			aa_emu_reach_violation(COLUMNS, x, y, x0, y0);
#endif
		line = aa::clampi(line, 0, (COLUMNS ? COLS : ROWS) - 1);
		bit = aa::clampi(bit, 0, (COLUMNS ? COL_DWORDS : ROW_DWORDS) * 32 - 1);
		const int k = line * (COLUMNS ? COL_DWORDS : ROW_DWORDS) + (bit >> 5), s = bit & 31;
		return mk2(float(((COLUMNS ? col_r : row_r)[k] >> s) & 1u), float(((COLUMNS ? col_g : row_g)[k] >> s) & 1u));
	}

	// A tap whose coordinate across the edge is the centre of pixel (px, py) -- a texel fetch on that axis -- and whose coordinate
	// along the edge is `t` (u for a horizontal edge, COLUMNS = false; v for a vertical one): two texels, one lerp.
	template <bool COLUMNS>
	__device__ __forceinline__ v2 sample_along(float t, int px, int py, int ox, int oy) const
	{
		int i;
		float a;
		aa::linear_axis(t * float(COLUMNS ? h : w) - 0.5f, i, a);
		const int ix = (COLUMNS ? px : i) + ox, iy = (COLUMNS ? i : py) + oy;
		const v2 t0 = texel<COLUMNS>(ix, iy), t1 = COLUMNS ? texel<COLUMNS>(ix, iy + 1) : texel<COLUMNS>(ix + 1, iy);
		return t0 * (1.0f - a) + t1 * a;
	}

	// LinearClamp over the edge texture: the sampler of the byte image (aa.hip: Tex8::sample) with the texels taken from the bits.
	// Straight-line code -- four texels, three lerps -- whatever the weights: with 0.0 / 1.0 texels t * (1 - a) + t' * a is exact for
	// a snapped weight (a = 0), and the lanes of a wave take their taps in lockstep.
	template <bool COLUMNS = false>
	__device__ __forceinline__ v4 sample(v2 uv, int ox = 0, int oy = 0) const
	{
		int ix, iy;
		float a, b;
		aa::linear_axis(uv.x * float(w) - 0.5f, ix, a);
		aa::linear_axis(uv.y * float(h) - 0.5f, iy, b);
		ix += ox;
		iy += oy;
		const v2 t00 = texel<COLUMNS>(ix, iy), t10 = texel<COLUMNS>(ix + 1, iy), t01 = texel<COLUMNS>(ix, iy + 1), t11 = texel<COLUMNS>(ix + 1, iy + 1);
		const v2 top = t00 * (1.0f - a) + t10 * a;
		const v2 bot = t01 * (1.0f - a) + t11 * a;
		const v2 r = top * (1.0f - b) + bot * b;
		return mk4(r.x, r.y, 0.0f, 1.0f);
	}
};

struct SmaaWeightsBitsArgs
{
	const uint8_t *edges;
	uint32_t edges_pitch;
	int w, h;
	SmaaBitPlanes planes;
	TexF<2> area;
	TexF<1> search;
	v4 rt;
	SmaaPreset P;
	int centres_snap;
	int diag_walks_exact;
};

// SMAABlendingWeightCalculationPS over the bit planes.  The reference runs the quad under a depth mask EQUAL to the edge pass's
// non-discarded pixels (smaa.cpp:101-112,170-177); the mask is the edge texel itself: zero edge => zero weights.  Edge pixels are
// a few per cent of a frame and take different paths through the pass (horizontal / vertical / diagonal patterns), so the
// workgroup first writes its edge pixels into a list and then walks the list with all of its lanes: full waves of edge pixels
// instead of a few live lanes per wave.
// Register cap: 64 VGPRs (the kernel wants 69: two dwords go to scratch).  A workgroup is two waves per SIMD, and beside four resident
// lighting waves of 96 registers a SIMD has 128 left: at 72 a workgroup only starts where a lighting wave has retired.
#if defined(__HIP_DEVICE_COMPILE__)
#define SMAA_WEIGHTS_OCCUPANCY __attribute__((amdgpu_waves_per_eu(8)))
#else
#define SMAA_WEIGHTS_OCCUPANCY
#endif
__global__ __launch_bounds__(FAST_BW *FAST_BH) SMAA_WEIGHTS_OCCUPANCY void k_smaa_weights_bits(SmaaWeightsBitsArgs A, uint8_t *out, uint32_t out_pitch, RowSpan rows)
{
	using T = EdgeBitTiles;
	constexpr int THREADS = FAST_BW * FAST_BH, WAVES = THREADS / 64;
	__shared__ uint32_t s_row_r[T::ROWS * T::ROW_DWORDS + 2], s_row_g[T::ROWS * T::ROW_DWORDS + 2]; // + 2: window() reads two words ahead
	__shared__ uint32_t s_col_r[T::COLS * T::COL_DWORDS + 2], s_col_g[T::COLS * T::COL_DWORDS + 2];
	__shared__ uint16_t s_list[THREADS];
	__shared__ uint32_t s_wave_count[WAVES];
	const int bx = blockIdx.x * FAST_BW, by = int(rows.first) + blockIdx.y * FAST_BH;
	const int tid = threadIdx.y * FAST_BW + threadIdx.x, wave = tid >> 6, lane = tid & 63;
	{
		const int x = bx + threadIdx.x, y = by + threadIdx.y;
		const bool inside = x < A.w && y < int(rows.end);
		uint32_t e = 0;
		if (inside)
			e = *reinterpret_cast<const uint16_t *>(A.edges + (uint32_t(y) * A.edges_pitch + uint32_t(x) * 2u));
		const uint64_t mine = __ballot(e != 0u);
		if (lane == 0)
			s_wave_count[wave] = uint32_t(__popcll(mine));
		if (inside && e == 0u)
			*reinterpret_cast<uint32_t *>(out + (uint32_t(y) * out_pitch + uint32_t(x) * 4u)) = 0u;
		__syncthreads();
		uint32_t total = 0, before = 0;
		for (int i = 0; i < WAVES; i++)
		{
			const uint32_t count = AA_WAVE_UNIFORM(s_wave_count[i]);
			before += i < int(AA_WAVE_UNIFORM(wave)) ? count : 0u;
			total += count;
		}
		if (total == 0u)
			return;
		if (e != 0u)
			s_list[before + uint32_t(__popcll(mine & ((1ull << lane) - 1ull)))] = uint16_t(tid);
	}
	// the planes as 32-bit words: bx is a multiple of 32 and so is the pad, the staged rows start on a word; the staged columns
	// start on the word holding row by - HALO
	const uint32_t *row_r32 = reinterpret_cast<const uint32_t *>(A.planes.row_r), *row_g32 = reinterpret_cast<const uint32_t *>(A.planes.row_g);
	const uint32_t *col_r32 = reinterpret_cast<const uint32_t *>(A.planes.col_r), *col_g32 = reinterpret_cast<const uint32_t *>(A.planes.col_g);
	const int row_word0 = (bx - T::HALO + SMAA_BITS_PAD) >> 5;
	const int col_word0 = (by - T::HALO + SMAA_BITS_PAD) >> 5, col_y0 = (col_word0 << 5) - SMAA_BITS_PAD;
	// without the diagonal searches nothing reaches beyond two rows above / one below the block's pixels
	const int row_first = A.P.diag ? 0 : 16, row_count = A.P.diag ? T::ROWS : FAST_BH + 4;
	for (int i = tid; i < row_count * T::ROW_DWORDS; i += THREADS)
	{
		const int r = row_first + i / T::ROW_DWORDS, k = i % T::ROW_DWORDS;
		const size_t word = size_t(by - 18 + r + SMAA_BITS_PAD) * (A.planes.row_words * 2) + row_word0 + k;
		s_row_r[r * T::ROW_DWORDS + k] = row_r32[word];
		s_row_g[r * T::ROW_DWORDS + k] = row_g32[word];
	}
	for (int i = tid; i < T::COLS * T::COL_DWORDS; i += THREADS)
	{
		const int c = i / T::COL_DWORDS, k = i % T::COL_DWORDS;
		const size_t word = size_t(bx - 2 + c + SMAA_BITS_PAD) * (A.planes.col_words * 2) + col_word0 + k;
		s_col_r[i] = col_r32[word];
		s_col_g[i] = col_g32[word];
	}
	__syncthreads();
	uint32_t total = 0;
	for (int i = 0; i < WAVES; i++)
		total += AA_WAVE_UNIFORM(s_wave_count[i]);
	SmaaWeights<EdgeBitTiles> S = {{s_row_r, s_row_g, s_col_r, s_col_g, bx, by, col_y0, A.w, A.h, A.centres_snap != 0, A.diag_walks_exact != 0}, A.area, A.search, A.rt, A.P};
	for (uint32_t i = uint32_t(tid); i < total; i += uint32_t(THREADS))
	{
		const int t = s_list[i], x = bx + (t & (FAST_BW - 1)), y = by + (t / FAST_BW);
		const v4 wgt = S.weights_at(x, y);
		*reinterpret_cast<uint32_t *>(out + (uint32_t(y) * out_pitch + uint32_t(x) * 4u)) =
		    aa::unorm8_encode(wgt.x) | (aa::unorm8_encode(wgt.y) << 8) | (aa::unorm8_encode(wgt.z) << 16) | (aa::unorm8_encode(wgt.w) << 24);
	}
}
