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
	__device__ v2 search_diag1(v2 texcoord, v2 dir, v2 &e) const
	{
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
// four planes of 64-bit words -- R and G along rows, R and G along columns -- padded by SMAA_BITS_PAD texels of clamp-to-edge
// replicas on every side, so that nothing downstream clamps.  A workgroup of the weight pass stages the words around its
// 32 x 16 pixels into LDS: rows y0 - 18 .. y0 + 33 over x0 - 128 .. x0 + 191 (the horizontal searches, the diagonal searches, every
// near fetch of a horizontal edge) and columns x0 - 2 .. x0 + 33 over y0 - 128 .. y0 + 191 (the vertical searches and what follows
// them).  A search step of the shader -- "edge continues on both texels of this pair, no crossing edge on the four" -- is then a
// bit pair of  C = G(y) & ~R(y) & ~R(y - 1)  (columns:  R(x) & ~G(x) & ~G(x - 1)),  and the run length a count of trailing ones.
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

// One wave per 64 x 64 tile of the padded domain: 64 coalesced row reads, the row words by ballot, the column words collected
// per lane.  tile_y0 .. : only the tiles a row band needs are written.
__global__ __launch_bounds__(256) void k_smaa_pack_edges(const uint8_t *edges, uint32_t pitch, int w, int h, SmaaBitPlanes planes, int tile_y_first,
                                                          int tile_y_count)
{
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + wave;
	if (tile >= planes.row_words * tile_y_count)
		return;
	const int ty = tile_y_first + tile / planes.row_words, tx = tile % planes.row_words;
	const int x = aa::clampi(tx * 64 - SMAA_BITS_PAD + lane, 0, w - 1);
	uint64_t col_r = 0, col_g = 0, my_r = 0, my_g = 0;
	for (int r = 0; r < 64; r++)
	{
		const int y = aa::clampi(ty * 64 - SMAA_BITS_PAD + r, 0, h - 1);
		const uint32_t t = *reinterpret_cast<const uint16_t *>(edges + (uint32_t(y) * pitch + uint32_t(x) * 2u));
		const bool er = (t & 255u) != 0u, eg = (t >> 8) != 0u;
		const uint64_t word_r = __ballot(er), word_g = __ballot(eg);
		if (lane == r)
		{
			my_r = word_r;
			my_g = word_g;
		}
		col_r |= uint64_t(er) << r;
		col_g |= uint64_t(eg) << r;
	}
	planes.row_r[size_t(ty * 64 + lane) * planes.row_words + tx] = my_r;
	planes.row_g[size_t(ty * 64 + lane) * planes.row_words + tx] = my_g;
	planes.col_r[size_t(tx * 64 + lane) * planes.col_words + ty] = col_r;
	planes.col_g[size_t(tx * 64 + lane) * planes.col_words + ty] = col_g;
}

struct EdgeBitTiles
{
	static constexpr bool HAS_RUNS = true;
	static constexpr int ROWS = FAST_BH + 36, COLS = FAST_BW + 4, WORDS = 5; // 320 bits per staged row / column
	const uint64_t *row_r, *row_g; // [ROWS][WORDS]: bit b of row i is texel (x0 - 128 + b, y0 - 18 + i)
	const uint64_t *col_r, *col_g; // [COLS][WORDS]: bit b of column i is texel (x0 - 2 + i, y0 - 128 + b)
	int x0, y0;
	int w, h;
	const uint8_t *image; // the RG8 edge texture itself, for a fetch outside the staged words (not expected)
	uint32_t pitch;

	// 128 bits starting at bit p of a staged row / column (p + 128 <= 320)
	__device__ __forceinline__ static void window(const uint64_t *words, int p, uint64_t &lo, uint64_t &hi)
	{
		const int wi = p >> 6, sh = p & 63;
		const uint64_t a = words[wi], b = words[wi + 1], c = (wi + 2 < WORDS) ? words[wi + 2] : 0ull;
		lo = sh ? ((a >> sh) | (b << (64 - sh))) : a;
		hi = sh ? ((b >> sh) | (c << (64 - sh))) : b;
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
	// number of leading steps j = 0, 1, .. whose texel pair (bits 2j, 2j + 1 of the 128-bit window) is set on both
	__device__ __forceinline__ static int leading_pairs(uint64_t lo, uint64_t hi)
	{
		const uint64_t even = 0x5555555555555555ull;
		const uint64_t miss_lo = ~(lo & (lo >> 1)) & even, miss_hi = ~(hi & (hi >> 1)) & even;
		if (miss_lo)
			return __builtin_ctzll(miss_lo) >> 1;
		return 32 + (miss_hi ? (__builtin_ctzll(miss_hi) >> 1) : 32);
	}
	__device__ __forceinline__ void row_condition(int x_first, int y, uint64_t &lo, uint64_t &hi) const
	{
		const int i = y - (y0 - 18), p = x_first - (x0 - 128);
		uint64_t g_lo, g_hi, r_lo, r_hi, q_lo, q_hi;
		window(row_g + i * WORDS, p, g_lo, g_hi);
		window(row_r + i * WORDS, p, r_lo, r_hi);
		window(row_r + (i - 1) * WORDS, p, q_lo, q_hi);
		lo = g_lo & ~r_lo & ~q_lo;
		hi = g_hi & ~r_hi & ~q_hi;
	}
	__device__ __forceinline__ void column_condition(int x, int y_first, uint64_t &lo, uint64_t &hi) const
	{
		const int i = x - (x0 - 2), p = y_first - (y0 - 128);
		uint64_t r_lo, r_hi, g_lo, g_hi, q_lo, q_hi;
		window(col_r + i * WORDS, p, r_lo, r_hi);
		window(col_g + i * WORDS, p, g_lo, g_hi);
		window(col_g + (i - 1) * WORDS, p, q_lo, q_hi);
		lo = r_lo & ~g_lo & ~q_lo;
		hi = r_hi & ~g_hi & ~q_hi;
	}
	// step j of the left search samples texels (x - 1 - 2j, x - 2j) of rows y - 1, y: the window [x - 127, x], mirrored
	__device__ __forceinline__ int run_left(int x, int y) const
	{
		uint64_t lo, hi;
		row_condition(x - 127, y, lo, hi);
		return leading_pairs(reverse64(hi), reverse64(lo));
	}
	// step j of the right search samples texels (x + 1 + 2j, x + 2 + 2j)
	__device__ __forceinline__ int run_right(int x, int y) const
	{
		uint64_t lo, hi;
		row_condition(x + 1, y, lo, hi);
		return leading_pairs(lo, hi);
	}
	__device__ __forceinline__ int run_up(int x, int y) const
	{
		uint64_t lo, hi;
		column_condition(x, y - 127, lo, hi);
		return leading_pairs(reverse64(hi), reverse64(lo));
	}
	__device__ __forceinline__ int run_down(int x, int y) const
	{
		uint64_t lo, hi;
		column_condition(x, y + 1, lo, hi);
		return leading_pairs(lo, hi);
	}

	// (R, G) of one texel as 0.0 / 1.0.  COLUMNS = false: from the staged rows (everything a horizontal edge and the diagonal
	// searches touch); true: from the staged columns (a vertical edge's searches and what follows them).  Every reach of the pass
	// is bounded by its search limits and lies inside the staged words; a texel outside them (not expected) is read from the image.
	template <bool COLUMNS>
	__device__ __forceinline__ v2 texel(int x, int y) const
	{
		const int line = COLUMNS ? x - (x0 - 2) : y - (y0 - 18), bit = COLUMNS ? y - (y0 - 128) : x - (x0 - 128);
		if (unsigned(line) >= unsigned(COLUMNS ? COLS : ROWS) || unsigned(bit) >= unsigned(WORDS * 64))
			return texel_from_image(x, y);
		const uint32_t *r32 = reinterpret_cast<const uint32_t *>(COLUMNS ? col_r : row_r), *g32 = reinterpret_cast<const uint32_t *>(COLUMNS ? col_g : row_g);
		const int k = line * (WORDS * 2) + (bit >> 5), s = bit & 31;
		return mk2(float((r32[k] >> s) & 1u), float((g32[k] >> s) & 1u));
	}
	__device__ __attribute__((noinline)) v2 texel_from_image(int x, int y) const
	{
		const uint32_t t = *reinterpret_cast<const uint16_t *>(image + (uint32_t(aa::clampi(y, 0, h - 1)) * pitch + uint32_t(aa::clampi(x, 0, w - 1)) * 2u));
		return mk2((t & 255u) ? 1.0f : 0.0f, (t >> 8) ? 1.0f : 0.0f);
	}

	// LinearClamp over the edge texture: the sampler of the byte image (aa.hip: Tex8::sample) with the texels taken from the bits.
	// 0.0 / 1.0 texels make t * (1 - a) + t' * a exact for a = 0, so the snapped cases need fewer texels, not another formula.
	template <bool COLUMNS = false>
	__device__ __forceinline__ v4 sample(v2 uv, int ox = 0, int oy = 0) const
	{
		int ix, iy;
		float a, b;
		aa::linear_axis(uv.x * float(w) - 0.5f, ix, a);
		aa::linear_axis(uv.y * float(h) - 0.5f, iy, b);
		ix += ox;
		iy += oy;
		const v2 t00 = texel<COLUMNS>(ix, iy);
		v2 top = t00;
		if (a != 0.0f)
		{
			const v2 t10 = texel<COLUMNS>(ix + 1, iy);
			top = t00 * (1.0f - a) + t10 * a;
		}
		if (b != 0.0f)
		{
			const v2 t01 = texel<COLUMNS>(ix, iy + 1);
			v2 bot = t01;
			if (a != 0.0f)
			{
				const v2 t11 = texel<COLUMNS>(ix + 1, iy + 1);
				bot = t01 * (1.0f - a) + t11 * a;
			}
			top = top * (1.0f - b) + bot * b;
		}
		return mk4(top.x, top.y, 0.0f, 1.0f);
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
};

// SMAABlendingWeightCalculationPS over the bit planes.  The reference runs the quad under a depth mask EQUAL to the edge pass's
// non-discarded pixels (smaa.cpp:101-112,170-177); the mask is the edge texel itself: zero edge => zero weights.
__global__ __launch_bounds__(FAST_BW *FAST_BH) void k_smaa_weights_bits(SmaaWeightsBitsArgs A, uint8_t *out, uint32_t out_pitch, RowSpan rows)
{
	using T = EdgeBitTiles;
	__shared__ uint64_t s_row_r[T::ROWS * T::WORDS], s_row_g[T::ROWS * T::WORDS], s_col_r[T::COLS * T::WORDS], s_col_g[T::COLS * T::WORDS];
	const int bx = blockIdx.x * FAST_BW, by = int(rows.first) + blockIdx.y * FAST_BH;
	const int x = bx + threadIdx.x, y = by + threadIdx.y;
	const bool inside = x < A.w && y < int(rows.end);
	uint32_t e = 0;
	if (inside)
		e = *reinterpret_cast<const uint16_t *>(A.edges + (uint32_t(y) * A.edges_pitch + uint32_t(x) * 2u));
	uint32_t *dst = reinterpret_cast<uint32_t *>(out + (uint32_t(y) * out_pitch + uint32_t(x) * 4u));
	if (!__syncthreads_or(e != 0u))
	{
		if (inside)
			*dst = 0u;
		return;
	}
	const int tid = threadIdx.y * FAST_BW + threadIdx.x;
	// words of the padded planes: bit 0 of the staged rows is texel x0 - 128 = plane bit x0 + 64, i.e. word bx / 64 + 1 when
	// bx is a multiple of 64, and a word-aligned start otherwise needs the shift below
	const int row_bit0 = bx - 128 + SMAA_BITS_PAD, col_bit0 = by - 128 + SMAA_BITS_PAD;
	for (int i = tid; i < T::ROWS * T::WORDS; i += FAST_BW * FAST_BH)
	{
		const int r = i / T::WORDS, k = i - r * T::WORDS;
		const size_t base = size_t(by - 18 + r + SMAA_BITS_PAD) * A.planes.row_words;
		const int bit = row_bit0 + 64 * k, wi = bit >> 6, sh = bit & 63;
		const uint64_t r0 = A.planes.row_r[base + wi], g0 = A.planes.row_g[base + wi];
		uint64_t vr = r0, vg = g0;
		if (sh)
		{
			vr = (r0 >> sh) | (A.planes.row_r[base + wi + 1] << (64 - sh));
			vg = (g0 >> sh) | (A.planes.row_g[base + wi + 1] << (64 - sh));
		}
		s_row_r[i] = vr;
		s_row_g[i] = vg;
	}
	for (int i = tid; i < T::COLS * T::WORDS; i += FAST_BW * FAST_BH)
	{
		const int c = i / T::WORDS, k = i - c * T::WORDS;
		const size_t base = size_t(bx - 2 + c + SMAA_BITS_PAD) * A.planes.col_words;
		const int bit = col_bit0 + 64 * k, wi = bit >> 6, sh = bit & 63;
		const uint64_t r0 = A.planes.col_r[base + wi], g0 = A.planes.col_g[base + wi];
		uint64_t vr = r0, vg = g0;
		if (sh)
		{
			vr = (r0 >> sh) | (A.planes.col_r[base + wi + 1] << (64 - sh));
			vg = (g0 >> sh) | (A.planes.col_g[base + wi + 1] << (64 - sh));
		}
		s_col_r[i] = vr;
		s_col_g[i] = vg;
	}
	__syncthreads();
	if (!inside)
		return;
	uint32_t packed = 0u;
	if (e != 0u)
	{
		SmaaWeights<EdgeBitTiles> S = {{s_row_r, s_row_g, s_col_r, s_col_g, bx, by, A.w, A.h, A.edges, A.edges_pitch}, A.area, A.search, A.rt, A.P};
		const v4 wgt = S.weights_at(x, y);
		packed = aa::unorm8_encode(wgt.x) | (aa::unorm8_encode(wgt.y) << 8) | (aa::unorm8_encode(wgt.z) << 16) | (aa::unorm8_encode(wgt.w) << 24);
	}
	*dst = packed;
}
