// The fast anti-aliasing kernels for gfx950: tiling, LDS staging and launch geometry around the per-pixel arithmetic of
// aa_core.hpp.  Included by aa.hip (after the HIP runtime and device_common.hpp) and, unchanged, by the host emulation the CPU
// tests run (tests/cpp/hip_emu.hpp: one std::thread per lane, barriers and wave votes as rendezvous) -- so indexing, halos,
// clamping and band handling are checked against the oracle before a GPU sees them.  Uses only what both environments
// provide: threadIdx / blockIdx, __shared__, __syncthreads*, float4 / uint4, min / max.
#pragma once
#include "aa_core.hpp"
#include "packed_float.hpp"
#include "row_span.hpp"

// ---- fast forms: pixel-centre taps as texel fetches (aa_core.hpp) ----------------------------------------------------------------
// Under the sampler model a tap on a pixel centre IS the texel, so FXAA's corner lumas, every tap of the SMAA edge pass and
// the weight fetches of the SMAA blend are exact reads of values that are decoded once.  The launchers use these kernels when
// aa::axis_taps_exact() holds for both axes of the image (any size below ~16K), the *_generic ones above otherwise.
constexpr int FAST_BW = 32, FAST_BH = 16; // 512 threads: 8 waves of two 32-pixel rows

// Tile order.  Workgroups take tiles in screen order.  An order that keeps whole tile rows on one XCD (so that the halo columns of
// horizontally adjacent tiles come out of one L2 instead of being fetched once per XCD: FETCH_SIZE is 2.1 - 2.9 x the algorithmic
// reads for these kernels) was built and measured in round 4 (profiles/r04_aa_tile_order_ab.txt): no kernel got faster, FXAA on the
// test card slower -- the second fetch is served by the memory-side cache.  Not kept.

__device__ __forceinline__ uint32_t load_rgba8_clamped(const uint8_t *ptr, uint32_t pitch, int w, int h, int x, int y)
{
	x = aa::clampi(x, 0, w - 1);
	y = aa::clampi(y, 0, h - 1);
	return *reinterpret_cast<const uint32_t *>(ptr + (uint32_t(y) * pitch + uint32_t(x) * 4u));
}

struct FxaaTile
{
	static constexpr int HALO = 6; // edge taps reach FXAA_SPAN_MAX / 2 = 4 texels, + 1 for the bilinear footprint, + 1 of slack
	static constexpr int W = FAST_BW + 2 * HALO, H = FAST_BH + 2 * HALO;
	const float4 *texels; // [H][W]: r, g, b, luma
	int ox, oy;
	__device__ __forceinline__ aa::f4 texel(int x, int y) const
	{
		const float4 t = texels[(y - oy) * W + (x - ox)];
		return {t.x, t.y, t.z, t.w};
	}
};

// Pixels whose four corners carry the same colour are copied (dir == 0: most of a rendered frame), decided on the bytes straight
// from the image: a workgroup of such pixels never stages or decodes anything.  The others are a small share of a workgroup that
// has any: they go into a list that the workgroup walks with all of its lanes (full waves on the 300-instruction path instead of
// a few live lanes per wave).
__global__ __launch_bounds__(FAST_BW *FAST_BH) void k_fxaa_fast(const uint8_t *in, uint32_t in_pitch, int w, int h, uint8_t *out, uint32_t out_pitch,
                                                                 float inv_w, float inv_h, RowSpan rows)
{
	constexpr int HALO = FxaaTile::HALO, TW = FxaaTile::W, TH = FxaaTile::H, THREADS = FAST_BW * FAST_BH, WAVES = THREADS / 64;
	__shared__ float4 s_dec[TW * TH];
	__shared__ uint16_t s_list[THREADS];
	__shared__ uint32_t s_wave_count[WAVES];
	const int bx = blockIdx.x * FAST_BW, by = int(rows.first) + blockIdx.y * FAST_BH;
	const int tid = threadIdx.y * FAST_BW + threadIdx.x, wave = tid >> 6, lane = tid & 63;
	// A workgroup whose tile (halo included) lies inside the image -- all but the frame of workgroups along the border -- needs no
	// clamping: neighbours at constant offsets from one address, the tile staged two texels per 8-byte load.
	const bool interior = bx >= HALO && by >= HALO && bx + FAST_BW + HALO <= w && by + FAST_BH + HALO <= h && (in_pitch & 7u) == 0u &&
	                      (reinterpret_cast<uintptr_t>(in) & 7u) == 0u;
	{
		const int x = bx + threadIdx.x, y = by + threadIdx.y;
		const bool inside = x < w && y < int(rows.end);
		uint32_t centre;
		bool flat;
		if (interior)
		{
			const uint8_t *p = in + (uint32_t(y) * in_pitch + uint32_t(x) * 4u);
			centre = *reinterpret_cast<const uint32_t *>(p);
			flat = aa::fxaa_corners_equal(*reinterpret_cast<const uint32_t *>(p - in_pitch - 4), *reinterpret_cast<const uint32_t *>(p - in_pitch + 4),
			                              *reinterpret_cast<const uint32_t *>(p + in_pitch - 4), *reinterpret_cast<const uint32_t *>(p + in_pitch + 4));
		}
		else
		{
			centre = load_rgba8_clamped(in, in_pitch, w, h, x, y);
			flat = aa::fxaa_corners_equal(load_rgba8_clamped(in, in_pitch, w, h, x - 1, y - 1), load_rgba8_clamped(in, in_pitch, w, h, x + 1, y - 1),
			                              load_rgba8_clamped(in, in_pitch, w, h, x - 1, y + 1), load_rgba8_clamped(in, in_pitch, w, h, x + 1, y + 1));
		}
		const bool work = inside && !flat;
		const uint64_t mine = __ballot(work);
		if (lane == 0)
			s_wave_count[wave] = uint32_t(__popcll(mine));
		if (inside && flat)
			*reinterpret_cast<uint32_t *>(out + (uint32_t(y) * out_pitch + uint32_t(x) * 4u)) = centre | 0xff000000u;
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
		if (work)
			s_list[before + uint32_t(__popcll(mine & ((1ull << lane) - 1ull)))] = uint16_t(tid);
	}
	const auto decode = [](uint32_t t) {
		const float r = aa::unorm8_decode(t & 255u), g = aa::unorm8_decode((t >> 8) & 255u), b = aa::unorm8_decode((t >> 16) & 255u);
		return make_float4(r, g, b, aa::luma_of(r, g, b, aa::FXAA_LUMA_R, aa::FXAA_LUMA_G, aa::FXAA_LUMA_B));
	};
	if (interior)
	{
		static_assert(TW % 2 == 0, "two texels per load");
		const uint8_t *origin = in + (uint32_t(by - HALO) * in_pitch + uint32_t(bx - HALO) * 4u);
		for (int i = tid; i < (TW / 2) * TH; i += THREADS)
		{
			const int ty = i / (TW / 2), tx = 2 * (i - ty * (TW / 2));
			const uint2 t = *reinterpret_cast<const uint2 *>(origin + (uint32_t(ty) * in_pitch + uint32_t(tx) * 4u));
			s_dec[ty * TW + tx] = decode(t.x);
			s_dec[ty * TW + tx + 1] = decode(t.y);
		}
	}
	else
		for (int i = tid; i < TW * TH; i += THREADS)
		{
			const int ty = i / TW, tx = i - ty * TW;
			s_dec[i] = decode(load_rgba8_clamped(in, in_pitch, w, h, bx - HALO + tx, by - HALO + ty));
		}
	__syncthreads();
	uint32_t total = 0;
	for (int i = 0; i < WAVES; i++)
		total += AA_WAVE_UNIFORM(s_wave_count[i]);
	const FxaaTile tile = {s_dec, bx - HALO, by - HALO};
	for (uint32_t i = uint32_t(tid); i < total; i += uint32_t(THREADS))
	{
		const int t = total == uint32_t(THREADS) ? int(i) : int(s_list[i]), x = bx + (t & (FAST_BW - 1)), y = by + (t / FAST_BW);
		*reinterpret_cast<uint32_t *>(out + (uint32_t(y) * out_pitch + uint32_t(x) * 4u)) = aa::fxaa_pixel(tile, x, y, inv_w, inv_h, float(w), float(h));
	}
}

struct EdgeLumaTile
{
	static constexpr int W = FAST_BW + 3, H = FAST_BH + 3; // two texels to the left / above, one to the right / below
	const float *values;
	int ox, oy;
	__device__ __forceinline__ float luma(int x, int y) const { return values[(y - oy) * W + (x - ox)]; }
};

__global__ __launch_bounds__(FAST_BW *FAST_BH) void k_smaa_edges_fast(const uint8_t *in, uint32_t in_pitch, int w, int h, uint8_t *edges, uint32_t edges_pitch,
                                                                       float threshold, RowSpan rows)
{
	constexpr int TW = EdgeLumaTile::W, TH = EdgeLumaTile::H;
	__shared__ float s_luma[TW * TH];
	const int bx = blockIdx.x * FAST_BW, by = int(rows.first) + blockIdx.y * FAST_BH;
	for (int i = threadIdx.y * FAST_BW + threadIdx.x; i < TW * TH; i += FAST_BW * FAST_BH)
	{
		const int ty = i / TW, tx = i - ty * TW;
		const uint32_t t = load_rgba8_clamped(in, in_pitch, w, h, bx - 2 + tx, by - 2 + ty);
		s_luma[i] = aa::luma_of(aa::unorm8_decode(t & 255u), aa::unorm8_decode((t >> 8) & 255u), aa::unorm8_decode((t >> 16) & 255u), aa::SMAA_LUMA_R,
		                        aa::SMAA_LUMA_G, aa::SMAA_LUMA_B);
	}
	__syncthreads();
	const int x = bx + threadIdx.x, y = by + threadIdx.y;
	if (x >= w || y >= int(rows.end))
		return;
	const EdgeLumaTile tile = {s_luma, bx - 2, by - 2};
	*reinterpret_cast<uint16_t *>(edges + (uint32_t(y) * edges_pitch + uint32_t(x) * 2u)) = uint16_t(aa::smaa_edges_pixel(tile, x, y, threshold));
}

struct ColorImage
{
	const uint8_t *ptr;
	uint32_t pitch;
	int w, h;
	__device__ __forceinline__ uint32_t raw(int x, int y) const { return load_rgba8_clamped(ptr, pitch, w, h, x, y); }
};

// PX pixels per lane along x (PX = 4: 16-byte accesses; needs width % 4 == 0 and 16-byte aligned rows).  No LDS: a pixel
// without weights -- nearly all of them -- is three weight texels and a copy; the few with weights take two one-axis lerps
// of colour texels straight from the image.
template <int PX>
__global__ __launch_bounds__(256) void k_smaa_blend_fast(ColorImage color, ColorImage weights, uint8_t *out, uint32_t out_pitch, float rt_x, float rt_y,
                                                          RowSpan rows)
{
	const int x0 = (blockIdx.x * 64 + threadIdx.x) * PX, y = int(rows.first) + blockIdx.y * 4 + threadIdx.y;
	if (x0 >= color.w || y >= int(rows.end))
		return;
	const int yb = min(y + 1, color.h - 1);
	uint32_t wc[PX + 1], wb[PX], col[PX];
	if (PX == 4)
	{
		const uint4 a = *reinterpret_cast<const uint4 *>(weights.ptr + (uint32_t(y) * weights.pitch + uint32_t(x0) * 4u));
		const uint4 b = *reinterpret_cast<const uint4 *>(weights.ptr + (uint32_t(yb) * weights.pitch + uint32_t(x0) * 4u));
		const uint4 c = *reinterpret_cast<const uint4 *>(color.ptr + (uint32_t(y) * color.pitch + uint32_t(x0) * 4u));
		wc[0] = a.x, wc[1] = a.y, wc[2] = a.z, wc[3] = a.w;
		wb[0] = b.x, wb[1] = b.y, wb[2] = b.z, wb[3] = b.w;
		col[0] = c.x, col[1] = c.y, col[2] = c.z, col[3] = c.w;
		wc[PX] = weights.raw(x0 + PX, y);
	}
	else
	{
		wc[0] = weights.raw(x0, y);
		wc[1] = weights.raw(x0 + 1, y);
		wb[0] = weights.raw(x0, yb);
		col[0] = color.raw(x0, y);
	}
	uint32_t result[PX];
#pragma unroll
	for (int i = 0; i < PX; i++)
	{
		// a = (right.a, bottom.g, this.b, this.r): all zero -> the colour texel itself
		if (((wc[i + 1] >> 24) | ((wb[i] >> 8) & 255u) | ((wc[i] >> 16) & 255u) | (wc[i] & 255u)) == 0u)
			result[i] = col[i];
		else
			result[i] = aa::smaa_blend_pixel(color, wc[i], wc[i + 1], wb[i], x0 + i, y, rt_x, rt_y, float(color.w), float(color.h));
	}
	uint8_t *dst = out + (uint32_t(y) * out_pitch + uint32_t(x0) * 4u);
	if (PX == 4)
		*reinterpret_cast<uint4 *>(dst) = make_uint4(result[0], result[1], result[2], result[3]);
	else
		*reinterpret_cast<uint32_t *>(dst) = result[0];
}


// ---- TAA resolve ---------------------------------------------------------------------------------------------------------------
// The 3 x 3 neighbourhood of every pixel of a 32 x 16 block comes out of LDS: the 34 x 18 texels around the block are fetched,
// clamped to the image and converted to the resolve's colour space ONCE, depth in the fourth component (one ds_read_b128 per
// neighbour).  The history taps go to the image: sixteen texels per pixel at TAAQuality High (aa_core.hpp: taa_pixel).
struct TaaImages
{
	const uint8_t *current, *depth, *mv, *history;
	uint8_t *out_color, *out_history;
	uint32_t current_pitch, depth_pitch, mv_pitch, history_pitch, out_color_pitch, out_history_pitch;
	int w, h;
	// B10G11R11_UFLOAT_PACK32 instead of RGBA16F: the input when the HDR targets are packed (renderTargetFp16 = false), the colour
	// output as the reference declares it (temporal.cpp:211-213).  The history is RGBA16F either way (temporal.cpp:216).
	int current_b10, color_b10;
	// Row bands: history rows [hist_first, hist_end) hold last frame's values on this rank; a pixel whose reprojection fetches a row
	// outside sets *reach_flag (host-visible memory) -- the frame is then not the single-device frame and the executor says so.
	// reach_flag == nullptr: the whole history is there, nothing to check.
	int hist_first, hist_end;
	uint32_t *reach_flag;
};

// one texel of the current frame as RGBA16F dwords, whatever its storage
__device__ __forceinline__ uint2 taa_load_current(const TaaImages &im, int x, int y)
{
	if (im.current_b10)
	{
		uint32_t rg, ba;
		expand_b10g11r11(*reinterpret_cast<const uint32_t *>(im.current + (uint32_t(y) * im.current_pitch + uint32_t(x) * 4u)), rg, ba);
		return make_uint2(rg, ba);
	}
	return *reinterpret_cast<const uint2 *>(im.current + (uint32_t(y) * im.current_pitch + uint32_t(x) * 8u));
}

struct TaaTile
{
	static constexpr int W = FAST_BW + 2, H = FAST_BH + 2;
	const float4 *texels;
	int lx, ly; // position of the pixel inside the tile
	__device__ __forceinline__ aa::f4 cur(int ox, int oy) const
	{
		const float4 t = texels[(ly + oy) * W + (lx + ox)];
		return {t.x, t.y, t.z, t.w};
	}
};
struct TaaMotion
{
	const uint8_t *ptr;
	uint32_t pitch;
	int w, h;
	__device__ __forceinline__ uint32_t mv(int x, int y) const
	{
		return *reinterpret_cast<const uint32_t *>(ptr + (uint32_t(aa::clampi(y, 0, h - 1)) * pitch + uint32_t(aa::clampi(x, 0, w - 1)) * 4u));
	}
};
struct TaaHistory
{
	const uint8_t *ptr;
	uint32_t pitch;
	__device__ __forceinline__ aa::u2 texel(int x, int y) const
	{
		const uint2 t = *reinterpret_cast<const uint2 *>(ptr + (uint32_t(y) * pitch + uint32_t(x) * 8u));
		return {t.x, t.y};
	}
	// texels (x .. x + 3, y), all inside the image: two 16-byte loads from one address (8-byte aligned: the hardware takes it)
	__device__ __forceinline__ void row4(int x, int y, aa::u2 (&out)[4]) const
	{
#if defined(__HIP_DEVICE_COMPILE__)
		typedef uint32_t words4 __attribute__((ext_vector_type(4), aligned(8)));
		const words4 *p = reinterpret_cast<const words4 *>(ptr + (uint32_t(y) * pitch + uint32_t(x) * 8u));
		const words4 a = p[0], b = p[1];
		out[0] = {a.x, a.y}, out[1] = {a.z, a.w}, out[2] = {b.x, b.y}, out[3] = {b.z, b.w};
#else
		for (int i = 0; i < 4; i++)
			out[i] = texel(x + i, y);
#endif
	}
};

template <int QUALITY, bool HISTORY>
__global__ __launch_bounds__(FAST_BW *FAST_BH) void k_taa_fast(TaaImages im, aa::TaaPush push, RowSpan rows)
{
	constexpr int TW = TaaTile::W, TH = TaaTile::H;
	__shared__ float4 s_cur[TW * TH];
	const int bx = blockIdx.x * FAST_BW, by = int(rows.first) + blockIdx.y * FAST_BH;
	const int x = bx + threadIdx.x, y = by + threadIdx.y;
	if (!HISTORY)
	{
		// first frame (REPROJECTION_HISTORY = 0): the colour goes through the resolve's colour space and back
		if (x >= im.w || y >= int(rows.end))
			return;
		const uint2 t = taa_load_current(im, x, y);
		const aa::f3 c = aa::taa_from_hdr(aa::half_lo(t.x), aa::half_hi(t.x), aa::half_lo(t.y));
		const aa::f3 o = aa::taa_to_hdr(c);
		if (im.color_b10)
			*reinterpret_cast<uint32_t *>(im.out_color + (uint32_t(y) * im.out_color_pitch + uint32_t(x) * 4u)) = pack_b10g11r11(o.x, o.y, o.z);
		else
			*reinterpret_cast<uint2 *>(im.out_color + (uint32_t(y) * im.out_color_pitch + uint32_t(x) * 8u)) =
			    make_uint2(aa::pack_half2_rne(o.x, o.y), aa::pack_half2_rne(o.z, 1.0f));
		*reinterpret_cast<uint2 *>(im.out_history + (uint32_t(y) * im.out_history_pitch + uint32_t(x) * 8u)) =
		    make_uint2(aa::pack_half2_rne(c.x, c.y), aa::pack_half2_rne(c.z, 1.0f));
		return;
	}
	for (int i = threadIdx.y * FAST_BW + threadIdx.x; i < TW * TH; i += FAST_BW * FAST_BH)
	{
		const int ty = i / TW, tx = i - ty * TW;
		const int px = aa::clampi(bx + tx - 1, 0, im.w - 1), py = aa::clampi(by + ty - 1, 0, im.h - 1);
		const uint2 t = taa_load_current(im, px, py);
		const aa::f3 c = aa::taa_from_hdr(aa::half_lo(t.x), aa::half_hi(t.x), aa::half_lo(t.y));
		s_cur[i] = make_float4(c.x, c.y, c.z, *reinterpret_cast<const float *>(im.depth + (uint32_t(py) * im.depth_pitch + uint32_t(px) * 4u)));
	}
	__syncthreads();
	if (x >= im.w || y >= int(rows.end))
		return;
	const TaaTile tile = {s_cur, int(threadIdx.x) + 1, int(threadIdx.y) + 1};
	const TaaMotion motion = {im.mv, im.mv_pitch, im.w, im.h};
	const TaaHistory history = {im.history, im.history_pitch};
	aa::u2 color, hist;
	aa::f3 color_f;
	int row_first, row_last;
	aa::taa_pixel<QUALITY>(tile, motion, history, x, y, im.w, im.h, push, color, hist, color_f, row_first, row_last);
	if (im.reach_flag && (row_first < im.hist_first || row_last >= im.hist_end))
		*im.reach_flag = 1u;
	if (im.color_b10)
		*reinterpret_cast<uint32_t *>(im.out_color + (uint32_t(y) * im.out_color_pitch + uint32_t(x) * 4u)) = pack_b10g11r11(color_f.x, color_f.y, color_f.z);
	else
		*reinterpret_cast<uint2 *>(im.out_color + (uint32_t(y) * im.out_color_pitch + uint32_t(x) * 8u)) = make_uint2(color.x, color.y);
	*reinterpret_cast<uint2 *>(im.out_history + (uint32_t(y) * im.out_history_pitch + uint32_t(x) * 8u)) = make_uint2(hist.x, hist.y);
}
