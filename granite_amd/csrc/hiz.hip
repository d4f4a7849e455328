// Depth hierarchy for gfx950: HiZPassState::build_render_pass (renderer/post/spd.cpp:141-194) + assets/shaders/post/hiz.comp.
// The work is a stream: 4 B read per texel, 4 B (+1/3) written, nothing re-read from HBM.
//
// Mapping (own design, not the shader's): a workgroup of four waves owns a 64 x 64 texel tile, a wave a 32 x 32 quadrant, a
// lane a 4 x 4 block (four 16-byte loads; eight lanes cover a 128-byte line of every row).  Lanes are numbered along a
// Morton curve inside the quadrant, so mips 1 and 2 are per-lane register work, mips 3 / 4 / 5 are xor-shuffles over lane
// bits (0,1) / (2,3) / (4,5) -- no LDS, no barrier -- and mip 6 is one LDS exchange between the four waves.  A one-workgroup
// tail launch reduces mip 6 to the end of the chain in LDS (k_hiz_tail).
//
// (Non-temporal loads / stores were tried: +9 % at 8K with the full-resolution level, -10..15 % elsewhere; not kept.)
//
// Max-reductions are exact in any order; the depth transform is fp32 with IEEE division and no contraction
// (-ffp-contract=off for this file), so the chain is bit-identical to the oracle.
#include "ctx.hpp"
#include "device_common.hpp"

namespace
{
constexpr int HIZ_MAX_MIPS = 16;

struct HizParams
{
	const uint8_t *depth;
	int iw, ih;
	uint32_t pitch;
	int aligned; // rows can be fetched as 16-byte vectors
	float *chain;
	int res_w, res_h; // push.resolution: mip 0 size, multiples of 64
	int mips;         // push.mips
	int top;          // WRITE_TOP_LEVEL
	float m0, m1, m2, m3;
	uint32_t offset[HIZ_MAX_MIPS]; // float offset of mip m inside the chain
};

__device__ __forceinline__ float transform_z(const HizParams &p, float z)
{
	// hiz.comp:65-72: (num, den) = z_transform * vec2(z, 1); min(num / den, 1e30) with GLSL's min(x, y) = y < x ? y : x.
	const float num = p.m0 * z + p.m2;
	const float den = p.m1 * z + p.m3;
	const float q = num / den;
	return (1e30f < q) ? 1e30f : q;
}

__device__ __forceinline__ float max4(float a, float b, float c, float d) { return fmaxf(fmaxf(a, b), fmaxf(c, d)); }

// One texel of mip `mip` from the finer level (pw x ph, row-major at `src`): 2 x 2 footprint, plus the row / column that
// halving an odd size would drop when this is the last texel (hiz.comp:178-247); coordinates clamp to the finer level.
template <typename Fetch>
__device__ __forceinline__ float reduce_folded(Fetch fetch, int x, int y, int w, int h, int pw, int ph)
{
	const int nx = 2 + ((x + 1 == w && (pw & 1)) ? 1 : 0);
	const int ny = 2 + ((y + 1 == h && (ph & 1)) ? 1 : 0);
	float r = fetch(min(2 * x, pw - 1), min(2 * y, ph - 1));
	for (int j = 0; j < ny; j++)
		for (int i = 0; i < nx; i++)
			r = fmaxf(r, fetch(min(2 * x + i, pw - 1), min(2 * y + j, ph - 1)));
	return r;
}

__global__ __launch_bounds__(256) void k_hiz_tiles(HizParams p)
{
	__shared__ float wave_top[4];

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int tx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4);
	const int ty = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
	const int qx = wave & 1, qy = wave >> 1;
	const int x0 = int(blockIdx.x) * 64 + qx * 32 + tx * 4;
	const int y0 = int(blockIdx.y) * 64 + qy * 32 + ty * 4;

	float v[4][4];
	const bool vector_rows = p.aligned && x0 + 3 < p.iw;
#pragma unroll
	for (int j = 0; j < 4; j++)
	{
		const uint8_t *row = p.depth + size_t(min(y0 + j, p.ih - 1)) * p.pitch;
		if (vector_rows)
		{
			const f32x4 t = *reinterpret_cast<const f32x4 *>(row + size_t(x0) * 4u);
			v[j][0] = t.x, v[j][1] = t.y, v[j][2] = t.z, v[j][3] = t.w;
		}
		else
		{
#pragma unroll
			for (int i = 0; i < 4; i++)
				v[j][i] = *reinterpret_cast<const float *>(row + size_t(min(x0 + i, p.iw - 1)) * 4u);
		}
#pragma unroll
		for (int i = 0; i < 4; i++)
			v[j][i] = transform_z(p, v[j][i]);
	}

	if (p.top)
	{
		float *l0 = p.chain + p.offset[0];
#pragma unroll
		for (int j = 0; j < 4; j++)
			*reinterpret_cast<f32x4 *>(l0 + size_t(y0 + j) * p.res_w + x0) = f32x4{v[j][0], v[j][1], v[j][2], v[j][3]};
	}
	if (p.mips <= 1)
		return;

	float h[2][2];
#pragma unroll
	for (int j = 0; j < 2; j++)
#pragma unroll
		for (int i = 0; i < 2; i++)
			h[j][i] = max4(v[2 * j][2 * i], v[2 * j][2 * i + 1], v[2 * j + 1][2 * i], v[2 * j + 1][2 * i + 1]);
	{
		float *l1 = p.chain + p.offset[1];
		const int w1 = p.res_w >> 1;
#pragma unroll
		for (int j = 0; j < 2; j++)
			*reinterpret_cast<float2 *>(l1 + size_t((y0 >> 1) + j) * w1 + (x0 >> 1)) = make_float2(h[j][0], h[j][1]);
	}
	if (p.mips <= 2)
		return;

	float r = max4(h[0][0], h[0][1], h[1][0], h[1][1]);
	p.chain[p.offset[2] + size_t(y0 >> 2) * (p.res_w >> 2) + (x0 >> 2)] = r;
	if (p.mips <= 3)
		return;

	const int bx = int(blockIdx.x), by = int(blockIdx.y);
	r = fmaxf(r, __shfl_xor(r, 1, 64));
	r = fmaxf(r, __shfl_xor(r, 2, 64));
	if ((lane & 3) == 0)
		p.chain[p.offset[3] + size_t(by * 8 + qy * 4 + (ty >> 1)) * (p.res_w >> 3) + (bx * 8 + qx * 4 + (tx >> 1))] = r;
	if (p.mips <= 4)
		return;

	r = fmaxf(r, __shfl_xor(r, 4, 64));
	r = fmaxf(r, __shfl_xor(r, 8, 64));
	if ((lane & 15) == 0)
		p.chain[p.offset[4] + size_t(by * 4 + qy * 2 + (ty >> 2)) * (p.res_w >> 4) + (bx * 4 + qx * 2 + (tx >> 2))] = r;
	if (p.mips <= 5)
		return;

	r = fmaxf(r, __shfl_xor(r, 16, 64));
	r = fmaxf(r, __shfl_xor(r, 32, 64));
	if (lane == 0)
	{
		p.chain[p.offset[5] + size_t(by * 2 + qy) * (p.res_w >> 5) + (bx * 2 + qx)] = r;
		wave_top[wave] = r;
	}
	if (p.mips <= 6)
		return;

	__syncthreads();
	if (threadIdx.x == 0)
		p.chain[p.offset[6] + size_t(by) * (p.res_w >> 6) + bx] = max4(wave_top[0], wave_top[1], wave_top[2], wave_top[3]);
}

// Mips 7 and up: what is left after the tiles (at most 128 x 128 texels of mip 6 for a 8192 x 8192 target) is reduced by one
// workgroup through LDS, folding the odd row / column of non-power-of-two levels.  A second launch on the same stream rather
// than the shader's "last workgroup takes a ticket" (hiz.comp:355-365): the tiles are written through eight XCD-private L2s,
// so handing mip 6 to one workgroup inside the launch costs every workgroup an agent-scope release (an L2 write-back) plus a
// memory-side atomic on one address -- measured ~80 ns per workgroup, serialised: 155 us at 4K against 20 us of streaming.
// The kernel boundary does the same hand-over once.
__global__ __launch_bounds__(256) void k_hiz_tail(HizParams p)
{
	extern __shared__ float tail_lds[];
	const int w6 = p.res_w >> 6, h6 = p.res_h >> 6;
	int pw = max(p.res_w >> 7, 1), ph = max(p.res_h >> 7, 1);
	float *cur = tail_lds;
	float *next = tail_lds + pw * ph;
	{
		const float *l6 = p.chain + p.offset[6];
		float *l7 = p.chain + p.offset[7];
		auto fetch6 = [&](int x, int y) { return l6[size_t(y) * w6 + x]; };
		for (int i = threadIdx.x; i < pw * ph; i += 256)
		{
			const int y = i / pw, x = i - y * pw;
			const float t = reduce_folded(fetch6, x, y, pw, ph, w6, h6);
			cur[i] = t;
			l7[i] = t;
		}
	}
	for (int mip = 8; mip < p.mips; mip++)
	{
		__syncthreads();
		const int w = max(p.res_w >> mip, 1), h = max(p.res_h >> mip, 1);
		float *out = p.chain + p.offset[mip];
		const float *src = cur;
		auto fetch = [&](int x, int y) { return src[y * pw + x]; };
		for (int i = threadIdx.x; i < w * h; i += 256)
		{
			const int y = i / w, x = i - y * w;
			const float t = reduce_folded(fetch, x, y, w, h, pw, ph);
			next[i] = t;
			out[i] = t;
		}
		float *swap = cur;
		cur = next;
		next = swap;
		pw = w;
		ph = h;
	}
}
} // namespace

extern "C" {

size_t gr_mip_chain_offset(uint32_t width, uint32_t height, uint32_t bytes_per_texel, uint32_t level)
{
	size_t texels = 0;
	for (uint32_t l = 0; l < level; l++)
	{
		const uint32_t w = width >> l, h = height >> l;
		texels += size_t(w ? w : 1u) * size_t(h ? h : 1u);
	}
	return texels * bytes_per_texel;
}

size_t gr_mip_chain_size(uint32_t width, uint32_t height, uint32_t bytes_per_texel, uint32_t levels)
{
	return gr_mip_chain_offset(width, height, bytes_per_texel, levels);
}

int gr_hiz(gr_ctx *ctx, gr_stream stream, const gr_hiz_args *args)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, args && args->depth.ptr && args->chain && args->counter);
	GR_CHECK_ARG(ctx, args->depth.format == GR_FORMAT_D32_SFLOAT || args->depth.format == GR_FORMAT_R32_SFLOAT);
	GR_CHECK_ARG(ctx, args->depth.width > 0 && args->depth.height > 0 && args->depth.pitch_bytes >= args->depth.width * 4u);
	const uint32_t ds = args->output_downsample ? 1u : 0u;
	const uint32_t res_w = args->chain_width << ds, res_h = args->chain_height << ds;
	GR_CHECK_ARG(ctx, res_w > 0 && res_h > 0 && (res_w & 63u) == 0 && (res_h & 63u) == 0);
	GR_CHECK_ARG(ctx, res_w >= args->depth.width && res_h >= args->depth.height);
	const uint32_t mips = args->chain_levels + ds;
	GR_CHECK_ARG(ctx, args->chain_levels >= 1 && mips <= uint32_t(HIZ_MAX_MIPS));

	HizParams p = {};
	p.depth = static_cast<const uint8_t *>(args->depth.ptr);
	p.iw = int(args->depth.width);
	p.ih = int(args->depth.height);
	p.pitch = args->depth.pitch_bytes;
	p.aligned = ((reinterpret_cast<uintptr_t>(args->depth.ptr) & 15u) == 0 && (args->depth.pitch_bytes & 15u) == 0) ? 1 : 0;
	p.chain = static_cast<float *>(args->chain);
	p.res_w = int(res_w);
	p.res_h = int(res_h);
	p.mips = int(mips);
	p.top = ds ? 0 : 1;
	p.m0 = args->z_transform[0];
	p.m1 = args->z_transform[1];
	p.m2 = args->z_transform[2];
	p.m3 = args->z_transform[3];
	for (uint32_t m = ds; m < mips; m++)
		p.offset[m] = uint32_t(gr_mip_chain_offset(args->chain_width, args->chain_height, 1, m - ds));

	size_t lds = 0;
	if (mips > 7)
	{
		const size_t t7 = size_t(std::max(res_w >> 7, 1u)) * std::max(res_h >> 7, 1u);
		const size_t t8 = size_t(std::max(res_w >> 8, 1u)) * std::max(res_h >> 8, 1u);
		lds = (t7 + t8) * sizeof(float);
		if (lds > 60u * 1024u)
			return ctx->fail(GR_ERR_INVALID_ARGUMENT, "gr_hiz: %u x %u is beyond the single-launch tail (mip 7 must fit LDS)", res_w, res_h);
	}
	gr_scoped_timing timing(ctx, gr_to_stream(stream), "hiz");
	hipLaunchKernelGGL(k_hiz_tiles, dim3(res_w / 64u, res_h / 64u), dim3(256), 0, gr_to_stream(stream), p);
	GR_CHECK_LAUNCH(ctx);
	if (mips > 7)
	{
		hipLaunchKernelGGL(k_hiz_tail, dim3(1), dim3(256), lds, gr_to_stream(stream), p);
		GR_CHECK_LAUNCH(ctx);
	}
	return GR_OK;
}

} // extern "C"
