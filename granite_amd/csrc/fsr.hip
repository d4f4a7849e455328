// Spatial upscaling after the post chain for gfx950: setup_after_post_chain_upscaling (renderer/post/aa.cpp:75-174) with
// assets/shaders/post/ffx-fsr/{upscale,sharpen}.{vert,frag}.  The arithmetic is AMD FidelityFX FSR 1.0 -- EASU (edge
// adaptive 12-tap Lanczos-like resampling) and RCAS (contrast adaptive sharpening) -- restated from the published
// algorithm; see oracle/oracle_fsr.cpp for the statement both sides are held to.
//
// The 12 taps of EASU are texel fetches at fixed offsets from floor(pp) (upscale.frag gathers with NearestClamp from the
// UNORM alias), so the kernel reads raw RGBA8 words.  FP16 = 1 (what the reference selects on hardware with fp16
// arithmetic, aa.cpp:118-119) maps onto packed v_pk_*_f16: the shader's "two taps per operation" is two lanes of a half2.
// All operations are correctly rounded and uncontracted (-ffp-contract=off); the two divisions of the half path go
// through fp32 (exactly rounded for half operands); min / max are IEEE minNum / maxNum.  EASU is therefore bit-exact
// against the oracle; RCAS into an *_SRGB target differs by the device's pow in the encode (<= 1 LSB).
#include "ctx.hpp"
#include "device_common.hpp"

namespace
{
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
constexpr int FSR_BLOCK_X = 32;
constexpr int FSR_BLOCK_Y = 8;
static_assert(FSR_BLOCK_X * FSR_BLOCK_Y == 256, "the RCAS kernel stages a 256-entry table with one entry per thread");

struct Rgba8
{
	const uint8_t *ptr;
	int w, h;
	uint32_t pitch;
	__device__ __forceinline__ uint32_t at(int x, int y) const
	{
		return *reinterpret_cast<const uint32_t *>(ptr + size_t(clampi(y, 0, h - 1)) * pitch + size_t(clampi(x, 0, w - 1)) * 4u);
	}
};

__device__ __forceinline__ float fast_rcp32(float a) { return __uint_as_float(0x7ef07ebbu - __float_as_uint(a)); }
__device__ __forceinline__ float fast_rsq32(float a) { return __uint_as_float(0x5f347d74u - (__float_as_uint(a) >> 1)); }
__device__ __forceinline__ float medium_rcp32(float a)
{
	const float b = __uint_as_float(0x7ef19fffu - __float_as_uint(a));
	return b * (-b * a + 2.0f);
}
__device__ __forceinline__ float sat32(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ uint32_t to_unorm8(float v)
{
	if (!(v > 0.0f))
		return 0u;
	if (v >= 1.0f)
		return 255u;
	return uint32_t(int(v * 255.0f + 0.5f));
}

__device__ __forceinline__ uint16_t hbits(_Float16 v) { return __builtin_bit_cast(uint16_t, v); }
__device__ __forceinline__ _Float16 hfrom(uint16_t b) { return __builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ _Float16 fast_rcp16(_Float16 a) { return hfrom(uint16_t(0x7784u - hbits(a))); }
__device__ __forceinline__ _Float16 fast_rsq16(_Float16 a) { return hfrom(uint16_t(0x59a3u - (hbits(a) >> 1))); }
__device__ __forceinline__ h2 hmax2(h2 a, h2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ h2 hmin2(h2 a, h2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ h2 habs2(h2 a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ _Float16 hmax1(_Float16 a, _Float16 b) { return __builtin_fmaxf16(a, b); }
__device__ __forceinline__ _Float16 hmin1(_Float16 a, _Float16 b) { return __builtin_fminf16(a, b); }
__device__ __forceinline__ h2 hsat2(h2 a) { return hmin2(hmax2(a, h2{0, 0}), h2{1, 1}); }
// Exactly rounded half division through fp32 (24 >= 2 * 11 + 2 significand bits).
__device__ __forceinline__ _Float16 hdiv(_Float16 a, _Float16 b) { return _Float16(float(a) / float(b)); }
__device__ __forceinline__ h2 hrcp2(h2 a) { return h2{hdiv(_Float16(1), a.x), hdiv(_Float16(1), a.y)}; }

struct EasuArgs
{
	Rgba8 in;
	uint8_t *out;
	uint32_t out_pitch;
	int ow, oh;
	float scale_x, scale_y, offset_x, offset_y;
};

// Tap order:      0:b 1:c
//            2:e 3:f 4:g 5:h
//            6:i 7:j 8:k 9:l
//                10:n 11:o
__device__ __forceinline__ void fetch_taps(const Rgba8 &in, int fx, int fy, uint32_t t[12])
{
	t[0] = in.at(fx, fy - 1), t[1] = in.at(fx + 1, fy - 1);
	t[2] = in.at(fx - 1, fy), t[3] = in.at(fx, fy), t[4] = in.at(fx + 1, fy), t[5] = in.at(fx + 2, fy);
	t[6] = in.at(fx - 1, fy + 1), t[7] = in.at(fx, fy + 1), t[8] = in.at(fx + 1, fy + 1), t[9] = in.at(fx + 2, fy + 1);
	t[10] = in.at(fx, fy + 2), t[11] = in.at(fx + 1, fy + 2);
}

__device__ __forceinline__ float chan(uint32_t texel, int c) { return unorm8_to_float((texel >> (8 * c)) & 255u); }

// ---- EASU, fp32 ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FSR_BLOCK_X *FSR_BLOCK_Y) void k_fsr_easu_f32(EasuArgs a)
{
	const int x = blockIdx.x * FSR_BLOCK_X + threadIdx.x;
	const int y = blockIdx.y * FSR_BLOCK_Y + threadIdx.y;
	if (x >= a.ow || y >= a.oh)
		return;
	float px = float(x) * a.scale_x + a.offset_x;
	float py = float(y) * a.scale_y + a.offset_y;
	const float fx = floorf(px), fy = floorf(py);
	px -= fx;
	py -= fy;

	uint32_t texel[12];
	fetch_taps(a.in, int(fx), int(fy), texel);
	float r[12], g[12], b[12], l[12];
#pragma unroll
	for (int t = 0; t < 12; t++)
	{
		r[t] = chan(texel[t], 0), g[t] = chan(texel[t], 1), b[t] = chan(texel[t], 2);
		l[t] = b[t] * 0.5f + (r[t] * 0.5f + g[t]);
	}

	float dir_x = 0.0f, dir_y = 0.0f, len = 0.0f;
	auto analyse = [&](float weight, float up, float left, float centre, float right, float down) {
		const float dc = right - centre, cb = centre - left;
		float lx = fast_rcp32(fmaxf(fabsf(dc), fabsf(cb)));
		const float gx = right - left;
		dir_x += gx * weight;
		lx = sat32(fabsf(gx) * lx);
		lx *= lx;
		len += lx * weight;
		const float ec = down - centre, ca = centre - up;
		float ly = fast_rcp32(fmaxf(fabsf(ec), fabsf(ca)));
		const float gy = down - up;
		dir_y += gy * weight;
		ly = sat32(fabsf(gy) * ly);
		ly *= ly;
		len += ly * weight;
	};
	analyse((1.0f - px) * (1.0f - py), l[0], l[2], l[3], l[4], l[7]);
	analyse(px * (1.0f - py), l[1], l[3], l[4], l[5], l[8]);
	analyse((1.0f - px) * py, l[3], l[6], l[7], l[8], l[10]);
	analyse(px * py, l[4], l[7], l[8], l[9], l[11]);

	float norm = dir_x * dir_x + dir_y * dir_y;
	const bool flat = norm < float(1.0 / 32768.0);
	norm = fast_rsq32(norm);
	if (flat)
	{
		norm = 1.0f;
		dir_x = 1.0f;
	}
	dir_x *= norm;
	dir_y *= norm;
	len = len * 0.5f;
	len *= len;
	const float stretch = (dir_x * dir_x + dir_y * dir_y) * fast_rcp32(fmaxf(fabsf(dir_x), fabsf(dir_y)));
	const float len_x = 1.0f + (stretch - 1.0f) * len, len_y = 1.0f + -0.5f * len;
	const float lobe = 0.5f + float((1.0 / 4.0 - 0.04) - 0.5) * len;
	const float clip = fast_rcp32(lobe);

	float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f, acc_w = 0.0f;
	auto tap = [&](int t, float dx, float dy) {
		const float ox = dx - px, oy = dy - py;
		float vx = ox * dir_x + oy * dir_y;
		float vy = ox * (-dir_y) + oy * dir_x;
		vx *= len_x;
		vy *= len_y;
		const float d2 = fminf(vx * vx + vy * vy, clip);
		float base = float(2.0 / 5.0) * d2 + -1.0f;
		float window = lobe * d2 + -1.0f;
		base *= base;
		window *= window;
		base = float(25.0 / 16.0) * base + float(-(25.0 / 16.0 - 1.0));
		const float w = base * window;
		acc_r += r[t] * w;
		acc_g += g[t] * w;
		acc_b += b[t] * w;
		acc_w += w;
	};
	tap(0, 0.0f, -1.0f), tap(1, 1.0f, -1.0f), tap(6, -1.0f, 1.0f), tap(7, 0.0f, 1.0f), tap(3, 0.0f, 0.0f), tap(2, -1.0f, 0.0f);
	tap(8, 1.0f, 1.0f), tap(9, 2.0f, 1.0f), tap(5, 2.0f, 0.0f), tap(4, 1.0f, 0.0f), tap(11, 1.0f, 2.0f), tap(10, 0.0f, 2.0f);

	const float inv_w = 1.0f / acc_w;
	auto dering = [&](const float *c, float acc) {
		const float lo = fminf(fminf(c[3], fminf(c[4], c[7])), c[8]);
		const float hi = fmaxf(fmaxf(c[3], fmaxf(c[4], c[7])), c[8]);
		return fminf(hi, fmaxf(lo, acc * inv_w));
	};
	const uint32_t packed = to_unorm8(dering(r, acc_r)) | (to_unorm8(dering(g, acc_g)) << 8) | (to_unorm8(dering(b, acc_b)) << 16) | 0xff000000u;
	*reinterpret_cast<uint32_t *>(a.out + size_t(y) * a.out_pitch + size_t(x) * 4u) = packed;
}

// ---- EASU, packed fp16 ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FSR_BLOCK_X *FSR_BLOCK_Y) void k_fsr_easu_f16(EasuArgs a)
{
	const int x = blockIdx.x * FSR_BLOCK_X + threadIdx.x;
	const int y = blockIdx.y * FSR_BLOCK_Y + threadIdx.y;
	if (x >= a.ow || y >= a.oh)
		return;
	float fpx = float(x) * a.scale_x + a.offset_x;
	float fpy = float(y) * a.scale_y + a.offset_y;
	const float fx = floorf(fpx), fy = floorf(fpy);
	fpx -= fx;
	fpy -= fy;
	const _Float16 px = _Float16(fpx), py = _Float16(fpy);
	const _Float16 one = 1, half = _Float16(0.5f);

	uint32_t texel[12];
	fetch_taps(a.in, int(fx), int(fy), texel);
	_Float16 r[12], g[12], b[12], l[12];
#pragma unroll
	for (int t = 0; t < 12; t++)
	{
		r[t] = _Float16(chan(texel[t], 0)), g[t] = _Float16(chan(texel[t], 1)), b[t] = _Float16(chan(texel[t], 2));
		l[t] = b[t] * half + (r[t] * half + g[t]);
	}

	// Two of the four bilinear corners per packed operation.
	h2 dir_x = {0, 0}, dir_y = {0, 0}, len2 = {0, 0};
	auto analyse = [&](_Float16 row_weight, h2 up, h2 left, h2 centre, h2 right, h2 down) {
		const h2 w = (h2{1, 0} + h2{-px, px}) * h2{row_weight, row_weight};
		const h2 dc = right - centre, cb = centre - left;
		h2 lx = hrcp2(hmax2(habs2(dc), habs2(cb)));
		const h2 gx = right - left;
		dir_x = dir_x + gx * w;
		lx = hsat2(habs2(gx) * lx);
		lx = lx * lx;
		len2 = len2 + lx * w;
		const h2 ec = down - centre, ca = centre - up;
		h2 ly = hrcp2(hmax2(habs2(ec), habs2(ca)));
		const h2 gy = down - up;
		dir_y = dir_y + gy * w;
		ly = hsat2(habs2(gy) * ly);
		ly = ly * ly;
		len2 = len2 + ly * w;
	};
	analyse(one - py, h2{l[0], l[1]}, h2{l[2], l[3]}, h2{l[3], l[4]}, h2{l[4], l[5]}, h2{l[7], l[8]});
	analyse(py, h2{l[3], l[4]}, h2{l[6], l[7]}, h2{l[7], l[8]}, h2{l[8], l[9]}, h2{l[10], l[11]});
	_Float16 dx = dir_x.x + dir_x.y, dy = dir_y.x + dir_y.y;
	_Float16 len = len2.x + len2.y;

	_Float16 norm = dx * dx + dy * dy;
	const bool flat = float(norm) < float(1.0 / 32768.0);
	norm = fast_rsq16(norm);
	if (flat)
	{
		norm = one;
		dx = one;
	}
	dx = dx * norm;
	dy = dy * norm;
	len = len * half;
	len = len * len;
	const _Float16 stretch = (dx * dx + dy * dy) * fast_rcp16(hmax1(__builtin_fabsf16(dx), __builtin_fabsf16(dy)));
	const _Float16 len_x = one + (stretch - one) * len, len_y = one + _Float16(-0.5f) * len;
	const _Float16 lobe = half + _Float16(float((1.0 / 4.0 - 0.04) - 0.5)) * len;
	const _Float16 clip = fast_rcp16(lobe);

	h2 acc_r = {0, 0}, acc_g = {0, 0}, acc_b = {0, 0}, acc_w = {0, 0};
	const h2 k_base = {_Float16(float(2.0 / 5.0)), _Float16(float(2.0 / 5.0))};
	const h2 k_a = {_Float16(1.5625f), _Float16(1.5625f)}, k_b = {_Float16(-0.5625f), _Float16(-0.5625f)}, minus_one = {-1, -1};
	const h2 dxx = {dx, dx}, dyy = {dy, dy}, lobe2 = {lobe, lobe}, clip2 = {clip, clip};
	auto taps = [&](int t0, int t1, h2 offx, h2 offy) {
		const h2 ox = offx - h2{px, px}, oy = offy - h2{py, py};
		h2 vx = ox * dxx + oy * dyy;
		h2 vy = ox * (-dyy) + oy * dxx;
		vx = vx * h2{len_x, len_x};
		vy = vy * h2{len_y, len_y};
		const h2 d2 = hmin2(vx * vx + vy * vy, clip2);
		h2 base = k_base * d2 + minus_one;
		h2 window = lobe2 * d2 + minus_one;
		base = base * base;
		window = window * window;
		base = k_a * base + k_b;
		const h2 w = base * window;
		acc_r = acc_r + h2{r[t0], r[t1]} * w;
		acc_g = acc_g + h2{g[t0], g[t1]} * w;
		acc_b = acc_b + h2{b[t0], b[t1]} * w;
		acc_w = acc_w + w;
	};
	taps(0, 1, h2{0, 1}, h2{-1, -1});  // b c
	taps(6, 7, h2{-1, 0}, h2{1, 1});   // i j
	taps(3, 2, h2{0, -1}, h2{0, 0});   // f e
	taps(8, 9, h2{1, 2}, h2{1, 1});    // k l
	taps(5, 4, h2{2, 1}, h2{0, 0});    // h g
	taps(11, 10, h2{1, 0}, h2{2, 2});  // o n

	const _Float16 inv_w = hdiv(one, acc_w.x + acc_w.y);
	auto dering = [&](const _Float16 *c, h2 acc) {
		const _Float16 lo = hmin1(hmin1(c[3], c[4]), hmin1(c[7], c[8]));
		const _Float16 hi = hmax1(hmax1(c[3], c[4]), hmax1(c[7], c[8]));
		return float(hmin1(hi, hmax1(lo, (acc.x + acc.y) * inv_w)));
	};
	const uint32_t packed = to_unorm8(dering(r, acc_r)) | (to_unorm8(dering(g, acc_g)) << 8) | (to_unorm8(dering(b, acc_b)) << 16) | 0xff000000u;
	*reinterpret_cast<uint32_t *>(a.out + size_t(y) * a.out_pitch + size_t(x) * 4u) = packed;
}

// ---- RCAS (fp32) -----------------------------------------------------------------------------------------------------------
template <bool SRGB>
__global__ __launch_bounds__(FSR_BLOCK_X *FSR_BLOCK_Y) void k_fsr_rcas(Rgba8 in, uint8_t *out, uint32_t out_pitch, float sharpness, const float *srgb_lut)
{
	// The sRGB view's decode table sits in LDS: 15 dependent lookups per pixel.
	__shared__ float lut[256];
	if (SRGB)
	{
		lut[threadIdx.y * FSR_BLOCK_X + threadIdx.x] = srgb_lut[threadIdx.y * FSR_BLOCK_X + threadIdx.x];
		__syncthreads();
	}
	const int x = blockIdx.x * FSR_BLOCK_X + threadIdx.x;
	const int y = blockIdx.y * FSR_BLOCK_Y + threadIdx.y;
	if (x >= in.w || y >= in.h)
		return;
	const uint32_t texel[5] = {in.at(x, y - 1), in.at(x - 1, y), in.at(x, y), in.at(x + 1, y), in.at(x, y + 1)}; // up left centre right down
	float rgb[3];
	float tap[5][3];
#pragma unroll
	for (int t = 0; t < 5; t++)
#pragma unroll
		for (int c = 0; c < 3; c++)
		{
			const uint32_t byte = (texel[t] >> (8 * c)) & 255u;
			tap[t][c] = SRGB ? lut[byte] : unorm8_to_float(byte);
		}
	float lobe_c[3];
#pragma unroll
	for (int c = 0; c < 3; c++)
	{
		const float ring_min = fminf(fminf(tap[0][c], fminf(tap[1][c], tap[3][c])), tap[4][c]);
		const float ring_max = fmaxf(fmaxf(tap[0][c], fmaxf(tap[1][c], tap[3][c])), tap[4][c]);
		const float hit_min = ring_min * (1.0f / (4.0f * ring_max));
		const float hit_max = (1.0f - ring_max) * (1.0f / (4.0f * ring_min + -4.0f));
		lobe_c[c] = fmaxf(-hit_min, hit_max);
	}
	const float limit = float(0.25 - (1.0 / 16.0));
	const float widest = fmaxf(fmaxf(lobe_c[0], lobe_c[1]), lobe_c[2]);
	const float lobe = fmaxf(-limit, fminf(widest, 0.0f)) * sharpness;
	const float inv = medium_rcp32(4.0f * lobe + 1.0f);
#pragma unroll
	for (int c = 0; c < 3; c++)
		rgb[c] = (lobe * tap[0][c] + lobe * tap[1][c] + lobe * tap[4][c] + lobe * tap[3][c] + tap[2][c]) * inv;
	uint32_t packed = 0xff000000u;
#pragma unroll
	for (int c = 0; c < 3; c++)
		packed |= (SRGB ? encode_srgb8(rgb[c]) : to_unorm8(rgb[c])) << (8 * c);
	*reinterpret_cast<uint32_t *>(out + size_t(y) * out_pitch + size_t(x) * 4u) = packed;
}

bool rgba8(const gr_image *img)
{
	return img && img->ptr && img->width && img->height && img->pitch_bytes >= img->width * 4u &&
	       (img->format == GR_FORMAT_R8G8B8A8_UNORM || img->format == GR_FORMAT_R8G8B8A8_SRGB);
}
} // namespace

extern "C" {

int gr_fsr_upscale(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, int fp16)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, rgba8(in) && rgba8(out) && in->ptr != out->ptr);
	EasuArgs a;
	a.in = {static_cast<const uint8_t *>(in->ptr), int(in->width), int(in->height), in->pitch_bytes};
	a.out = static_cast<uint8_t *>(out->ptr);
	a.out_pitch = out->pitch_bytes;
	a.ow = int(out->width);
	a.oh = int(out->height);
	// FsrEasuCon with viewport = input size (aa.cpp:106-108)
	a.scale_x = float(in->width) * (1.0f / float(out->width));
	a.scale_y = float(in->height) * (1.0f / float(out->height));
	a.offset_x = 0.5f * float(in->width) * (1.0f / float(out->width)) - 0.5f;
	a.offset_y = 0.5f * float(in->height) * (1.0f / float(out->height)) - 0.5f;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "fsr_upscale"};
	const dim3 grid(gr_div_up(out->width, FSR_BLOCK_X), gr_div_up(out->height, FSR_BLOCK_Y)), block(FSR_BLOCK_X, FSR_BLOCK_Y);
	if (fp16)
		hipLaunchKernelGGL(k_fsr_easu_f16, grid, block, 0, gr_to_stream(stream), a);
	else
		hipLaunchKernelGGL(k_fsr_easu_f32, grid, block, 0, gr_to_stream(stream), a);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_fsr_sharpen(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, float sharpness)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, rgba8(in) && rgba8(out) && in->ptr != out->ptr && in->width == out->width && in->height == out->height);
	const Rgba8 src = {static_cast<const uint8_t *>(in->ptr), int(in->width), int(in->height), in->pitch_bytes};
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "fsr_sharpen"};
	const dim3 grid(gr_div_up(out->width, FSR_BLOCK_X), gr_div_up(out->height, FSR_BLOCK_Y)), block(FSR_BLOCK_X, FSR_BLOCK_Y);
	if (out->format == GR_FORMAT_R8G8B8A8_SRGB)
		hipLaunchKernelGGL(k_fsr_rcas<true>, grid, block, 0, gr_to_stream(stream), src, static_cast<uint8_t *>(out->ptr), out->pitch_bytes, sharpness,
		                   ctx->srgb_decode_lut);
	else
		hipLaunchKernelGGL(k_fsr_rcas<false>, grid, block, 0, gr_to_stream(stream), src, static_cast<uint8_t *>(out->ptr), out->pitch_bytes, sharpness,
		                   ctx->srgb_decode_lut);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

} // extern "C"
