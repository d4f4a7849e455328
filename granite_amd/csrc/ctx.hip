// Context, memory and timing entry points of the C ABI (include/granite_hip.h).
#include "ctx.hpp"
#include "device_common.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace
{
struct MixArgs
{
	uint32_t *out;
	uint32_t out_dwords;
	const uint32_t *in[4];
	uint32_t in_dwords[4];
	uint32_t count, salt;
};

// Executor self-test operation: every output dword is a hash of its index, a salt and one dword of each input, so a pass that
// ran before its producers, on a recycled allocation that is still in use, or on the wrong copy of a hand-over ring leaves a
// different image behind (tests/cpp/graph_cases.cpp --execute).
__global__ __launch_bounds__(256) void k_debug_mix(MixArgs a)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= a.out_dwords)
		return;
	uint32_t h = (i * 0x9E3779B1u) ^ a.salt;
	for (uint32_t k = 0; k < a.count; k++)
	{
		const uint32_t v = a.in[k][(i + k * 977u) % a.in_dwords[k]];
		h = (h ^ v) * 0x85EBCA6Bu;
		h ^= h >> 13;
	}
	a.out[i] = h;
}
} // namespace

// ---- 24-bit transport form of an RGBA8 target whose alpha is 255 everywhere ----------------------------------------------------
// Row-band tiling sends every band of the finished frame to every other device (collective B); over point-to-point
// xGMI that all-gather, not the kernels, bounds the multi-GPU frame rate, and the alpha byte of a tonemapped frame is a
// constant.  Four pixels per lane: 16 B in / 12 B out (pack) and back (unpack, alpha = 255).
__global__ __launch_bounds__(256) void k_pack_rgb8(const uint8_t *image, uint32_t pitch, uint32_t width, uint32_t row_first, uint32_t rows, uint8_t *packed)
{
	const uint32_t groups = (width + 3u) / 4u;
	const uint32_t g = blockIdx.x * 256u + threadIdx.x;
	const uint32_t y = row_first + blockIdx.y;
	if (g >= groups || blockIdx.y >= rows)
		return;
	const uint8_t *src = image + size_t(y) * pitch + size_t(g) * 16u;
	uint8_t *dst = packed + (size_t(y) * width + size_t(g) * 4u) * 3u;
	if (g * 4u + 4u <= width && ((size_t(y) * width * 3u) & 3u) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0)
	{
		const uint4 p = *reinterpret_cast<const uint4 *>(src);
		const uint32_t a = p.x & 0xffffffu, b = p.y & 0xffffffu, c = p.z & 0xffffffu, d = p.w & 0xffffffu;
		uint32_t *out = reinterpret_cast<uint32_t *>(dst);
		out[0] = a | (b << 24);
		out[1] = (b >> 8) | (c << 16);
		out[2] = (c >> 16) | (d << 8);
	}
	else
		for (uint32_t i = 0; i < 4u && g * 4u + i < width; i++)
			for (uint32_t ch = 0; ch < 3u; ch++)
				dst[i * 3u + ch] = src[i * 4u + ch];
}

__global__ __launch_bounds__(256) void k_unpack_rgb8(const uint8_t *packed, uint8_t *image, uint32_t pitch, uint32_t width, uint32_t row_first, uint32_t rows)
{
	const uint32_t groups = (width + 3u) / 4u;
	const uint32_t g = blockIdx.x * 256u + threadIdx.x;
	const uint32_t y = row_first + blockIdx.y;
	if (g >= groups || blockIdx.y >= rows)
		return;
	const uint8_t *src = packed + (size_t(y) * width + size_t(g) * 4u) * 3u;
	uint8_t *dst = image + size_t(y) * pitch + size_t(g) * 16u;
	if (g * 4u + 4u <= width && ((size_t(y) * width * 3u) & 3u) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0)
	{
		const uint32_t *in = reinterpret_cast<const uint32_t *>(src);
		const uint32_t w0 = in[0], w1 = in[1], w2 = in[2];
		uint4 p;
		p.x = (w0 & 0xffffffu) | 0xff000000u;
		p.y = (w0 >> 24) | ((w1 & 0xffffu) << 8) | 0xff000000u;
		p.z = (w1 >> 16) | ((w2 & 0xffu) << 16) | 0xff000000u;
		p.w = (w2 >> 8) | 0xff000000u;
		*reinterpret_cast<uint4 *>(dst) = p;
	}
	else
		for (uint32_t i = 0; i < 4u && g * 4u + i < width; i++)
		{
			for (uint32_t ch = 0; ch < 3u; ch++)
				dst[i * 4u + ch] = src[i * 3u + ch];
			dst[i * 4u + 3u] = 255;
		}
}

extern "C" {

int gr_abi_version(void)
{
	return GR_ABI_VERSION;
}

// Linear -> sRGB8 as an *_SRGB attachment store does it (assets/shaders/inc/srgb.h:12-18 + UNORM8 rounding), in fp32.
static uint32_t srgb8_encode_host(float c)
{
	if (!(c > 0.0f))
		return 0;
	c = fminf(c, 1.0f);
	float r = (c <= 0.0031308f) ? (c * 12.92f) : (1.055f * powf(c, 1.0f / 2.4f) - 0.055f);
	r = fminf(fmaxf(r, 0.0f), 1.0f);
	return r >= 1.0f ? 255u : uint32_t(int(r * 255.0f + 0.5f));
}

// tonemap.frag:42-53 for one channel, evaluated in fp32 operation by operation (no contraction), into the sRGB8 byte.
static uint32_t tonemap_srgb8_host(float x)
{
#pragma clang fp contract(off)
	const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f, W = 11.2f;
	auto uncharted2 = [=](float v) { return ((v * (A * v + C * B) + D * E) / (v * (A * v + B) + D * F)) - E / F; };
	const float white_scale = 1.0f / uncharted2(W);
	return srgb8_encode_host(uncharted2(x) * white_scale);
}

// Staircase tables share one builder: entry i = {bits of the smallest float of bucket i that encodes above the bucket's
// first float (or +inf), byte of the bucket's first float}.  -1 if a bucket holds more than one step.
static int build_staircase(uint32_t (*enc)(float), uint32_t min_bits, uint32_t shift, uint32_t entries, uint32_t *entries_xy)
{
	auto at = [enc](uint32_t bits) { return enc(__builtin_bit_cast(float, bits)); };
	for (uint32_t i = 0; i < entries; i++)
	{
		const uint32_t lo = min_bits + (i << shift);
		const uint32_t hi = lo + (1u << shift) - 1u; // last float of the bucket
		const uint32_t value = at(lo);
		uint32_t threshold = 0x7f800000u; // +inf
		if (i + 1 < entries && at(hi) != value)
		{
			if (at(hi) != value + 1)
				return -1;
			uint32_t a = lo, b = hi; // at(a) == value, at(b) == value + 1
			while (b - a > 1)
			{
				const uint32_t mid = a + (b - a) / 2;
				if (at(mid) == value)
					a = mid;
				else
					b = mid;
			}
			threshold = b;
		}
		entries_xy[2 * i] = threshold;
		entries_xy[2 * i + 1] = value;
	}
	return 0;
}

extern "C" int gr_tonemap_srgb8_table(uint32_t *entries_xy, uint32_t count)
{
	if (!entries_xy || count != TONEMAP_TABLE_ENTRIES)
		return -1;
	return build_staircase(tonemap_srgb8_host, TONEMAP_TABLE_MIN_BITS, TONEMAP_TABLE_BUCKET_SHIFT, TONEMAP_TABLE_ENTRIES, entries_xy);
}

// One entry per bucket of SRGB_ENCODE_BUCKET_SHIFT mantissa bits: {bits of the smallest float of the bucket that encodes
// to value + 1 (or +inf), value at the bucket's lower bound}.  Returns 0, or -1 if a bucket holds more than one step.
extern "C" int gr_srgb_encode_table(uint32_t *entries_xy, uint32_t count)
{
	if (!entries_xy || count != SRGB_ENCODE_ENTRIES)
		return -1;
	return build_staircase(srgb8_encode_host, SRGB_ENCODE_MIN_BITS, SRGB_ENCODE_BUCKET_SHIFT, SRGB_ENCODE_ENTRIES, entries_xy);
}

gr_ctx *gr_create(int device)
{
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
		return nullptr;
	if (hipSetDevice(device) != hipSuccess)
		return nullptr;

	auto *ctx = new gr_ctx;
	ctx->device = device;

	// sRGB8 -> linear decode table; formula of assets/shaders/inc/srgb.h:4-10 (the *_SRGB sampler view).
	float lut[256];
	for (int i = 0; i < 256; i++)
	{
		float c = float(i) / 255.0f;
		float r = (c <= 0.0404482362771082f) ? (c / 12.92f) : powf((c + 0.055f) / 1.055f, 2.4f);
		lut[i] = fminf(fmaxf(r, 0.0f), 1.0f);
	}
	static uint32_t encode_table[2 * SRGB_ENCODE_ENTRIES];
	static const int encode_table_status = gr_srgb_encode_table(encode_table, SRGB_ENCODE_ENTRIES);
	static uint32_t tonemap_table[2 * TONEMAP_TABLE_ENTRIES];
	static const int tonemap_table_status = gr_tonemap_srgb8_table(tonemap_table, TONEMAP_TABLE_ENTRIES);
	// cos / sin of phi = 2 * M_PI_SIC * (byte / 255), the expression of sssr_util.h's SampleGGXVNDF for its second random number
	float2 azimuth[256];
	for (volatile int i = 0; i < 256; i++)
	{
		const float u2 = float(int(i)) / 255.0f;
		volatile float phi = 2.0f * 3.1415628f * u2;
		azimuth[i] = make_float2(cosf(phi), sinf(phi));
	}
	if (encode_table_status != 0 || tonemap_table_status != 0 ||
	    hipMalloc(reinterpret_cast<void **>(&ctx->ssr_azimuth_lut), sizeof(azimuth)) != hipSuccess ||
	    hipMemcpy(ctx->ssr_azimuth_lut, azimuth, sizeof(azimuth), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMalloc(reinterpret_cast<void **>(&ctx->tonemap_srgb8_lut), sizeof(tonemap_table)) != hipSuccess ||
	    hipMemcpy(ctx->tonemap_srgb8_lut, tonemap_table, sizeof(tonemap_table), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMalloc(reinterpret_cast<void **>(&ctx->pyramid_sync), gr_ctx::PYRAMID_SYNC_SLOTS * 4 * sizeof(uint32_t)) != hipSuccess ||
	    hipMemset(ctx->pyramid_sync, 0, gr_ctx::PYRAMID_SYNC_SLOTS * 4 * sizeof(uint32_t)) != hipSuccess ||
	    hipMalloc(reinterpret_cast<void **>(&ctx->srgb_decode_lut), sizeof(lut)) != hipSuccess ||
	    hipMemcpy(ctx->srgb_decode_lut, lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMalloc(reinterpret_cast<void **>(&ctx->srgb_encode_lut), sizeof(encode_table)) != hipSuccess ||
	    hipMemcpy(ctx->srgb_encode_lut, encode_table, sizeof(encode_table), hipMemcpyHostToDevice) != hipSuccess)
	{
		delete ctx;
		return nullptr;
	}
	return ctx;
}

void gr_destroy(gr_ctx *ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	(void)hipDeviceSynchronize();
	for (auto &s : ctx->spans)
	{
		(void)hipEventDestroy(s.start);
		(void)hipEventDestroy(s.stop);
	}
	for (auto &e : ctx->event_pool)
		(void)hipEventDestroy(e);
	if (ctx->pyramid_sync)
		(void)hipFree(ctx->pyramid_sync);
	if (ctx->srgb_decode_lut)
		(void)hipFree(ctx->srgb_decode_lut);
	if (ctx->srgb_encode_lut)
		(void)hipFree(ctx->srgb_encode_lut);
	if (ctx->tonemap_srgb8_lut)
		(void)hipFree(ctx->tonemap_srgb8_lut);
	if (ctx->ssr_azimuth_lut)
		(void)hipFree(ctx->ssr_azimuth_lut);
	if (ctx->smaa_area)
		(void)hipFree(ctx->smaa_area);
	if (ctx->smaa_search)
		(void)hipFree(ctx->smaa_search);
	for (auto &bits : ctx->smaa_bits)
		if (bits.second.memory)
			(void)hipFree(bits.second.memory);
	delete ctx;
}

const char *gr_last_error(gr_ctx *ctx)
{
	return ctx ? ctx->last_error.c_str() : "null context";
}

int gr_sync(gr_ctx *ctx, gr_stream stream)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_HIP(ctx, hipStreamSynchronize(gr_to_stream(stream)));
	return GR_OK;
}

int gr_alloc(gr_ctx *ctx, size_t bytes, void **dptr)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, dptr != nullptr && bytes != 0);
	hipError_t err = hipMalloc(dptr, bytes);
	if (err != hipSuccess)
		return ctx->fail(GR_ERR_OUT_OF_MEMORY, "gr_alloc: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
	// hipMemset on device memory is asynchronous with respect to the host and runs on the null stream, which the executor's
	// non-blocking streams do not synchronise with: wait for it, or the zeros can land on top of a kernel's first writes.
	GR_CHECK_HIP(ctx, hipMemset(*dptr, 0, bytes));
	GR_CHECK_HIP(ctx, hipStreamSynchronize(nullptr));
	return GR_OK;
}

int gr_free(gr_ctx *ctx, void *dptr)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	if (dptr)
	{
		GR_CHECK_HIP(ctx, hipFree(dptr));
	}
	return GR_OK;
}

int gr_upload(gr_ctx *ctx, gr_stream stream, void *dst, const void *src_host, size_t bytes)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, dst && src_host);
	GR_CHECK_HIP(ctx, hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, gr_to_stream(stream)));
	return GR_OK;
}

namespace
{
struct UploadBatch
{
	uint32_t *dst[GR_MAX_UPLOAD_RANGES];
	const uint32_t *src[GR_MAX_UPLOAD_RANGES];
	uint32_t dwords[GR_MAX_UPLOAD_RANGES];
	uint32_t count;
};

// blockIdx.y selects the range; 16-byte accesses where both pointers allow it.
__global__ __launch_bounds__(256) void k_upload_batch(UploadBatch b)
{
	const uint32_t r = blockIdx.y;
	if (r >= b.count)
		return;
	uint32_t *dst = b.dst[r];
	const uint32_t *src = b.src[r];
	const uint32_t n = b.dwords[r];
	const uint32_t stride = gridDim.x * blockDim.x;
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0)
	{
		const uint32_t n4 = n / 4;
		for (uint32_t i = tid; i < n4; i += stride)
			reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
		for (uint32_t i = n4 * 4 + tid; i < n; i += stride)
			dst[i] = src[i];
	}
	else
		for (uint32_t i = tid; i < n; i += stride)
			dst[i] = src[i];
}
} // namespace

int gr_upload_batch(gr_ctx *ctx, gr_stream stream, const gr_upload_range *ranges, uint32_t count)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, ranges != nullptr || count == 0);
	GR_CHECK_ARG(ctx, count <= GR_MAX_UPLOAD_RANGES);
	UploadBatch b = {};
	uint32_t max_dwords = 0;
	for (uint32_t i = 0; i < count; i++)
	{
		if (ranges[i].bytes == 0)
			continue;
		GR_CHECK_ARG(ctx, ranges[i].dst && ranges[i].src_pinned && (ranges[i].bytes & 3u) == 0 && ranges[i].bytes < (size_t(1) << 33));
		GR_CHECK_ARG(ctx, (reinterpret_cast<uintptr_t>(ranges[i].dst) & 3u) == 0 && (reinterpret_cast<uintptr_t>(ranges[i].src_pinned) & 3u) == 0);
		void *mapped = nullptr;
		GR_CHECK_HIP(ctx, hipHostGetDevicePointer(&mapped, const_cast<void *>(ranges[i].src_pinned), 0));
		b.dst[b.count] = static_cast<uint32_t *>(ranges[i].dst);
		b.src[b.count] = static_cast<const uint32_t *>(mapped);
		b.dwords[b.count] = uint32_t(ranges[i].bytes / 4);
		max_dwords = max_dwords > b.dwords[b.count] ? max_dwords : b.dwords[b.count];
		b.count++;
	}
	if (b.count == 0)
		return GR_OK;
	const unsigned blocks_x = gr_div_up(gr_div_up(max_dwords, 4u), 256u) < 64u ? gr_div_up(gr_div_up(max_dwords, 4u), 256u) : 64u;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "upload_batch"};
	hipLaunchKernelGGL(k_upload_batch, dim3(blocks_x ? blocks_x : 1, b.count), dim3(256), 0, gr_to_stream(stream), b);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_alloc_host(gr_ctx *ctx, size_t bytes, void **hptr)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, hptr != nullptr && bytes != 0);
	hipError_t err = hipHostMalloc(hptr, bytes, hipHostMallocDefault);
	if (err != hipSuccess)
		return ctx->fail(GR_ERR_OUT_OF_MEMORY, "gr_alloc_host: hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
	return GR_OK;
}

int gr_free_host(gr_ctx *ctx, void *hptr)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	if (hptr)
		GR_CHECK_HIP(ctx, hipHostFree(hptr));
	return GR_OK;
}

int gr_download(gr_ctx *ctx, gr_stream stream, void *dst_host, const void *src, size_t bytes)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, dst_host && src);
	GR_CHECK_HIP(ctx, hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, gr_to_stream(stream)));
	GR_CHECK_HIP(ctx, hipStreamSynchronize(gr_to_stream(stream)));
	return GR_OK;
}

int gr_copy(gr_ctx *ctx, gr_stream stream, void *dst, const void *src, size_t bytes)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, dst && src);
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "copy"};
	GR_CHECK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, gr_to_stream(stream)));
	return GR_OK;
}

int gr_fill_zero(gr_ctx *ctx, gr_stream stream, void *dst, size_t bytes)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, dst);
	GR_CHECK_HIP(ctx, hipMemsetAsync(dst, 0, bytes, gr_to_stream(stream)));
	return GR_OK;
}

int gr_fill_byte(gr_ctx *ctx, gr_stream stream, void *dst, int value, size_t bytes)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, dst);
	GR_CHECK_HIP(ctx, hipMemsetAsync(dst, value & 0xff, bytes, gr_to_stream(stream)));
	return GR_OK;
}

static bool rgba8_target(const gr_image *image)
{
	return image && image->ptr && image->width && image->height && image->pitch_bytes >= image->width * 4u &&
	       (image->format == GR_FORMAT_R8G8B8A8_SRGB || image->format == GR_FORMAT_R8G8B8A8_UNORM);
}

int gr_pack_rgb8_rows(gr_ctx *ctx, gr_stream stream, const gr_image *image, const gr_rows *rows, void *packed)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, rgba8_target(image) && packed && (reinterpret_cast<uintptr_t>(packed) & 3u) == 0);
	const RowSpan span = resolve_rows(rows, image->height);
	if (span.count() == 0)
		return GR_OK;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "pack_rgb8"};
	hipLaunchKernelGGL(k_pack_rgb8, dim3(gr_div_up(gr_div_up(image->width, 4u), 256u), span.count()), dim3(256), 0, gr_to_stream(stream),
	                   static_cast<const uint8_t *>(image->ptr), image->pitch_bytes, image->width, span.first, span.count(), static_cast<uint8_t *>(packed));
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_unpack_rgb8_rows(gr_ctx *ctx, gr_stream stream, const void *packed, const gr_image *image, const gr_rows *rows)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, rgba8_target(image) && packed && (reinterpret_cast<uintptr_t>(packed) & 3u) == 0);
	const RowSpan span = resolve_rows(rows, image->height);
	if (span.count() == 0)
		return GR_OK;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "unpack_rgb8"};
	hipLaunchKernelGGL(k_unpack_rgb8, dim3(gr_div_up(gr_div_up(image->width, 4u), 256u), span.count()), dim3(256), 0, gr_to_stream(stream),
	                   static_cast<const uint8_t *>(packed), static_cast<uint8_t *>(image->ptr), image->pitch_bytes, image->width, span.first, span.count());
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_fill_u32(gr_ctx *ctx, gr_stream stream, void *dst, uint32_t value, size_t count)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, dst && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0);
	GR_CHECK_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dst), int(value), count, gr_to_stream(stream)));
	return GR_OK;
}

int gr_debug_mix(gr_ctx *ctx, gr_stream stream, void *out, size_t out_dwords, const void *const *inputs, const size_t *input_dwords,
                 uint32_t input_count, uint32_t salt)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, out && out_dwords > 0 && out_dwords <= 0xffffffffull && input_count <= 4 && (input_count == 0 || (inputs && input_dwords)));
	MixArgs a{};
	a.out = static_cast<uint32_t *>(out);
	a.out_dwords = uint32_t(out_dwords);
	a.count = input_count;
	a.salt = salt;
	for (uint32_t k = 0; k < input_count; k++)
	{
		GR_CHECK_ARG(ctx, inputs[k] && input_dwords[k] > 0 && input_dwords[k] <= 0xffffffffull);
		a.in[k] = static_cast<const uint32_t *>(inputs[k]);
		a.in_dwords[k] = uint32_t(input_dwords[k]);
	}
	hipLaunchKernelGGL(k_debug_mix, dim3(unsigned((out_dwords + 255) / 256)), dim3(256), 0, gr_to_stream(stream), a);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

// The attachment store conversion of B10G11R11_UFLOAT_PACK32 on its own (what the lighting and TAA kernels apply to their
// results), so that it can be held to the oracle's statement word for word.
__global__ void k_pack_b10g11r11(const float *rgb, uint32_t *out, uint32_t count)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < count)
		out[i] = pack_b10g11r11(rgb[3u * i], rgb[3u * i + 1u], rgb[3u * i + 2u]);
}

int gr_pack_b10g11r11(gr_ctx *ctx, gr_stream stream, const float *rgb, uint32_t *out, uint32_t texels)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, rgb && out);
	if (texels == 0)
		return GR_OK;
	hipLaunchKernelGGL(k_pack_b10g11r11, dim3((texels + 255u) / 256u), dim3(256), 0, gr_to_stream(stream), rgb, out, texels);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}

int gr_get_device_info(gr_ctx *ctx, char *name, size_t name_capacity, uint32_t *driver_version)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, name && name_capacity > 0);
	hipDeviceProp_t props;
	GR_CHECK_HIP(ctx, hipGetDeviceProperties(&props, ctx->device));
	// Containers without the amdgpu ids table report an empty marketing name; the ISA name is always there.
	snprintf(name, name_capacity, "%s", props.name[0] ? props.name : props.gcnArchName);
	if (driver_version)
	{
		int version = 0;
		GR_CHECK_HIP(ctx, hipDriverGetVersion(&version));
		*driver_version = uint32_t(version);
	}
	return GR_OK;
}

namespace
{
// One workgroup per 16 KiB: four independent 16-byte accesses per lane, 256 lanes side by side (every access of a wave is
// one contiguous 1 KiB), a grid of n / 1024 workgroups.
__global__ __launch_bounds__(256) void k_probe_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
	const size_t i = size_t(blockIdx.x) * 1024u + threadIdx.x;
	if (i + 768u < n)
	{
		const float4 v0 = a[i], v1 = a[i + 256u], v2 = a[i + 512u], v3 = a[i + 768u];
		b[i] = v0;
		b[i + 256u] = v1;
		b[i + 512u] = v2;
		b[i + 768u] = v3;
	}
	else
		for (size_t j = i; j < n; j += 256u)
			b[j] = a[j];
}
__device__ __forceinline__ float4 triad4(float4 x, float4 y, float s)
{
	return make_float4(fmaf(s, y.x, x.x), fmaf(s, y.y, x.y), fmaf(s, y.z, x.z), fmaf(s, y.w, x.w));
}
__global__ __launch_bounds__(256) void k_probe_triad(float4 *__restrict__ a, const float4 *__restrict__ b, const float4 *__restrict__ c,
                                                     float s, size_t n)
{
	const size_t i = size_t(blockIdx.x) * 512u + threadIdx.x;
	if (i + 256u < n)
	{
		const float4 x0 = b[i], y0 = c[i], x1 = b[i + 256u], y1 = c[i + 256u];
		a[i] = triad4(x0, y0, s);
		a[i + 256u] = triad4(x1, y1, s);
	}
	else
		for (size_t j = i; j < n; j += 256u)
			a[j] = triad4(b[j], c[j], s);
}
} // namespace

int gr_bandwidth_probe(gr_ctx *ctx, size_t bytes, int repeats, double *copy_GBps, double *triad_GBps)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, bytes >= (size_t(1) << 20) && repeats > 0 && copy_GBps && triad_GBps);
	const size_t n = bytes / sizeof(float4);
	float4 *buf[3] = {};
	hipEvent_t e0 = nullptr, e1 = nullptr;
	int status = GR_OK;
	auto cleanup = [&]() {
		for (auto *b : buf)
			if (b)
				(void)hipFree(b);
		if (e0)
			(void)hipEventDestroy(e0);
		if (e1)
			(void)hipEventDestroy(e1);
	};
	for (auto &b : buf)
		if (hipMalloc(reinterpret_cast<void **>(&b), n * sizeof(float4)) != hipSuccess || hipMemset(b, 0, n * sizeof(float4)) != hipSuccess)
		{
			cleanup();
			return ctx->fail(GR_ERR_OUT_OF_MEMORY, "gr_bandwidth_probe: 3 x %zu bytes not available", bytes);
		}
	if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
	{
		cleanup();
		return ctx->fail(GR_ERR_HIP, "gr_bandwidth_probe: hipEventCreate failed");
	}
	const dim3 block(256), grid_copy(unsigned((n + 1023u) / 1024u)), grid_triad(unsigned((n + 511u) / 512u));
	double best[2] = {1e30, 1e30};
	for (int r = 0; r < repeats + 1 && status == GR_OK; r++) // first round warms up
		for (int which = 0; which < 2; which++)
		{
			(void)hipEventRecord(e0, nullptr);
			if (which == 0)
				hipLaunchKernelGGL(k_probe_copy, grid_copy, block, 0, nullptr, buf[0], buf[1], n);
			else
				hipLaunchKernelGGL(k_probe_triad, grid_triad, block, 0, nullptr, buf[0], buf[1], buf[2], 0.5f, n);
			(void)hipEventRecord(e1, nullptr);
			float ms = 0.0f;
			if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess)
			{
				status = ctx->fail(GR_ERR_HIP, "gr_bandwidth_probe: launch failed");
				break;
			}
			if (r > 0 && ms < best[which])
				best[which] = ms;
		}
	cleanup();
	if (status != GR_OK)
		return status;
	*copy_GBps = 2.0 * double(n * sizeof(float4)) / (best[0] * 1e-3) / 1e9;
	*triad_GBps = 3.0 * double(n * sizeof(float4)) / (best[1] * 1e-3) / 1e9;
	return GR_OK;
}

int gr_timing_enable(gr_ctx *ctx, int enable)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	ctx->timing_enabled = enable != 0;
	return GR_OK;
}

int gr_timing_set_filter(gr_ctx *ctx, const char *name)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	ctx->timing_filter = name ? name : "";
	return GR_OK;
}

int gr_timing_brackets(gr_ctx *ctx, const char *name)
{
	if (!ctx || !name)
		return 0;
	return ctx->timing_enabled && ctx->timing_matches(name) ? 1 : 0;
}

int gr_timing_set_sampling(gr_ctx *ctx, uint32_t every_nth)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	ctx->timing_every = every_nth ? every_nth : 1;
	ctx->timing_seen = 0;
	return GR_OK;
}

static int drain_spans(gr_ctx *ctx)
{
	// GR_TIMING_DUMP=<file>: also append every span as "name start_us stop_us" relative to the first span drained in this
	// process -- a GPU-side timeline across streams that, unlike a profiler's API interception, does not slow the host.
	static FILE *dump = getenv("GR_TIMING_DUMP") ? fopen(getenv("GR_TIMING_DUMP"), "w") : nullptr;
	static hipEvent_t origin = nullptr;
	for (auto &s : ctx->spans)
	{
		if (dump)
		{
			(void)hipEventSynchronize(s.stop);
			if (!origin)
			{
				origin = s.start; // first span's start event is kept (never recycled) as the time origin
			}
			float a = 0.0f, b = 0.0f;
			if (hipEventElapsedTime(&a, origin, s.start) == hipSuccess && hipEventElapsedTime(&b, origin, s.stop) == hipSuccess)
				fprintf(dump, "%s %.1f %.1f\n", s.name, a * 1000.0f, b * 1000.0f);
		}
		GR_CHECK_HIP(ctx, hipEventSynchronize(s.stop));
		float ms = 0.0f;
		GR_CHECK_HIP(ctx, hipEventElapsedTime(&ms, s.start, s.stop));
		auto itr = ctx->accum.find(s.name);
		if (itr == ctx->accum.end())
		{
			ctx->accum_order.push_back(s.name);
			itr = ctx->accum.emplace(s.name, gr_ctx::Accum{}).first;
		}
		itr->second.count++;
		itr->second.ms += ms;
		itr->second.max_ms = std::max(itr->second.max_ms, double(ms));
		if (s.start != origin)
			ctx->event_pool.push_back(s.start);
		ctx->event_pool.push_back(s.stop);
	}
	if (dump)
		fflush(dump);
	ctx->spans.clear();
	return GR_OK;
}

// A bracket around work that is not one of this library's launches (a collective on its stream, a copy): the same hipEvent pair, the
// same accumulator, under the caller's name.  `name` must outlive the context (a string literal).
int gr_timing_span_begin(gr_ctx *ctx, gr_stream stream, const char *name, void **span)
{
	if (!ctx || !name || !span)
		return GR_ERR_INVALID_ARGUMENT;
	*span = nullptr;
	if (!ctx->timing_enabled || !ctx->timing_matches(name))
		return GR_OK;
	std::lock_guard<std::mutex> holder{ctx->lock};
	auto *s = new gr_timing_span{name, ctx->get_event(), ctx->get_event()};
	if (!s->start || !s->stop)
	{
		// one of the two could not be created: the other goes back to the pool
		if (s->start)
			ctx->event_pool.push_back(s->start);
		if (s->stop)
			ctx->event_pool.push_back(s->stop);
		delete s;
		return GR_OK;
	}
	(void)hipEventRecord(s->start, gr_to_stream(stream));
	*span = s;
	return GR_OK;
}

int gr_timing_span_end(gr_ctx *ctx, gr_stream stream, void *span)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	if (!span)
		return GR_OK;
	auto *s = static_cast<gr_timing_span *>(span);
	(void)hipEventRecord(s->stop, gr_to_stream(stream));
	std::lock_guard<std::mutex> holder{ctx->lock};
	ctx->spans.push_back(*s);
	delete s;
	return GR_OK;
}

// Longest single bracket recorded under `name` since the last reset (gr_timing_query drains the pending brackets first).
int gr_timing_max_ms(gr_ctx *ctx, const char *name, double *max_ms)
{
	if (!ctx || !name || !max_ms)
		return GR_ERR_INVALID_ARGUMENT;
	int ret = drain_spans(ctx);
	if (ret < 0)
		return ret;
	auto itr = ctx->accum.find(name);
	*max_ms = itr == ctx->accum.end() ? 0.0 : itr->second.max_ms;
	return GR_OK;
}

int gr_timing_reset(gr_ctx *ctx)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	int ret = drain_spans(ctx);
	ctx->accum.clear();
	ctx->accum_order.clear();
	return ret;
}

int gr_timing_query(gr_ctx *ctx, gr_timing_entry *entries, int max_entries)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	int ret = drain_spans(ctx);
	if (ret < 0)
		return ret;
	int n = 0;
	for (auto &name : ctx->accum_order)
	{
		if (n >= max_entries)
			break;
		auto itr = ctx->accum.find(name);
		entries[n].name = itr->first.c_str();
		entries[n].count = itr->second.count;
		entries[n].total_ms = itr->second.ms;
		n++;
	}
	return n;
}
}
