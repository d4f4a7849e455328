// Context, memory and timing entry points of the C ABI (include/granite_hip.h).
#include "ctx.hpp"
#include <cmath>
#include <cstring>

extern "C" {

int gr_abi_version(void)
{
	return GR_ABI_VERSION;
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
	if (hipMalloc(reinterpret_cast<void **>(&ctx->srgb_decode_lut), sizeof(lut)) != hipSuccess ||
	    hipMemcpy(ctx->srgb_decode_lut, lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess)
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
	if (ctx->srgb_decode_lut)
		(void)hipFree(ctx->srgb_decode_lut);
	if (ctx->smaa_area)
		(void)hipFree(ctx->smaa_area);
	if (ctx->smaa_search)
		(void)hipFree(ctx->smaa_search);
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
	GR_CHECK_HIP(ctx, hipMemset(*dptr, 0, bytes));
	return GR_OK;
}

int gr_free(gr_ctx *ctx, void *dptr)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	if (dptr)
		GR_CHECK_HIP(ctx, hipFree(dptr));
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

static int drain_spans(gr_ctx *ctx)
{
	for (auto &s : ctx->spans)
	{
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
		ctx->event_pool.push_back(s.start);
		ctx->event_pool.push_back(s.stop);
	}
	ctx->spans.clear();
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
