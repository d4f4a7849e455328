// Internal context of the C ABI (include/granite_hip.h).  Not part of the public interface.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/granite_hip.h"
#include "row_span.hpp"

struct gr_timing_span
{
	const char *name;
	hipEvent_t start, stop;
};

struct gr_ctx
{
	int device = 0;
	std::mutex lock;
	std::string last_error;

	// 256-entry sRGB8 -> linear table (assets/shaders/inc/srgb.h:4-10 semantics, what the sampler hardware does for
	// an *_SRGB view), built on the host and resident in HBM.
	float *srgb_decode_lut = nullptr;
	// Linear -> sRGB8 staircase table ({threshold bits, byte below it} per bucket; device_common.hpp: encode_srgb8_lut).
	uint2 *srgb_encode_lut = nullptr;
	// Exposed colour -> tonemapped sRGB8 staircase (device_common.hpp: tonemap_srgb8_lut).
	uint2 *tonemap_srgb8_lut = nullptr;
	// SSR (ssr.hip): cos / sin of the 256 azimuths 2 pi u / 255 the dither texture's byte can select (sssr_util.h: SampleGGXVNDF),
	// evaluated once on the host: the traced directions then do not depend on the device's trigonometric approximations.
	float2 *ssr_azimuth_lut = nullptr;

	// gr_bloom_pyramid (post.hip): 64 x {phase counters 0..2, give-ups}, zero-initialised, one slot per launch in rotation.
	static constexpr unsigned PYRAMID_SYNC_SLOTS = 64;
	uint32_t *pyramid_sync = nullptr;
	std::atomic<unsigned> pyramid_launches{0};

	// SMAA lookup tables (assets/textures/smaa/{area,search}.gtx payloads), uploaded through gr_smaa_set_luts.
	void *smaa_area = nullptr;   // 160 x 560 x 2 floats (the RG8 area texture, decoded)
	void *smaa_search = nullptr; // 64 x 16 floats (the R8 search texture, decoded)
	// Bit planes of the SMAA edge texture (smaa_weights.hpp), one set per launch stream, grown on demand.
	struct SmaaBits
	{
		void *memory = nullptr;
		size_t bytes = 0;
	};
	std::map<void *, SmaaBits> smaa_bits;

	// aa.hip: (axis length, 1 / length bits) -> "pixel-centre taps along this axis are texel fetches" (aa_core.hpp: axis_taps_exact)
	std::map<uint64_t, bool> centre_taps_exact, diag_walk_exact_x, diag_walk_exact_y;

	bool timing_enabled = false;
	std::string timing_filter; // empty = every launcher
	unsigned timing_every = 1;  // bracket every n-th matching launch (gr_timing_set_sampling)
	unsigned timing_seen = 0;
	std::vector<gr_timing_span> spans;
	std::vector<hipEvent_t> event_pool;
	struct Accum { uint64_t count = 0; double ms = 0.0, max_ms = 0.0; };
	// timing_filter: empty = every name; otherwise a comma-separated list of the names that get a bracket
	bool timing_matches(const char *name) const
	{
		if (timing_filter.empty())
			return true;
		const size_t n = strlen(name);
		for (size_t at = 0; at <= timing_filter.size();)
		{
			size_t end = timing_filter.find(',', at);
			if (end == std::string::npos)
				end = timing_filter.size();
			if (end - at == n && timing_filter.compare(at, n, name) == 0)
				return true;
			at = end + 1;
		}
		return false;
	}
	std::map<std::string, Accum> accum;
	std::vector<std::string> accum_order;

	int fail(int code, const char *fmt, ...)
	{
		char buf[512];
		va_list va;
		va_start(va, fmt);
		vsnprintf(buf, sizeof(buf), fmt, va);
		va_end(va);
		std::lock_guard<std::mutex> holder{lock};
		last_error = buf;
		return code;
	}

	hipEvent_t get_event()
	{
		if (!event_pool.empty())
		{
			hipEvent_t e = event_pool.back();
			event_pool.pop_back();
			return e;
		}
		// timing brackets order nothing and publish nothing to the host: no system-scope fence (GRANITE_TIMING_EVENT_SYSTEM_FENCE=1
		// restores the default event)
		static const unsigned flags = getenv("GRANITE_TIMING_EVENT_SYSTEM_FENCE") ? unsigned(hipEventDefault) : unsigned(hipEventDisableSystemFence);
		hipEvent_t e;
		if (hipEventCreateWithFlags(&e, flags) != hipSuccess)
			return nullptr;
		return e;
	}
};

// RAII bracket used by every launcher: records hipEvents on the launch stream around the kernel when timing is on.
struct gr_scoped_timing
{
	gr_ctx *ctx;
	hipStream_t stream;
	gr_timing_span span{};
	bool active = false;
	gr_scoped_timing(gr_ctx *ctx_, hipStream_t stream_, const char *name) : ctx(ctx_), stream(stream_)
	{
		if (!ctx->timing_enabled)
			return;
		if (!ctx->timing_matches(name))
			return;
		std::lock_guard<std::mutex> holder{ctx->lock};
		if (ctx->timing_every > 1 && (ctx->timing_seen++ % ctx->timing_every) != 0)
			return;
		span.name = name;
		span.start = ctx->get_event();
		span.stop = ctx->get_event();
		if (!span.start || !span.stop)
		{
			// one of the two could not be created: the other goes back to the pool
			if (span.start)
				ctx->event_pool.push_back(span.start);
			if (span.stop)
				ctx->event_pool.push_back(span.stop);
			return;
		}
		active = true;
		(void)hipEventRecord(span.start, stream);
	}
	~gr_scoped_timing()
	{
		if (!active)
			return;
		(void)hipEventRecord(span.stop, stream);
		std::lock_guard<std::mutex> holder{ctx->lock};
		ctx->spans.push_back(span);
	}
};

#define GR_CHECK_ARG(ctx, cond)                                                                   \
	do                                                                                            \
	{                                                                                             \
		if (!(cond))                                                                              \
			return (ctx)->fail(GR_ERR_INVALID_ARGUMENT, "%s: invalid argument: %s", __func__, #cond); \
	} while (0)

#define GR_CHECK_HIP(ctx, expr)                                                                        \
	do                                                                                                 \
	{                                                                                                  \
		hipError_t err__ = (expr);                                                                     \
		if (err__ != hipSuccess)                                                                       \
			return (ctx)->fail(GR_ERR_HIP, "%s: %s failed: %s", __func__, #expr, hipGetErrorString(err__)); \
	} while (0)

#define GR_CHECK_LAUNCH(ctx)                                                                          \
	do                                                                                                \
	{                                                                                                 \
		hipError_t err__ = hipGetLastError();                                                         \
		if (err__ != hipSuccess)                                                                      \
			return (ctx)->fail(GR_ERR_HIP, "%s: kernel launch failed: %s", __func__, hipGetErrorString(err__)); \
	} while (0)

static inline hipStream_t gr_to_stream(gr_stream s) { return static_cast<hipStream_t>(s); }
static inline unsigned gr_div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

// Resolves a render area (gr_rows) against `height` output rows: [first, end), empty when the band lies outside the image.
static inline RowSpan resolve_rows(const gr_rows *rows, uint32_t height)
{
	if (!rows)
		return {0, height};
	if (rows->count == 0)
		return {0, 0}; // an empty band: the launcher returns without launching
	const uint32_t first = rows->first < height ? rows->first : height;
	const uint64_t end = uint64_t(rows->first) + rows->count;
	return {first, end < height ? uint32_t(end) : height};
}

// A/B switches read from the environment select between implementations that are all parity-tested (results do not change, launch
// times do).  An inherited environment must not change timing silently: the first read of a set switch says so on stderr.
static inline const char *gr_measurement_switch(const char *name)
{
	const char *value = getenv(name);
	if (value)
		fprintf(stderr, "[granite-hip] note: measurement switch %s=%s is set (results unchanged, timing differs)\n", name, value);
	return value;
}
