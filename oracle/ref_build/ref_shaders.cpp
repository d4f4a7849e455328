// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own post-chain shaders on the CPU: each `gen/*.inc` below (build scratch) is the shader source from
// /root/reference/assets/shaders/post re-spelled by glsl2cpp.py at build time, compiled against the GLSL
// environment in glsl_cpu.hpp, one namespace per shader variant.  The entry points mirror the oracle's (oracle_post.cpp) so
// the two can be compared buffer for buffer: what is being checked is the oracle's reading of the shader text.
#include <barrier>
#include <thread>
#include <vector>
#include "glsl_cpu.hpp"

using namespace glsl;

namespace
{
std::barrier<> *workgroup_barrier = nullptr;
}
static inline void memoryBarrierShared() {}
static inline void barrier()
{
	if (workgroup_barrier)
		workgroup_barrier->arrive_and_wait();
}

#define DYNAMIC_EXPOSURE 1
namespace bloom_threshold_dynamic
{
#include "gen/bloom_threshold.inc"
}
namespace tonemap_dynamic
{
#include "gen/tonemap.inc"
}
#undef DYNAMIC_EXPOSURE
#define DYNAMIC_EXPOSURE 0
namespace bloom_threshold_static
{
#include "gen/bloom_threshold.inc"
}
namespace tonemap_static
{
#include "gen/tonemap.inc"
}
namespace blit
{
#include "gen/blit.inc"
}
#undef DYNAMIC_EXPOSURE

#define FEEDBACK 1
namespace bloom_downsample_feedback
{
#include "gen/bloom_downsample.inc"
}
#undef FEEDBACK
#define FEEDBACK 0
namespace bloom_downsample_plain
{
#include "gen/bloom_downsample.inc"
}
#undef FEEDBACK
namespace bloom_upsample
{
#include "gen/bloom_upsample.inc"
}
namespace pq10
{
#include "gen/pq10_encode.inc"
}
namespace luminance
{
#include "gen/luminance.inc"
#undef STEP
}

namespace
{
Texture tex16(const uint16_t *data, int w, int h, Filter filter = Filter::Linear)
{
	Texture t;
	t.data = data;
	t.w = w;
	t.h = h;
	t.format = Format::RGBA16F;
	t.filter = filter;
	return t;
}

Image img16(uint16_t *data, int w, int h)
{
	Image i;
	i.data = data;
	i.w = w;
	i.h = h;
	i.format = Format::RGBA16F;
	return i;
}

// One invocation per thread of every 8 x 8 workgroup covering w x h (the shaders bounds-check themselves).  Rows of invocations run on
// the host's cores (OpenMP in the RUNNER: per-invocation state -- gl_*, stage inputs / outputs -- is thread_local, resources are bound
// before the loop and only read inside it, every invocation stores to its own texel; the shader text is not touched).
template <typename Main>
void dispatch_8x8(int w, int h, Main main_fn)
{
	const int gw = (w + 7) & ~7, gh = (h + 7) & ~7;
#pragma omp parallel for schedule(dynamic, 4)
	for (int y = 0; y < gh; y++)
		for (int x = 0; x < gw; x++)
		{
			gl_GlobalInvocationID = uvec3(uint(x), uint(y), 0u);
			gl_LocalInvocationID = uvec3(uint(x & 7), uint(y & 7), 0u);
			gl_WorkGroupID = uvec3(uint(x >> 3), uint(y >> 3), 0u);
			gl_LocalInvocationIndex = uint((y & 7) * 8 + (x & 7));
			main_fn();
		}
}
} // namespace

extern "C" {

// bloom_threshold.comp with the push constants of bloom_threshold_build_compute (renderer/post/hdr.cpp:133-142).
void ref_bloom_threshold(const uint16_t *hdr, int iw, int ih, uint16_t *out, int ow, int oh, const float *lum3)
{
	if (lum3)
	{
		namespace s = bloom_threshold_dynamic;
		s::average_log_luminance = lum3[0], s::average_linear_luminance = lum3[1], s::average_inv_linear_luminance = lum3[2];
		s::uHDR = tex16(hdr, iw, ih);
		s::uOutput = img16(out, ow, oh);
		s::registers.num_threads = uvec2(uint(ow), uint(oh));
		s::registers.inv_output_size = vec2(1.0f / float(ow), 1.0f / float(oh));
		dispatch_8x8(ow, oh, s::main);
	}
	else
	{
		namespace s = bloom_threshold_static;
		s::uHDR = tex16(hdr, iw, ih);
		s::uOutput = img16(out, ow, oh);
		s::registers.num_threads = uvec2(uint(ow), uint(oh));
		s::registers.inv_output_size = vec2(1.0f / float(ow), 1.0f / float(oh));
		dispatch_8x8(ow, oh, s::main);
	}
}

// bloom_downsample.comp, push constants hdr.cpp:169-185.  history != NULL: FEEDBACK variant (sampled NearestClamp? no:
// hdr.cpp:158-163 binds it LinearClamp at the same size, so filtering degenerates to the texel).
void ref_bloom_downsample(const uint16_t *in, int iw, int ih, uint16_t *out, int ow, int oh, const uint16_t *history, float lerp)
{
	if (history)
	{
		namespace s = bloom_downsample_feedback;
		s::uSampler = tex16(in, iw, ih);
		s::uSamplerHistory = tex16(history, ow, oh, Filter::Nearest);
		s::uOutput = img16(out, ow, oh);
		s::registers.num_threads = uvec2(uint(ow), uint(oh));
		s::registers.inv_output_size = vec2(1.0f / float(ow), 1.0f / float(oh));
		s::registers.inv_input_size = vec2(1.0f / float(iw), 1.0f / float(ih));
		s::registers.lerp = lerp;
		dispatch_8x8(ow, oh, s::main);
	}
	else
	{
		namespace s = bloom_downsample_plain;
		s::uSampler = tex16(in, iw, ih);
		s::uOutput = img16(out, ow, oh);
		s::registers.num_threads = uvec2(uint(ow), uint(oh));
		s::registers.inv_output_size = vec2(1.0f / float(ow), 1.0f / float(oh));
		s::registers.inv_input_size = vec2(1.0f / float(iw), 1.0f / float(ih));
		dispatch_8x8(ow, oh, s::main);
	}
}

// bloom_upsample.comp, push constants hdr.cpp:202-214.
void ref_bloom_upsample(const uint16_t *in, int iw, int ih, uint16_t *out, int ow, int oh)
{
	namespace s = bloom_upsample;
	s::uSampler = tex16(in, iw, ih);
	s::uOutput = img16(out, ow, oh);
	s::registers.num_threads = uvec2(uint(ow), uint(oh));
	s::registers.inv_output_size = vec2(1.0f / float(ow), 1.0f / float(oh));
	s::registers.inv_input_size = vec2(1.0f / float(iw), 1.0f / float(ih));
	dispatch_8x8(ow, oh, s::main);
}

// luminance.comp: ONE 8 x 8 workgroup with real barriers (64 threads), push constants hdr.cpp:84-96 (size = d3 / 2).
void ref_luminance(const uint16_t *d3, int w, int h, float *lum3, float lerp, float min_loglum, float max_loglum)
{
	namespace s = luminance;
	s::average_log_luminance = lum3[0], s::average_linear_luminance = lum3[1], s::average_inv_linear_luminance = lum3[2];
	s::uImage = tex16(d3, w, h);
	s::registers.size = ivec2(w / 2, h / 2);
	s::registers.lerp = lerp;
	s::registers.min_loglum = min_loglum;
	s::registers.max_loglum = max_loglum;
	std::barrier<> sync(64);
	workgroup_barrier = &sync;
	std::vector<std::thread> threads;
	for (int i = 0; i < 64; i++)
		threads.emplace_back([i]() {
			gl_LocalInvocationID = uvec3(uint(i & 7), uint(i >> 3), 0u);
			gl_GlobalInvocationID = gl_LocalInvocationID;
			gl_LocalInvocationIndex = uint(i);
			s::main();
		});
	for (auto &t : threads)
		t.join();
	workgroup_barrier = nullptr;
	lum3[0] = s::average_log_luminance, lum3[1] = s::average_linear_luminance, lum3[2] = s::average_inv_linear_luminance;
}

// tonemap.frag on a full-screen quad: the interpolated varying is vUV = (pixel + 0.5) * (1 / size) (quad.vert; the oracle's
// convention for every full-screen pass), output attachment R8G8B8A8_SRGB, alpha written as 1.
void ref_tonemap(const uint16_t *hdr, int w, int h, const uint16_t *bloom, int bw, int bh, const float *lum3, float dynamic_exposure, uint8_t *out)
{
	Image target;
	target.data = out;
	target.w = w;
	target.h = h;
	target.format = Format::RGBA8_SRGB;
	const bool dynamic = lum3 != nullptr;
	if (dynamic)
	{
		namespace s = tonemap_dynamic;
		s::average_log_luminance = lum3[0], s::average_linear_luminance = lum3[1], s::average_inv_linear_luminance = lum3[2];
		s::uHDR = tex16(hdr, w, h);
		s::uBloom = tex16(bloom, bw, bh);
		s::registers.dynamic_exposure = dynamic_exposure;
	}
	else
	{
		namespace s = tonemap_static;
		s::uHDR = tex16(hdr, w, h);
		s::uBloom = tex16(bloom, bw, bh);
		s::registers.dynamic_exposure = dynamic_exposure;
	}
#pragma omp parallel for schedule(dynamic, 4)
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			const vec2 uv = (vec2(float(x), float(y)) + vec2(0.5f, 0.5f)) * vec2(1.0f / float(w), 1.0f / float(h));
			vec3 color;
			if (dynamic)
			{
				namespace s = tonemap_dynamic;
				s::vUV = uv;
				s::main();
				color = s::FragColor;
			}
			else
			{
				namespace s = tonemap_static;
				s::vUV = uv;
				s::main();
				color = s::FragColor;
			}
			imageStore(target, ivec2(x, y), vec4(color, 1.0f));
		}
}

// pq10_encode.frag with the UBO of PQEncoder::build_render_pass (hdr.cpp:626-636): hdr RGBA16F, ui through its sRGB view,
// output A2B10G10R10 (alpha 1).  conversion9: column-major mat3.
void ref_pq10_encode(const uint16_t *hdr, const uint8_t *ui_srgb8, int w, int h, const float *conversion9, float hdr_pre_exposure, float ui_pre_exposure,
                     float max_light_level, uint32_t *out)
{
	namespace s = pq10;
	s::uHDR = tex16(hdr, w, h, Filter::Nearest);
	s::uUI.data = ui_srgb8;
	s::uUI.w = w;
	s::uUI.h = h;
	s::uUI.format = Format::RGBA8_SRGB;
	s::uUI.filter = Filter::Nearest;
	for (int col = 0; col < 4; col++)
		s::config.primary_conversion.c[col] = col < 3 ? vec4(conversion9[3 * col], conversion9[3 * col + 1], conversion9[3 * col + 2], 0.0f) : vec4(0.0f, 0.0f, 0.0f, 1.0f);
	s::config.hdr_pre_exposure = hdr_pre_exposure;
	s::config.ui_pre_exposure = ui_pre_exposure;
	s::config.max_light_level = max_light_level;
	s::config.inv_max_light_level = 1.0f / max_light_level;
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			gl_FragCoord = vec4(float(x) + 0.5f, float(y) + 0.5f, 0.0f, 1.0f);
			s::main();
			auto q = [](float v) -> uint32_t { return !(v > 0.0f) ? 0u : (v >= 1.0f ? 1023u : uint32_t(int(v * 1023.0f + 0.5f))); };
			out[size_t(y) * w + x] = q(s::FragColor.x) | (q(s::FragColor.y) << 10) | (q(s::FragColor.z) << 20) | (3u << 30);
		}
}

// blit.frag on a full-screen quad (tools/aa_bench.cpp:97-105,138-147): formats 0 = RGBA16F, 1 = RGBA8_UNORM, 2 = RGBA8_SRGB on
// either side (the sampled view decodes, the attachment store encodes), LinearClamp or NearestClamp.
void ref_blit(const void *in, int iw, int ih, int in_format, void *out, int ow, int oh, int out_format, int linear)
{
	namespace s = blit;
	static const Format formats[3] = {Format::RGBA16F, Format::RGBA8_UNORM, Format::RGBA8_SRGB};
	s::uImage.data = in;
	s::uImage.w = iw;
	s::uImage.h = ih;
	s::uImage.format = formats[in_format];
	s::uImage.filter = linear ? Filter::Linear : Filter::Nearest;
	Image target;
	target.data = out;
	target.w = ow;
	target.h = oh;
	target.format = formats[out_format];
	for (int y = 0; y < oh; y++)
		for (int x = 0; x < ow; x++)
		{
			s::vUV = (vec2(float(x), float(y)) + vec2(0.5f, 0.5f)) * vec2(1.0f / float(ow), 1.0f / float(oh));
			s::main();
			imageStore(target, ivec2(x, y), s::FragColor);
		}
}

} // extern "C"
