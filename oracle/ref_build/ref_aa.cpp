// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own anti-aliasing fragment shaders on the CPU (post/fxaa.frag, post/taa_resolve.frag with
// reprojection.h / reprojection_color_space.h), re-spelled into gen/ at build time.  Bindings as renderer/post/fxaa.cpp:28-55
// and renderer/post/temporal.cpp:199-266 make them: FXAA reads the UNORM alias of its input with LinearClamp; TAA binds the
// current frame, depth and motion vectors NearestClamp and the history LinearClamp.  vUV = (pixel + 0.5) * (1 / size).
#include <vector>
#include "glsl_cpu.hpp"

using namespace glsl;

#define FXAA_TARGET_SRGB 0
namespace fxaa_unorm
{
#include "gen/fxaa.inc"
}
#undef FXAA_TARGET_SRGB
#define FXAA_TARGET_SRGB 1
namespace fxaa_srgb
{
#include "gen/fxaa.inc"
}
#undef FXAA_TARGET_SRGB

#define TAA_QUALITY 0
#define REPROJECTION_HISTORY 1
namespace taa_q0
{
#include "gen/taa_resolve.inc"
}
#undef TAA_QUALITY
#define TAA_QUALITY 1
namespace taa_q1
{
#include "gen/taa_resolve.inc"
}
#undef TAA_QUALITY
#define TAA_QUALITY 2
namespace taa_q2
{
#include "gen/taa_resolve.inc"
}
#undef REPROJECTION_HISTORY
#define REPROJECTION_HISTORY 0
namespace taa_first_frame
{
#include "gen/taa_resolve.inc"
}
#undef REPROJECTION_HISTORY
#undef TAA_QUALITY

namespace
{
Texture make(const void *data, int w, int h, Format f, Filter filter)
{
	Texture t;
	t.data = data;
	t.w = w;
	t.h = h;
	t.format = f;
	t.filter = filter;
	return t;
}

Image target(void *data, int w, int h, Format f)
{
	Image i;
	i.data = data;
	i.w = w;
	i.h = h;
	i.format = f;
	return i;
}

vec2 pixel_uv(int x, int y, int w, int h) { return (vec2(float(x), float(y)) + vec2(0.5f, 0.5f)) * vec2(1.0f / float(w), 1.0f / float(h)); }
} // namespace

#define RUN_FXAA(NS, FORMAT)                                                      \
	{                                                                             \
		namespace s = NS;                                                         \
		s::uInput = make(in, w, h, Format::RGBA8_UNORM, Filter::Linear);          \
		s::registers.inv_resolution = vec2(1.0f / float(w), 1.0f / float(h));     \
		Image out_image = target(out, w, h, FORMAT);                              \
		for (int y = 0; y < h; y++)                                               \
			for (int x = 0; x < w; x++)                                           \
			{                                                                     \
				s::vUV = pixel_uv(x, y, w, h);                                    \
				s::main();                                                        \
				imageStore(out_image, ivec2(x, y), vec4(s::FragColor, 1.0f));     \
			}                                                                     \
	}

#define RUN_TAA(NS)                                                                                   \
	{                                                                                                 \
		namespace s = NS;                                                                             \
		s::CurrentFrame = make(current, w, h, Format::RGBA16F, Filter::Nearest);                      \
		s::CurrentDepth = make(depth, w, h, Format::R32F, Filter::Nearest);                           \
		s::MVs = make(mv, w, h, Format::RG16F, Filter::Nearest);                                      \
		s::PreviousFrame = make(history, w, h, Format::RGBA16F, Filter::Linear);                      \
		for (int c = 0; c < 4; c++)                                                                   \
			s::registers.reproj.c[c] = vec4(reproj16[4 * c], reproj16[4 * c + 1], reproj16[4 * c + 2], reproj16[4 * c + 3]); \
		s::registers.rt_metrics = vec4(1.0f / float(w), 1.0f / float(h), float(w), float(h));         \
		for (int y = 0; y < h; y++)                                                                   \
			for (int x = 0; x < w; x++)                                                               \
			{                                                                                         \
				s::vUV = pixel_uv(x, y, w, h);                                                        \
				s::main();                                                                            \
				imageStore(color_image, ivec2(x, y), vec4(s::Color, 1.0f));                           \
				imageStore(history_image, ivec2(x, y), vec4(s::HistoryColor, 1.0f));                  \
			}                                                                                         \
	}

extern "C" {

void ref_fxaa(const uint8_t *in, int w, int h, uint8_t *out, int target_srgb)
{
	if (target_srgb)
		RUN_FXAA(fxaa_srgb, Format::RGBA8_SRGB)
	else
		RUN_FXAA(fxaa_unorm, Format::RGBA8_UNORM)
}

// history == NULL: REPROJECTION_HISTORY = 0 (first frame).  Outputs: RGBA16F colour + history, alpha written as 1.
// color_b10g11r11: the colour attachment is B10G11R11_UFLOAT_PACK32 (temporal.cpp:211-213); out_color receives its exact RGBA16F texels.
void ref_taa_resolve_fmt(const uint16_t *current, const float *depth, const uint16_t *mv, const uint16_t *history, int w, int h,
                         const float *reproj16, int quality, uint16_t *out_color, uint16_t *out_history, int color_b10g11r11);
void ref_taa_resolve(const uint16_t *current, const float *depth, const uint16_t *mv, const uint16_t *history, int w, int h,
                     const float *reproj16, int quality, uint16_t *out_color, uint16_t *out_history)
{
	ref_taa_resolve_fmt(current, depth, mv, history, w, h, reproj16, quality, out_color, out_history, 0);
}
void ref_taa_resolve_fmt(const uint16_t *current, const float *depth, const uint16_t *mv, const uint16_t *history, int w, int h,
                         const float *reproj16, int quality, uint16_t *out_color, uint16_t *out_history, int color_b10g11r11)
{
	std::vector<uint32_t> packed(color_b10g11r11 ? size_t(w) * h : 0);
	Image color_image = color_b10g11r11 ? target(packed.data(), w, h, Format::B10G11R11_UFLOAT) : target(out_color, w, h, Format::RGBA16F);
	Image history_image = target(out_history, w, h, Format::RGBA16F);
	struct Unpack
	{
		std::vector<uint32_t> &words;
		uint16_t *out;
		~Unpack()
		{
			for (size_t i = 0; i < words.size(); i++)
			{
				const orc::vec4 v = orc::unpack_b10g11r11(words[i]);
				out[4 * i] = orc::float_to_half_rne(v.x), out[4 * i + 1] = orc::float_to_half_rne(v.y), out[4 * i + 2] = orc::float_to_half_rne(v.z);
				out[4 * i + 3] = 0x3c00u;
			}
		}
	} unpack_at_exit{packed, out_color};
	if (!history)
	{
		namespace s = taa_first_frame;
		s::CurrentFrame = make(current, w, h, Format::RGBA16F, Filter::Nearest);
		s::registers.rt_metrics = vec4(1.0f / float(w), 1.0f / float(h), float(w), float(h));
		for (int y = 0; y < h; y++)
			for (int x = 0; x < w; x++)
			{
				s::vUV = pixel_uv(x, y, w, h);
				s::main();
				imageStore(color_image, ivec2(x, y), vec4(s::Color, 1.0f));
				imageStore(history_image, ivec2(x, y), vec4(s::HistoryColor, 1.0f));
			}
	}
	else if (quality == 0)
		RUN_TAA(taa_q0)
	else if (quality == 1)
		RUN_TAA(taa_q1)
	else
		RUN_TAA(taa_q2)
}

} // extern "C"
