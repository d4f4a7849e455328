// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own SMAA 1x passes on the CPU: post/smaa_{edge_detection,blend_weight,neighbor_blend}.{vert,frag} with
// post/SMAA.hlsl in its GLSL 4 spelling (smaa_common.h), re-spelled into gen/ at build time.  Pass set-up as
// renderer/post/smaa.cpp:32-208: every texture LinearClamp, colour read through its UNORM alias, edges cleared to zero and
// discarded fragments left untouched, the weight pass masked by the stencil the edge pass wrote (= edge texel non-zero).  The
// vertex shaders' outputs are affine in the position, so they are evaluated at each pixel centre
// (TexCoord = (pixel + 0.5) * rt_metrics.xy) instead of being interpolated.
#include "glsl_cpu.hpp"

using namespace glsl;

#define SMAA_SUBPIXEL_MODE 0

// SMAA.hlsl turns the preset into a set of macros that stay defined, so every preset gets its own translation unit: this
// file is compiled four times with -DSMAA_QUALITY=0..3 (Makefile) and exports ref_smaa_{edges,weights}_q<N>; the
// neighbourhood blend does not depend on the preset and is exported by the SMAA_QUALITY == 3 unit only.
#ifndef SMAA_QUALITY
#error "compile with -DSMAA_QUALITY=0..3"
#endif
#define CONCAT2(a, b) a##b
#define CONCAT(a, b) CONCAT2(a, b)
#define QNS CONCAT(smaa_q, SMAA_QUALITY)
#define SMAA_TARGET_SRGB 0
namespace QNS
{
namespace edge_vs
{
#include "gen/smaa_edge_detection.vert.inc"
}
namespace edge_ps
{
#include "gen/smaa_edge_detection.frag.inc"
}
namespace weight_vs
{
#include "gen/smaa_blend_weight.vert.inc"
}
namespace weight_ps
{
#include "gen/smaa_blend_weight.frag.inc"
}
#if SMAA_QUALITY == 3
namespace blend_vs
{
#include "gen/smaa_neighbor_blend.vert.inc"
}
namespace blend_ps
{
#include "gen/smaa_neighbor_blend.frag.inc"
}
#endif
}
#if SMAA_QUALITY == 3
#undef SMAA_TARGET_SRGB
#define SMAA_TARGET_SRGB 1
namespace smaa_blend_srgb
{
namespace blend_ps
{
#include "gen/smaa_neighbor_blend.frag.inc"
}
}
#endif

namespace
{
Texture make(const void *data, int w, int h, Format f)
{
	Texture t;
	t.data = data;
	t.w = w;
	t.h = h;
	t.format = f;
	t.filter = Filter::Linear;
	return t;
}
vec4 metrics(int w, int h) { return vec4(1.0f / float(w), 1.0f / float(h), float(w), float(h)); }
vec2 centre(int x, int y, const vec4 &rt) { return vec2((float(x) + 0.5f) * rt.x, (float(y) + 0.5f) * rt.y); }
} // namespace

#define RUN_EDGES(Q)                                                                          \
	{                                                                                         \
		namespace vs = Q::edge_vs;                                                            \
		namespace ps = Q::edge_ps;                                                            \
		vs::registers.rt_metrics = ps::registers.rt_metrics = rt;                             \
		ps::ColorTex = make(color, w, h, Format::RGBA8_UNORM);                                \
		for (int y = 0; y < h; y++)                                                           \
			for (int x = 0; x < w; x++)                                                       \
			{                                                                                 \
				vec4 offsets[3];                                                              \
				const vec2 uv = centre(x, y, rt);                                             \
				vs::SMAAEdgeDetectionVS(uv, offsets);                                         \
				ps::vTex = uv;                                                                \
				ps::vOffset0 = offsets[0], ps::vOffset1 = offsets[1], ps::vOffset2 = offsets[2]; \
				try                                                                           \
				{                                                                             \
					ps::main();                                                               \
				}                                                                             \
				catch (const Discard &)                                                       \
				{                                                                             \
					continue;                                                                 \
				}                                                                             \
				uint8_t *o = edges_rg8 + (size_t(y) * w + x) * 2;                             \
				o[0] = orc::float_to_unorm8(ps::Edges.x);                                     \
				o[1] = orc::float_to_unorm8(ps::Edges.y);                                     \
			}                                                                                 \
	}

#define RUN_WEIGHTS(Q)                                                                                     \
	{                                                                                                      \
		namespace vs = Q::weight_vs;                                                                       \
		namespace ps = Q::weight_ps;                                                                       \
		vs::registers.rt_metrics = ps::registers.rt_metrics = rt;                                          \
		ps::EdgesTex = make(edges_rg8, w, h, Format::RG8_UNORM);                                           \
		ps::AreaTex = make(area_rg8, 160, 560, Format::RG8_UNORM);                                         \
		ps::SearchTex = make(search_r8, 64, 16, Format::R8_UNORM);                                         \
		for (int y = 0; y < h; y++)                                                                        \
			for (int x = 0; x < w; x++)                                                                    \
			{                                                                                              \
				const uint8_t *e = edges_rg8 + (size_t(y) * w + x) * 2;                                    \
				if (!(e[0] | e[1]))                                                                        \
					continue;                                                                              \
				vec4 offsets[3];                                                                           \
				vec2 pixcoord;                                                                             \
				const vec2 uv = centre(x, y, rt);                                                          \
				vs::SMAABlendingWeightCalculationVS(uv, pixcoord, offsets);                                \
				ps::vTex = uv;                                                                             \
				ps::vPixCoord = pixcoord;                                                                  \
				ps::vOffset0 = offsets[0], ps::vOffset1 = offsets[1], ps::vOffset2 = offsets[2];           \
				ps::main();                                                                                \
				uint8_t *o = weights_rgba8 + (size_t(y) * w + x) * 4;                                      \
				for (int c = 0; c < 4; c++)                                                                \
					o[c] = orc::float_to_unorm8(ps::Weights.d[c]);                                         \
			}                                                                                              \
	}

extern "C" {

void CONCAT(ref_smaa_edges_q, SMAA_QUALITY)(const uint8_t *color, int w, int h, uint8_t *edges_rg8)
{
	const vec4 rt = metrics(w, h);
	memset(edges_rg8, 0, size_t(w) * h * 2);
	RUN_EDGES(QNS)
}

void CONCAT(ref_smaa_weights_q, SMAA_QUALITY)(const uint8_t *edges_rg8, int w, int h, const uint8_t *area_rg8, const uint8_t *search_r8, uint8_t *weights_rgba8)
{
	const vec4 rt = metrics(w, h);
	memset(weights_rgba8, 0, size_t(w) * h * 4);
	RUN_WEIGHTS(QNS)
}

#if SMAA_QUALITY == 3
void ref_smaa_blend(const uint8_t *color, const uint8_t *weights_rgba8, int w, int h, uint8_t *out, int target_srgb)
{
	const vec4 rt = metrics(w, h);
	namespace vs = QNS::blend_vs;
	vs::registers.rt_metrics = rt;
	Image image;
	image.data = out;
	image.w = w;
	image.h = h;
	image.format = target_srgb ? Format::RGBA8_SRGB : Format::RGBA8_UNORM;
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			const vec2 uv = centre(x, y, rt);
			vec4 offset;
			vs::SMAANeighborhoodBlendingVS(uv, offset);
			vec4 result;
			if (target_srgb)
			{
				namespace ps = smaa_blend_srgb::blend_ps;
				ps::registers.rt_metrics = rt;
				ps::ColorTex = make(color, w, h, Format::RGBA8_UNORM);
				ps::BlendTex = make(weights_rgba8, w, h, Format::RGBA8_UNORM);
				ps::vTex = uv;
				ps::vOffset = offset;
				ps::main();
				result = ps::Color;
			}
			else
			{
				namespace ps = QNS::blend_ps;
				ps::registers.rt_metrics = rt;
				ps::ColorTex = make(color, w, h, Format::RGBA8_UNORM);
				ps::BlendTex = make(weights_rgba8, w, h, Format::RGBA8_UNORM);
				ps::vTex = uv;
				ps::vOffset = offset;
				ps::main();
				result = ps::Color;
			}
			imageStore(image, ivec2(x, y), result);
		}
}
#endif

} // extern "C"
