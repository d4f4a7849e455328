// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own deferred-lighting shaders on the CPU: lights/directional.frag and lights/clustering.frag with
// everything they include (lighting.h, pbr.h, clusterer_bindless.h, point.h, spot.h ...), re-spelled by glsl2cpp.py at build
// time into gen/ (scratch, removed after the compile) and compiled against glsl_cpu.hpp.  Defines as DeferredLightRenderer::render_light sets them
// on this path (renderer.cpp:1020-1056,1125-1147): VOLUMETRIC_DIFFUSE_FALLBACK (+ AMBIENT_OCCLUSION) for the directional
// quad, nothing for the clustered quad; STAGE_FRAGMENT from the shader compiler (compiler/compiler.cpp:289).
//
// Each fragment runs alone, so subgroupMin / Max / Or are the identity: the exact per-pixel light set (the oracle's
// wave_tile = 0 form; a wider subgroup only adds lights whose contribution is exactly zero).  What the rasteriser provides
// is stated as the oracle states it: gl_FragCoord = pixel centre, vClip = inv_view_projection * (ndc, 0, 1) at the pixel
// centre, depth test NOT_EQUAL against the quad's z = 0, blending ONE / ONE into the RGBA16F target after each quad.
#include <vector>
#include "glsl_cpu.hpp"

using namespace glsl;

#define STAGE_FRAGMENT 1
namespace clustering
{
#include "gen/clustering.inc"
}
namespace fog_quad
{
#include "gen/fog.inc"
}
namespace directional_plain
{
#include "gen/directional.inc"
}
#define VOLUMETRIC_DIFFUSE_FALLBACK 1
namespace directional_fallback
{
#include "gen/directional.inc"
}
#define AMBIENT_OCCLUSION 1
namespace directional_fallback_ao
{
#include "gen/directional.inc"
}
#undef AMBIENT_OCCLUSION
#undef VOLUMETRIC_DIFFUSE_FALLBACK

namespace
{
// Same layout as OrcLightingArgs (oracle_lights.cpp): the test hands both the same structure.
struct LightInfo
{
	float color[3];
	uint32_t spot_scale_bias;
	float position[3];
	uint32_t offset_radius;
	float direction[3];
	float inv_radius;
};
struct ClusterParams
{
	float transform[16];
	float clip_scale[4];
	float camera_base[3], pad0;
	float camera_front[3], pad1;
	float xy_scale[2];
	int32_t resolution_xy[2];
	float inv_resolution_xy[2];
	int32_t num_lights, num_lights_32, num_decals, num_decals_32, decals_texture_offset, z_max_index;
	float z_scale;
	float pad2[3];
};
struct RenderParams
{
	float projection[16], view[16], view_projection[16], inv_projection[16], inv_view[16], inv_view_projection[16];
	float camera_position[3], camera_front[3];
	float z_near, z_far;
};
struct LightingArgs
{
	int32_t width, height;
	const uint32_t *albedo;
	const uint32_t *normal;
	const uint16_t *pbr;
	const float *depth;
	uint16_t *hdr;
	const RenderParams *rp;
	const ClusterParams *cluster;
	const LightInfo *lights;
	const uint32_t *type_mask;
	const uint32_t *bitmask;
	const uint32_t *range;
	float dir_color[3];
	float dir_direction[3];
	int32_t enable_directional, enable_clustered, ambient_fallback, wave_tile;
	const uint8_t *ambient_occlusion;
	int32_t ao_width, ao_height;
	int32_t hdr_b10g11r11; // the target is a B10G11R11_UFLOAT_PACK32 attachment, held as its exact RGBA16F texels
	float fog_color[3];    // the fog quad of render_light (renderer.cpp:1179-1196), drawn when fog_falloff > 0
	float fog_falloff;
};

Texture make(const void *data, int w, int h, Format f, Filter filter = Filter::Nearest)
{
	Texture t;
	t.data = data;
	t.w = w;
	t.h = h;
	t.format = f;
	t.filter = filter;
	return t;
}

mat4 load_mat4(const float *m)
{
	mat4 r;
	for (int c = 0; c < 4; c++)
		r.c[c] = vec4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
	return r;
}

vec3 ld3(const float *v) { return vec3(v[0], v[1], v[2]); }

// ONE / ONE blend into the colour attachment: the sum is converted by the attachment's store (fp16 round to nearest even, or the
// packed-float rule of B10G11R11_UFLOAT_PACK32).
void blend_one_one(const LightingArgs *a, int x, int y, const vec3 &src)
{
	uint16_t *p = a->hdr + (size_t(y) * a->width + x) * 4;
	const float r = orc::half_to_float(p[0]) + src.d[0], g = orc::half_to_float(p[1]) + src.d[1], b = orc::half_to_float(p[2]) + src.d[2];
	if (a->hdr_b10g11r11)
	{
		Image target;
		uint32_t word = 0;
		target.data = &word, target.w = 1, target.h = 1, target.format = Format::B10G11R11_UFLOAT;
		imageStore(target, ivec2(0, 0), vec4(r, g, b, 1.0f));
		const orc::vec4 q = orc::unpack_b10g11r11(word);
		p[0] = orc::float_to_half_rne(q.x), p[1] = orc::float_to_half_rne(q.y), p[2] = orc::float_to_half_rne(q.z), p[3] = 0x3c00u;
		return;
	}
	p[0] = orc::float_to_half_rne(r), p[1] = orc::float_to_half_rne(g), p[2] = orc::float_to_half_rne(b);
}

// ONE_MINUS_SRC_ALPHA / SRC_ALPHA on colour and alpha (cmd.set_blend_factors sets both): src * (1 - src.a) + dst * src.a, converted
// by the attachment's store.
void blend_fog(const LightingArgs *a, int x, int y, const vec4 &src)
{
	uint16_t *p = a->hdr + (size_t(y) * a->width + x) * 4;
	float out[4];
	for (int c = 0; c < 4; c++)
		out[c] = src.d[c] * (1.0f - src.d[3]) + orc::half_to_float(p[c]) * src.d[3];
	Image target;
	if (a->hdr_b10g11r11)
	{
		uint32_t word = 0;
		target.data = &word, target.w = 1, target.h = 1, target.format = Format::B10G11R11_UFLOAT;
		imageStore(target, ivec2(0, 0), vec4(out[0], out[1], out[2], out[3]));
		const orc::vec4 q = orc::unpack_b10g11r11(word);
		p[0] = orc::float_to_half_rne(q.x), p[1] = orc::float_to_half_rne(q.y), p[2] = orc::float_to_half_rne(q.z), p[3] = 0x3c00u;
		return;
	}
	target.data = p, target.w = 1, target.h = 1, target.format = Format::RGBA16F;
	imageStore(target, ivec2(0, 0), vec4(out[0], out[1], out[2], out[3]));
}

template <typename Setup>
void bind_gbuffer(const LightingArgs *a, Setup set)
{
	set(make(a->albedo, a->width, a->height, Format::RGBA8_SRGB), make(a->normal, a->width, a->height, Format::A2B10G10R10_UNORM),
	    make(a->pbr, a->width, a->height, Format::RG8_UNORM), make(a->depth, a->width, a->height, Format::R32F));
}
} // namespace

#define RUN_DIRECTIONAL(NS)                                                                                         \
	{                                                                                                               \
		namespace s = NS;                                                                                           \
		bind_gbuffer(a, [](const Texture &c, const Texture &n, const Texture &p, const Texture &d) {               \
			s::BaseColor = c, s::Normal = n, s::PBR = p, s::Depth = d;                                              \
		});                                                                                                         \
		s::registers.inverse_view_projection_col2 = inv_vp.c[2];                                                    \
		s::registers.color = ld3(a->dir_color);                                                                     \
		s::registers.direction = ld3(a->dir_direction);                                                             \
		s::registers.camera_pos = ld3(a->rp->camera_position);                                                      \
		s::registers.camera_front = ld3(a->rp->camera_front);                                                       \
		s::registers.inv_resolution = inv_resolution;                                                               \
		_Pragma("omp parallel for schedule(dynamic, 4)")                                                            \
		for (int y = 0; y < H; y++)                                                                                 \
			for (int x = 0; x < W; x++)                                                                             \
			{                                                                                                       \
				if (a->depth[size_t(y) * W + x] == 0.0f)                                                            \
					continue;                                                                                       \
				gl_FragCoord = vec4(float(x) + 0.5f, float(y) + 0.5f, 0.0f, 1.0f);                                  \
				s::vClip = clip_at(x, y);                                                                           \
				s::main();                                                                                          \
				blend_one_one(a, x, y, s::FragColor);                                                       \
			}                                                                                                       \
	}

extern "C" void ref_lighting(const LightingArgs *a)
{
	const int W = a->width, H = a->height;
	const mat4 inv_vp = load_mat4(a->rp->inv_view_projection);
	const vec2 inv_resolution(1.0f / float(W), 1.0f / float(H));
	auto clip_at = [&](int x, int y) {
		const vec2 ndc(2.0f * ((float(x) + 0.5f) * inv_resolution.x) - 1.0f, 2.0f * ((float(y) + 0.5f) * inv_resolution.y) - 1.0f);
		return inv_vp * vec4(ndc.x, ndc.y, 0.0f, 1.0f);
	};

	if (a->enable_directional)
	{
		if (a->ambient_fallback && a->ambient_occlusion)
		{
			directional_fallback_ao::uAmbientOcclusion = make(a->ambient_occlusion, a->ao_width, a->ao_height, Format::R8_UNORM, Filter::Linear);
			RUN_DIRECTIONAL(directional_fallback_ao)
		}
		else if (a->ambient_fallback)
			RUN_DIRECTIONAL(directional_fallback)
		else
			RUN_DIRECTIONAL(directional_plain)
	}

	if (a->enable_clustered && a->cluster->num_lights > 0)
	{
		namespace s = clustering;
		bind_gbuffer(a, [](const Texture &c, const Texture &n, const Texture &p, const Texture &d) {
			s::BaseColor = c, s::Normal = n, s::PBR = p, s::Depth = d;
		});
		const ClusterParams &cl = *a->cluster;
		s::cluster.transform = load_mat4(cl.transform);
		s::cluster.clip_scale = vec4(cl.clip_scale[0], cl.clip_scale[1], cl.clip_scale[2], cl.clip_scale[3]);
		s::cluster.camera_base = ld3(cl.camera_base);
		s::cluster.camera_front = ld3(cl.camera_front);
		s::cluster.xy_scale = vec2(cl.xy_scale[0], cl.xy_scale[1]);
		s::cluster.resolution_xy = ivec2(cl.resolution_xy[0], cl.resolution_xy[1]);
		s::cluster.inv_resolution_xy = vec2(cl.inv_resolution_xy[0], cl.inv_resolution_xy[1]);
		s::cluster.num_lights = cl.num_lights;
		s::cluster.num_lights_32 = cl.num_lights_32;
		s::cluster.num_decals = cl.num_decals;
		s::cluster.num_decals_32 = cl.num_decals_32;
		s::cluster.decals_texture_offset = cl.decals_texture_offset;
		s::cluster.z_max_index = cl.z_max_index;
		s::cluster.z_scale = cl.z_scale;
		for (int i = 0; i < cl.num_lights; i++)
		{
			auto &dst = s::cluster_transforms.lights[i];
			const LightInfo &src = a->lights[i];
			dst.color = ld3(src.color);
			dst.spot_scale_bias = src.spot_scale_bias;
			dst.position = ld3(src.position);
			dst.offset_radius = src.offset_radius;
			dst.direction = ld3(src.direction);
			dst.inv_radius = src.inv_radius;
		}
		for (int i = 0; i < 128; i++)
			s::cluster_transforms.type_mask[i] = a->type_mask[i];
		s::cluster_bitmask = const_cast<uint32_t *>(a->bitmask); // read-only in the shader; run-time sized arrays are plain pointers
		s::cluster_range = reinterpret_cast<uvec2 *>(const_cast<uint32_t *>(a->range));
		s::registers.inverse_view_projection_col2 = inv_vp.c[2];
		s::registers.camera_pos = ld3(a->rp->camera_position);
		s::registers.inv_resolution = inv_resolution;
		// rows of fragments on the host's cores: per-fragment state (gl_FragCoord, vClip, FragColor) is thread_local, the bound
		// resources are only read, a fragment blends into its own texel
#pragma omp parallel for schedule(dynamic, 4)
		for (int y = 0; y < H; y++)
			for (int x = 0; x < W; x++)
			{
				if (a->depth[size_t(y) * W + x] == 0.0f)
					continue;
				gl_FragCoord = vec4(float(x) + 0.5f, float(y) + 0.5f, 0.0f, 1.0f);
				s::vClip = clip_at(x, y);
				s::main();
				blend_one_one(a, x, y, s::FragColor);
			}
	}

	if (a->fog_falloff > 0.0f)
	{
		namespace s = fog_quad;
		s::Depth = make(a->depth, W, H, Format::R32F);
		s::registers.inverse_view_projection = inv_vp;
		s::registers.camera_pos = ld3(a->rp->camera_position);
		s::registers.color = ld3(a->fog_color);
		s::registers.falloff = a->fog_falloff;
#pragma omp parallel for schedule(dynamic, 4)
		for (int y = 0; y < H; y++)
			for (int x = 0; x < W; x++)
			{
				if (a->depth[size_t(y) * W + x] == 0.0f)
					continue; // the depth state of the lighting quads (NOT_EQUAL against z = 0) is still bound
				gl_FragCoord = vec4(float(x) + 0.5f, float(y) + 0.5f, 0.0f, 1.0f);
				s::vClip = clip_at(x, y); // fog.vert: inverse_view_projection * (Position, 0, 1), interpolated = evaluated at the centre
				s::main();
				blend_fog(a, x, y, s::FragColor);
			}
	}
}
