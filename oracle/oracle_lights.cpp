// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED by the reference's own tests.
//
// Clustered deferred lighting restated from:
//   host:   renderer/lights/lights.cpp:63-146,196-220,330-370 (light packing, z ranges),
//           renderer/lights/clusterer.cpp:646-703,781-827,1265-1346 (parameters, uint ranges),
//           renderer/threaded_scene.cpp:141-150 (front-to-back sort)
//   device: assets/shaders/lights/clusterer_bindless_{spot_transform,setup,binning,z_range}.comp,
//           clusterer_bindless{,_buffers}.h, clustering.{vert,frag}, directional.{vert,frag}, lighting.h,
//           point.h, spot.h, pbr.h; blend/depth state renderer/renderer.cpp:1004-1156.
#include "oracle_common.h"
#include <vector>
#include <numeric>
#include <limits>

using namespace orc;

namespace
{
// ---- byte layouts shared with the product through the C ABI -----------------------
struct LightInfo // PositionalFragmentInfo, renderer/lights/light_info.hpp:35-44 (48 B)
{
	float color[3];
	uint32_t spot_scale_bias; // 2 x fp16 (scale | bias << 16)
	float position[3];
	uint32_t offset_radius; // 2 x fp16
	float direction[3];
	float inv_radius;
};
static_assert(sizeof(LightInfo) == 48, "LightInfo");

struct Affine { float rows[3][4]; }; // mat_affine, 3 row vec4
static_assert(sizeof(Affine) == 48, "Affine");

struct LightDesc // scene-level description handed to both oracle and product by the harness
{
	int32_t type; // 0 = spot, 1 = point
	float color[3];
	float inner_cone, outer_cone; // cosines (spot only)
	float cutoff_range; // PositionalLight::set_maximum_range
	float pad;
	Affine transform; // world transform of the light node
};
static_assert(sizeof(LightDesc) == 80, "LightDesc");

struct ClusterParams // ClustererParametersBindless std140 (clusterer_data.h:20-39), 176 B
{
	float transform[16];
	float clip_scale[4];
	float camera_base[3]; float pad0;
	float camera_front[3]; float pad1;
	float xy_scale[2];
	int32_t resolution_xy[2];
	float inv_resolution_xy[2];
	int32_t num_lights, num_lights_32, num_decals, num_decals_32, decals_texture_offset, z_max_index;
	float z_scale;
	float pad2[3];
};
static_assert(sizeof(ClusterParams) == 176, "ClusterParams");

struct RenderParams // subset of Granite::RenderParameters (math/render_parameters.hpp:37-59), tightly packed floats
{
	float projection[16], view[16], view_projection[16], inv_projection[16], inv_view[16], inv_view_projection[16];
	float camera_position[3], camera_front[3];
	float z_near, z_far;
};
static_assert(sizeof(RenderParams) == (96 + 8) * 4, "RenderParams");

struct TransformedSpot { vec4 clip[5]; vec4 z; };
struct CullSetup { vec4 data[32]; };

static inline mat4 load_mat4(const float *m)
{
	mat4 r;
	for (int c = 0; c < 4; c++)
		r.c[c] = V4(m[4 * c + 0], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
	return r;
}
static inline vec3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
static inline uint32_t pack_half2_muglm(float a, float b)
{
	return uint32_t(float_to_half_muglm(a)) | (uint32_t(float_to_half_muglm(b)) << 16);
}
static inline vec2 unpack_half2(uint32_t v) { return V2(half_to_float(uint16_t(v & 0xffffu)), half_to_float(uint16_t(v >> 16))); }

static inline vec3 aff_translation(const Affine &m) { return V3(m.rows[0][3], m.rows[1][3], m.rows[2][3]); }
static inline vec3 aff_forward(const Affine &m) { return V3(-m.rows[0][2], -m.rows[1][2], -m.rows[2][2]); }
static inline vec3 aff_right(const Affine &m) { return V3(m.rows[0][0], m.rows[1][0], m.rows[2][0]); }
static inline vec3 aff_up(const Affine &m) { return V3(m.rows[0][1], m.rows[1][1], m.rows[2][1]); }
static inline float aff_uniform_scale(const Affine &m) { return length(V3(m.rows[0][0], m.rows[0][1], m.rows[0][2])); }

// PositionalLight::recompute_range (lights.cpp:63-70): falloff range where attenuation drops below 0.1.
static inline float falloff_range(const LightDesc &d)
{
	float max_color = std::max(std::max(d.color[0], d.color[1]), d.color[2]);
	return sqrtf(max_color / 0.1f);
}
} // namespace

extern "C" {

// Front-to-back sort (threaded_scene.cpp:141-150), pack PositionalFragmentInfo (lights.cpp:105-146,203-220),
// model matrices (lights.cpp:97-103, clusterer.cpp:636-650), type mask (clusterer.cpp:689), cap at 4096 (clusterer.cpp:672,685).
// Returns the number of packed lights.  order[i] receives the source index of packed light i.
int orc_pack_lights(const LightDesc *descs, int count, const float *camera_front, LightInfo *lights, Affine *model,
                    uint32_t *type_mask128, int32_t *order)
{
	vec3 front = ld3(camera_front);
	std::vector<int> idx(count);
	std::iota(idx.begin(), idx.end(), 0);
	std::vector<float> key(count);
	for (int i = 0; i < count; i++)
		key[i] = dot(aff_translation(descs[i].transform), front);
	std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[a] < key[b]; });

	memset(type_mask128, 0, 128 * sizeof(uint32_t));
	int n = std::min(count, 4096);
	for (int i = 0; i < n; i++)
	{
		const LightDesc &d = descs[idx[i]];
		order[i] = idx[i];
		LightInfo &o = lights[i];
		float scale_factor = aff_uniform_scale(d.transform);
		float max_range = std::min(falloff_range(d), d.cutoff_range) * scale_factor;
		vec3 pos = aff_translation(d.transform);
		float s2 = scale_factor * scale_factor;
		o.color[0] = d.color[0] * s2; o.color[1] = d.color[1] * s2; o.color[2] = d.color[2] * s2;
		o.position[0] = pos.x; o.position[1] = pos.y; o.position[2] = pos.z;
		o.inv_radius = 1.0f / max_range;
		if (d.type == 0)
		{
			float inner_cone = clampf(d.inner_cone, 0.001f, 1.0f);
			float outer_cone = clampf(d.outer_cone, 0.001f, 1.0f);
			float spot_scale = 1.0f / std::max(0.001f, inner_cone - outer_cone);
			float spot_bias = -outer_cone * spot_scale;
			float tan2 = (1.0f - outer_cone * outer_cone) / (outer_cone * outer_cone);
			float center_distance = ((tan2 + 1.0f) * max_range) * 0.5f;
			float spot_offset, spot_radius;
			if (center_distance < max_range)
			{
				spot_offset = center_distance;
				spot_radius = center_distance;
			}
			else
			{
				spot_offset = max_range;
				spot_radius = sqrtf(tan2) * max_range;
			}
			o.spot_scale_bias = pack_half2_muglm(spot_scale, spot_bias);
			o.offset_radius = pack_half2_muglm(spot_offset, spot_radius);
			vec3 dir = normalize(aff_forward(d.transform));
			o.direction[0] = dir.x; o.direction[1] = dir.y; o.direction[2] = dir.z;

			// SpotLight::build_model_matrix: transform * scale(xy_range*R, xy_range*R, R) with R = min(falloff, cutoff)
			// (no scale_factor here, lights.cpp:97-103); xy_range = sqrt(1-oc^2)/oc (lights.cpp:84-87).
			float r = std::min(falloff_range(d), d.cutoff_range);
			float xy_range = sqrtf(1.0f - outer_cone * outer_cone) / outer_cone;
			float sx = xy_range * r, sz = r;
			for (int row = 0; row < 3; row++)
			{
				model[i].rows[row][0] = d.transform.rows[row][0] * sx;
				model[i].rows[row][1] = d.transform.rows[row][1] * sx;
				model[i].rows[row][2] = d.transform.rows[row][2] * sz;
				model[i].rows[row][3] = d.transform.rows[row][3];
			}
		}
		else
		{
			o.spot_scale_bias = 0;
			o.offset_radius = pack_half2_muglm(0.0f, max_range);
			vec3 dir = aff_forward(d.transform);
			o.direction[0] = dir.x; o.direction[1] = dir.y; o.direction[2] = dir.z;
			memset(&model[i], 0, sizeof(Affine));
			model[i].rows[0][0] = pos.x; model[i].rows[0][1] = pos.y; model[i].rows[0][2] = pos.z;
			model[i].rows[0][3] = 1.0f / o.inv_radius;
			type_mask128[i >> 5] |= 1u << (i & 31);
		}
	}
	return n;
}

// LightClusterer::refresh_bindless_prepare (clusterer.cpp:803-826) + get_z_slice_extent (:700-703).
void orc_cluster_params(const RenderParams *rp, int res_x, int res_y, int res_z, int num_lights, ClusterParams *out)
{
	memset(out, 0, sizeof(*out));
	mat4 vp = load_mat4(rp->view_projection);
	// translate(0.5,0.5,0) * scale(0.5,0.5,1) * VP: row x' = 0.5*x + 0.5*w, y' = 0.5*y + 0.5*w.
	for (int c = 0; c < 4; c++)
	{
		vec4 col = vp.c[c];
		out->transform[4 * c + 0] = 0.5f * col.x + 0.5f * col.w;
		out->transform[4 * c + 1] = 0.5f * col.y + 0.5f * col.w;
		out->transform[4 * c + 2] = col.z;
		out->transform[4 * c + 3] = col.w;
	}
	out->clip_scale[0] = rp->projection[0];
	out->clip_scale[1] = -rp->projection[5];
	out->clip_scale[2] = rp->inv_projection[0];
	out->clip_scale[3] = -rp->inv_projection[5];
	for (int i = 0; i < 3; i++)
	{
		out->camera_base[i] = rp->camera_position[i];
		out->camera_front[i] = rp->camera_front[i];
	}
	out->xy_scale[0] = float(res_x); out->xy_scale[1] = float(res_y);
	out->resolution_xy[0] = res_x; out->resolution_xy[1] = res_y;
	out->inv_resolution_xy[0] = 1.0f / float(res_x); out->inv_resolution_xy[1] = 1.0f / float(res_y);
	out->num_lights = num_lights;
	out->num_lights_32 = (num_lights + 31) / 32;
	out->z_max_index = res_z - 1;
	float z_slice_size = std::min(0.5f, rp->z_far / float(res_z));
	out->z_scale = 1.0f / z_slice_size;
}

// update_bindless_range_buffer_gpu CPU half (clusterer.cpp:1322-1346) + compute_uint_range (:1265-1275) +
// point_light_z_range / spot_light_z_range (lights.cpp:330-370).
void orc_light_z_ranges(const RenderParams *rp, const LightInfo *lights, const Affine *model, const uint32_t *type_mask,
                        int num_lights, int res_z, uint32_t *ranges /* uvec2[n] */)
{
	vec3 cam = ld3(rp->camera_position), front = ld3(rp->camera_front);
	float extent = std::min(0.5f, rp->z_far / float(res_z));
	for (int i = 0; i < num_lights; i++)
	{
		vec2 range;
		bool point = (type_mask[i >> 5] >> (i & 31)) & 1u;
		if (point)
		{
			float radius = 1.0f / lights[i].inv_radius;
			float z = dot(ld3(lights[i].position) - cam, front);
			range = V2(z - radius, z + radius);
		}
		else
		{
			float lo = std::numeric_limits<float>::infinity(), hi = -lo;
			vec3 base = aff_translation(model[i]);
			vec3 xo = aff_right(model[i]), yo = aff_up(model[i]), zo = aff_forward(model[i]);
			vec3 zb = base + zo;
			vec3 wp[5] = {base, zb + xo + yo, zb - xo + yo, zb + xo - yo, zb - xo - yo};
			for (auto &p : wp)
			{
				float z = dot(p - cam, front);
				lo = std::min(z, lo);
				hi = std::max(z, hi);
			}
			range = V2(lo, hi);
		}
		range = V2(range.x / extent, range.y / extent);
		if (range.y < 0.0f)
		{
			ranges[2 * i] = 0xffffffffu;
			ranges[2 * i + 1] = 0u;
			continue;
		}
		range.x = std::max(range.x, 0.0f);
		uint32_t ux = uint32_t(range.x), uy = uint32_t(range.y);
		uy = std::min(uy, uint32_t(res_z - 1));
		ranges[2 * i] = ux;
		ranges[2 * i + 1] = uy;
	}
}

// clusterer_bindless_spot_transform.comp:38-73 (runs for every light index, points included — their rows[1..2] are zero).
void orc_cluster_spot_transform(const RenderParams *rp, const Affine *model, int num_lights, float *out /* 24 floats per light */)
{
	mat4 vp = load_mat4(rp->view_projection);
	vec3 cam = ld3(rp->camera_position), front = ld3(rp->camera_front);
	TransformedSpot *ts = reinterpret_cast<TransformedSpot *>(out);
	for (int index = 0; index < num_lights; index++)
	{
		const Affine &m = model[index];
		vec3 p[5];
		p[0] = aff_translation(m);
		vec3 pz = p[0] + aff_forward(m);
		vec3 right = aff_right(m), up = aff_up(m);
		p[1] = pz + right + up;
		p[2] = pz - right + up;
		p[3] = pz - right - up;
		p[4] = pz + right - up;
		float z[5];
		for (int i = 0; i < 5; i++)
			z[i] = dot(p[i] - cam, front);
		float z_lo = z[0], z_hi = z[0];
		for (int i = 1; i < 5; i++)
		{
			z_lo = std::min(z_lo, z[i]);
			z_hi = std::max(z_hi, z[i]);
		}
		float cull;
		if (z_lo <= rp->z_near && z_hi >= rp->z_far)
			cull = 0.0f;
		else if (z_lo <= rp->z_near)
			cull = -1.0f;
		else
			cull = 1.0f;
		for (int i = 0; i < 5; i++)
			ts[index].clip[i] = mul(vp, V4(p[i], 1.0f));
		ts[index].z = V4(cull, z_lo, z_hi, 0.0f);
	}
}
} // extern "C"

// ---- clusterer_bindless_setup.comp ---------------------------------------------------------------------------------------
namespace
{
struct mat3x2 { vec2 c[3]; };
struct mat3c { vec3 c[3]; };

static vec2 project_sphere_flat(float view_xy, float view_z, float radius)
{
	float len = length(V2(view_xy, view_z));
	float sin_xy = radius / len;
	vec2 result;
	if (sin_xy < 0.999f)
	{
		float cos_xy = sqrtf(1.0f - sin_xy * sin_xy);
		// mat2(c, s, -s, c) * v = (c,s)*v.x + (-s,c)*v.y
		vec2 rot_lo = V2(cos_xy * view_xy + (-sin_xy) * view_z, sin_xy * view_xy + cos_xy * view_z);
		vec2 rot_hi = V2(cos_xy * view_xy + sin_xy * view_z, (-sin_xy) * view_xy + cos_xy * view_z);
		if (rot_lo.y <= 0.0f)
			rot_lo = V2(-1.0f, 0.0f);
		if (rot_hi.y <= 0.0f)
			rot_hi = V2(+1.0f, 0.0f);
		result = V2(rot_lo.x / rot_lo.y, rot_hi.x / rot_hi.y);
	}
	else
		result = V2(-std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity());
	return result;
}

static inline vec3 mix3(vec3 a, vec3 b, float t) { return mix(a, b, t); }
static inline vec4 mix4(vec4 a, vec4 b, float t) { return mix(a, b, t); }
static inline vec3 xyz(vec4 v) { return V3(v.x, v.y, v.z); }
static inline vec2 xy(vec3 v) { return V2(v.x, v.y); }
static inline float cross_2d(vec2 a, vec2 b) { return a.x * b.y - a.y * b.x; }
static inline float signf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

static void clip_single_output(mat3x2 &clipped, vec3 c0, vec3 c1, vec3 c2, float target)
{
	float la = (target - c0.z) / (c2.z - c0.z);
	float lb = (target - c1.z) / (c2.z - c1.z);
	c0 = mix3(c0, c2, la);
	c1 = mix3(c1, c2, lb);
	clipped = {{xy(c0), xy(c1), xy(c2)}};
}
static void clip_dual_output(mat3x2 &clipped0, mat3x2 &clipped1, vec3 c0, vec3 c1, vec3 c2, float target)
{
	float l_ab = (target - c0.z) / (c1.z - c0.z);
	float l_ac = (target - c0.z) / (c2.z - c0.z);
	vec3 ab = mix3(c0, c1, l_ab);
	vec3 ac = mix3(c0, c2, l_ac);
	clipped0 = {{xy(ab), xy(c1), xy(ac)}};
	clipped1 = {{xy(ac), xy(c1), xy(c2)}};
}
static void clip_single_output(mat3c &clipped, vec4 c0, vec4 c1, vec4 c2, float target)
{
	float la = (target - c0.w) / (c2.w - c0.w);
	float lb = (target - c1.w) / (c2.w - c1.w);
	c0 = mix4(c0, c2, la);
	c1 = mix4(c1, c2, lb);
	clipped = {{xyz(c0) / target, xyz(c1) / target, xyz(c2) / c2.w}};
}
static void clip_dual_output(mat3c &clipped0, mat3c &clipped1, vec4 c0, vec4 c1, vec4 c2, float target)
{
	float l_ab = (target - c0.w) / (c1.w - c0.w);
	float l_ac = (target - c0.w) / (c2.w - c0.w);
	vec4 ab = mix4(c0, c1, l_ab);
	vec4 ac = mix4(c0, c2, l_ac);
	clipped0 = {{xyz(ab) / target, xyz(c1) / c1.w, xyz(ac) / target}};
	clipped1 = {{xyz(ac) / target, xyz(c1) / c1.w, xyz(c2) / c2.w}};
}

static void setup_triangle_2d(CullSetup &cs, uint32_t &num_triangles, const mat3x2 &tri, float cull)
{
	vec2 c0 = tri.c[0], c1 = tri.c[1], c2 = tri.c[2];
	vec2 ab = c1 - c0, bc = c2 - c1, ca = c0 - c2;
	float z = cross_2d(ab, -ca);
	if (fabsf(z) < 0.000001f || signf(cull) == signf(z))
		return;
	float inv_z = 1.0f / z;
	vec3 base = inv_z * V3(cross_2d(ab, -c0), cross_2d(bc, -c1), cross_2d(ca, -c2));
	vec3 dx = inv_z * V3(-ab.y, -bc.y, -ca.y);
	vec3 dy = inv_z * V3(ab.x, bc.x, ca.x);
	if (num_triangles < 8u)
	{
		cs.data[4u * num_triangles] = V4(base, 0.0f);
		cs.data[4u * num_triangles + 1u] = V4(dx, z);
		cs.data[4u * num_triangles + 2u] = V4(dy, inv_z);
		cs.data[4u * num_triangles + 3u] =
		    V4(std::min(std::min(c0.x, c1.x), c2.x), std::min(std::min(c0.y, c1.y), c2.y),
		       std::max(std::max(c0.x, c1.x), c2.x), std::max(std::max(c0.y, c1.y), c2.y));
	}
	num_triangles++;
}

static void setup_triangle_3d(CullSetup &cs, uint32_t &num_triangles, const mat3c &tri, float cull)
{
	vec3 c0 = tri.c[0], c1 = tri.c[1], c2 = tri.c[2];
	uint32_t clip_code = uint32_t(c0.z < 0.0f) + uint32_t(c1.z < 0.0f) * 2u + uint32_t(c2.z < 0.0f) * 4u;
	mat3x2 clipped0{}, clipped1{};
	bool dual = false;
	switch (clip_code)
	{
	case 0: clipped0 = {{xy(c0), xy(c1), xy(c2)}}; break;
	case 1: clip_dual_output(clipped0, clipped1, c0, c1, c2, 0.0f); dual = true; break;
	case 2: clip_dual_output(clipped0, clipped1, c1, c2, c0, 0.0f); dual = true; break;
	case 4: clip_dual_output(clipped0, clipped1, c2, c0, c1, 0.0f); dual = true; break;
	case 3: clip_single_output(clipped0, c0, c1, c2, 0.0f); break;
	case 5: clip_single_output(clipped0, c2, c0, c1, 0.0f); break;
	case 6: clip_single_output(clipped0, c1, c2, c0, 0.0f); break;
	case 7: return;
	}
	setup_triangle_2d(cs, num_triangles, clipped0, cull);
	if (dual)
		setup_triangle_2d(cs, num_triangles, clipped1, cull);
}

static void setup_triangle_4d(CullSetup &cs, uint32_t &num_triangles, vec4 c0, vec4 c1, vec4 c2, float cull)
{
	const float MIN_W = 1.0f / 1024.0f;
	uint32_t clip_code = uint32_t(c0.w < MIN_W) + uint32_t(c1.w < MIN_W) * 2u + uint32_t(c2.w < MIN_W) * 4u;
	mat3c clipped0{}, clipped1{};
	bool dual = false;
	switch (clip_code)
	{
	case 0: clipped0 = {{xyz(c0) / c0.w, xyz(c1) / c1.w, xyz(c2) / c2.w}}; break;
	case 1: clip_dual_output(clipped0, clipped1, c0, c1, c2, MIN_W); dual = true; break;
	case 2: clip_dual_output(clipped0, clipped1, c1, c2, c0, MIN_W); dual = true; break;
	case 4: clip_dual_output(clipped0, clipped1, c2, c0, c1, MIN_W); dual = true; break;
	case 3: clip_single_output(clipped0, c0, c1, c2, MIN_W); break;
	case 5: clip_single_output(clipped0, c2, c0, c1, MIN_W); break;
	case 6: clip_single_output(clipped0, c1, c2, c0, MIN_W); break;
	case 7: return;
	}
	setup_triangle_3d(cs, num_triangles, clipped0, cull);
	if (dual)
		setup_triangle_3d(cs, num_triangles, clipped1, cull);
}

// clusterer_bindless_binning.comp:41-92
static bool test_point_light(const ClusterParams &prm, const CullSetup &cs, vec2 uv, vec2 uv_stride)
{
	vec4 screen_bb = cs.data[0];
	vec4 tr = cs.data[1];
	vec4 ct = cs.data[2];
	vec4 eir = cs.data[3];
	if (eir.x != 0.0f)
	{
		vec2 center = 0.5f * (V2(tr.x, tr.z) + V2(tr.y, tr.w));
		vec2 clip_lo = uv, clip_hi = uv + uv_stride;
		vec2 cszw = V2(prm.clip_scale[2], prm.clip_scale[3]);
		clip_lo = clip_lo * cszw;
		clip_hi = clip_hi * cszw;
		auto xf = [&](float x, float y) {
			// mat2(ct.xy, ct.zw) * (x, y) = ct.xy * x + ct.zw * y
			return V2(ct.x * x + ct.z * y, ct.y * x + ct.w * y) - center;
		};
		vec2 d00 = xf(clip_lo.x, clip_lo.y), d01 = xf(clip_lo.x, clip_hi.y);
		vec2 d10 = xf(clip_hi.x, clip_lo.y), d11 = xf(clip_hi.x, clip_hi.y);
		vec2 inv_r = V2(eir.y, eir.z);
		d00 = d00 * inv_r; d01 = d01 * inv_r; d10 = d10 * inv_r; d11 = d11 * inv_r;
		float max_diag = std::max(distance(d00, d11), distance(d01, d10));
		float min_sq_dist = 1.0f + max_diag;
		min_sq_dist *= min_sq_dist;
		return dot(d00, d00) < min_sq_dist && dot(d01, d01) < min_sq_dist && dot(d10, d10) < min_sq_dist &&
		       dot(d11, d11) < min_sq_dist;
	}
	vec2 hi = uv + uv_stride;
	return hi.x > screen_bb.x && hi.y > screen_bb.y && uv.x < screen_bb.z && uv.y < screen_bb.w;
}

// clusterer_bindless_binning.comp:94-130
static bool test_spot_light(const CullSetup &cs, vec2 uv, vec2 uv_stride)
{
	uint32_t num_triangles = f2u(cs.data[0].w);
	if (num_triangles > 8u)
		return true;
	for (uint32_t i = 0; i < num_triangles; i++)
	{
		vec4 bb = cs.data[4u * i + 3u];
		vec2 hi = uv + uv_stride;
		if (hi.x > bb.x && hi.y > bb.y && uv.x < bb.z && uv.y < bb.w)
		{
			vec3 base = xyz(cs.data[4u * i]);
			vec3 dx = xyz(cs.data[4u * i + 1u]);
			vec3 dy = xyz(cs.data[4u * i + 2u]);
			base = base + dx * uv.x;
			base = base + dy * uv.y;
			base = base + V3(dx.x > 0.0f ? uv_stride.x * dx.x : 0.0f, dx.y > 0.0f ? uv_stride.x * dx.y : 0.0f,
			                 dx.z > 0.0f ? uv_stride.x * dx.z : 0.0f);
			base = base + V3(dy.x > 0.0f ? uv_stride.y * dy.x : 0.0f, dy.y > 0.0f ? uv_stride.y * dy.y : 0.0f,
			                 dy.z > 0.0f ? uv_stride.y * dy.z : 0.0f);
			if (base.x > 0.0f && base.y > 0.0f && base.z > 0.0f)
				return true;
		}
	}
	return false;
}
} // namespace

extern "C" {

// clusterer_bindless_setup.comp:252-322.  cull_setup must be zero-initialised by the caller (the graph zero-fills
// buffers on creation, render_graph.cpp:2586-2587); only the fields the shader writes are touched.
void orc_cluster_setup(const RenderParams *rp, const ClusterParams *prm, const LightInfo *lights, const uint32_t *type_mask,
                       const float *transformed_spots, int num_lights, float *cull_setup /* 128 floats per light */)
{
	mat4 view = load_mat4(rp->view);
	const TransformedSpot *ts = reinterpret_cast<const TransformedSpot *>(transformed_spots);
	CullSetup *cs = reinterpret_cast<CullSetup *>(cull_setup);
	for (int index = 0; index < num_lights; index++)
	{
		bool point = (type_mask[index >> 5] & (1u << (index & 31))) != 0u;
		if (point)
		{
			vec3 pos = ld3(lights[index].position);
			float radius = 1.0f / lights[index].inv_radius;
			vec4 v4 = mul(view, V4(pos, 1.0f));
			vec3 v = V3(v4.x, -v4.y, -v4.z);
			vec2 r0 = project_sphere_flat(v.x, v.z, radius);
			vec2 r1 = project_sphere_flat(v.y, v.z, radius);
			vec4 ranges = V4(r0.x, r0.y, r1.x, r1.y);
			float xy_length = length(V2(v.x, v.y));
			vec2 ct0, ct1; // columns
			if (xy_length < 0.00001f)
			{
				ct0 = V2(1.0f, 0.0f);
				ct1 = V2(0.0f, 1.0f);
			}
			else
			{
				float inv_xy_length = 1.0f / xy_length;
				ct0 = V2(v.x, -v.y) * inv_xy_length;
				ct1 = V2(v.y, v.x) * inv_xy_length;
			}
			vec2 txy = V2(ct0.x * v.x + ct1.x * v.y, ct0.y * v.x + ct1.y * v.y);
			vec2 t0 = project_sphere_flat(txy.x, v.z, radius);
			vec2 t1 = project_sphere_flat(txy.y, v.z, radius);
			vec4 tr = V4(t0.x, t0.y, t1.x, t1.y);
			bool ellipsis = !std::isinf(tr.x) && !std::isinf(tr.y) && !std::isinf(tr.z) && !std::isinf(tr.w);
			vec2 center = (V2(tr.x, tr.z) + V2(tr.y, tr.w)) * 0.5f;
			vec2 ellipse_radius = V2(tr.y, tr.w) - center;
			ranges = ranges * V4(prm->clip_scale[0], prm->clip_scale[0], prm->clip_scale[1], prm->clip_scale[1]);
			cs[index].data[0] = V4(ranges.x, ranges.z, ranges.y, ranges.w);
			cs[index].data[1] = tr;
			cs[index].data[2] = V4(ct0.x, ct0.y, ct1.x, ct1.y);
			cs[index].data[3] = V4(ellipsis ? 1.0f : 0.0f, 1.0f / ellipse_radius.x, 1.0f / ellipse_radius.y, 0.0f);
		}
		else
		{
			vec4 z = ts[index].z;
			if (z.x != 0.0f)
			{
				uint32_t n = 0;
				vec4 c0 = ts[index].clip[0], c1 = ts[index].clip[1], c2 = ts[index].clip[2], c3 = ts[index].clip[3],
				     c4 = ts[index].clip[4];
				setup_triangle_4d(cs[index], n, c0, c1, c2, z.x);
				setup_triangle_4d(cs[index], n, c0, c2, c3, z.x);
				setup_triangle_4d(cs[index], n, c0, c3, c4, z.x);
				setup_triangle_4d(cs[index], n, c0, c4, c1, z.x);
				setup_triangle_4d(cs[index], n, c2, c1, c3, z.x);
				setup_triangle_4d(cs[index], n, c4, c3, c1, z.x);
				cs[index].data[0].w = u2f(n);
			}
			else
				cs[index].data[0].w = u2f(0xffffffffu);
		}
	}
}

// clusterer_bindless_binning.comp:136-214.  subgroup_tile = 8 selects the SUBGROUPS path as it executes on a wave64
// device (tile 8x8 cells: coarse tile test ANDed with per-cell test, clusterer.cpp:1546-1552); 4 selects the wave32 tile
// (8x4); 0 selects the non-subgroup fallback (per-cell test only).  Lights >= num_lights never pass (equivalent to the
// zero-initialised cull-setup buffer).
void orc_cluster_binning(const ClusterParams *prm, const uint32_t *type_mask, const float *cull_setup, uint32_t *bitmask,
                         int subgroup_tile_h)
{
	const CullSetup *cs = reinterpret_cast<const CullSetup *>(cull_setup);
	int res_x = prm->resolution_xy[0], res_y = prm->resolution_xy[1];
	int n32 = prm->num_lights_32;
	vec2 inv_res = V2(prm->inv_resolution_xy[0], prm->inv_resolution_xy[1]);
	int tile_w = subgroup_tile_h ? 8 : 1, tile_h = subgroup_tile_h ? subgroup_tile_h : 1;
#pragma omp parallel for schedule(dynamic, 1)
	for (int ty = 0; ty < res_y / tile_h; ty++)
	{
		for (int tx = 0; tx < res_x / tile_w; tx++)
		{
			for (int chunk = 0; chunk < n32; chunk++)
			{
				uint32_t tmask = type_mask[chunk];
				uint32_t ballot = 0xffffffffu;
				if (subgroup_tile_h)
				{
					vec2 tile_uv = 2.0f * V2(float(tx * tile_w), float(ty * tile_h)) * inv_res - V2(1.0f, 1.0f);
					vec2 tile_stride = (2.0f * V2(float(tile_w), float(tile_h))) * inv_res;
					ballot = 0;
					for (int l = 0; l < 32; l++)
					{
						int li = 32 * chunk + l;
						if (li >= prm->num_lights)
							continue;
						bool passed = (tmask >> l) & 1u ? test_point_light(*prm, cs[li], tile_uv, tile_stride)
						                                : test_spot_light(cs[li], tile_uv, tile_stride);
						if (passed)
							ballot |= 1u << l;
					}
				}
				for (int py = 0; py < tile_h; py++)
				{
					for (int px = 0; px < tile_w; px++)
					{
						int cx = tx * tile_w + px, cy = ty * tile_h + py;
						vec2 uv = 2.0f * V2(float(cx), float(cy)) * inv_res - V2(1.0f, 1.0f);
						vec2 uv_stride = 2.0f * inv_res;
						uint32_t pixel_mask = 0;
						for (int l = 0; l < 32; l++)
						{
							if (!((ballot >> l) & 1u))
								continue;
							int li = 32 * chunk + l;
							if (li >= prm->num_lights)
								continue;
							bool passed = (tmask >> l) & 1u ? test_point_light(*prm, cs[li], uv, uv_stride)
							                                : test_spot_light(cs[li], uv, uv_stride);
							if (passed)
								pixel_mask |= 1u << l;
						}
						bitmask[size_t(cy * res_x + cx) * n32 + chunk] = pixel_mask;
					}
				}
			}
		}
	}
}

// clusterer_bindless_z_range.comp:22-50: first and last light index whose [lo,hi] slice interval contains z.
void orc_cluster_z_range(const uint32_t *light_ranges, int num_lights, int num_ranges, uint32_t *out)
{
	for (int zi = 0; zi < num_ranges; zi++)
	{
		uint32_t z = uint32_t(zi);
		uint32_t z_lo = 0xffffffffu, z_hi = 0u;
		for (int i = 0; i < num_lights; i++)
		{
			if (z >= light_ranges[2 * i] && z <= light_ranges[2 * i + 1])
			{
				z_lo = uint32_t(i);
				break;
			}
		}
		int z_lo_int = std::max(int(z_lo), 0);
		for (int i = num_lights - 1; i >= z_lo_int; i--)
		{
			if (z >= light_ranges[2 * i] && z <= light_ranges[2 * i + 1])
			{
				z_hi = uint32_t(i);
				break;
			}
		}
		out[2 * zi] = z_lo;
		out[2 * zi + 1] = z_hi;
	}
}
} // extern "C"

// ---- per-pixel shading ---------------------------------------------------------------------------------------------------
namespace
{
const float PI_SIC = 3.1415628f; // pbr.h:4-6 (sic)

struct Material { vec3 base; float ambient; vec3 N; float metallic, roughness; };

static inline float D_GGX(float roughness, vec3 N, vec3 H)
{
	float NoH = clampf(dot(N, H), 0.0001f, 1.0f);
	float m = roughness * roughness;
	float m2 = m * m;
	float d = (NoH * m2 - NoH) * NoH + 1.0f;
	return m2 / (PI_SIC * d * d);
}
static inline float G_schlick(float roughness, float NoV, float NoL)
{
	float r = roughness + 1.0f;
	float k = r * r * (1.0f / 8.0f);
	float V = NoV * (1.0f - k) + k;
	float L = NoL * (1.0f - k) + k;
	return 0.25f / std::max(V * L, 0.001f);
}
static inline vec3 fresnel(vec3 F0, float HoV) { return mix(F0, V3(1.0f), powf(1.0f - HoV, 5.0f)); }
static inline vec3 compute_F0(vec3 base, float metallic) { return mix(V3(0.04f), base, metallic); }

// Shared tail of compute_point_light / compute_spot_light / compute_lighting (point.h:119-142, spot.h:122-145, lighting.h:26-45).
static inline vec3 brdf(const Material &m, vec3 L, vec3 world_pos, vec3 camera_pos)
{
	float roughness = m.roughness * 0.75f + 0.25f;
	vec3 V = normalize(camera_pos - world_pos);
	vec3 H = normalize(V + L);
	vec3 N = m.N;
	float NoV = clampf(dot(N, V), 0.001f, 1.0f);
	float NoL = clampf(dot(N, L), 0.001f, 1.0f);
	float HoV = clampf(dot(H, V), 0.001f, 1.0f);
	vec3 F0 = compute_F0(m.base, m.metallic);
	vec3 specular_fresnel = fresnel(F0, HoV);
	vec3 spec = specular_fresnel * G_schlick(roughness, NoV, NoL) * D_GGX(roughness, N, H);
	vec3 specref = NoL * spec;
	vec3 diffref = NoL * (V3(1.0f) - specular_fresnel) * (1.0f / PI_SIC);
	vec3 diffuse_light = diffref * m.base * (1.0f - m.metallic);
	return specref + diffuse_light;
}

// compute_lighting (lighting.h:26-47) for the directional quad: the same terms as brdf(), but the light colour multiplies
// from the left -- light_color * NoL * shadow_term * (...) -- which rounds differently from colour * (NoL * (...)).  Found by
// running the reference's own directional.frag on the CPU (oracle/ref_build): one fp32 ulp, visible after the fp16 store in
// about 1 texel in 20 000.
static inline vec3 directional_lighting(const Material &m, vec3 light_color, vec3 L, vec3 world_pos, vec3 camera_pos)
{
	const float shadow_term = 1.0f;
	float roughness = m.roughness * 0.75f + 0.25f;
	vec3 V = normalize(camera_pos - world_pos);
	vec3 H = normalize(V + L);
	vec3 N = m.N;
	float NoV = clampf(dot(N, V), 0.001f, 1.0f);
	float NoL = clampf(dot(N, L), 0.001f, 1.0f);
	float HoV = clampf(dot(H, V), 0.001f, 1.0f);
	vec3 F0 = compute_F0(m.base, m.metallic);
	vec3 specular_fresnel = fresnel(F0, HoV);
	vec3 cook_torrance = specular_fresnel * G_schlick(roughness, NoV, NoL) * D_GGX(roughness, N, H);
	vec3 specref = light_color * NoL * shadow_term * cook_torrance;
	vec3 diffref = light_color * NoL * shadow_term * (V3(1.0f) - specular_fresnel) * (1.0f / PI_SIC);
	vec3 diffuse_light = diffref * m.base * (1.0f - m.metallic);
	return specref + diffuse_light;
}

// point.h:33-84 (no shadows)
static inline vec3 compute_point_light(const LightInfo &pt, const Material &m, vec3 world_pos, vec3 camera_pos)
{
	vec3 light_dir_full = world_pos - ld3(pt.position);
	vec3 light_dir = normalize(-light_dir_full);
	float light_dist = std::max(0.1f, length(light_dir_full));
	float static_falloff = 1.0f - smoothstep(0.9f, 1.0f, light_dist * pt.inv_radius);
	vec3 point_color = V3(0.0f);
	if (static_falloff > 0.0f)
		point_color = ld3(pt.color) * (1.0f * static_falloff) / (light_dist * light_dist);
	if (point_color.x == 0.0f && point_color.y == 0.0f && point_color.z == 0.0f)
		return V3(0.0f);
	return point_color * brdf(m, light_dir, world_pos, camera_pos);
}

// spot.h:34-93 (no shadows)
static inline vec3 compute_spot_light(const LightInfo &sp, const Material &m, vec3 world_pos, vec3 camera_pos)
{
	vec3 light_pos = ld3(sp.position);
	vec3 light_dir_full = light_pos - world_pos;
	vec3 light_dir = normalize(light_dir_full);
	float light_dist = std::max(0.1f, length(light_dir_full));
	float cone_angle = dot(normalize(world_pos - light_pos), ld3(sp.direction));
	vec2 sb = unpack_half2(sp.spot_scale_bias);
	float cone_falloff = clampf(cone_angle * sb.x + sb.y, 0.0f, 1.0f);
	cone_falloff *= cone_falloff;
	cone_falloff *= 1.0f - smoothstep(0.9f, 1.0f, light_dist * sp.inv_radius);
	vec3 spot_color = V3(0.0f);
	if (cone_falloff > 0.0f)
		spot_color = ld3(sp.color) * ((cone_falloff * 1.0f) / (light_dist * light_dist));
	if (spot_color.x == 0.0f && spot_color.y == 0.0f && spot_color.z == 0.0f)
		return spot_color;
	return spot_color * brdf(m, light_dir, world_pos, camera_pos);
}

// clusterer_bindless_buffers.h:17-27
static inline uint32_t cluster_mask_range(uint32_t mask, uint32_t rx, uint32_t ry, uint32_t start_index)
{
	rx = std::min(std::max(rx, start_index), start_index + 32u);
	ry = std::min(std::max(ry + 1u, rx), start_index + 32u);
	uint32_t num_bits = ry - rx;
	uint32_t range_mask = num_bits == 32u ? 0xffffffffu : ((1u << num_bits) - 1u) << (rx - start_index);
	return mask & range_mask;
}
} // namespace

extern "C" {

struct OrcLightingArgs
{
	int32_t width, height;
	const uint32_t *albedo;   // RGBA8 sRGB (+ linear alpha = ambient)
	const uint32_t *normal;   // A2B10G10R10 UNORM
	const uint16_t *pbr;      // RG8 UNORM (metallic, roughness)
	const float *depth;       // D32F, reverse-Z (0 = far plane)
	uint16_t *hdr;            // RGBA16F, read-modify-write: emissive in, HDR out
	const RenderParams *rp;
	const ClusterParams *cluster;
	const LightInfo *lights;
	const uint32_t *type_mask;
	const uint32_t *bitmask;
	const uint32_t *range;    // uvec2[res_z]
	float dir_color[3];
	float dir_direction[3];
	int32_t enable_directional;
	int32_t enable_clustered;
	int32_t ambient_fallback; // VOLUMETRIC_DIFFUSE_FALLBACK (renderer.cpp:1049-1055, directional.frag:62-64)
	int32_t wave_tile;        // 0/1: exact per-pixel light set; N>1: N x N pixel tile emulating the subgroup union (clusterer_bindless.h:49-56)
	const uint8_t *ambient_occlusion; // AMBIENT_OCCLUSION (renderer.cpp:1050-1051, directional.frag:52-64): R8_UNORM or NULL
	int32_t ao_width, ao_height;
	// 1: the target is a B10G11R11_UFLOAT_PACK32 attachment (renderTargetFp16 = false, scene_viewer_application.cpp:881-883).
	// `hdr` still holds RGBA16F texels -- every packed value is exactly a half float -- but each blend rounds to the packed format.
	int32_t hdr_b10g11r11;
	// The fog quad behind the clustered one (renderer.cpp:1179-1196, lights/fog.{vert,frag}, fog.h), when fog_falloff > 0:
	// src = (fog_color, exp2(-|pos - camera|^2 falloff)) blended ONE_MINUS_SRC_ALPHA / SRC_ALPHA, under the same depth test.
	float fog_color[3];
	float fog_falloff;
};

static inline void store_hdr(const OrcLightingArgs *a, int x, int y, vec4 v)
{
	if (a->hdr_b10g11r11)
		store_rgba16f_as_b10g11r11(a->hdr, a->width, x, y, v);
	else
		store_rgba16f(a->hdr, a->width, x, y, v);
}

// textureLod(uAmbientOcclusion, gl_FragCoord.xy * inv_resolution, 0).x with StockSampler::LinearClamp (renderer.cpp:611-612).
static float sample_ambient_occlusion(const OrcLightingArgs *a, float u, float v)
{
	const int w = a->ao_width, h = a->ao_height;
	float wx, wy;
	int x0, y0;
	linear_axis(u * float(w) - 0.5f, x0, wx);
	linear_axis(v * float(h) - 0.5f, y0, wy);
	auto texel = [&](int x, int y) { return float(a->ambient_occlusion[size_t(clampi(y, 0, h - 1)) * w + clampi(x, 0, w - 1)]) / 255.0f; };
	return linear_combine(texel(x0, y0), texel(x0 + 1, y0), texel(x0, y0 + 1), texel(x0 + 1, y0 + 1), wx, wy);
}

// DeferredLightRenderer::render_light (renderer.cpp:1004-1156): directional quad then clustered quad, each blended
// ONE/ONE into the RGBA16F target (two separate fp16 roundings), depth test NOT_EQUAL against the quad at z = 0 so
// reverse-Z far-plane pixels are untouched.  Alpha is left as-is (shader outputs vec3).
void orc_lighting(const OrcLightingArgs *a)
{
	const int W = a->width, H = a->height;
	mat4 inv_vp = load_mat4(a->rp->inv_view_projection);
	vec3 camera_pos = ld3(a->rp->camera_position);
	vec3 camera_front = ld3(a->rp->camera_front);
	(void)camera_front;
	const ClusterParams &cl = *a->cluster;
	vec2 inv_resolution = V2(1.0f / float(W), 1.0f / float(H));
	vec3 cl_base = ld3(cl.camera_base), cl_front = ld3(cl.camera_front);
	const int tile = a->wave_tile > 1 ? a->wave_tile : 1;
	const int tiles_x = (W + tile - 1) / tile, tiles_y = (H + tile - 1) / tile;

#pragma omp parallel for schedule(dynamic, 4)
	for (int t = 0; t < tiles_x * tiles_y; t++)
	{
		int tx = t % tiles_x, ty = t / tiles_x;
		int x0 = tx * tile, y0 = ty * tile;
		int x1 = std::min(x0 + tile, W), y1 = std::min(y0 + tile, H);
		const int maxpix = tile * tile;
		std::vector<Material> mats(maxpix);
		std::vector<vec3> poss(maxpix);
		std::vector<uint8_t> act(maxpix, 0);
		std::vector<int> cbase(maxpix);
		std::vector<uint32_t> zr(maxpix * 2);
		uint32_t zmin = 0xffffffffu, zmax = 0;

		for (int y = y0; y < y1; y++)
			for (int x = x0; x < x1; x++)
			{
				int li = (y - y0) * tile + (x - x0);
				size_t p = size_t(y) * W + x;
				float depth = a->depth[p];
				if (depth == 0.0f)
					continue; // NOT_EQUAL vs quad z = 0
				act[li] = 1;
				uint32_t alb = a->albedo[p];
				Material &m = mats[li];
				m.base = V3(srgb8_to_float(alb & 255u), srgb8_to_float((alb >> 8) & 255u), srgb8_to_float((alb >> 16) & 255u));
				m.ambient = float(alb >> 24) / 255.0f;
				vec4 n = unpack_a2b10g10r10(a->normal[p]);
				m.N = V3(n.x, n.y, n.z) * 2.0f - V3(1.0f);
				uint16_t mr = a->pbr[p];
				m.metallic = float(mr & 255u) / 255.0f;
				m.roughness = float(mr >> 8) / 255.0f;

				// vClip = invVP * (ndc.xy, 0, 1) interpolated from the full-screen triangle (clustering.vert:10-13);
				// ndc = 2*(pixel+0.5)/size - 1.
				vec2 ndc = V2(2.0f * ((float(x) + 0.5f) * inv_resolution.x) - 1.0f, 2.0f * ((float(y) + 0.5f) * inv_resolution.y) - 1.0f);
				vec4 vclip = mul(inv_vp, V4(ndc.x, ndc.y, 0.0f, 1.0f));
				vec4 clip = vclip + depth * inv_vp.c[2];
				poss[li] = V3(clip.x / clip.w, clip.y / clip.w, clip.z / clip.w);

				// clusterer_bindless.h:39-47
				vec2 fc = V2(float(x) + 0.5f, float(y) + 0.5f);
				int ccx = int(fc.x * inv_resolution.x * cl.xy_scale[0]);
				int ccy = int(fc.y * inv_resolution.y * cl.xy_scale[1]);
				ccx = clampi(ccx, 0, cl.resolution_xy[0] - 1);
				ccy = clampi(ccy, 0, cl.resolution_xy[1] - 1);
				cbase[li] = (ccy * cl.resolution_xy[0] + ccx) * cl.num_lights_32;
				float z = dot(poss[li] - cl_base, cl_front);
				int z_index = clampi(int(z * cl.z_scale), 0, cl.z_max_index);
				zr[2 * li] = a->range[2 * z_index];
				zr[2 * li + 1] = a->range[2 * z_index + 1];
				zmin = std::min(zmin, zr[2 * li]);
				zmax = std::max(zmax, zr[2 * li + 1]);
			}

		// Directional pass (directional.frag:41-65, lighting.h:9-47, LIGHTING_NO_AMBIENT) -> blend, round to fp16.
		if (a->enable_directional)
		{
			vec3 dcol = ld3(a->dir_color), ddir = ld3(a->dir_direction);
			for (int y = y0; y < y1; y++)
				for (int x = x0; x < x1; x++)
				{
					int li = (y - y0) * tile + (x - x0);
					if (!act[li])
						continue;
					const Material &m = mats[li];
					// light_color * NoL * shadow_term * (...) — light_color multiplies from the left in lighting.h:41-42.
					vec3 lit = directional_lighting(m, dcol, ddir, poss[li], camera_pos);
					if (a->ambient_fallback)
					{
						const float base_ambient = a->ambient_occlusion
						                               ? sample_ambient_occlusion(a, (float(x) + 0.5f) * inv_resolution.x, (float(y) + 0.5f) * inv_resolution.y)
						                               : 1.0f;
						lit = lit + base_ambient * m.base * V3(0.05f);
					}
					vec4 dst = load_rgba16f(a->hdr, W, x, y);
					store_hdr(a, x, y, V4(dst.x + lit.x, dst.y + lit.y, dst.z + lit.z, dst.w));
				}
		}

		// Clustered pass (clustering.frag:29-43, clusterer_bindless.h:29-84).
		if (a->enable_clustered && cl.num_lights > 0)
		{
			std::vector<vec3> acc(maxpix, V3(0.0f));
			bool any = false;
			for (int li = 0; li < maxpix; li++)
				any = any || act[li];
			if (any)
			{
				int z_start = int(zmin >> 5), z_end = int(zmax >> 5);
				for (int i = z_start; i <= z_end && i < cl.num_lights_32; i++)
				{
					// per-lane trimmed mask, then union over the emulated subgroup
					uint32_t uni = 0;
					std::vector<uint32_t> lane_mask(maxpix, 0);
					for (int li = 0; li < maxpix; li++)
					{
						if (!act[li])
							continue;
						uint32_t mask = a->bitmask[cbase[li] + i];
						mask = cluster_mask_range(mask, zr[2 * li], zr[2 * li + 1], 32u * uint32_t(i));
						lane_mask[li] = mask;
						uni |= mask;
					}
					uint32_t tmask = a->type_mask[i];
					for (int li = 0; li < maxpix; li++)
					{
						if (!act[li])
							continue;
						uint32_t mask = tile > 1 ? uni : lane_mask[li];
						while (mask)
						{
							int bit = __builtin_ctz(mask);
							int index = 32 * i + bit;
							if ((tmask >> bit) & 1u)
								acc[li] += compute_point_light(a->lights[index], mats[li], poss[li], camera_pos);
							else
								acc[li] += compute_spot_light(a->lights[index], mats[li], poss[li], camera_pos);
							mask &= mask - 1u;
						}
					}
				}
			}
			for (int y = y0; y < y1; y++)
				for (int x = x0; x < x1; x++)
				{
					int li = (y - y0) * tile + (x - x0);
					if (!act[li])
						continue;
					vec4 dst = load_rgba16f(a->hdr, W, x, y);
					store_hdr(a, x, y, V4(dst.x + acc[li].x, dst.y + acc[li].y, dst.z + acc[li].z, dst.w));
				}
		}

		// Fog quad (fog.frag:17-25, fog.h:4-8): FragColor = (color, fog_factor); the blend unit computes, per channel and for alpha,
		// src * (1 - src.a) + dst * src.a and the attachment store rounds once more.
		if (a->fog_falloff > 0.0f)
		{
			const vec3 fog = ld3(a->fog_color);
			for (int y = y0; y < y1; y++)
				for (int x = x0; x < x1; x++)
				{
					int li = (y - y0) * tile + (x - x0);
					if (!act[li])
						continue;
					const vec3 eye_vec = poss[li] - camera_pos;
					const float distance = dot(eye_vec, eye_vec);
					const float f = exp2f(-distance * a->fog_falloff);
					const vec4 dst = load_rgba16f(a->hdr, W, x, y);
					store_hdr(a, x, y, V4(fog.x * (1.0f - f) + dst.x * f, fog.y * (1.0f - f) + dst.y * f, fog.z * (1.0f - f) + dst.z * f,
					                      f * (1.0f - f) + dst.w * f));
				}
		}
	}
}

// Brute force: every light evaluated for every pixel in index order (no clustering).  Used to show the clustered result
// equals the unclustered sum (conservative culling), i.e. that the subgroup footprint cannot matter.
void orc_lighting_bruteforce_clustered(const OrcLightingArgs *a)
{
	const int W = a->width, H = a->height;
	mat4 inv_vp = load_mat4(a->rp->inv_view_projection);
	vec3 camera_pos = ld3(a->rp->camera_position);
	vec2 inv_resolution = V2(1.0f / float(W), 1.0f / float(H));
#pragma omp parallel for schedule(dynamic, 1)
	for (int y = 0; y < H; y++)
		for (int x = 0; x < W; x++)
		{
			size_t p = size_t(y) * W + x;
			float depth = a->depth[p];
			if (depth == 0.0f)
				continue;
			uint32_t alb = a->albedo[p];
			Material m;
			m.base = V3(srgb8_to_float(alb & 255u), srgb8_to_float((alb >> 8) & 255u), srgb8_to_float((alb >> 16) & 255u));
			m.ambient = float(alb >> 24) / 255.0f;
			vec4 n = unpack_a2b10g10r10(a->normal[p]);
			m.N = V3(n.x, n.y, n.z) * 2.0f - V3(1.0f);
			uint16_t mr = a->pbr[p];
			m.metallic = float(mr & 255u) / 255.0f;
			m.roughness = float(mr >> 8) / 255.0f;
			vec2 ndc = V2(2.0f * ((float(x) + 0.5f) * inv_resolution.x) - 1.0f, 2.0f * ((float(y) + 0.5f) * inv_resolution.y) - 1.0f);
			vec4 vclip = mul(inv_vp, V4(ndc.x, ndc.y, 0.0f, 1.0f));
			vec4 clip = vclip + depth * inv_vp.c[2];
			vec3 pos = V3(clip.x / clip.w, clip.y / clip.w, clip.z / clip.w);
			vec3 acc = V3(0.0f);
			for (int index = 0; index < a->cluster->num_lights; index++)
			{
				if ((a->type_mask[index >> 5] >> (index & 31)) & 1u)
					acc += compute_point_light(a->lights[index], m, pos, camera_pos);
				else
					acc += compute_spot_light(a->lights[index], m, pos, camera_pos);
			}
			vec4 dst = load_rgba16f(a->hdr, W, x, y);
			store_hdr(a, x, y, V4(dst.x + acc.x, dst.y + acc.y, dst.z + acc.z, dst.w));
		}
}
}
