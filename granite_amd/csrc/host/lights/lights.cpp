// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
#include "lights.hpp"
#include <algorithm>
#include <limits>

namespace Granite
{
void PositionalLight::recompute_range()
{
	// Distance at which 1/d^2 attenuation of the brightest channel drops below 0.1.
	const float target_atten = 0.1f;
	float max_color = std::max(std::max(color.x, color.y), color.z);
	set_range(std::sqrt(max_color / target_atten));
}

void SpotLight::set_spot_parameters(float inner_cone_, float outer_cone_)
{
	inner_cone = std::min(std::max(inner_cone_, 0.001f), 1.0f);
	outer_cone = std::min(std::max(outer_cone_, 0.001f), 1.0f);
	recompute_range();
}

void SpotLight::set_range(float range)
{
	falloff_range = range;
	xy_range = std::sqrt(1.0f - outer_cone * outer_cone) / outer_cone; // tan(outer angle)
}

mat_affine SpotLight::build_model_matrix(const mat_affine &transform) const
{
	// transform * scale(xy_range * R, xy_range * R, R): the unit cone (apex at origin, base at z = -1, half-extent 1)
	// stretched to the light's reach.
	float max_range = std::min(falloff_range, cutoff_range);
	float sxy = xy_range * max_range;
	mat_affine res;
	for (int row = 0; row < 3; row++)
		res[row] = vec4(transform[row].x * sxy, transform[row].y * sxy, transform[row].z * max_range, transform[row].w);
	return res;
}

PositionalFragmentInfo SpotLight::get_shader_info(const mat_affine &transform) const
{
	// A uniformly scaled node scales the reach and (squared) the intensity.
	float scale_factor = transform.get_uniform_scale();
	float max_range = std::min(falloff_range, cutoff_range) * scale_factor;

	float spot_scale = 1.0f / std::max(0.001f, inner_cone - outer_cone);
	float spot_bias = -outer_cone * spot_scale;

	// Bounding sphere of the cone: centre at x = (tan^2 + 1) * R / 2 along the axis if that lies inside the cone,
	// otherwise the base disc's circumscribed sphere.
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
		spot_radius = std::sqrt(tan2) * max_range;
	}

	PositionalFragmentInfo info = {};
	vec3 c = color * (scale_factor * scale_factor);
	vec3 pos = transform.get_translation();
	vec3 dir = normalize(transform.get_forward());
	info.color[0] = c.x; info.color[1] = c.y; info.color[2] = c.z;
	info.spot_scale_bias = floatToHalf2(spot_scale, spot_bias);
	info.position[0] = pos.x; info.position[1] = pos.y; info.position[2] = pos.z;
	info.offset_radius = floatToHalf2(spot_offset, spot_radius);
	info.direction[0] = dir.x; info.direction[1] = dir.y; info.direction[2] = dir.z;
	info.inv_radius = 1.0f / max_range;
	return info;
}

PositionalFragmentInfo PointLight::get_shader_info(const mat_affine &transform) const
{
	float scale_factor = transform.get_uniform_scale();
	float max_range = std::min(falloff_range, cutoff_range) * scale_factor;

	PositionalFragmentInfo info = {};
	vec3 c = color * (scale_factor * scale_factor);
	vec3 pos = transform.get_translation();
	vec3 dir = transform.get_forward(); // unused by the point-light shading path
	info.color[0] = c.x; info.color[1] = c.y; info.color[2] = c.z;
	info.spot_scale_bias = 0;
	info.position[0] = pos.x; info.position[1] = pos.y; info.position[2] = pos.z;
	info.offset_radius = floatToHalf2(0.0f, max_range);
	info.direction[0] = dir.x; info.direction[1] = dir.y; info.direction[2] = dir.z;
	info.inv_radius = 1.0f / max_range;
	return info;
}

vec2 point_light_z_range(const RenderContext &context, const vec3 &center, float radius)
{
	return point_light_z_range(context.get_render_parameters(), center, radius);
}

vec2 spot_light_z_range(const RenderContext &context, const mat_affine &model)
{
	return spot_light_z_range(context.get_render_parameters(), model);
}

vec2 point_light_z_range(const RenderParameters &params, const vec3 &center, float radius)
{
	float z = dot(center - params.camera_position, params.camera_front);
	return vec2(z - radius, z + radius);
}

vec2 spot_light_z_range(const RenderParameters &params, const mat_affine &model)
{
	vec3 apex = model.get_translation();
	vec3 x_off = model.get_right(), y_off = model.get_up();
	vec3 base = apex + model.get_forward();
	const vec3 corners[5] = {apex, base + x_off + y_off, base - x_off + y_off, base + x_off - y_off, base - x_off - y_off};

	float lo = std::numeric_limits<float>::infinity();
	float hi = -lo;
	for (auto &p : corners)
	{
		float z = dot(p - params.camera_position, params.camera_front);
		lo = std::min(z, lo);
		hi = std::max(z, hi);
	}
	return vec2(lo, hi);
}
} // namespace Granite
