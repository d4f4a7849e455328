// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// PositionalLight / SpotLight / PointLight — the CPU-side light objects of renderer/lights/lights.{hpp,cpp} reduced to
// what feeds the clusterer: colour, cones, ranges, PositionalFragmentInfo packing, model matrices and view-Z ranges.
#pragma once
#include "../math.hpp"
#include "../render_context.hpp"
#include "../../../../include/granite_hip.h"

namespace Granite
{
using PositionalFragmentInfo = gr_light_info; // renderer/lights/light_info.hpp:35-44, 48 bytes
static_assert(sizeof(PositionalFragmentInfo) == 48, "PositionFragmentInfo is not 48 bytes.");

class PositionalLight
{
public:
	enum class Type { Spot, Point };
	explicit PositionalLight(Type type_) : type(type_) {}
	virtual ~PositionalLight() = default;
	Type get_type() const { return type; }

	void set_color(vec3 color_)
	{
		color = color_;
		recompute_range();
	}
	const vec3 &get_color() const { return color; }
	void set_maximum_range(float range)
	{
		cutoff_range = range;
		recompute_range();
	}
	float get_falloff_range() const { return falloff_range; }
	float get_cutoff_range() const { return cutoff_range; }

protected:
	vec3 color = vec3(1.0f);
	float falloff_range = 1.0f;
	float cutoff_range = 100.0f;
	void recompute_range(); // lights.cpp:63-70
	virtual void set_range(float range) = 0;

private:
	Type type;
};

class SpotLight : public PositionalLight
{
public:
	SpotLight() : PositionalLight(Type::Spot) {}
	void set_spot_parameters(float inner_cone, float outer_cone); // lights.cpp:72-77
	PositionalFragmentInfo get_shader_info(const mat_affine &transform) const; // lights.cpp:105-146
	mat_affine build_model_matrix(const mat_affine &transform) const;          // lights.cpp:97-103
	float get_inner_cone() const { return inner_cone; }
	float get_outer_cone() const { return outer_cone; }

private:
	float inner_cone = 0.4f;
	float outer_cone = 0.45f;
	float xy_range = 0.0f;
	void set_range(float range) override; // lights.cpp:79-89
};

class PointLight : public PositionalLight
{
public:
	PointLight() : PositionalLight(Type::Point) {}
	PositionalFragmentInfo get_shader_info(const mat_affine &transform) const; // lights.cpp:203-220

private:
	void set_range(float range) override { falloff_range = range; }
};

vec2 point_light_z_range(const RenderContext &context, const vec3 &center, float radius); // lights.cpp:330-337
vec2 spot_light_z_range(const RenderContext &context, const mat_affine &model);           // lights.cpp:339-370
// the same on bare render parameters (what the clusterer's refresh works from, on whichever thread it runs)
vec2 point_light_z_range(const RenderParameters &params, const vec3 &center, float radius);
vec2 spot_light_z_range(const RenderParameters &params, const mat_affine &model);
} // namespace Granite
