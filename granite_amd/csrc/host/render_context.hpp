// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// FrameParameters / RenderParameters / RenderContext / LightingParameters — the slices of
// renderer/render_context.{hpp,cpp} and math/render_parameters.hpp:37-59,155-162 the image-space passes read.
#pragma once
#include "hip_device.hpp"
#include "math.hpp"

namespace Granite
{
class LightClusterer;

struct FrameParameters
{
	double frame_time = 0.0;
	double elapsed_time = 0.0;
	bool discontinuous_camera = false;
};

struct RenderParameters
{
	mat4 projection;
	mat4 view;
	mat4 view_projection;
	mat4 inv_projection;
	mat4 inv_view;
	mat4 inv_view_projection;
	mat4 unjittered_view_projection;
	mat4 unjittered_inv_view_projection;
	mat4 unjittered_prev_view_projection;
	vec3 camera_position;
	vec3 camera_front;
	vec3 camera_right;
	vec3 camera_up;
	float z_near = 0.0f;
	float z_far = 0.0f;
};

struct DirectionalParameters
{
	vec3 color = vec3(0.0f);
	vec3 direction = vec3(0.0f, 1.0f, 0.0f);
};

struct FogParameters
{
	vec3 color = vec3(0.0f);
	float falloff = 0.0f;
};

struct LightingParameters
{
	FogParameters fog;
	DirectionalParameters directional;
	LightClusterer *cluster = nullptr;
	// LightingParameters::ambient_occlusion (renderer/lights/lights.hpp): the SSAO result the deferred lighting pass
	// samples, or null.  Set every frame from the graph's physical resource (scene_viewer_application.cpp:1571).
	const HIP::ImageView *ambient_occlusion = nullptr;
};

class RenderContext
{
public:
	// RenderContext::set_camera (render_context.cpp:53-86): inverses, camera basis, z_near / z_far from inv_projection.
	void set_camera(const mat4 &projection, const mat4 &view);
	// Harness hook: install precomputed parameters verbatim (keeps CPU oracle and device bit-identical inputs).
	void set_render_parameters(const RenderParameters &params) { camera = params; }
	const RenderParameters &get_render_parameters() const { return camera; }

	void set_frame_parameters(const FrameParameters &frame_) { frame = frame_; }
	const FrameParameters &get_frame_parameters() const { return frame; }

	void set_lighting_parameters(const LightingParameters *lighting_) { lighting = lighting_; }
	const LightingParameters *get_lighting_parameters() const { return lighting; }

private:
	RenderParameters camera;
	FrameParameters frame;
	const LightingParameters *lighting = nullptr;
};
} // namespace Granite
