// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
#include "renderer.hpp"
#include "lights/clusterer.hpp"
#include <cstring>

namespace Granite
{
void DeferredLightRenderer::render_light(HIP::CommandBuffer &cmd, const RenderContext &context, const DeferredLightAttachments &att,
                                         RendererOptionFlags options)
{
	auto *light = context.get_lighting_parameters();
	if (!light)
		throw std::logic_error("render_light needs lighting parameters on the context.");
	auto &rp = context.get_render_parameters();

	gr_lighting_args args = {};
	args.albedo = att.base_color->get_view();
	args.normal = att.normal->get_view();
	args.pbr = att.pbr->get_view();
	args.depth = att.depth->get_view();
	args.emissive = att.emissive->get_view();
	args.hdr = att.hdr->get_view();
	memcpy(args.inv_view_projection, rp.inv_view_projection.data(), sizeof(args.inv_view_projection));

	const float inv_w = 1.0f / float(att.hdr->get_width());
	const float inv_h = 1.0f / float(att.hdr->get_height());

	// DirectionalLightPush (renderer.cpp:1073-1103)
	auto &d = args.directional;
	for (int i = 0; i < 4; i++)
		d.inv_view_proj_col2[i] = rp.inv_view_projection[2][i];
	for (int i = 0; i < 3; i++)
	{
		d.color[i] = light->directional.color[i];
		d.direction[i] = light->directional.direction[i];
		d.camera_pos[i] = rp.camera_position[i];
		d.camera_front[i] = rp.camera_front[i];
	}
	d.inv_resolution[0] = inv_w;
	d.inv_resolution[1] = inv_h;

	// The directional quad is always drawn; VOLUMETRIC_DIFFUSE_FALLBACK is defined whenever the clusterer has no
	// volumetric diffuse (renderer.cpp:1049-1055), which is always the case on this path.
	args.flags = GR_LIGHTING_DIRECTIONAL_BIT | ((options & SHARE_REGISTERS_BIT) ? GR_LIGHTING_SHARE_REGISTERS_BIT : 0u);
	bool cluster_volumetric_diffuse = light->cluster && light->cluster->clusterer_has_volumetric_diffuse();
	if (!cluster_volumetric_diffuse)
	{
		args.flags |= GR_LIGHTING_AMBIENT_FALLBACK_BIT;
		if (light->ambient_occlusion) // AMBIENT_OCCLUSION define + BINDING_GLOBAL_AMBIENT_OCCLUSION (renderer.cpp:611-612,1050-1051)
		{
			args.flags |= GR_LIGHTING_AMBIENT_OCCLUSION_BIT;
			args.ambient_occlusion = light->ambient_occlusion->get_view();
		}
	}

	// Clustered lighting (renderer.cpp:1107-1156)
	if (light->cluster && light->cluster->get_cluster_bitmask_buffer())
	{
		auto &c = args.clustering;
		for (int i = 0; i < 4; i++)
			c.inv_view_proj_col2[i] = rp.inv_view_projection[2][i];
		for (int i = 0; i < 3; i++)
			c.camera_pos[i] = rp.camera_position[i];
		c.inv_resolution[0] = inv_w;
		c.inv_resolution[1] = inv_h;
		// set_cluster_parameters_bindless (renderer.cpp:476-481)
		args.cluster = light->cluster->get_cluster_parameters_bindless();
		args.transforms = light->cluster->get_cluster_transform_buffer()->get_device_pointer();
		args.bitmask = static_cast<const uint32_t *>(light->cluster->get_cluster_bitmask_buffer()->get_device_pointer());
		args.range = static_cast<const uint32_t *>(light->cluster->get_cluster_range_buffer()->get_device_pointer());
		args.flags |= GR_LIGHTING_CLUSTERED_BIT;
	}
	else
		args.clustering = {};
	if (!(args.flags & GR_LIGHTING_CLUSTERED_BIT))
	{
		for (int i = 0; i < 3; i++)
			args.clustering.camera_pos[i] = rp.camera_position[i];
		args.clustering.inv_resolution[0] = inv_w;
		args.clustering.inv_resolution[1] = inv_h;
	}

	// renderer.cpp:1179-1196: the fog quad when light.fog.falloff > 0 (volumetric fog is outside the path)
	if (light && light->fog.falloff > 0.0f)
	{
		for (int i = 0; i < 3; i++)
			args.fog_color[i] = light->fog.color[i];
		args.fog_falloff = light->fog.falloff;
	}

	if (att.rows && !att.rows->whole)
	{
		if (att.rows->count == 0)
			return; // empty band on this rank
		args.rows.first = att.rows->first;
		args.rows.count = att.rows->count;
	}
	cmd.check(gr_lighting(cmd.get_context(), cmd.get_stream(), &args), "lighting");
}

void RenderContext::set_camera(const mat4 &projection, const mat4 &view)
{
	camera.projection = projection;
	camera.view = view;
	camera.view_projection = projection * view;
	camera.inv_projection = inverse(projection);
	camera.inv_view = inverse(view);
	camera.inv_view_projection = inverse(camera.view_projection);
	camera.unjittered_view_projection = camera.view_projection;
	camera.unjittered_inv_view_projection = camera.inv_view_projection;
	camera.camera_position = camera.inv_view[3].xyz();
	camera.camera_up = camera.inv_view[1].xyz();
	camera.camera_right = camera.inv_view[0].xyz();
	camera.camera_front = -camera.inv_view[2].xyz();

	// z_near / z_far recovered from the inverse projection exactly as render_context.cpp:76-85 does.
	const vec4 &c2 = camera.inv_projection[2];
	const vec4 &c3 = camera.inv_projection[3];
	auto project = [&](float z, float w) {
		float vz = c2.z * z + c3.z * w;
		float vw = c2.w * z + c3.w * w;
		return -vz / vw;
	};
	bool infinite_z = camera.inv_view_projection[3][3] == 0.0f;
	camera.z_near = project(1.0f, 1.0f);
	camera.z_far = project(infinite_z ? 1e-10f : 0.0f, 1.0f);
}
} // namespace Granite
