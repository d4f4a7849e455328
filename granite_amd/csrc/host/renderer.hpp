// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// DeferredLightRenderer — renderer/renderer.hpp's static render_light, the lighting half of the deferred renderer.
#pragma once
#include "render_context.hpp"
#include "render_graph.hpp"
#include "strip_plan.hpp"

namespace Granite
{
// What the reference binds implicitly through the current render pass: input attachments 0..3 via
// cmd.set_input_attachments(3, 0) (renderer.cpp:1008) and colour attachment 0 as the blend target.
struct DeferredLightAttachments
{
	const HIP::ImageView *base_color = nullptr; // albedo-main, R8G8B8A8_SRGB
	const HIP::ImageView *normal = nullptr;     // normal-main, A2B10G10R10
	const HIP::ImageView *pbr = nullptr;        // pbr-main, R8G8
	const HIP::ImageView *depth = nullptr;      // depth-main, D32F
	const HIP::ImageView *emissive = nullptr;   // blend destination contents before the draws
	HIP::ImageView *hdr = nullptr;              // HDR-main; may be the same image as emissive (reference RMW)
	const RowRange *rows = nullptr;             // render area (row-band tiling); nullptr = whole target
};

class DeferredLightRenderer
{
public:
	using RendererOptionFlags = uint32_t;
	// Scheduling hint (GR_LIGHTING_SHARE_REGISTERS_BIT): register-heavy passes of other streams (TAA resolve, SMAA) run beside the lighting launch.
	static constexpr RendererOptionFlags SHARE_REGISTERS_BIT = 1u;
	// renderer.cpp:1004-1197: directional quad, clustered quad, and the fog quad when LightingParameters::fog.falloff > 0 (the volumetric-fog quad is outside the path).
	static void render_light(HIP::CommandBuffer &cmd, const RenderContext &context, const DeferredLightAttachments &attachments,
	                         RendererOptionFlags flags = 0);
};
} // namespace Granite
