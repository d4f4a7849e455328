// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// HDR bloom + tonemap pass setup — same free functions, names and options as renderer/post/hdr.hpp:29-49.
#pragma once
#include <string>
#include "../render_graph.hpp"
#include "../render_context.hpp"
#include "../strip_plan.hpp"

namespace Granite
{
struct HDRDynamicExposureInterface
{
	virtual ~HDRDynamicExposureInterface() = default;
	virtual float get_exposure() const = 0;
};

struct HDROptions
{
	bool dynamic_exposure = true;
	// HIP executor extension: row-band tiling of the frame across devices (strip_plan.hpp).  nullptr or an inactive plan
	// = the whole frame on this device.  Honoured by setup_hdr_postprocess_compute only.
	const StripPlan *strip = nullptr;
	// HIP executor extension, a scheduling hint (GR_BLOOM_BUSY_FRAME_BIT): other heavy passes (a temporal resolve, SMAA) share the back of the frame.
	bool busy_frame = false;
};

// Ten separate passes (hdr.cpp:402-561).
void setup_hdr_postprocess(RenderGraph &graph, const FrameParameters &frame, const std::string &input, const std::string &output,
                           const HDROptions &options, const HDRDynamicExposureInterface *iface = nullptr);
// One "bloom-compute" pass recording the whole pyramid + a "tonemap" pass (hdr.cpp:308-400).
void setup_hdr_postprocess_compute(RenderGraph &graph, const FrameParameters &frame, const std::string &input, const std::string &output,
                                   const HDROptions &options, const HDRDynamicExposureInterface *iface = nullptr);

// ---- HDR10 output (renderer/post/hdr.hpp:51-60, hdr.cpp:562-658) -----------------------------------------------------------
struct HDR10PQEncodingConfig
{
	float hdr_pre_exposure;
	float ui_pre_exposure;
};

// The fields of VkHdrMetadataEXT the encoder reads: display primaries and white point (CIE xy) and the content light level.
struct HdrMetadata
{
	float display_primary_red[2] = {0.708f, 0.292f}; // ST.2020
	float display_primary_green[2] = {0.170f, 0.797f};
	float display_primary_blue[2] = {0.131f, 0.046f};
	float white_point[2] = {0.3127f, 0.3290f};
	float max_luminance = 1000.0f;
	float min_luminance = 0.001f;
	float max_content_light_level = 1000.0f;
	float max_frame_average_light_level = 500.0f;
};

// RGB -> XYZ matrix of a set of primaries (math/transforms.cpp:353-370), column major 3 x 3; and the rec.709 -> display
// conversion the encoder applies (hdr.cpp:580-593).
void compute_xyz_matrix(const float red[2], const float green[2], const float blue[2], const float white_point[2], float out9[9]);
void compute_rec709_to_st2020(const HdrMetadata &metadata, float out9[9]);

// Pass "pq10": colour output `output` (swapchain format, A2B10G10R10 on an HDR10 swapchain) from texture inputs `hdr_input`
// (linear scene colour) and `ui_input` (sRGB UI layer, alpha = how much of the scene shows through).
void setup_hdr10_pq_encoding(RenderGraph &graph, const std::string &output, const std::string &hdr_input, const std::string &ui_input,
                             const HDR10PQEncodingConfig &config, const HdrMetadata &static_metadata);
} // namespace Granite
