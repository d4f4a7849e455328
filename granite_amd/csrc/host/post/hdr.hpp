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
};

// Ten separate passes (hdr.cpp:402-561).
void setup_hdr_postprocess(RenderGraph &graph, const FrameParameters &frame, const std::string &input, const std::string &output,
                           const HDROptions &options, const HDRDynamicExposureInterface *iface = nullptr);
// One "bloom-compute" pass recording the whole pyramid + a "tonemap" pass (hdr.cpp:308-400).
void setup_hdr_postprocess_compute(RenderGraph &graph, const FrameParameters &frame, const std::string &input, const std::string &output,
                                   const HDROptions &options, const HDRDynamicExposureInterface *iface = nullptr);
} // namespace Granite
