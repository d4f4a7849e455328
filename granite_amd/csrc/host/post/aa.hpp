// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// Anti-aliasing pass setup: renderer/post/{aa,fxaa,smaa,temporal}.hpp restated on the HIP executor.
// Live variants only: FXAA, SMAA 1x Low..Ultra, TAA Low/Medium/High.  FXAA-2phase and SMAA-T2X do not
// compile in the reference (missing GLSL functions), FSR2 needs an absent third_party module.
#pragma once
#include <string>
#include <vector>
#include "../render_context.hpp"
#include "../render_graph.hpp"
#include "../strip_plan.hpp"

namespace Granite
{
enum class PostAAType
{
	FXAA,
	FXAA_2Phase,
	SMAA_Low,
	SMAA_Medium,
	SMAA_High,
	SMAA_Ultra,
	SMAA_Ultra_T2X,
	TAA_Low,
	TAA_Medium,
	TAA_High,
	TAA_FSR2,
	None
};

enum class SMAAPreset { Low, Medium, High, Ultra, Ultra_T2X };
enum class TAAQuality { Low, Medium, High };

// renderer/post/temporal.{hpp,cpp}:40-197 — sub-pixel jitter tables + the ring of saved view-projections the TAA
// reprojection matrix is built from.
class TemporalJitter
{
public:
	enum class Type { FXAA_2Phase, SMAA_T2X, TAA_8Phase, TAA_16Phase, Custom, None };
	TemporalJitter();
	void reset() { phase = 0; }
	void init(Type type_, vec2 backbuffer_resolution);
	void step(const mat4 &projection, const mat4 &view);
	const mat4 &get_jitter_matrix() const { return jitter_table[phase]; }
	const mat4 &get_jittered_projection() const { return saved_jittered_projection; }
	const mat4 &get_history_view_proj(int frames) const { return saved_view_proj[get_offset_phase(frames)]; }
	const mat4 &get_history_inv_view_proj(int frames) const { return saved_inv_view_proj[get_offset_phase(frames)]; }
	const mat4 &get_history_jittered_view_proj(int frames) const { return saved_jittered_view_proj[get_offset_phase(frames)]; }
	unsigned get_jitter_phase() const { return phase; }
	// checkpoint / replay: the phase and the saved matrices of the last jitter_count frames
	struct State
	{
		unsigned phase = 0;
		std::vector<mat4> jittered_view_proj, view_proj, inv_view_proj;
		mat4 jittered_projection;
	};
	State get_state() const { return {phase, saved_jittered_view_proj, saved_view_proj, saved_inv_view_proj, saved_jittered_projection}; }
	void set_state(const State &s)
	{
		if (s.view_proj.size() != saved_view_proj.size() || s.inv_view_proj.size() != saved_inv_view_proj.size() ||
		    s.jittered_view_proj.size() != saved_jittered_view_proj.size())
			throw std::logic_error("TemporalJitter::set_state: jitter sequence length differs");
		if (jitter_count != 0 && s.phase >= jitter_count)
			throw std::logic_error("TemporalJitter::set_state: phase outside the jitter sequence");
		phase = s.phase;
		saved_jittered_view_proj = s.jittered_view_proj;
		saved_view_proj = s.view_proj;
		saved_inv_view_proj = s.inv_view_proj;
		saved_jittered_projection = s.jittered_projection;
	}
	unsigned get_jitter_count() const { return jitter_count; }
	Type get_jitter_type() const { return type; }

private:
	unsigned phase = 0;
	unsigned jitter_count = 0;
	std::vector<mat4> jitter_table;
	std::vector<mat4> saved_jittered_view_proj;
	std::vector<mat4> saved_view_proj;
	std::vector<mat4> saved_inv_view_proj;
	mat4 saved_jittered_projection;
	Type type = Type::None;
	unsigned get_offset_phase(int frames) const;
};

// fxaa.cpp:28-55
// HIP executor extension on every set-up function below: `strip` = the row-band plan of this instance (strip_plan.hpp),
// nullptr or an inactive plan = the whole frame, as the reference.
void setup_fxaa_postprocess(RenderGraph &graph, const std::string &input, const std::string &output, VkFormat output_format = VK_FORMAT_UNDEFINED,
                            const StripPlan *strip = nullptr);
// smaa.cpp:32-208
void setup_smaa_postprocess(RenderGraph &graph, TemporalJitter &jitter, float scaling_factor, const std::string &input,
                            const std::string &input_depth, const std::string &output, SMAAPreset preset, const StripPlan *strip = nullptr);
// temporal.cpp:199-266
void setup_taa_resolve(RenderGraph &graph, TemporalJitter &jitter, float scaling_factor, const std::string &input, const std::string &input_depth,
                       const std::string &input_mv, const std::string &output, TAAQuality quality, const StripPlan *strip = nullptr);

// aa.cpp:176-253
bool setup_before_post_chain_antialiasing(PostAAType type, RenderGraph &graph, TemporalJitter &jitter, const RenderContext &context,
                                          float scaling_factor, const std::string &input, const std::string &input_depth,
                                          const std::string &input_mv, const std::string &output, const StripPlan *strip = nullptr);
// aa.cpp:75-174: `output + "-scale"` (FSR 1.0 EASU from `input` to the swapchain size) and, with use_sharpen,
// `output + "-sharpen"` (RCAS, 0.5 stops).  fp16 = the FP16 shader variant (the reference picks it from the device's
// shaderFloat16 / FIDELITYFX_FSR_FP16, aa.cpp:118-119; on MI355X that is true).
bool setup_after_post_chain_upscaling(RenderGraph &graph, const std::string &input, const std::string &output, bool use_sharpen, bool fp16 = true);
bool setup_after_post_chain_antialiasing(PostAAType type, RenderGraph &graph, TemporalJitter &jitter, float scaling_factor,
                                         const std::string &input, const std::string &input_depth, const std::string &output,
                                         const StripPlan *strip = nullptr);
// SMAA_MAX_SEARCH_STEPS of a preset (SMAA.hlsl:304-324), 0 for anything that is not SMAA: what StripAA wants to know.
unsigned smaa_search_steps(PostAAType type);
PostAAType string_to_post_antialiasing_type(const char *type);
} // namespace Granite
