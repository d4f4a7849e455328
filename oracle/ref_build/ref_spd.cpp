// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own single-pass downsampler on the CPU: post/ffx-spd/spd.comp with the FidelityFX headers the
// reference vendors (ffx-a/ffx_a.h, ffx-spd/ffx_spd.h in their GLSL GPU spelling, wave-operation path), re-spelled into gen/
// at build time, under the defines and bindings of emit_single_pass_downsample (renderer/post/spd.cpp:56-102): SUBGROUP = 1,
// SINGLE_INPUT_TAP = 1, LinearClamp (NearestClamp in depth mode), twelve storage images with the spare bindings repeating the
// last level, the atomic counter buffer.  A workgroup is a team of 256 real threads; subgroups are 64 lanes; quad swaps
// exchange through a per-quad rendezvous (the shader only uses them with whole quads active); workgroups run one after the
// other, so the last one to take a ticket from the counter reduces levels 6.., as on a GPU.
#include <barrier>
#include <memory>
#include <thread>
#include <vector>
#include "glsl_cpu.hpp"

namespace glsl
{
namespace
{
std::barrier<> *team_barrier = nullptr;
std::vector<std::unique_ptr<std::barrier<>>> quad_barriers;
vec4 quad_slots[256];

vec4 quad_exchange(const vec4 &v, unsigned partner_xor)
{
	const unsigned lane = gl_LocalInvocationIndex;
	quad_slots[lane] = v;
	quad_barriers[lane >> 2]->arrive_and_wait();
	const vec4 other = quad_slots[lane ^ partner_xor];
	quad_barriers[lane >> 2]->arrive_and_wait();
	return other;
}
} // namespace
static inline void barrier()
{
	if (team_barrier)
		team_barrier->arrive_and_wait();
}
static inline void memoryBarrier() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline vec4 subgroupQuadSwapHorizontal(const vec4 &v) { return quad_exchange(v, 1); }
static inline vec4 subgroupQuadSwapVertical(const vec4 &v) { return quad_exchange(v, 2); }
static inline vec4 subgroupQuadSwapDiagonal(const vec4 &v) { return quad_exchange(v, 3); }

#define SUBGROUP 1
#define SINGLE_INPUT_TAP 1

#define COMPONENTS 4
#define FILTER_MOD 0
#define REDUCTION_MODE 0
namespace spd_c4
{
#include "gen/spd.inc"
}
#undef COMPONENTS
#undef FILTER_MOD
#undef SPD_LINEAR_SAMPLER

#define COMPONENTS 3
#define FILTER_MOD 1
namespace spd_c3_mod
{
#include "gen/spd.inc"
}
#undef COMPONENTS
#undef FILTER_MOD
#undef REDUCTION_MODE
#undef SPD_LINEAR_SAMPLER

#define COMPONENTS 1
#define FILTER_MOD 0
#define REDUCTION_MODE 1
namespace spd_depth
{
#include "gen/spd.inc"
}
#undef COMPONENTS
#undef FILTER_MOD
#undef REDUCTION_MODE
} // namespace glsl

using namespace glsl;

namespace
{
template <typename Main>
void dispatch(int groups_x, int groups_y, Main main_fn)
{
	quad_barriers.clear();
	for (int i = 0; i < 64; i++)
		quad_barriers.emplace_back(std::make_unique<std::barrier<>>(4));
	for (int gy = 0; gy < groups_y; gy++)
		for (int gx = 0; gx < groups_x; gx++)
		{
			std::barrier<> sync(256);
			team_barrier = &sync;
			std::vector<std::thread> threads;
			for (unsigned i = 0; i < 256; i++)
				threads.emplace_back([=]() {
					gl_WorkGroupID = uvec3(uint(gx), uint(gy), 0u);
					gl_LocalInvocationIndex = i;
					gl_LocalInvocationID = uvec3(i, 0u, 0u);
					gl_SubgroupSize = 64;
					gl_NumSubgroups = 4;
					gl_SubgroupID = i / 64;
					gl_SubgroupInvocationID = i % 64;
					main_fn();
				});
			for (auto &t : threads)
				t.join();
			team_barrier = nullptr;
		}
}

size_t level_offset(int w0, int h0, int level)
{
	size_t texels = 0;
	for (int l = 0; l < level; l++)
		texels += size_t(std::max(w0 >> l, 1)) * size_t(std::max(h0 >> l, 1));
	return texels * 4;
}
} // namespace

#define BIND_SPD(NS, FILTER)                                                                       \
	{                                                                                              \
		namespace s = glsl::NS;                                                                    \
		s::uInput.data = input;                                                                    \
		s::uInput.w = iw;                                                                          \
		s::uInput.h = ih;                                                                          \
		s::uInput.format = Format::RGBA16F;                                                        \
		s::uInput.filter = FILTER;                                                                 \
		for (int i = 0; i < 12; i++)                                                               \
		{                                                                                          \
			const int l = std::min(i, mips - 1); /* spd.cpp:70-71 */                               \
			s::uImages[i].data = chain + level_offset(w0, h0, l);                                  \
			s::uImages[i].w = std::max(w0 >> l, 1);                                                \
			s::uImages[i].h = std::max(h0 >> l, 1);                                                \
			s::uImages[i].format = Format::RGBA16F;                                                \
		}                                                                                          \
		s::spdGlobalAtomic.counter = &counter;                                                     \
		s::base_image_resolution = ivec2(w0, h0);                                                  \
		s::inv_resolution = vec2(1.0f / float(iw), 1.0f / float(ih));                              \
		s::mips = uint(mips);                                                                      \
		s::num_workgroups = uint(groups_x * groups_y);                                             \
	}

// Same signature as orc_spd (oracle/oracle_spd.cpp).  Supported define sets: (components 4, no filter_mods, colour),
// (components 3, filter_mods, colour), (components 1, no filter_mods, depth); anything else returns -1.
extern "C" int ref_spd(const uint16_t *input, int iw, int ih, int w0, int h0, int mips, int components, int depth_mode,
                       const float *filter_mods, uint16_t *chain)
{
	uint counter = 0;
	const int groups_x = (w0 + 31) / 32, groups_y = (h0 + 31) / 32; // spd.cpp:91-93
	if (!depth_mode && components == 4 && !filter_mods)
	{
		BIND_SPD(spd_c4, Filter::Linear)
		dispatch(groups_x, groups_y, glsl::spd_c4::main);
	}
	else if (!depth_mode && components == 3 && filter_mods)
	{
		BIND_SPD(spd_c3_mod, Filter::Linear)
		for (int i = 0; i < mips; i++)
			glsl::spd_c3_mod::filter_mods[i] = vec4(filter_mods[4 * i], filter_mods[4 * i + 1], filter_mods[4 * i + 2], filter_mods[4 * i + 3]);
		dispatch(groups_x, groups_y, glsl::spd_c3_mod::main);
	}
	else if (depth_mode && components == 1 && !filter_mods)
	{
		BIND_SPD(spd_depth, Filter::Nearest)
		dispatch(groups_x, groups_y, glsl::spd_depth::main);
	}
	else
		return -1;
	return int(counter); // the shader resets it to 0 when levels 6.. were reduced (ffx_spd.h:830)
}
