// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own depth-hierarchy shader on the CPU: post/hiz.comp (re-spelled into gen/ at build time) with the
// bindings and push constants of HiZPassState::build_render_pass (renderer/post/spd.cpp:141-194).  A workgroup is a team of
// 256 real threads; subgroups are 64 lanes; quad swaps exchange through a per-quad rendezvous (the shader only ever uses them
// with whole quads active); workgroups run one after the other, so the last one to take a ticket from the atomic counter does
// the tail, as on a GPU.
#include <barrier>
#include <memory>
#include <thread>
#include <vector>
#include "glsl_cpu.hpp"

using namespace glsl;

namespace
{
std::barrier<> *team_barrier = nullptr;
std::vector<std::unique_ptr<std::barrier<>>> quad_barriers;
float quad_slots[256];

float quad_exchange(float v, unsigned partner_xor)
{
	const unsigned lane = gl_LocalInvocationIndex;
	quad_slots[lane] = v;
	quad_barriers[lane >> 2]->arrive_and_wait();
	const float other = quad_slots[lane ^ partner_xor];
	quad_barriers[lane >> 2]->arrive_and_wait();
	return other;
}
} // namespace
static inline void barrier()
{
	if (team_barrier)
		team_barrier->arrive_and_wait();
}
static inline float subgroupQuadSwapHorizontal(float v) { return quad_exchange(v, 1); }
static inline float subgroupQuadSwapVertical(float v) { return quad_exchange(v, 2); }
static inline float subgroupQuadSwapDiagonal(float v) { return quad_exchange(v, 3); }

#define LAYERED 0
#define WRITE_TOP_LEVEL 1
namespace hiz_top
{
#include "gen/hiz.inc"
}
#undef WRITE_TOP_LEVEL
#define WRITE_TOP_LEVEL 0
namespace hiz_downsample
{
#include "gen/hiz.inc"
}
#undef WRITE_TOP_LEVEL

namespace
{
size_t chain_offset(int w, int h, int level)
{
	size_t o = 0;
	for (int l = 0; l < level; l++)
		o += size_t(std::max(w >> l, 1)) * size_t(std::max(h >> l, 1));
	return o;
}

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
} // namespace

#define RUN_HIZ(NS, TOP)                                                                                     \
	{                                                                                                        \
		namespace s = NS;                                                                                    \
		s::uTexture.data = depth;                                                                            \
		s::uTexture.w = iw;                                                                                  \
		s::uTexture.h = ih;                                                                                  \
		s::uTexture.format = Format::R32F;                                                                   \
		s::uTexture.filter = Filter::Nearest;                                                                \
		for (int i = 0; i < 12; i++)                                                                         \
		{                                                                                                    \
			/* binding i holds chain level i (+1 with a top level); spare bindings repeat level 0 (spd.cpp:177-183) */ \
			const int level = std::min(i + (TOP ? 1 : 0), chain_levels - 1);                                 \
			const bool spare = i + (TOP ? 1 : 0) >= chain_levels;                                            \
			const int l = spare ? 0 : level;                                                                 \
			s::uImages[i].data = chain + chain_offset(chain_w, chain_h, l);                                  \
			s::uImages[i].w = std::max(chain_w >> l, 1);                                                     \
			s::uImages[i].h = std::max(chain_h >> l, 1);                                                     \
			s::uImages[i].format = Format::R32F;                                                             \
		}                                                                                                    \
		s::registers.z_transform = mat2(z_transform[0], z_transform[1], z_transform[2], z_transform[3]);     \
		s::registers.resolution = ivec2(res_w, res_h);                                                       \
		s::registers.inv_resolution = vec2(1.0f / float(iw), 1.0f / float(ih));                              \
		s::registers.mips = mips;                                                                            \
		s::registers.target_counter = uint((res_w / 64) * (res_h / 64));                                     \
		s::atomic_counter = &counter;                                                                        \
	}

// depth iw x ih; chain: chain_levels levels, level 0 = chain_w x chain_h, tightly packed; write_top_level = !output_downsample.
extern "C" void ref_hiz(const float *depth, int iw, int ih, int res_w, int res_h, int mips, const float *z_transform, int write_top_level,
                        float *chain, int chain_w, int chain_h, int chain_levels)
{
	uint counter = 0;
	if (write_top_level)
	{
		RUN_HIZ(hiz_top, true)
		hiz_top::uImageTop.data = chain;
		hiz_top::uImageTop.w = chain_w;
		hiz_top::uImageTop.h = chain_h;
		hiz_top::uImageTop.format = Format::R32F;
		dispatch(res_w / 64, res_h / 64, hiz_top::main);
	}
	else
	{
		RUN_HIZ(hiz_downsample, false)
		dispatch(res_w / 64, res_h / 64, hiz_downsample::main);
	}
}
