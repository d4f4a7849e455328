// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own screen-space-reflection shaders on the CPU: post/ffx-sssr/{classify,build_indirect,trace_primary}.comp
// and apply.frag with sssr_util.h, inc/project_direction.h and lights/pbr.h, re-spelled by glsl2cpp.py at build time into gen/
// (scratch, removed after the compile) and compiled against glsl_cpu.hpp; bindings as SSRState::build_render_pass and the
// apply pass set them (renderer/post/ssr.cpp:84-173,286-322).
//
// What a GPU leaves open is fixed the way oracle_ssr.cpp states it, so that the two can be compared bit for bit:
//   * classify.comp appends rays with atomicAdd in whatever order the hardware reaches them.  Here workgroups (8 x 8 tiles) run
//     one after the other in row-major order and, inside a workgroup, the 64 invocations take their atomicAdd turn in lane
//     order (the append is wrapped in a turnstile): the list comes out in tile order, Z-order inside a tile.
//   * trace_primary.comp: a workgroup = a subgroup = 64 real threads; subgroupBallot(true) inside the traversal loop is a
//     rendezvous of the lanes still in the loop (a lane that has left the loop drops out of it for good), so
//     subgroupBallotBitCount gives what a wave in lockstep sees.
//   * texelFetch outside a level returns 0; imageStore outside the image is dropped.
//   * copy conflicts (two rays of a quad copying into the same pixel, one horizontally, one vertically): the shader issues a
//     ray's own stores, then its horizontal, vertical and diagonal copies (trace_primary.comp:278-305); between rays the order
//     is a race.  The runner buffers the image stores of trace_primary.comp and replays them in those four phases over the
//     whole dispatch -- all own stores, then all horizontal copies, then all vertical ones, then the diagonal ones -- which
//     is the "vertical copy wins" rule of oracle_ssr.cpp (a horizontal copy survives exactly where no vertical copy lands).
#include <atomic>
#include <barrier>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "glsl_cpu.hpp"

using namespace glsl;

namespace
{
// ---- resources the stock environment does not model ----------------------------------------------------------------------------
// A texture with mip levels and robust (zero) out-of-range fetches: uDepth is the depth hierarchy, the others have one level.
struct MipTexture
{
	const void *level_data[16] = {};
	int level_w[16] = {}, level_h[16] = {};
	int levels = 0;
	Format format = Format::R32F;
};
struct ArrayTexture // texture2DArray: R8G8_UNORM layers
{
	const uint16_t *data = nullptr;
	int w = 0, h = 0, layers = 0;
};
} // namespace

static inline vec4 texelFetch(const MipTexture &t, const ivec2 &p, int lod)
{
	if (lod < 0 || lod >= t.levels || p.x < 0 || p.y < 0 || p.x >= t.level_w[lod] || p.y >= t.level_h[lod])
		return vec4(0.0f);
	Texture view;
	view.data = t.level_data[lod];
	view.w = t.level_w[lod];
	view.h = t.level_h[lod];
	view.format = t.format;
	return view.texel(p.x, p.y);
}
static inline vec4 texelFetch(const ArrayTexture &t, const ivec3 &p, int)
{
	if (p.x < 0 || p.y < 0 || p.z < 0 || p.x >= t.w || p.y >= t.h || p.z >= t.layers)
		return vec4(0.0f);
	const uint16_t v = t.data[(size_t(p.z) * t.h + p.y) * t.w + p.x];
	return vec4(float(v & 255u) / 255.0f, float(v >> 8) / 255.0f, 0.0f, 1.0f);
}

// ---- execution environment of one workgroup of 64 threads ----------------------------------------------------------------------
namespace
{
struct Team
{
	std::barrier<> *sync = nullptr; // barrier()
	// quad swaps
	std::vector<std::unique_ptr<std::barrier<>>> quad_barriers;
	int quad_slots[64] = {};
	// atomicAdd turnstile: lane order
	std::mutex turn_lock;
	std::condition_variable turn_cv;
	unsigned next_lane = 0;
	// ballot rendezvous over the lanes that are still running
	std::atomic<unsigned> ballot_arrivals{0};
	unsigned ballot_snapshot = 0;
	std::unique_ptr<std::barrier<std::function<void()>>> ballot;
} team;

bool quad_exchange(bool v, unsigned partner_xor)
{
	const unsigned lane = gl_LocalInvocationIndex;
	team.quad_slots[lane] = v ? 1 : 0;
	team.quad_barriers[lane >> 2]->arrive_and_wait();
	const bool other = team.quad_slots[lane ^ partner_xor] != 0;
	team.quad_barriers[lane >> 2]->arrive_and_wait();
	return other;
}
} // namespace
static inline void barrier()
{
	if (team.sync)
		team.sync->arrive_and_wait();
}
static inline bool subgroupQuadSwapHorizontal(bool v) { return quad_exchange(v, 1); }
static inline bool subgroupQuadSwapVertical(bool v) { return quad_exchange(v, 2); }
static inline bool subgroupQuadSwapDiagonal(bool v) { return quad_exchange(v, 3); }
struct BallotResult { unsigned count; };
static inline BallotResult subgroupBallot(bool)
{
	team.ballot_arrivals.fetch_add(1);
	team.ballot->arrive_and_wait(); // completes when every lane that has not finished is here; the completion step takes the count
	return {team.ballot_snapshot};
}
static inline uint subgroupBallotBitCount(const BallotResult &b) { return b.count; }
static inline float ldexp(float v, int e) { return ldexpf(v, e); }
static inline vec3 reflect(const vec3 &i, const vec3 &n) { return i - n * (2.0f * dot(n, i)); }
static inline ivec2 operator^(const ivec2 &a, int b) { return ivec2(a.x ^ b, a.y ^ b); }
template <int N, int A, int B, bool C>
static inline ivec2 operator&(const swz2<int, N, A, B, C> &s, int mask)
{
	const ivec2 v = s;
	return ivec2(v.x & mask, v.y & mask);
}
static inline float exp2(int v) { return glsl::exp2(float(v)); }
#undef M_PI // sssr_util.h declares its own constant of that name

// atomicAdd on the ray counter goes through the turnstile (classify only); shared-memory counters take the plain one.
namespace classify_env
{
static uint *ordered_counter = nullptr;
static inline uint atomicAdd(uint &mem, uint v)
{
	if (&mem != ordered_counter)
		return glsl::atomicAdd(mem, v);
	std::unique_lock<std::mutex> holder{team.turn_lock};
	// lanes that do not append never come here: wait until every lower lane has either appended or declared that it will not
	team.turn_cv.wait(holder, [] { return team.next_lane == gl_LocalInvocationIndex; });
	const uint old = mem;
	mem += v;
	return old;
}
} // namespace classify_env

// image stores of trace_primary.comp are recorded per invocation and replayed in phases (header comment)
namespace
{
struct BufferedImage
{
	Image target;
};
struct StoreRecord
{
	BufferedImage *image;
	ivec2 coord;
	vec4 value;
};
thread_local std::vector<StoreRecord> lane_stores;
}
static inline void imageStore(BufferedImage &img, const ivec2 &p, const vec4 &v) { lane_stores.push_back({&img, p, v}); }

#define texture2D MipTexture
#define texture2DArray ArrayTexture

// classify.comp names a local after the function that initialises it (legal GLSL, not C++): the calls get another spelling
#define is_base_ray(c) is_base_ray_fn(c)
namespace sssr_classify
{
using classify_env::atomicAdd;
#include "gen/sssr_classify.inc"
}
#undef is_base_ray
namespace sssr_build_indirect
{
#include "gen/sssr_build_indirect.inc"
}
#define image2D BufferedImage
namespace sssr_trace
{
#include "gen/sssr_trace_primary.inc"
}
#undef image2D
#undef texture2D
namespace sssr_apply
{
#include "gen/sssr_apply.inc"
}
#undef texture2DArray

namespace
{
// Same layout as OrcSSRArgs (oracle_ssr.cpp): the test hands both the same structure.
struct SSRArgs
{
	int32_t width, height;
	const float *hier;
	int32_t hier_w, hier_h, hier_levels;
	const uint16_t *pbr;
	const uint32_t *normal;
	const uint16_t *light;
	const uint16_t *noise;
	int32_t frame;
	const float *view_projection;
	const float *inv_view_projection;
	float camera_position[3];
	uint16_t *output;
	uint16_t *ray_length;
	uint8_t *confidence;
	uint32_t *ray_list;
	uint32_t *ray_counter;
};

mat4 load_mat(const float *m)
{
	return mat4(vec4(m[0], m[1], m[2], m[3]), vec4(m[4], m[5], m[6], m[7]), vec4(m[8], m[9], m[10], m[11]), vec4(m[12], m[13], m[14], m[15]));
}

MipTexture single_level(const void *data, int w, int h, Format format)
{
	MipTexture t;
	t.level_data[0] = data;
	t.level_w[0] = w;
	t.level_h[0] = h;
	t.levels = 1;
	t.format = format;
	return t;
}

MipTexture hierarchy(const SSRArgs *a)
{
	MipTexture t;
	t.format = Format::R32F;
	t.levels = a->hier_levels;
	size_t offset = 0;
	for (int l = 0; l < a->hier_levels; l++)
	{
		t.level_w[l] = std::max(a->hier_w >> l, 1);
		t.level_h[l] = std::max(a->hier_h >> l, 1);
		t.level_data[l] = a->hier + offset;
		offset += size_t(t.level_w[l]) * size_t(t.level_h[l]);
	}
	return t;
}

template <typename UBO>
void fill_ubo(UBO &ubo, const SSRArgs *a)
{
	ubo.view_projection = load_mat(a->view_projection);
	ubo.inv_view_projection = load_mat(a->inv_view_projection);
	ubo.float_resolution = vec2(float(a->width), float(a->height));
	ubo.inv_resolution = vec2(1.0f) / ubo.float_resolution;
	ubo.resolution = uvec2(uint(a->width), uint(a->height));
	ubo.camera_position = vec3(a->camera_position[0], a->camera_position[1], a->camera_position[2]);
	ubo.max_lod = a->hier_levels - 1;
	ubo.frame = a->frame;
	ubo.resolution_1d = uint(a->width) * uint(a->height);
}

Image image_of(void *data, int w, int h, Format format)
{
	Image img;
	img.data = data;
	img.w = w;
	img.h = h;
	img.format = format;
	return img;
}

template <typename Main>
void run_team(unsigned wg_x, unsigned wg_y, unsigned global_base, Main main_fn, bool with_ballot)
{
	std::barrier<> sync(64);
	team.sync = &sync;
	team.next_lane = 0;
	team.ballot_arrivals = 0;
	if (with_ballot)
		team.ballot = std::make_unique<std::barrier<std::function<void()>>>(64, std::function<void()>([] {
			team.ballot_snapshot = team.ballot_arrivals.exchange(0);
		}));
	std::vector<std::thread> threads;
	for (unsigned i = 0; i < 64; i++)
		threads.emplace_back([=]() {
			gl_WorkGroupID = uvec3(wg_x, wg_y, 0u);
			gl_LocalInvocationIndex = i;
			gl_LocalInvocationID = uvec3(i, 0u, 0u);
			gl_GlobalInvocationID = uvec3(global_base + i, 0u, 0u);
			gl_SubgroupSize = 64;
			gl_NumSubgroups = 1;
			gl_SubgroupID = 0;
			gl_SubgroupInvocationID = i;
			main_fn();
			if (with_ballot)
				team.ballot->arrive_and_drop(); // this lane is out of every later rendezvous
		});
	for (auto &t : threads)
		t.join();
	team.sync = nullptr;
}
} // namespace

// classify.comp + build_indirect.comp
extern "C" void ref_ssr_classify(const SSRArgs *a)
{
	namespace s = sssr_classify;
	const MipTexture depth = hierarchy(a);
	fill_ubo(s::sssr, a);
	s::uDepth = depth;
	s::uPBR = single_level(a->pbr, a->width, a->height, Format::RG8_UNORM);
	s::uOutput = image_of(a->output, a->width, a->height, Format::RGBA16F);
	s::uRayConfidence = image_of(a->confidence, a->width, a->height, Format::R8_UNORM);
	s::ray_counter.indirect = uvec4(0u);
	s::ray_counter.atomic_count = 0u;
	s::ray_counter.copied_count = 0u;
	s::ray_list.data = a->ray_list;
	classify_env::ordered_counter = &s::ray_counter.atomic_count;
	team.quad_barriers.clear();
	for (int i = 0; i < 16; i++)
		team.quad_barriers.emplace_back(std::make_unique<std::barrier<>>(4));

	const unsigned groups_x = unsigned(a->width + 7) / 8, groups_y = unsigned(a->height + 7) / 8;
	for (unsigned gy = 0; gy < groups_y; gy++)
		for (unsigned gx = 0; gx < groups_x; gx++)
			run_team(gx, gy, 0, [] {
				sssr_classify::main();
				// turnstile: this lane is done with (or never needed) its append; let the next lane go
				std::unique_lock<std::mutex> holder{team.turn_lock};
				team.turn_cv.wait(holder, [] { return team.next_lane == gl_LocalInvocationIndex; });
				team.next_lane++;
				team.turn_cv.notify_all();
			}, false);

	namespace b = sssr_build_indirect;
	b::ray_counter.indirect = s::ray_counter.indirect;
	b::ray_counter.atomic_count = s::ray_counter.atomic_count;
	b::ray_counter.copied_count = s::ray_counter.copied_count;
	b::main();
	a->ray_counter[0] = b::ray_counter.indirect.x;
	a->ray_counter[1] = b::ray_counter.indirect.y;
	a->ray_counter[2] = b::ray_counter.indirect.z;
	a->ray_counter[3] = b::ray_counter.indirect.w;
	a->ray_counter[4] = b::ray_counter.atomic_count;
	a->ray_counter[5] = b::ray_counter.copied_count;
}

// trace_primary.comp over the list ref_ssr_classify left, dispatch_indirect(ray_counter).
extern "C" void ref_ssr_trace(const SSRArgs *a)
{
	namespace s = sssr_trace;
	const MipTexture depth = hierarchy(a);
	fill_ubo(s::sssr, a);
	s::uDepth = depth;
	s::uPBR = single_level(a->pbr, a->width, a->height, Format::RG8_UNORM);
	s::uNormal = single_level(a->normal, a->width, a->height, Format::A2B10G10R10_UNORM);
	s::uLight = single_level(a->light, a->width, a->height, Format::RGBA16F);
	s::uBaseColor = MipTexture();
	s::uNoise.data = a->noise;
	s::uNoise.w = 128;
	s::uNoise.h = 128;
	s::uNoise.layers = 64;
	s::uOutput.target = image_of(a->output, a->width, a->height, Format::RGBA16F);
	s::uRayLength.target = image_of(a->ray_length, a->width, a->height, Format::R16F);
	s::uRayConfidence.target = image_of(a->confidence, a->width, a->height, Format::R8_UNORM);
	s::ray_counter.indirect = uvec4(a->ray_counter[0], a->ray_counter[1], a->ray_counter[2], a->ray_counter[3]);
	s::ray_counter.atomic_count = a->ray_counter[4];
	s::ray_counter.copied_count = a->ray_counter[5];
	s::ray_list.data = a->ray_list;
	// phase 0: a ray's own pixel, 1 / 2 / 3: its horizontal / vertical / diagonal copy (by the offset from its first store)
	std::vector<StoreRecord> phases[4];
	std::mutex phases_lock;
	const unsigned groups = a->ray_counter[0];
	for (unsigned g = 0; g < groups; g++)
		run_team(g, 0, g * 64u, [&] {
			lane_stores.clear();
			sssr_trace::main();
			if (lane_stores.empty())
				return;
			const ivec2 own = lane_stores.front().coord;
			std::lock_guard<std::mutex> holder{phases_lock};
			for (auto &r : lane_stores)
				phases[((r.coord.x ^ own.x) & 1) | (((r.coord.y ^ own.y) & 1) << 1)].push_back(r);
		}, true);
	for (auto &phase : phases)
		for (auto &r : phase)
			glsl::imageStore(r.image->target, r.coord, r.value);
}

// apply.frag blended ONE / ONE into hdr, depth test NOT_EQUAL against the quad at z = 1.
extern "C" void ref_ssr_apply(int width, int height, const uint16_t *reflected, const uint32_t *albedo_srgb, const uint32_t *normal,
                              const uint16_t *pbr, const float *depth, const uint16_t *brdf_lut_rg16f, int lut_w, int lut_h,
                              const float *inv_view_projection, const float *camera_position, uint16_t *hdr)
{
	namespace s = sssr_apply;
	s::sssr.inv_view_projection = load_mat(inv_view_projection);
	s::sssr.camera_position = vec3(camera_position[0], camera_position[1], camera_position[2]);
	auto tex = [&](Texture &t, const void *data, int w, int h, Format f, Filter filter) {
		t.data = data;
		t.w = w;
		t.h = h;
		t.format = f;
		t.filter = filter;
	};
	tex(s::uBaseColor, albedo_srgb, width, height, Format::RGBA8_SRGB, Filter::Nearest);
	tex(s::uNormal, normal, width, height, Format::A2B10G10R10_UNORM, Filter::Nearest);
	tex(s::uPBR, pbr, width, height, Format::RG8_UNORM, Filter::Nearest);
	tex(s::uDepth, depth, width, height, Format::R32F, Filter::Nearest);
	tex(s::uReflected, reflected, width, height, Format::RGBA16F, Filter::Nearest);
	tex(s::uBRDFLut, brdf_lut_rg16f, lut_w, lut_h, Format::RG16F, Filter::Linear);
	for (int y = 0; y < height; y++)
		for (int x = 0; x < width; x++)
		{
			if (depth[size_t(y) * width + x] == 1.0f)
				continue;
			gl_FragCoord = vec4(float(x) + 0.5f, float(y) + 0.5f, 1.0f, 1.0f);
			// apply.vert: vUV = Attr * 0.5 + 0.5 interpolated over the full-screen triangle = the pixel centre in [0, 1]
			s::vUV = (vec2(float(x), float(y)) + vec2(0.5f)) * vec2(1.0f / float(width), 1.0f / float(height));
			s::main();
			uint16_t *p = hdr + (size_t(y) * width + x) * 4;
			p[0] = orc::float_to_half_rne(orc::half_to_float(p[0]) + s::FragColor.x);
			p[1] = orc::float_to_half_rne(orc::half_to_float(p[1]) + s::FragColor.y);
			p[2] = orc::float_to_half_rne(orc::half_to_float(p[2]) + s::FragColor.z);
		}
}
