// CPU-only exercise of LightClusterer's threaded per-frame refresh (one frame ahead on helper threads, adopted by the next
// refresh() when its prediction held): no device.  Built by tests/test_host_sanitizers_cpu.py against a libgranite_host.so compiled
// with -fsanitize=thread / address (make SANITIZE=...), so that races and lifetime errors of the prefetch hand-over show up here;
// the program itself checks that a prefetched refresh packs exactly what a synchronous one packs.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <vector>
#include "host/lights/clusterer.hpp"
#include "host/lights/lights.hpp"
#include "host/math.hpp"
#include "host/render_context.hpp"

using namespace Granite;

static mat4 look_along_z(float x)
{
	return translate(vec3(-x, -2.0f, -8.0f));
}

int main()
{
	const unsigned count = 3000;
	std::mt19937 rng(1234);
	std::uniform_real_distribution<float> u(-1.0f, 1.0f);
	std::vector<std::unique_ptr<PositionalLight>> lights;
	std::vector<mat_affine> transforms(count);
	PositionalLightList list;
	for (unsigned i = 0; i < count; i++)
	{
		if (i % 4 == 0)
		{
			auto spot = std::make_unique<SpotLight>();
			spot->set_spot_parameters(0.94f, 0.87f);
			lights.push_back(std::move(spot));
		}
		else
			lights.push_back(std::make_unique<PointLight>());
		lights.back()->set_color(vec3(1.0f + 10.0f * std::fabs(u(rng)), 2.0f, 3.0f));
		lights.back()->set_maximum_range(4.0f);
		mat_affine &t = transforms[i];
		t[0] = vec4(1.0f, 0.0f, 0.0f, 12.0f * u(rng));
		t[1] = vec4(0.0f, 1.0f, 0.0f, 6.0f * u(rng));
		t[2] = vec4(0.0f, 0.0f, 1.0f, -20.0f + 19.0f * u(rng));
		list.push_back({lights.back().get(), &transforms[i]});
	}
	const mat4 projection = perspective(1.0471976f, 16.0f / 9.0f, 0.1f, 100.0f);
	RenderContext context;
	TaskComposer composer;
	LightClusterer threaded, plain;
	for (LightClusterer *c : {&threaded, &plain})
	{
		c->set_resolution(128, 64, 4096);
		c->set_scene_lights(&list);
		c->set_base_render_context(&context);
	}
	unsigned mismatches = 0;
	for (int frame = 0; frame < 60; frame++)
	{
		const float x = 0.01f * float(frame);
		context.set_camera(projection, look_along_z(x));
		if (frame % 17 == 5)
		{
			// the scene changes between frames: the prefetch in flight must be dropped, not adopted
			threaded.invalidate_prefetch();
			transforms[frame].operator[](0).w += 0.5f;
		}
		threaded.refresh(context, composer);
		plain.refresh(context, composer);
		// what the next frame will use (every third prediction is wrong on purpose)
		RenderContext next = context;
		next.set_camera(projection, look_along_z(frame % 3 == 2 ? x + 0.5f : x + 0.01f));
		threaded.prefetch(next.get_render_parameters());
		const auto &a = threaded.get_packed_lights(), &b = plain.get_packed_lights();
		if (a.size() != b.size() || memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) != 0 ||
		    memcmp(threaded.get_type_mask(), plain.get_type_mask(), sizeof(uint32_t) * 128) != 0 ||
		    threaded.get_volume_index_range().size() != plain.get_volume_index_range().size() ||
		    memcmp(threaded.get_volume_index_range().data(), plain.get_volume_index_range().data(),
		           plain.get_volume_index_range().size() * sizeof(uvec2)) != 0)
			mismatches++;
	}
	printf("{\"frames\":60,\"lights\":%u,\"prefetch_hits\":%llu,\"mismatches\":%u}\n", count, (unsigned long long)threaded.get_prefetch_hits(), mismatches);
	return mismatches ? 1 : 0;
}
