// CPU-only exercise of the restated RenderGraph declaration / bake API (no device): prints one JSON object per case on
// stdout, consumed by tests/test_render_graph_cases_cpu.py.  The first case declares the topology of the reference's
// tests/render_graph_sandbox.cpp (depth -> first -> async compute -> final, backbuffer "back") with this repo's own code.
#include "../../granite_amd/csrc/host/render_graph.hpp"
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <string>
#include <vector>

using namespace Granite;

static ResourceDimensions backbuffer(unsigned w, unsigned h)
{
	ResourceDimensions dim;
	dim.width = w;
	dim.height = h;
	dim.format = VK_FORMAT_R8G8B8A8_SRGB;
	return dim;
}

static void print_case(const char *name, const std::string &json)
{
	printf("{\"case\":\"%s\",\"graph\":%s}\n", name, json.c_str());
}

static void print_error(const char *name, const std::function<void()> &fn)
{
	try
	{
		fn();
		printf("{\"case\":\"%s\",\"error\":null}\n", name);
	}
	catch (const std::logic_error &e)
	{
		printf("{\"case\":\"%s\",\"error\":\"%s\"}\n", name, e.what());
	}
}

// A random but valid frame graph: `count` passes, each writing one or two fresh images (colour outputs on graphics passes,
// storage images on compute passes; a handful of size classes and formats so that aliasing candidates exist) and reading up
// to three images written by earlier passes; a final graphics pass composes the backbuffer from a few of them.  Some passes
// end up unreferenced and must be culled.  Printed with the declaration so that the checker knows every edge.
struct Execution
{
	RenderGraph *graph = nullptr;
	const unsigned *frame = nullptr;
	bool force_graphics = false; // every pass on the graphics queue: the serial reference run
};

static void mix_pass(RenderGraph &graph, HIP::CommandBuffer &cmd, const std::vector<RenderTextureResource *> &outs,
                     const std::vector<RenderTextureResource *> &ins, unsigned salt, RenderTextureResource *feedback = nullptr)
{
	const void *in_ptr[4] = {};
	size_t in_dwords[4] = {};
	unsigned n = 0;
	if (feedback)
	{
		// null on the first frame (render_graph.cpp get_physical_history_texture_resource): the pass then has one input less
		if (auto *history = graph.get_physical_history_texture_resource(*feedback))
		{
			in_ptr[n] = history->get_device_pointer();
			in_dwords[n++] = history->get_size_bytes() / 4;
		}
	}
	for (auto *r : ins)
	{
		if (n == 4)
			break;
		auto &view = graph.get_physical_texture_resource(*r);
		in_ptr[n] = view.get_device_pointer();
		in_dwords[n++] = view.get_size_bytes() / 4;
	}
	unsigned k = 0;
	for (auto *r : outs)
	{
		auto &view = graph.get_physical_texture_resource(*r);
		cmd.check(gr_debug_mix(cmd.get_context(), cmd.get_stream(), view.get_device_pointer(), view.get_size_bytes() / 4, in_ptr, in_dwords, n,
		                       salt * 31u + k++), "debug mix");
	}
}

static std::string declare_random(RenderGraph &graph, unsigned seed, bool alias, const Execution *exec)
{
	std::mt19937 rng(seed);
	auto pick = [&](unsigned n) { return unsigned(rng() % n); };
	graph.set_backbuffer_dimensions(backbuffer(1280, 720));
	graph.set_alias_disjoint_images(alias);
	const unsigned count = 4 + pick(12);
	struct Produced { std::string name; bool color; unsigned size_class; };
	std::vector<Produced> produced;
	std::string decl = "[";
	static const float scales[3] = {1.0f, 0.5f, 0.25f};
	static const VkFormat formats[2] = {VK_FORMAT_R16G16B16A16_SFLOAT, VK_FORMAT_R8G8B8A8_UNORM};
	for (unsigned i = 0; i < count; i++)
	{
		const unsigned kind = pick(4); // 0, 1: graphics, 2: compute, 3: async compute
		const RenderGraphQueueFlagBits declared_queue = kind < 2 ? RENDER_GRAPH_QUEUE_GRAPHICS_BIT : kind == 2 ? RENDER_GRAPH_QUEUE_COMPUTE_BIT : RENDER_GRAPH_QUEUE_ASYNC_COMPUTE_BIT;
		const RenderGraphQueueFlagBits queue = exec && exec->force_graphics ? RENDER_GRAPH_QUEUE_GRAPHICS_BIT : declared_queue;
		std::vector<RenderTextureResource *> ins, outs;
		const std::string name = "p" + std::to_string(i);
		auto &pass = graph.add_pass(name, queue);
		decl += std::string(i ? "," : "") + "{\"name\":\"" + name + "\",\"queue\":" + std::to_string(unsigned(queue)) + ",\"reads\":[";
		const unsigned reads = produced.empty() ? 0 : pick(4);
		std::vector<std::string> seen;
		for (unsigned r = 0; r < reads; r++)
		{
			const auto &src = produced[pick(unsigned(produced.size()))];
			bool dup = false;
			for (auto &n : seen)
				dup = dup || n == src.name;
			if (dup)
				continue;
			ins.push_back(&pass.add_texture_input(src.name));
			decl += std::string(seen.empty() ? "" : ",") + "\"" + src.name + "\"";
			seen.push_back(src.name);
		}
		decl += "],\"writes\":[";
		const unsigned writes = 1 + pick(2);
		const unsigned size_class = pick(3); // colour outputs of one pass must agree in size
		for (unsigned w = 0; w < writes; w++)
		{
			AttachmentInfo info;
			info.size_class = SizeClass::SwapchainRelative;
			info.size_x = info.size_y = scales[size_class];
			info.format = formats[pick(2)];
			const std::string out = name + "-o" + std::to_string(w);
			if (kind < 2)
				outs.push_back(&pass.add_color_output(out, info));
			else
				outs.push_back(&pass.add_storage_texture_output(out, info));
			produced.push_back({out, kind < 2, size_class});
			decl += std::string(w ? "," : "") + "\"" + out + "\"";
		}
		// Feedback like the bloom chain's: every third pass also reads last frame's version of its own first output.
		RenderTextureResource *feedback = pick(3) == 0 ? &pass.add_history_input(name + "-o0") : nullptr;
		decl += std::string("],\"feedback\":") + (feedback ? "true" : "false") + "}";
		if (exec)
			pass.set_build_render_pass([exec, outs, ins, i, feedback](HIP::CommandBuffer &cmd) {
				mix_pass(*exec->graph, cmd, outs, ins, *exec->frame * 1000u + i, feedback);
			});
	}
	auto &final_pass = graph.add_pass("final", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
	AttachmentInfo back;
	std::vector<RenderTextureResource *> final_ins, final_outs;
	final_outs.push_back(&final_pass.add_color_output("back", back));
	decl += ",{\"name\":\"final\",\"queue\":1,\"reads\":[";
	const unsigned taps = 1 + pick(3);
	std::vector<std::string> seen;
	for (unsigned r = 0; r < taps; r++)
	{
		// bias towards late results so that most of the graph stays alive
		const unsigned lo = unsigned(produced.size()) > 4 ? unsigned(produced.size()) - 4 : 0;
		const auto &src = produced[lo + pick(unsigned(produced.size()) - lo)];
		bool dup = false;
		for (auto &n : seen)
			dup = dup || n == src.name;
		if (dup)
			continue;
		final_ins.push_back(&final_pass.add_texture_input(src.name));
		decl += std::string(seen.empty() ? "" : ",") + "\"" + src.name + "\"";
		seen.push_back(src.name);
	}
	decl += "],\"writes\":[\"back\"]}]";
	if (exec)
		final_pass.set_build_render_pass([exec, final_outs, final_ins](HIP::CommandBuffer &cmd) { mix_pass(*exec->graph, cmd, final_outs, final_ins, *exec->frame * 1000u + 999u); });
	graph.set_backbuffer_source("back");
	return decl;
}

static void random_graph(unsigned seed, bool alias)
{
	RenderGraph graph;
	const std::string decl = declare_random(graph, seed, alias, nullptr);
	graph.bake();
	printf("{\"case\":\"random-%u-%d\",\"declared\":%s,\"graph\":%s}\n", seed, int(alias), decl.c_str(), graph.dump_json().c_str());
}

// Runs the random graph of `seed` for four frames without any host synchronisation in between and returns one hash per frame
// of the swapchain image.  serial = every pass on the graphics queue, nothing hoisted, nothing aliased: one in-order stream.
static std::vector<uint64_t> execute_random(HIP::Device &device, unsigned seed, bool serial)
{
	RenderGraph graph;
	unsigned frame = 0;
	Execution exec;
	exec.graph = &graph;
	exec.frame = &frame;
	exec.force_graphics = serial;
	graph.set_device(&device);
	declare_random(graph, seed, !serial, &exec);
	if (serial)
		graph.set_hoist_independent_compute(false);
	graph.bake();
	std::vector<HIP::ImageHandle> swapchain;
	for (unsigned i = 0; i < 4; i++)
		swapchain.push_back(device.create_image(1280, 720, VK_FORMAT_R8G8B8A8_SRGB, "swapchain-" + std::to_string(i)));
	TaskComposer composer;
	for (frame = 0; frame < 4; frame++)
	{
		graph.setup_attachments(device, swapchain[frame].get());
		graph.enqueue_render_passes(device, composer);
	}
	device.wait_idle();
	std::vector<uint64_t> hashes;
	std::vector<uint32_t> host(1280 * 720);
	for (unsigned i = 0; i < 4; i++)
	{
		if (gr_download(device.get_context(), nullptr, host.data(), swapchain[i]->get_device_pointer(), host.size() * 4) < 0 ||
		    gr_sync(device.get_context(), nullptr) < 0)
			throw std::runtime_error(gr_last_error(device.get_context()));
		uint64_t h = 1469598103934665603ull;
		for (uint32_t v : host)
			h = (h ^ v) * 1099511628211ull;
		hashes.push_back(h);
	}
	return hashes;
}

int main(int argc, char **argv)
{
	if (argc >= 3 && std::string(argv[1]) == "--random")
	{
		const unsigned n = unsigned(atoi(argv[2]));
		for (unsigned seed = 0; seed < n; seed++)
		{
			random_graph(seed, true);
			random_graph(seed, false);
		}
		return 0;
	}
	if (argc >= 3 && std::string(argv[1]) == "--execute")
	{
		HIP::Device device(0);
		const unsigned n = unsigned(atoi(argv[2]));
		for (unsigned seed = 0; seed < n; seed++)
		{
			const auto pipelined = execute_random(device, seed, false), serial = execute_random(device, seed, true);
			printf("{\"case\":\"execute-%u\",\"pipelined\":[", seed);
			for (size_t i = 0; i < pipelined.size(); i++)
				printf("%s\"%016llx\"", i ? "," : "", (unsigned long long)pipelined[i]);
			printf("],\"serial\":[");
			for (size_t i = 0; i < serial.size(); i++)
				printf("%s\"%016llx\"", i ? "," : "", (unsigned long long)serial[i]);
			printf("]}\n");
		}
		return 0;
	}
	{
		RenderGraph graph;
		graph.set_backbuffer_dimensions(backbuffer(1920, 1080));
		AttachmentInfo back; // swapchain relative, swapchain format
		AttachmentInfo im;
		im.format = VK_FORMAT_R8G8B8A8_UNORM;
		im.size_x = 1280.0f;
		im.size_y = 720.0f;
		im.size_class = SizeClass::Absolute;

		auto &depth = graph.add_pass("depth", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		depth.add_color_output("depth", back);
		auto &graphics = graph.add_pass("first", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		auto &first = graphics.add_color_output("first", back);
		graphics.add_texture_input("depth");
		auto &compute = graph.add_pass("compute", RENDER_GRAPH_QUEUE_ASYNC_COMPUTE_BIT);
		auto &image = compute.add_storage_texture_output("image", im);
		compute.add_texture_input("first");
		auto &swap = graph.add_pass("final", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		swap.add_color_output("back", back);
		swap.add_texture_input("image");
		swap.add_texture_input("first");
		// add_pass is idempotent by name; resource references stay valid
		if (&graph.add_pass("first", RENDER_GRAPH_QUEUE_GRAPHICS_BIT) != &graphics)
			return 2;
		if (&graph.get_texture_resource("first") != &first || &graph.get_texture_resource("image") != &image)
			return 3;
		graph.set_backbuffer_source("back");
		graph.bake();
		print_case("sandbox", graph.dump_json());
	}
	{
		// A pass nobody needs is culled; an unused branch does not survive the back-to-front walk.
		RenderGraph graph;
		graph.set_backbuffer_dimensions(backbuffer(640, 360));
		AttachmentInfo info;
		auto &a = graph.add_pass("a", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		a.add_color_output("a-out", info);
		auto &dead = graph.add_pass("dead", RENDER_GRAPH_QUEUE_COMPUTE_BIT);
		AttachmentInfo half;
		half.size_x = half.size_y = 0.5f;
		half.format = VK_FORMAT_R16G16B16A16_SFLOAT;
		dead.add_storage_texture_output("dead-out", half);
		dead.add_texture_input("a-out");
		auto &b = graph.add_pass("b", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		b.add_color_output("b-out", info);
		b.add_texture_input("a-out");
		graph.set_backbuffer_source("b-out");
		graph.bake();
		print_case("culling", graph.dump_json());
	}
	{
		// Read-modify-write chain: both names share one physical image; InputRelative sizes follow ceil(input * scale).
		RenderGraph graph;
		graph.set_backbuffer_dimensions(backbuffer(1001, 333));
		AttachmentInfo hdr;
		hdr.format = VK_FORMAT_R16G16B16A16_SFLOAT;
		auto &fill = graph.add_pass("fill", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		fill.add_color_output("base", hdr);
		auto &add = graph.add_pass("add", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		add.add_color_output("sum", hdr, "base");
		AttachmentInfo quarter;
		quarter.format = VK_FORMAT_R16G16B16A16_SFLOAT;
		quarter.size_class = SizeClass::InputRelative;
		quarter.size_relative_name = "sum";
		quarter.size_x = quarter.size_y = 0.25f;
		auto &down = graph.add_pass("down", RENDER_GRAPH_QUEUE_COMPUTE_BIT);
		down.add_storage_texture_output("small", quarter);
		down.add_texture_input("sum");
		AttachmentInfo out;
		auto &present = graph.add_pass("present", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		present.add_color_output("screen", out);
		present.add_texture_input("small");
		present.add_texture_input("sum");
		graph.set_backbuffer_source("screen");
		graph.bake();
		print_case("rmw", graph.dump_json());
	}
	{
		// Attachment images of identical geometry with disjoint lifetimes share one allocation (build_aliases); storage
		// images never do.  "c" moves into a's allocation; "d" overlaps c (pass p3 reads c while writing d), so it must
		// not follow it there even though it is disjoint with a itself -- it lands in b's.
		RenderGraph graph;
		graph.set_backbuffer_dimensions(backbuffer(512, 256));
		AttachmentInfo half;
		half.format = VK_FORMAT_R16G16B16A16_SFLOAT;
		half.size_x = half.size_y = 0.5f;
		const char *names[] = {"a", "b", "c", "d"};
		for (unsigned i = 0; i < 4; i++)
		{
			auto &pass = graph.add_pass(std::string("p") + std::to_string(i), RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
			pass.add_color_output(names[i], half);
			if (i)
				pass.add_texture_input(names[i - 1]);
		}
		auto &store = graph.add_pass("p4", RENDER_GRAPH_QUEUE_COMPUTE_BIT);
		store.add_storage_texture_output("s", half);
		store.add_texture_input("d");
		auto &present = graph.add_pass("p5", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		AttachmentInfo out;
		present.add_color_output("screen", out);
		present.add_texture_input("s");
		graph.set_backbuffer_source("screen");
		graph.bake();
		print_case("aliasing-pipelined", graph.dump_json());
		graph.set_hoist_independent_compute(false); // whole frame on one stream
		graph.bake();
		print_case("aliasing", graph.dump_json());
		graph.set_alias_disjoint_images(false);
		graph.bake();
		print_case("aliasing-off", graph.dump_json());
	}
	print_error("no-writer", []() {
		RenderGraph graph;
		graph.set_backbuffer_dimensions(backbuffer(64, 64));
		auto &p = graph.add_pass("p", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		AttachmentInfo info;
		p.add_color_output("out", info);
		p.add_texture_input("never-written");
		graph.set_backbuffer_source("out");
		graph.bake();
	});
	print_error("missing-backbuffer", []() {
		RenderGraph graph;
		graph.set_backbuffer_dimensions(backbuffer(64, 64));
		auto &p = graph.add_pass("p", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		AttachmentInfo info;
		p.add_color_output("out", info);
		graph.set_backbuffer_source("nope");
		graph.bake();
	});
	print_error("cycle", []() {
		RenderGraph graph;
		graph.set_backbuffer_dimensions(backbuffer(64, 64));
		AttachmentInfo info;
		auto &a = graph.add_pass("a", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		a.add_color_output("a-out", info);
		a.add_texture_input("b-out");
		auto &b = graph.add_pass("b", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		b.add_color_output("b-out", info);
		b.add_texture_input("a-out");
		graph.set_backbuffer_source("b-out");
		graph.bake();
	});
	print_error("rmw-size-mismatch", []() {
		RenderGraph graph;
		graph.set_backbuffer_dimensions(backbuffer(64, 64));
		AttachmentInfo full, half;
		half.size_x = half.size_y = 0.5f;
		auto &a = graph.add_pass("a", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		a.add_color_output("a-out", half);
		auto &b = graph.add_pass("b", RENDER_GRAPH_QUEUE_GRAPHICS_BIT);
		b.add_color_output("b-out", full, "a-out");
		graph.set_backbuffer_source("b-out");
		graph.bake();
	});
	print_error("texture-as-buffer", []() {
		RenderGraph graph;
		graph.get_texture_resource("x");
		graph.get_buffer_resource("x");
	});
	return 0;
}
