// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// Granite::RenderGraph on HIP streams — see render_graph.hpp for what is kept from / dropped against
// renderer/render_graph.cpp.  Reference line numbers are cited per function.
#include "render_graph.hpp"
#include "timeline_trace.hpp"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>

namespace Granite
{
static constexpr RenderGraphQueueFlags compute_queues = RENDER_GRAPH_QUEUE_ASYNC_COMPUTE_BIT | RENDER_GRAPH_QUEUE_COMPUTE_BIT;

// ---------------------------------------------------------------------------------------------------------------------
// RenderPass declarations (render_graph.cpp:87-421).  Each call records: the resource's queue set, which passes read /
// write it, its usage bits, and the pass-side slot it occupies.
// ---------------------------------------------------------------------------------------------------------------------
template <typename Res>
static Res &declare_read(Res &res, RenderGraphQueueFlagBits queue, unsigned pass)
{
	res.add_queue(queue);
	res.read_in_pass(pass);
	return res;
}

template <typename Res>
static Res &declare_write(Res &res, RenderGraphQueueFlagBits queue, unsigned pass)
{
	res.add_queue(queue);
	res.written_in_pass(pass);
	return res;
}

void RenderPass::add_proxy_output(const std::string &name, VkPipelineStageFlags2 stages, VkAccessFlags2 access, const std::string &input)
{
	AccessedProxyResource acc;
	acc.proxy = &declare_write(graph.get_proxy_resource(name), queue, index);
	acc.stages = stages;
	acc.access = access;
	if (!input.empty())
		acc.alias_input = &declare_read(graph.get_proxy_resource(input), queue, index);
	proxy_outputs.push_back(acc);
}

void RenderPass::add_proxy_input(const std::string &name, VkPipelineStageFlags2 stages, VkAccessFlags2 access)
{
	AccessedProxyResource acc;
	acc.proxy = &declare_read(graph.get_proxy_resource(name), queue, index);
	acc.stages = stages;
	acc.access = access;
	proxy_inputs.push_back(acc);
}

RenderTextureResource &RenderPass::add_attachment_input(const std::string &name)
{
	auto &res = declare_read(graph.get_texture_resource(name), queue, index);
	res.add_image_usage(VK_IMAGE_USAGE_INPUT_ATTACHMENT_BIT);
	attachments_inputs.push_back(&res);
	return res;
}

RenderTextureResource &RenderPass::add_history_input(const std::string &name)
{
	// Not a read in this frame's dependency sense: it names LAST frame's contents (render_graph.cpp:97-105).
	auto &res = graph.get_texture_resource(name);
	res.add_queue(queue);
	res.add_image_usage(VK_IMAGE_USAGE_SAMPLED_BIT);
	history_inputs.push_back(&res);
	return res;
}

RenderBufferResource &RenderPass::add_generic_buffer_input(const std::string &name, VkPipelineStageFlags2 stages, VkAccessFlags2 access,
                                                           VkBufferUsageFlags usage)
{
	auto &res = declare_read(graph.get_buffer_resource(name), queue, index);
	res.add_buffer_usage(usage);
	AccessedBufferResource acc;
	acc.buffer = &res;
	acc.access = access;
	acc.stages = stages;
	generic_buffer.push_back(acc);
	return res;
}

RenderBufferResource &RenderPass::add_vertex_buffer_input(const std::string &name)
{
	return add_generic_buffer_input(name, 0, 0, VK_BUFFER_USAGE_VERTEX_BUFFER_BIT);
}

RenderBufferResource &RenderPass::add_index_buffer_input(const std::string &name)
{
	return add_generic_buffer_input(name, 0, 0, VK_BUFFER_USAGE_INDEX_BUFFER_BIT);
}

RenderBufferResource &RenderPass::add_indirect_buffer_input(const std::string &name)
{
	return add_generic_buffer_input(name, 0, 0, VK_BUFFER_USAGE_INDIRECT_BUFFER_BIT);
}

static VkPipelineStageFlags2 default_stage(RenderGraphQueueFlagBits queue, VkPipelineStageFlags2 stages)
{
	if (stages != 0)
		return stages;
	return (queue & compute_queues) != 0 ? VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT : VK_PIPELINE_STAGE_FRAGMENT_SHADER_BIT;
}

RenderBufferResource &RenderPass::add_uniform_input(const std::string &name, VkPipelineStageFlags2 stages)
{
	return add_generic_buffer_input(name, default_stage(queue, stages), VK_ACCESS_UNIFORM_READ_BIT, VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT);
}

RenderBufferResource &RenderPass::add_storage_read_only_input(const std::string &name, VkPipelineStageFlags2 stages)
{
	return add_generic_buffer_input(name, default_stage(queue, stages), VK_ACCESS_2_SHADER_STORAGE_READ_BIT,
	                                VK_BUFFER_USAGE_STORAGE_BUFFER_BIT);
}

RenderBufferResource &RenderPass::add_storage_output(const std::string &name, const BufferInfo &info, const std::string &input)
{
	auto &res = declare_write(graph.get_buffer_resource(name), queue, index);
	res.set_buffer_info(info);
	res.add_buffer_usage(VK_BUFFER_USAGE_STORAGE_BUFFER_BIT);
	storage_outputs.push_back(&res);
	RenderBufferResource *rmw = nullptr;
	if (!input.empty())
	{
		rmw = &declare_read(graph.get_buffer_resource(input), queue, index);
		rmw->add_buffer_usage(VK_BUFFER_USAGE_STORAGE_BUFFER_BIT);
	}
	storage_inputs.push_back(rmw);
	return res;
}

RenderBufferResource &RenderPass::add_transfer_output(const std::string &name, const BufferInfo &info)
{
	auto &res = declare_write(graph.get_buffer_resource(name), queue, index);
	res.set_buffer_info(info);
	res.add_buffer_usage(VK_BUFFER_USAGE_TRANSFER_DST_BIT);
	transfer_outputs.push_back(&res);
	return res;
}

RenderTextureResource &RenderPass::add_texture_input(const std::string &name, VkPipelineStageFlags2 stages)
{
	auto &res = declare_read(graph.get_texture_resource(name), queue, index);
	res.add_image_usage(VK_IMAGE_USAGE_SAMPLED_BIT);
	// Duplicate add_texture_input of one resource is allowed and collapses (render_graph.cpp:208-214).
	for (auto &acc : generic_texture)
		if (acc.texture == &res)
			return res;
	AccessedTextureResource acc;
	acc.texture = &res;
	acc.access = VK_ACCESS_2_SHADER_SAMPLED_READ_BIT;
	acc.stages = default_stage(queue, stages);
	generic_texture.push_back(acc);
	return res;
}

RenderTextureResource &RenderPass::add_resolve_output(const std::string &name, const AttachmentInfo &info)
{
	auto &res = declare_write(graph.get_texture_resource(name), queue, index);
	res.set_attachment_info(info);
	res.add_image_usage(VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT);
	resolve_outputs.push_back(&res);
	return res;
}

RenderTextureResource &RenderPass::add_color_output(const std::string &name, const AttachmentInfo &info, const std::string &input)
{
	auto &res = declare_write(graph.get_texture_resource(name), queue, index);
	res.set_attachment_info(info);
	res.add_image_usage(VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT);
	if (info.levels != 1)
		res.add_image_usage(VK_IMAGE_USAGE_TRANSFER_DST_BIT | VK_IMAGE_USAGE_TRANSFER_SRC_BIT);
	color_outputs.push_back(&res);
	RenderTextureResource *rmw = nullptr;
	if (!input.empty())
	{
		rmw = &declare_read(graph.get_texture_resource(input), queue, index);
		rmw->add_image_usage(VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT);
	}
	color_inputs.push_back(rmw);
	color_scale_inputs.push_back(nullptr);
	return res;
}

RenderTextureResource &RenderPass::add_storage_texture_output(const std::string &name, const AttachmentInfo &info, const std::string &input)
{
	auto &res = declare_write(graph.get_texture_resource(name), queue, index);
	res.set_attachment_info(info);
	res.add_image_usage(VK_IMAGE_USAGE_STORAGE_BIT);
	storage_texture_outputs.push_back(&res);
	RenderTextureResource *rmw = nullptr;
	if (!input.empty())
	{
		rmw = &declare_read(graph.get_texture_resource(input), queue, index);
		rmw->add_image_usage(VK_IMAGE_USAGE_STORAGE_BIT);
	}
	storage_texture_inputs.push_back(rmw);
	return res;
}

void RenderTextureResource::become_write_alias_of(const RenderTextureResource &other, unsigned writer_pass)
{
	info = other.info;
	image_usage = other.image_usage;
	reset_pass_sets(other.get_used_queues());
	written_in_pass(writer_pass);
}

void RenderPass::add_fake_resource_write_alias(const std::string &from, const std::string &to)
{
	auto &from_res = graph.get_texture_resource(from);
	auto &to_res = graph.get_texture_resource(to);
	to_res.become_write_alias_of(from_res, index);
	fake_resource_alias.emplace_back(&from_res, &to_res);
}

RenderTextureResource &RenderPass::set_depth_stencil_output(const std::string &name, const AttachmentInfo &info)
{
	auto &res = declare_write(graph.get_texture_resource(name), queue, index);
	res.set_attachment_info(info);
	res.add_image_usage(VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT);
	depth_stencil_output = &res;
	return res;
}

RenderTextureResource &RenderPass::set_depth_stencil_input(const std::string &name)
{
	auto &res = declare_read(graph.get_texture_resource(name), queue, index);
	res.add_image_usage(VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT);
	depth_stencil_input = &res;
	return res;
}

// ---------------------------------------------------------------------------------------------------------------------
// RenderGraph: name tables (render_graph.cpp:450-546)
// ---------------------------------------------------------------------------------------------------------------------
RenderTextureResource &RenderGraph::get_texture_resource(const std::string &name)
{
	auto itr = resource_to_index.find(name);
	if (itr != resource_to_index.end())
	{
		if (resources[itr->second]->get_type() != RenderResource::Type::Texture)
			throw std::logic_error("Resource is not a texture: " + name);
		return static_cast<RenderTextureResource &>(*resources[itr->second]);
	}
	unsigned index = unsigned(resources.size());
	resources.emplace_back(new RenderTextureResource(index));
	resources.back()->set_name(name);
	resource_to_index[name] = index;
	return static_cast<RenderTextureResource &>(*resources.back());
}

RenderBufferResource &RenderGraph::get_buffer_resource(const std::string &name)
{
	auto itr = resource_to_index.find(name);
	if (itr != resource_to_index.end())
	{
		if (resources[itr->second]->get_type() != RenderResource::Type::Buffer)
			throw std::logic_error("Resource is not a buffer: " + name);
		return static_cast<RenderBufferResource &>(*resources[itr->second]);
	}
	unsigned index = unsigned(resources.size());
	resources.emplace_back(new RenderBufferResource(index));
	resources.back()->set_name(name);
	resource_to_index[name] = index;
	return static_cast<RenderBufferResource &>(*resources.back());
}

RenderResource &RenderGraph::get_proxy_resource(const std::string &name)
{
	auto itr = resource_to_index.find(name);
	if (itr != resource_to_index.end())
	{
		if (resources[itr->second]->get_type() != RenderResource::Type::Proxy)
			throw std::logic_error("Resource is not a proxy: " + name);
		return *resources[itr->second];
	}
	unsigned index = unsigned(resources.size());
	resources.emplace_back(new RenderResource(RenderResource::Type::Proxy, index));
	resources.back()->set_name(name);
	resource_to_index[name] = index;
	return *resources.back();
}

RenderPass &RenderGraph::add_pass(const std::string &name, RenderGraphQueueFlagBits queue)
{
	auto itr = pass_to_index.find(name);
	if (itr != pass_to_index.end())
		return *passes[itr->second]; // idempotent by name
	unsigned index = unsigned(passes.size());
	passes.emplace_back(new RenderPass(*this, index, queue));
	passes.back()->set_name(name);
	pass_to_index[name] = index;
	return *passes.back();
}

RenderPass *RenderGraph::find_pass(const std::string &name)
{
	auto itr = pass_to_index.find(name);
	return itr != pass_to_index.end() ? passes[itr->second].get() : nullptr;
}

void RenderGraph::set_backbuffer_source(const std::string &name)
{
	backbuffer_source = name;
}

HIP::BufferHandle RenderGraph::consume_persistent_physical_buffer_resource(unsigned index) const
{
	if (index >= physical_buffers.size())
		return {};
	return physical_buffers[index];
}

void RenderGraph::install_persistent_physical_buffer_resource(unsigned index, HIP::BufferHandle buffer)
{
	if (index >= physical_buffers.size())
		throw std::logic_error("Out of range.");
	physical_buffers[index] = std::move(buffer);
}

RenderGraph::~RenderGraph()
{
	for (void *e : pass_done_event)
		if (e)
			(void)hipEventDestroy(static_cast<hipEvent_t>(e));
	for (void *e : event_pool)
		if (e)
			(void)hipEventDestroy(static_cast<hipEvent_t>(e));
}

void RenderGraph::reset()
{
	passes.clear();
	resources.clear();
	pass_to_index.clear();
	resource_to_index.clear();
	pass_stack.clear();
	physical_dimensions.clear();
	physical_attachments.clear();
	physical_buffers.clear();
	physical_image_attachments.clear();
	physical_history_image_attachments.clear();
	physical_image_has_history.clear();
	for (auto &v : physical_images_alternate)
		v.clear();
	for (auto &v : physical_buffers_alternate)
		v.clear();
	swapchain_physical_index = RenderResource::Unused;
}

// ---------------------------------------------------------------------------------------------------------------------
// Size resolution (render_graph.cpp:3113-3191)
// ---------------------------------------------------------------------------------------------------------------------
ResourceDimensions RenderGraph::get_resource_dimensions(const RenderBufferResource &resource) const
{
	ResourceDimensions dim;
	auto &info = resource.get_buffer_info();
	dim.buffer_info = info;
	dim.buffer_info.usage |= resource.get_buffer_usage();
	dim.flags |= info.flags;
	dim.name = resource.get_name();
	return dim;
}

static unsigned scaled_extent(unsigned base, float scale)
{
	// ceil(base * scale) in fp32 like muglm::ceil(info.size_x * dim), at least 1.
	return std::max(unsigned(std::ceil(float(base) * scale)), 1u);
}

ResourceDimensions RenderGraph::get_resource_dimensions(const RenderTextureResource &resource) const
{
	ResourceDimensions dim;
	auto &info = resource.get_attachment_info();
	dim.layers = info.layers;
	dim.samples = info.samples;
	dim.format = info.format;
	dim.queues = resource.get_used_queues();
	dim.image_usage = info.aux_usage | resource.get_image_usage();
	dim.name = resource.get_name();
	dim.flags = info.flags & ~ATTACHMENT_INFO_SUPPORTS_PREROTATE_BIT; // no pre-rotation off a swapchain-less device

	switch (info.size_class)
	{
	case SizeClass::SwapchainRelative:
		dim.width = scaled_extent(swapchain_dimensions.width, info.size_x);
		dim.height = scaled_extent(swapchain_dimensions.height, info.size_y);
		dim.depth = std::max(unsigned(std::ceil(info.size_z)), 1u);
		break;

	case SizeClass::Absolute:
		dim.width = std::max(unsigned(info.size_x), 1u);
		dim.height = std::max(unsigned(info.size_y), 1u);
		dim.depth = std::max(unsigned(info.size_z), 1u);
		break;

	case SizeClass::InputRelative:
	{
		auto itr = resource_to_index.find(info.size_relative_name);
		if (itr == resource_to_index.end())
			throw std::logic_error("Resource does not exist.");
		auto &input = static_cast<RenderTextureResource &>(*resources[itr->second]);
		auto input_dim = get_resource_dimensions(input);
		dim.width = scaled_extent(input_dim.width, info.size_x);
		dim.height = scaled_extent(input_dim.height, info.size_y);
		dim.depth = std::max(unsigned(std::ceil(float(input_dim.depth) * info.size_z)), 1u);
		break;
	}
	}

	if (dim.format == VK_FORMAT_UNDEFINED)
		dim.format = swapchain_dimensions.format;

	unsigned max_dim = std::max(std::max(dim.width, dim.height), dim.depth);
	unsigned full_chain = 0;
	for (; max_dim; max_dim >>= 1)
		full_chain++;
	dim.levels = std::min(full_chain, info.levels == 0 ? ~0u : info.levels);
	return dim;
}

// ---------------------------------------------------------------------------------------------------------------------
// Validation (render_graph.cpp:562-622)
// ---------------------------------------------------------------------------------------------------------------------
void RenderGraph::validate_passes()
{
	for (auto &pass_ptr : passes)
	{
		auto &pass = *pass_ptr;
		if (pass.get_color_inputs().size() != pass.get_color_outputs().size())
			throw std::logic_error("Size of color inputs must match color outputs.");
		if (pass.get_storage_inputs().size() != pass.get_storage_outputs().size())
			throw std::logic_error("Size of storage inputs must match storage outputs.");
		if (pass.get_storage_texture_inputs().size() != pass.get_storage_texture_outputs().size())
			throw std::logic_error("Size of storage texture inputs must match storage texture outputs.");
		if (!pass.get_resolve_outputs().empty() && pass.get_resolve_outputs().size() != pass.get_color_outputs().size())
			throw std::logic_error("Must have one resolve output for each color output.");

		for (unsigned i = 0; i < pass.get_color_inputs().size(); i++)
		{
			auto *in = pass.get_color_inputs()[i];
			if (in && get_resource_dimensions(*in) != get_resource_dimensions(*pass.get_color_outputs()[i]))
				pass.make_color_input_scaled(i);
		}

		for (unsigned i = 0; i < pass.get_storage_outputs().size(); i++)
		{
			auto *in = pass.get_storage_inputs()[i];
			if (in && pass.get_storage_outputs()[i]->get_buffer_info() != in->get_buffer_info())
				throw std::logic_error("Doing RMW on a storage buffer, but usage and sizes do not match.");
		}

		for (unsigned i = 0; i < pass.get_storage_texture_outputs().size(); i++)
		{
			auto *in = pass.get_storage_texture_inputs()[i];
			if (in && get_resource_dimensions(*pass.get_storage_texture_outputs()[i]) != get_resource_dimensions(*in))
				throw std::logic_error("Doing RMW on a storage texture image, but sizes do not match.");
		}

		if (pass.get_depth_stencil_input() && pass.get_depth_stencil_output())
			if (get_resource_dimensions(*pass.get_depth_stencil_input()) != get_resource_dimensions(*pass.get_depth_stencil_output()))
				throw std::logic_error("Dimension mismatch.");
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// Dependency walk (render_graph.cpp:2767-2870)
// ---------------------------------------------------------------------------------------------------------------------
void RenderGraph::depend_passes_recursive(const RenderPass &self, const std::unordered_set<unsigned> &written_passes, unsigned stack_count,
                                          bool no_check, bool ignore_self, bool merge_dependency)
{
	if (!no_check && written_passes.empty())
		throw std::logic_error("No pass exists which writes to resource.");
	if (stack_count > passes.size())
		throw std::logic_error("Cycle detected.");

	for (unsigned writer : written_passes)
	{
		if (writer == self.get_index())
			continue;
		pass_dependencies[self.get_index()].insert(writer);
		if (merge_dependency)
			pass_merge_dependencies[self.get_index()].insert(writer);
	}

	stack_count++;
	for (unsigned writer : written_passes)
	{
		if (writer == self.get_index())
		{
			if (ignore_self)
				continue;
			throw std::logic_error("Pass depends on itself.");
		}
		pass_stack.push_back(writer);
		traverse_dependencies(*passes[writer], stack_count);
	}
}

void RenderGraph::traverse_dependencies(const RenderPass &pass, unsigned stack_count)
{
	// Attachment-style inputs first (they were "merge" candidates in the reference; here the flag only feeds the
	// reorder heuristic so baked orders stay comparable).
	if (auto *ds = pass.get_depth_stencil_input())
		depend_passes_recursive(pass, ds->get_write_passes(), stack_count, false, false, true);

	for (auto *input : pass.get_attachment_inputs())
	{
		bool self_dependency = pass.get_depth_stencil_output() == input;
		auto &outs = pass.get_color_outputs();
		if (std::find(outs.begin(), outs.end(), input) != outs.end())
			self_dependency = true;
		if (!self_dependency)
			depend_passes_recursive(pass, input->get_write_passes(), stack_count, false, false, true);
	}

	for (auto *input : pass.get_color_inputs())
		if (input)
			depend_passes_recursive(pass, input->get_write_passes(), stack_count, false, false, true);
	for (auto *input : pass.get_color_scale_inputs())
		if (input)
			depend_passes_recursive(pass, input->get_write_passes(), stack_count, false, false, false);
	for (auto &input : pass.get_generic_texture_inputs())
		depend_passes_recursive(pass, input.texture->get_write_passes(), stack_count, false, false, false);

	for (auto *input : pass.get_storage_inputs())
	{
		if (!input)
			continue;
		// Feedback buffers may have no writer; readers of the old value must run before this RMW (WAR).
		depend_passes_recursive(pass, input->get_write_passes(), stack_count, true, false, false);
		depend_passes_recursive(pass, input->get_read_passes(), stack_count, true, true, false);
	}

	for (auto *input : pass.get_storage_texture_inputs())
		if (input)
			depend_passes_recursive(pass, input->get_write_passes(), stack_count, false, false, false);

	for (auto &input : pass.get_generic_buffer_inputs())
		depend_passes_recursive(pass, input.buffer->get_write_passes(), stack_count, true, false, false);

	// Proxies: ordering edges only (render_graph.cpp:2802-2812); a proxy RMW is ordered like a storage-buffer RMW.
	for (auto &input : pass.get_proxy_inputs())
		depend_passes_recursive(pass, input.proxy->get_write_passes(), stack_count, false, false, false);
	for (auto &output : pass.get_proxy_outputs())
		if (output.alias_input)
		{
			depend_passes_recursive(pass, output.alias_input->get_write_passes(), stack_count, true, false, false);
			depend_passes_recursive(pass, output.alias_input->get_read_passes(), stack_count, true, true, false);
		}
}

bool RenderGraph::depends_on_pass(unsigned dst_pass, unsigned src_pass)
{
	if (dst_pass == src_pass)
		return true;
	for (unsigned dep : pass_dependencies[dst_pass])
		if (depends_on_pass(dep, src_pass))
			return true;
	return false;
}

void RenderGraph::filter_passes(std::vector<unsigned> &list)
{
	std::unordered_set<unsigned> seen;
	std::vector<unsigned> unique;
	unique.reserve(list.size());
	for (unsigned p : list)
		if (seen.insert(p).second)
			unique.push_back(p);
	list.swap(unique);
}

// Greedy list scheduling (render_graph.cpp:2872-2977): next pass = the schedulable one that leaves the most already
// scheduled passes between itself and its nearest dependency (maximises overlap), ties resolved by original order.
void RenderGraph::reorder_passes(std::vector<unsigned> &flattened)
{
	for (unsigned pass_index = 0; pass_index < pass_merge_dependencies.size(); pass_index++)
	{
		auto &deps = pass_dependencies[pass_index];
		for (unsigned merge_dep : pass_merge_dependencies[pass_index])
			for (unsigned dependee : deps)
				if (!depends_on_pass(dependee, merge_dep) && merge_dep != dependee)
					pass_dependencies[merge_dep].insert(dependee);
	}

	if (flattened.size() <= 2)
		return;

	std::vector<unsigned> pending;
	pending.swap(flattened);
	flattened.reserve(pending.size());
	auto take = [&](size_t i) {
		flattened.push_back(pending[i]);
		pending.erase(pending.begin() + ptrdiff_t(i));
	};
	take(0);

	while (!pending.empty())
	{
		size_t best = 0;
		unsigned best_overlap = 0;
		for (size_t i = 0; i < pending.size(); i++)
		{
			unsigned overlap = 0;
			if (pass_merge_dependencies[pending[i]].count(flattened.back()))
				overlap = ~0u;
			else
			{
				for (auto itr = flattened.rbegin(); itr != flattened.rend(); ++itr)
				{
					if (depends_on_pass(pending[i], *itr))
						break;
					overlap++;
				}
			}
			if (overlap <= best_overlap)
				continue;

			bool blocked = false;
			for (size_t j = 0; j < i && !blocked; j++)
				blocked = depends_on_pass(pending[i], pending[j]);
			if (blocked)
				continue;
			best = i;
			best_overlap = overlap;
		}
		take(best);
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// Physical index assignment (render_graph.cpp:624-952): walk the baked order; first touch allocates a physical slot,
// RMW outputs take their input's slot, later touches OR in queue/usage bits.
// ---------------------------------------------------------------------------------------------------------------------
void RenderGraph::build_physical_resources()
{
	unsigned phys_index = 0;
	physical_dimensions.clear();

	auto touch_texture = [&](RenderTextureResource *res, VkImageUsageFlags extra_usage = 0) {
		if (res->get_physical_index() == RenderResource::Unused)
		{
			physical_dimensions.push_back(get_resource_dimensions(*res));
			res->set_physical_index(phys_index++);
		}
		else
		{
			auto &dim = physical_dimensions[res->get_physical_index()];
			dim.queues |= res->get_used_queues();
			dim.image_usage |= res->get_image_usage();
		}
		physical_dimensions[res->get_physical_index()].image_usage |= extra_usage;
	};
	auto touch_buffer = [&](RenderBufferResource *res) {
		if (res->get_physical_index() == RenderResource::Unused)
		{
			physical_dimensions.push_back(get_resource_dimensions(*res));
			res->set_physical_index(phys_index++);
		}
		else
		{
			auto &dim = physical_dimensions[res->get_physical_index()];
			dim.queues |= res->get_used_queues();
			dim.buffer_info.usage |= res->get_buffer_usage();
		}
	};
	auto touch_proxy = [&](RenderResource *res) {
		if (res->get_physical_index() == RenderResource::Unused)
		{
			ResourceDimensions dim;
			dim.flags |= ATTACHMENT_INFO_INTERNAL_PROXY_BIT; // no memory behind it (render_graph.cpp:758-767)
			dim.name = res->get_name();
			dim.queues = res->get_used_queues();
			physical_dimensions.push_back(dim);
			res->set_physical_index(phys_index++);
		}
		else
			physical_dimensions[res->get_physical_index()].queues |= res->get_used_queues();
	};
	auto alias_output = [&](RenderResource *output, RenderResource *input) {
		if (output->get_physical_index() == RenderResource::Unused)
			output->set_physical_index(input->get_physical_index());
		else if (output->get_physical_index() != input->get_physical_index())
			throw std::logic_error("Cannot alias resources. Index already claimed.");
	};

	for (unsigned pass_index : pass_stack)
	{
		auto &pass = *passes[pass_index];

		for (auto &input : pass.get_generic_texture_inputs())
			touch_texture(input.texture);
		for (auto &input : pass.get_generic_buffer_inputs())
			touch_buffer(input.buffer);
		for (auto *input : pass.get_color_scale_inputs())
			if (input)
				touch_texture(input, VK_IMAGE_USAGE_SAMPLED_BIT);

		for (unsigned i = 0; i < pass.get_color_inputs().size(); i++)
			if (auto *input = pass.get_color_inputs()[i])
			{
				touch_texture(input);
				alias_output(pass.get_color_outputs()[i], input);
			}
		for (unsigned i = 0; i < pass.get_storage_inputs().size(); i++)
			if (auto *input = pass.get_storage_inputs()[i])
			{
				touch_buffer(input);
				alias_output(pass.get_storage_outputs()[i], input);
			}
		for (unsigned i = 0; i < pass.get_storage_texture_inputs().size(); i++)
			if (auto *input = pass.get_storage_texture_inputs()[i])
			{
				touch_texture(input);
				alias_output(pass.get_storage_texture_outputs()[i], input);
			}

		for (auto &input : pass.get_proxy_inputs())
			touch_proxy(input.proxy);
		for (auto &output : pass.get_proxy_outputs())
			if (output.alias_input)
			{
				touch_proxy(output.alias_input);
				alias_output(output.proxy, output.alias_input);
			}

		for (auto *output : pass.get_color_outputs())
			touch_texture(output);
		for (auto *output : pass.get_resolve_outputs())
			touch_texture(output);
		for (auto &output : pass.get_proxy_outputs())
			touch_proxy(output.proxy);
		for (auto *output : pass.get_storage_outputs())
			touch_buffer(output);
		for (auto *output : pass.get_transfer_outputs())
			touch_buffer(output);
		for (auto *output : pass.get_storage_texture_outputs())
			touch_texture(output);

		auto *ds_output = pass.get_depth_stencil_output();
		auto *ds_input = pass.get_depth_stencil_input();
		if (ds_input)
		{
			touch_texture(ds_input);
			if (ds_output)
			{
				alias_output(ds_output, ds_input);
				auto &dim = physical_dimensions[ds_output->get_physical_index()];
				dim.queues |= ds_output->get_used_queues();
				dim.image_usage |= ds_output->get_image_usage();
			}
		}
		else if (ds_output)
			touch_texture(ds_output);

		// Input attachments last so they can pick up the slot of a colour/depth attachment of the same pass.
		for (auto *input : pass.get_attachment_inputs())
			touch_texture(input);
		for (auto &alias : pass.get_fake_resource_aliases())
			alias.second->set_physical_index(alias.first->get_physical_index());
	}

	physical_image_has_history.assign(physical_dimensions.size(), false);
	for (unsigned pass_index : pass_stack)
		for (auto *history : passes[pass_index]->get_history_inputs())
		{
			unsigned phys = history->get_physical_index();
			if (phys == RenderResource::Unused)
				throw std::logic_error("History input is used, but it was never written to.");
			physical_image_has_history[phys] = true;
		}
}

// ---------------------------------------------------------------------------------------------------------------------
// bake (render_graph.cpp:2993-3111)
// ---------------------------------------------------------------------------------------------------------------------
void RenderGraph::bake()
{
	for (auto &pass : passes)
		pass->setup_dependencies();

	validate_passes();

	auto itr = resource_to_index.find(backbuffer_source);
	if (itr == resource_to_index.end())
		throw std::logic_error("Backbuffer source does not exist.");

	pass_stack.clear();
	pass_dependencies.assign(passes.size(), {});
	pass_merge_dependencies.assign(passes.size(), {});
	for (auto &res : resources)
		res->set_physical_index(RenderResource::Unused);

	auto &backbuffer_resource = *resources[itr->second];
	if (backbuffer_resource.get_write_passes().empty())
		throw std::logic_error("No pass exists which writes to resource.");

	for (unsigned writer : backbuffer_resource.get_write_passes())
		pass_stack.push_back(writer);
	auto roots = pass_stack;
	for (unsigned root : roots)
		traverse_dependencies(*passes[root], 0);

	std::reverse(pass_stack.begin(), pass_stack.end());
	filter_passes(pass_stack);
	reorder_passes(pass_stack);
	build_physical_resources();

	// Backbuffer aliasing (render_graph.cpp:3048-3098): if the resource that feeds the backbuffer has the swapchain's
	// geometry and format it IS the externally provided swapchain image; otherwise it gets its own image and is copied
	// (same byte size) at the end of the frame.
	unsigned backbuffer_phys = backbuffer_resource.get_physical_index();
	auto &backbuffer_dim = physical_dimensions[backbuffer_phys];
	bool same_geometry = backbuffer_dim.width == swapchain_dimensions.width && backbuffer_dim.height == swapchain_dimensions.height &&
	                     backbuffer_dim.format == swapchain_dimensions.format;
	bool has_history = physical_image_has_history[backbuffer_phys];
	swapchain_physical_index = (same_geometry && !has_history) ? backbuffer_phys : unsigned(RenderResource::Unused);
	build_physical_passes();
	build_stream_assignment();
	build_aliases();

	if (device)
		for (unsigned pass_index : pass_stack)
			passes[pass_index]->setup(*device);
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-frame: attachments (render_graph.cpp:2577-2765)
// ---------------------------------------------------------------------------------------------------------------------
void RenderGraph::setup_physical_buffer(HIP::Device &device_, unsigned attachment)
{
	auto &att = physical_dimensions[attachment];
	auto &slot = physical_buffers[attachment];
	bool reuse = slot && (att.flags & ATTACHMENT_INFO_PERSISTENT_BIT) != 0 && slot->get_size() == att.buffer_info.size &&
	             (slot->get_usage() & att.buffer_info.usage) == att.buffer_info.usage;
	if (!reuse)
		slot = device_.create_buffer(att.buffer_info.size, att.buffer_info.usage, att.name); // zero-initialised
}

void RenderGraph::setup_physical_image(HIP::Device &device_, unsigned attachment)
{
	auto &att = physical_dimensions[attachment];
	auto &slot = physical_image_attachments[attachment];
	if (physical_aliases[attachment] != RenderResource::Unused)
	{
		// lower index => already set up this frame (render_graph.cpp:2612-2618)
		slot = physical_image_attachments[physical_aliases[attachment]];
		physical_attachments[attachment] = slot.get();
		return;
	}
	bool reuse = slot && (att.flags & ATTACHMENT_INFO_PERSISTENT_BIT) != 0 && slot->get_format() == att.format &&
	             slot->get_width() == att.width && slot->get_height() == att.height && slot->get_levels() == att.levels;
	if (!reuse)
		slot = device_.create_image(att.width, att.height, att.format, att.name, att.levels);
	physical_attachments[attachment] = slot.get();
}

void RenderGraph::setup_attachments(HIP::Device &device_, HIP::ImageView *swapchain)
{
	GRANITE_SCOPED_TIMELINE_EVENT("setup-attachments");
	const size_t count = physical_dimensions.size();
	physical_attachments.assign(count, nullptr);
	physical_buffers.resize(count);
	physical_image_attachments.resize(count);
	physical_history_image_attachments.resize(count);
	for (auto &v : physical_buffers_alternate)
		v.resize(count);
	for (auto &v : physical_images_alternate)
		v.resize(count);
	if (physical_sync.size() != count)
	{
		physical_sync.assign(count, {});
		for (auto &v : physical_sync_alternate)
			v.assign(count, {});
	}
	swapchain_attachment = swapchain;

	for (unsigned i = 0; i < count; i++)
	{
		// What was rendered last frame becomes this frame's history; the old history image (if any) is recycled as
		// the new render target.
		if (physical_image_has_history[i])
			std::swap(physical_history_image_attachments[i], physical_image_attachments[i]);

		auto &att = physical_dimensions[i];
		if (physical_buffer_is_double_buffered(i))
		{
			// rotate: current -> newest spare, oldest spare -> current
			constexpr int spares = HandOverCopies - 1;
			if (att.buffer_info.size != 0)
			{
				auto current = physical_buffers[i];
				physical_buffers[i] = physical_buffers_alternate[0][i];
				for (int k = 0; k + 1 < spares; k++)
					physical_buffers_alternate[k][i] = physical_buffers_alternate[k + 1][i];
				physical_buffers_alternate[spares - 1][i] = current;
			}
			else
			{
				auto current = physical_image_attachments[i];
				physical_image_attachments[i] = physical_images_alternate[0][i];
				for (int k = 0; k + 1 < spares; k++)
					physical_images_alternate[k][i] = physical_images_alternate[k + 1][i];
				physical_images_alternate[spares - 1][i] = current;
			}
			auto current_sync = physical_sync[i];
			physical_sync[i] = physical_sync_alternate[0][i];
			for (int k = 0; k + 1 < spares; k++)
				physical_sync_alternate[k][i] = physical_sync_alternate[k + 1][i];
			physical_sync_alternate[spares - 1][i] = current_sync;
		}
		if ((att.flags & ATTACHMENT_INFO_INTERNAL_PROXY_BIT) != 0)
			continue;
		if (att.buffer_info.size != 0)
			setup_physical_buffer(device_, i);
		else if (i == swapchain_physical_index && swapchain)
			physical_attachments[i] = swapchain;
		else
			setup_physical_image(device_, i);
	}
}

HIP::ImageView &RenderGraph::get_physical_texture_resource(unsigned index)
{
	if (index == RenderResource::Unused || index >= physical_attachments.size() || !physical_attachments[index])
		throw std::logic_error("Physical texture resource is not available (not baked / not set up / culled).");
	return *physical_attachments[index];
}

HIP::ImageView *RenderGraph::get_physical_history_texture_resource(unsigned index)
{
	if (index == RenderResource::Unused || index >= physical_history_image_attachments.size())
		return nullptr;
	return physical_history_image_attachments[index].get();
}

HIP::Buffer &RenderGraph::get_physical_buffer_resource(unsigned index)
{
	if (index == RenderResource::Unused || index >= physical_buffers.size() || !physical_buffers[index])
		throw std::logic_error("Physical buffer resource is not available (not baked / not set up / culled).");
	return *physical_buffers[index];
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-frame: execution (render_graph.cpp:2183-2575).  Every logical pass is a kernel sequence on the stream of its queue:
// GRAPHICS and COMPUTE share the generic stream (the reference maps plain compute onto the graphics queue too,
// render_graph.hpp:219-229); ASYNC_COMPUTE gets its own stream, ordered against the other by events at every switch.
// ---------------------------------------------------------------------------------------------------------------------
void *RenderGraph::acquire_event()
{
	if (!event_pool.empty())
	{
		void *e = event_pool.back();
		event_pool.pop_back();
		return e;
	}
	hipEvent_t e;
	if (hipEventCreate(&e) != hipSuccess)
		throw std::runtime_error("hipEventCreate failed");
	return e;
}

void RenderGraph::build_stream_assignment()
{
	pass_stream.assign(passes.size(), 0);
	pass_reads_physical.assign(passes.size(), {});
	pass_writes_physical.assign(passes.size(), {});
	uses_async_stream = false;

	auto add = [](std::vector<unsigned> &list, const RenderResource *res) {
		if (!res || res->get_physical_index() == RenderResource::Unused)
			return;
		if (std::find(list.begin(), list.end(), res->get_physical_index()) == list.end())
			list.push_back(res->get_physical_index());
	};

	for (unsigned pass_index : pass_stack)
	{
		auto &pass = *passes[pass_index];
		auto &reads = pass_reads_physical[pass_index];
		auto &writes = pass_writes_physical[pass_index];
		for (auto *r : pass.get_color_inputs()) add(reads, r);
		for (auto *r : pass.get_color_scale_inputs()) add(reads, r);
		for (auto *r : pass.get_storage_texture_inputs()) add(reads, r);
		for (auto *r : pass.get_attachment_inputs()) add(reads, r);
		for (auto *r : pass.get_history_inputs()) add(reads, r);
		for (auto *r : pass.get_storage_inputs()) add(reads, r);
		for (auto &r : pass.get_generic_texture_inputs()) add(reads, r.texture);
		for (auto &r : pass.get_generic_buffer_inputs()) add(reads, r.buffer);
		for (auto &r : pass.get_proxy_inputs()) add(reads, r.proxy);
		for (auto &r : pass.get_proxy_outputs()) add(reads, r.alias_input);
		for (auto &r : pass.get_proxy_outputs()) add(writes, r.proxy);
		add(reads, pass.get_depth_stencil_input());
		for (auto *r : pass.get_color_outputs()) add(writes, r);
		for (auto *r : pass.get_resolve_outputs()) add(writes, r);
		for (auto *r : pass.get_storage_texture_outputs()) add(writes, r);
		for (auto *r : pass.get_storage_outputs()) add(writes, r);
		for (auto *r : pass.get_transfer_outputs()) add(writes, r);
		add(writes, pass.get_depth_stencil_output());

		pass_stream[pass_index] = (pass.get_queue() & RENDER_GRAPH_QUEUE_ASYNC_COMPUTE_BIT) != 0 ? 1 : 0;
	}

	// Frame pipelining (HIP executor policy, set_hoist_independent_compute): the "front" of a frame is every pass that
	// does not depend, directly or through other passes, on anything carried over from the previous frame (history
	// inputs, buffers that are read before they are written within the frame) -- cluster build, G-buffer, lighting.  The
	// front runs on the second stream, so the front of frame N+1 executes while the back of frame N (bloom pyramid with
	// its feedback, exposure, tonemap, AA) is still in flight on the first stream.
	std::vector<bool> front(passes.size(), false);
	if (hoist_independent_compute)
	{
		std::vector<int> first_writer(physical_dimensions.size(), -1), any_writer(physical_dimensions.size(), 0);
		std::vector<int> order(passes.size(), -1);
		int position = 0;
		for (unsigned pass_index : pass_stack)
		{
			order[pass_index] = position++;
			for (unsigned w : pass_writes_physical[pass_index])
			{
				if (first_writer[w] < 0)
					first_writer[w] = int(pass_index);
				any_writer[w]++;
			}
		}
		std::vector<bool> written_by_back(physical_dimensions.size(), false);
		bool any_back = false;
		for (unsigned pass_index : pass_stack)
		{
			auto &pass = *passes[pass_index];
			bool ok = pass_stream[pass_index] == 0 && pass.get_history_inputs().empty() && !pass_writes_physical[pass_index].empty();
			for (unsigned w : pass_writes_physical[pass_index])
				ok = ok && w != swapchain_physical_index && !physical_image_has_history[w];
			for (unsigned r : pass_reads_physical[pass_index])
			{
				// Every producer of what it reads must already have run in this frame, on the front.
				const bool produced_before = first_writer[r] >= 0 && order[first_writer[r]] < order[pass_index];
				const bool rmw_of_own_output = std::find(pass_writes_physical[pass_index].begin(), pass_writes_physical[pass_index].end(), r) !=
				                               pass_writes_physical[pass_index].end();
				ok = ok && !written_by_back[r] && (produced_before || (any_writer[r] == 0)) && !(rmw_of_own_output && first_writer[r] == int(pass_index));
			}
			front[pass_index] = ok;
			if (!ok)
			{
				any_back = true;
				for (unsigned w : pass_writes_physical[pass_index])
					written_by_back[w] = true;
			}
		}
		if (!any_back) // nothing to overlap with: keep the whole frame on one stream
			front.assign(passes.size(), false);
	}
	for (unsigned pass_index : pass_stack)
	{
		if (front[pass_index])
			pass_stream[pass_index] = pass_reads_physical[pass_index].empty() ? 1 : 2;
		uses_async_stream = uses_async_stream || pass_stream[pass_index] != 0;
	}

	// The tail: the longest run of passes at the end of the baked order that (a) stand on the generic stream, (b) write nothing the next
	// frame reads (no image with history, nothing read before it is written within the frame: bloom's feedback, exposure) and (c) whose
	// first pass -- and with it the whole run -- takes exactly ONE physical resource from the passes in front of it.  For the application's
	// graphs that is post-tonemap anti-aliasing reading `tonemapped`; a frame that ends with the tonemap has no tail.  Frame N's tail then
	// runs beside frame N + 1's back instead of in front of it (the passes of the tail depend on nothing frame N + 1 produces and
	// produce nothing it consumes): TAA resolve N + 1 no longer queues behind SMAA N.
	// (a frame that ends in a blit to the swapchain keeps its end on the generic stream, where the blit is)
	if (hoist_independent_compute && split_tail && uses_async_stream && swapchain_physical_index != RenderResource::Unused)
	{
		std::vector<bool> carried(physical_dimensions.size(), false); // read by some pass before any pass of the frame has written it
		{
			std::vector<bool> written(physical_dimensions.size(), false);
			for (unsigned pass_index : pass_stack)
			{
				for (unsigned r : pass_reads_physical[pass_index])
					if (!written[r])
						carried[r] = true;
				for (unsigned w : pass_writes_physical[pass_index])
					written[w] = true;
			}
		}
		// the longest suffix satisfying (a) and (b)
		size_t begin = pass_stack.size();
		while (begin > 0)
		{
			const unsigned pass_index = pass_stack[begin - 1];
			bool ok = pass_stream[pass_index] == 0 && passes[pass_index]->get_history_inputs().empty() && !passes[pass_index]->may_not_need_render_pass();
			for (unsigned w : pass_writes_physical[pass_index])
				ok = ok && !physical_image_has_history[w] && !carried[w] && physical_dimensions[w].buffer_info.size == 0;
			if (!ok)
				break;
			begin--;
		}
		// (c): shrink from the front until what crosses into the run is one image, written once, by a back pass
		for (; begin < pass_stack.size(); begin++)
		{
			std::vector<bool> inside(physical_dimensions.size(), false);
			for (size_t i = begin; i < pass_stack.size(); i++)
				for (unsigned w : pass_writes_physical[pass_stack[i]])
					inside[w] = true;
			std::vector<unsigned> crossing;
			for (size_t i = begin; i < pass_stack.size(); i++)
				for (unsigned r : pass_reads_physical[pass_stack[i]])
					if (!inside[r] && std::find(crossing.begin(), crossing.end(), r) == crossing.end())
						crossing.push_back(r);
			if (crossing.size() != 1)
				continue;
			unsigned writers = 0;
			bool back_writer = true;
			for (size_t i = 0; i < begin; i++)
				for (unsigned w : pass_writes_physical[pass_stack[i]])
					if (w == crossing[0])
					{
						writers++;
						back_writer = back_writer && pass_stream[pass_stack[i]] == 0;
					}
			const auto &dim = physical_dimensions[crossing[0]];
			if (writers == 1 && back_writer && dim.buffer_info.size == 0 && !physical_image_has_history[crossing[0]] && !carried[crossing[0]] &&
			    crossing[0] != swapchain_physical_index)
				break;
		}
		// ... and something must be left in front of it on the generic stream for the run to overlap with
		bool back_in_front = false;
		for (size_t i = 0; i < begin && i < pass_stack.size(); i++)
			back_in_front = back_in_front || pass_stream[pass_stack[i]] == 0;
		if (back_in_front)
			for (size_t i = begin; i < pass_stack.size(); i++)
				pass_stream[pass_stack[i]] = 3;
	}
	physical_sync.assign(physical_dimensions.size(), {});
	for (auto &v : physical_sync_alternate)
		v.assign(physical_dimensions.size(), {});

	// Only passes that touch a resource which is also touched from the other stream take part in event ordering; every
	// other pass is ordered by its in-order stream alone and records nothing (event / barrier packets are not free:
	// each one is a command-processor round trip between two kernels).
	pass_needs_sync.assign(passes.size(), false);
	blit_needs_sync = false;
	if (uses_async_stream)
	{
		std::vector<uint8_t> touched(physical_dimensions.size(), 0); // bit s: touched from stream s
		auto several = [](uint8_t bits) { return (bits & (bits - 1)) != 0; };
		for (unsigned pass_index : pass_stack)
		{
			const uint8_t bit = uint8_t(1u << pass_stream[pass_index]);
			for (unsigned r : pass_reads_physical[pass_index])
				touched[r] |= bit;
			for (unsigned w : pass_writes_physical[pass_index])
				touched[w] |= bit;
		}
		if (swapchain_physical_index == RenderResource::Unused)
		{
			auto itr = resource_to_index.find(backbuffer_source);
			if (itr != resource_to_index.end() && resources[itr->second]->get_physical_index() != RenderResource::Unused)
			{
				touched[resources[itr->second]->get_physical_index()] |= 1; // final blit runs on the generic stream
				blit_needs_sync = several(touched[resources[itr->second]->get_physical_index()]);
			}
		}
		for (unsigned pass_index : pass_stack)
		{
			bool shared = false;
			for (unsigned r : pass_reads_physical[pass_index])
				shared = shared || several(touched[r]);
			for (unsigned w : pass_writes_physical[pass_index])
				shared = shared || several(touched[w]);
			// The front's stream alternates with the frame's parity (HIP::Device): what two front passes of different frames share (the
			// G-buffer targets a producer pass rewrites three frames after the lighting pass read them) is ordered by events too.
			pass_needs_sync[pass_index] = shared || (pass_stream[pass_index] == 2 && device && device->front_alternates());
		}
	}

	// What the front writes and the back reads exists twice and alternates per frame (like an image with history), so the
	// front of frame N+1 never waits for the back of frame N to finish reading: write-after-read across frames disappears.
	// Only resources with a single writer qualify (every byte the back reads is rewritten by the front each frame).
	physical_buffer_double.assign(physical_dimensions.size(), false);
	{
		std::vector<unsigned> writers(physical_dimensions.size(), 0);
		std::vector<uint8_t> reader_streams(physical_dimensions.size(), 0);
		for (unsigned pass_index : pass_stack)
		{
			for (unsigned w : pass_writes_physical[pass_index])
				writers[w]++;
			for (unsigned r : pass_reads_physical[pass_index])
				reader_streams[r] |= uint8_t(1u << pass_stream[pass_index]);
		}
		if (swapchain_physical_index == RenderResource::Unused)
		{
			auto itr = resource_to_index.find(backbuffer_source);
			if (itr != resource_to_index.end() && resources[itr->second]->get_physical_index() != RenderResource::Unused)
				reader_streams[resources[itr->second]->get_physical_index()] |= 1;
		}
		bool has_tail = false;
		for (unsigned pass_index : pass_stack)
			has_tail = has_tail || pass_stream[pass_index] == 3;
		for (unsigned pass_index : pass_stack)
		{
			// ... and what the back hands to the tail (`tonemapped`), for the same reason one stage further down the frame
			const bool hands_to_tail = has_tail && pass_stream[pass_index] == 0;
			if (front[pass_index] || hands_to_tail)
				for (unsigned w : pass_writes_physical[pass_index])
				{
					const uint8_t others = reader_streams[w] & ~uint8_t(1u << pass_stream[pass_index]);
					if (writers[w] == 1 && (front[pass_index] ? others != 0 : (others & uint8_t(1u << 3)) != 0))
						physical_buffer_double[w] = true;
				}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// Physical passes (render_graph.cpp:1221-1392): consecutive graphics passes on one queue become subpasses of one render
// pass when the later one consumes the earlier one's attachments "on tile" (colour RMW, shared depth, input attachments)
// and needs nothing from it through memory (sampled textures, storage images / buffers, scaled colour inputs), has no
// conflicting depth attachment and no mip generation in between.  A run grows while the candidate can join EVERY pass
// already in it.  Compute passes never merge.
// ---------------------------------------------------------------------------------------------------------------------
namespace
{
template <typename List, typename Res>
bool holds_physical(const List &list, const Res *res)
{
	if (!res)
		return false;
	for (auto *entry : list)
		if (entry && entry->get_physical_index() == res->get_physical_index())
			return true;
	return false;
}
} // namespace

void RenderGraph::build_physical_passes()
{
	constexpr RenderGraphQueueFlags compute_mask = RENDER_GRAPH_QUEUE_COMPUTE_BIT | RENDER_GRAPH_QUEUE_ASYNC_COMPUTE_BIT;

	auto through_memory = [&](const RenderPass &early, const RenderPass &late) {
		// anything `late` fetches that `early` produced other than as an attachment of the same pixel
		for (auto &in : late.get_generic_texture_inputs())
			if (holds_physical(early.get_color_outputs(), in.texture) || holds_physical(early.get_resolve_outputs(), in.texture) ||
			    holds_physical(early.get_storage_texture_outputs(), in.texture) || (in.texture && in.texture == early.get_depth_stencil_output()))
				return true;
		for (auto &in : late.get_generic_buffer_inputs())
			if (holds_physical(early.get_storage_outputs(), in.buffer))
				return true;
		for (auto *in : late.get_storage_inputs())
			if (holds_physical(early.get_storage_outputs(), in))
				return true;
		for (auto *in : late.get_storage_texture_inputs())
			if (holds_physical(early.get_storage_texture_outputs(), in))
				return true;
		for (auto *in : late.get_color_scale_inputs())
			if (holds_physical(early.get_storage_texture_outputs(), in) || holds_physical(early.get_color_outputs(), in) ||
			    holds_physical(early.get_resolve_outputs(), in))
				return true;
		for (auto *in : late.get_color_inputs())
			if (holds_physical(early.get_storage_texture_outputs(), in))
				return true;
		return false;
	};
	auto differs = [](const RenderResource *a, const RenderResource *b) { return a && b && a->get_physical_index() != b->get_physical_index(); };
	auto same = [](const RenderResource *a, const RenderResource *b) { return a && b && a->get_physical_index() == b->get_physical_index(); };

	auto joins = [&](const RenderPass &early, const RenderPass &late) {
		if ((early.get_queue() & compute_mask) != 0 || late.get_queue() != early.get_queue())
			return false;
		for (auto *out : early.get_color_outputs())
		{
			auto &dim = physical_dimensions[out->get_physical_index()];
			if (dim.levels > 1 && (dim.flags & ATTACHMENT_INFO_MIPGEN_BIT) != 0)
				return false; // mips are generated between the two
		}
		if (through_memory(early, late))
			return false;
		const RenderResource *early_ds[2] = {early.get_depth_stencil_input(), early.get_depth_stencil_output()};
		const RenderResource *late_ds[2] = {late.get_depth_stencil_input(), late.get_depth_stencil_output()};
		for (auto *a : late_ds)
			for (auto *b : early_ds)
				if (differs(a, b))
					return false;

		// Allowed; is there anything to gain?
		for (auto *in : late.get_color_inputs())
			if (holds_physical(early.get_color_outputs(), in) || holds_physical(early.get_resolve_outputs(), in))
				return true;
		if (same(late_ds[0], early_ds[0]) || same(late_ds[0], early_ds[1]))
			return true;
		for (auto *in : late.get_attachment_inputs())
			if (holds_physical(early.get_color_outputs(), in) || holds_physical(early.get_resolve_outputs(), in) ||
			    (in && in == early.get_depth_stencil_output()))
				return true;
		return false;
	};

	pass_physical_pass.assign(passes.size(), unsigned(RenderResource::Unused));
	physical_pass_count = 0;
	for (size_t begin = 0; begin < pass_stack.size();)
	{
		size_t end = begin + 1;
		while (end < pass_stack.size())
		{
			bool ok = true;
			for (size_t member = begin; member < end && ok; member++)
				ok = joins(*passes[pass_stack[member]], *passes[pass_stack[end]]);
			if (!ok)
				break;
			end++;
		}
		for (size_t member = begin; member < end; member++)
			pass_physical_pass[pass_stack[member]] = physical_pass_count;
		physical_pass_count++;
		begin = end;
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// Aliasing of attachment images (render_graph.cpp:1548-1746).  Two images of identical geometry whose uses within the
// frame do not overlap share one allocation.  As in the reference: buffers, images with history and storage images
// (implicitly preserved) never alias, nor does the output of a pass that may be skipped, nor an image that is read
// before it is completely written.  Lifetimes are measured in physical passes, as in the reference.
//
// The reference restricts aliasing to images used on one and the same single queue ("we can only use events to pass
// aliasing barriers").  The HIP executor's form of that rule: both images are touched from one and the same in-order
// stream only, where program order is the aliasing barrier -- within the frame and across frames -- and neither is a
// front-to-back hand-over ring, the swapchain image or the source of the final blit.
//
// One deliberate difference: the reference tests a candidate against one lower-indexed image only, so an image can
// join a chain through a member it is disjoint with while overlapping another member.  Here a candidate must be
// disjoint with every image already sharing the allocation.
// ---------------------------------------------------------------------------------------------------------------------
void RenderGraph::build_aliases()
{
	struct Range
	{
		unsigned first_write = ~0u, last_write = 0, first_read = ~0u, last_read = 0;
		bool block_alias = false;
		uint8_t streams = 0;
		bool has_writer() const { return first_write <= last_write; }
		bool has_reader() const { return first_read <= last_read; }
		bool is_used() const { return has_writer() || has_reader(); }
		bool can_alias() const
		{
			if (has_reader() && has_writer() && first_read <= first_write)
				return false; // read before completely written: contents must be preserved
			return !block_alias;
		}
		unsigned last_used() const { return std::max(has_writer() ? last_write : 0u, has_reader() ? last_read : 0u); }
		unsigned first_used() const { return std::min(has_writer() ? first_write : ~0u, has_reader() ? first_read : ~0u); }
		bool disjoint_lifetime(const Range &other) const
		{
			if (!is_used() || !other.is_used() || !can_alias() || !other.can_alias())
				return false;
			return last_used() < other.first_used() || other.last_used() < first_used();
		}
	};

	const size_t count = physical_dimensions.size();
	physical_aliases.assign(count, unsigned(RenderResource::Unused));
	if (!alias_disjoint_images)
		return;

	std::vector<Range> ranges(count);
	auto reader = [&](const RenderTextureResource *res, unsigned position, unsigned stream) {
		if (!res || res->get_physical_index() == RenderResource::Unused)
			return;
		auto &r = ranges[res->get_physical_index()];
		r.first_read = std::min(r.first_read, position);
		r.last_read = std::max(r.last_read, position);
		r.streams |= uint8_t(1u << stream);
	};
	auto writer = [&](const RenderTextureResource *res, unsigned position, unsigned stream, bool block) {
		if (!res || res->get_physical_index() == RenderResource::Unused)
			return;
		auto &r = ranges[res->get_physical_index()];
		r.first_write = std::min(r.first_write, position);
		r.last_write = std::max(r.last_write, position);
		r.block_alias = r.block_alias || block;
		r.streams |= uint8_t(1u << stream);
	};

	for (unsigned pass_index : pass_stack)
	{
		auto &pass = *passes[pass_index];
		const unsigned stream = pass_stream[pass_index];
		const unsigned position = pass_physical_pass[pass_index];
		for (auto *in : pass.get_color_inputs()) reader(in, position, stream);
		for (auto *in : pass.get_color_scale_inputs()) reader(in, position, stream);
		for (auto *in : pass.get_attachment_inputs()) reader(in, position, stream);
		for (auto &in : pass.get_generic_texture_inputs()) reader(in.texture, position, stream);
		for (auto *in : pass.get_storage_texture_inputs()) reader(in, position, stream);
		reader(pass.get_depth_stencil_input(), position, stream);

		const bool conditional = pass.may_not_need_render_pass();
		writer(pass.get_depth_stencil_output(), position, stream, conditional);
		for (auto *out : pass.get_color_outputs()) writer(out, position, stream, conditional);
		for (auto *out : pass.get_resolve_outputs()) writer(out, position, stream, conditional);
		for (auto *out : pass.get_storage_texture_outputs()) writer(out, position, stream, true);
	}

	unsigned blit_source = RenderResource::Unused;
	if (swapchain_physical_index == RenderResource::Unused)
	{
		auto itr = resource_to_index.find(backbuffer_source);
		if (itr != resource_to_index.end())
			blit_source = resources[itr->second]->get_physical_index();
	}
	auto eligible = [&](unsigned i) {
		const uint8_t s = ranges[i].streams;
		return physical_dimensions[i].buffer_info.size == 0 && !physical_image_has_history[i] && i != swapchain_physical_index &&
		       (physical_dimensions[i].flags & ATTACHMENT_INFO_INTERNAL_RETAINED_BIT) == 0 &&
		       i != blit_source && !physical_buffer_double[i] && s != 0 && (s & (s - 1)) == 0;
	};

	std::vector<std::vector<unsigned>> sharing(count); // owner -> images living in its allocation (owner first)
	for (unsigned i = 0; i < count; i++)
	{
		if (!eligible(i))
			continue;
		for (unsigned j = 0; j < i; j++)
		{
			if (!eligible(j) || physical_aliases[j] != RenderResource::Unused)
				continue;
			if (physical_dimensions[i] != physical_dimensions[j] || ranges[i].streams != ranges[j].streams)
				continue;
			bool disjoint = ranges[i].disjoint_lifetime(ranges[j]);
			for (unsigned other : sharing[j])
				disjoint = disjoint && ranges[i].disjoint_lifetime(ranges[other]);
			if (!disjoint)
				continue;
			physical_aliases[i] = j;
			if (sharing[j].empty())
				sharing[j].push_back(j);
			sharing[j].push_back(i);
			auto usage = physical_dimensions[j].image_usage | physical_dimensions[i].image_usage;
			physical_dimensions[i].image_usage = physical_dimensions[j].image_usage = usage;
			break;
		}
	}
}

void RenderGraph::enqueue_render_passes(HIP::Device &device_, TaskComposer &composer)
{
	GRANITE_SCOPED_TIMELINE_EVENT("enqueue-render-passes");
	const size_t ring_slot = size_t(frame_counter++ % EventRing);
	if (pass_done_event.size() < (passes.size() + 1) * EventRing)
		pass_done_event.resize((passes.size() + 1) * EventRing, nullptr);
	if (physical_sync.size() != physical_dimensions.size())
	{
		physical_sync.assign(physical_dimensions.size(), {});
		for (auto &v : physical_sync_alternate)
			v.assign(physical_dimensions.size(), {});
	}

	static const bool sync_debug = getenv("GRANITE_SYNC_DEBUG") != nullptr;
	const uint64_t this_frame = frame_counter - 1;
	// The runs published under the device's frame fences stay named for EventRing frames on the assumption that the device's ring and this
	// one advance together: one next_frame_context() per enqueued frame.  A caller that rotates the device faster only makes waits
	// stricter (a fence re-recorded early is a later point of its stream), i.e. costs barrier packets, never correctness: say so once.
	if (sync_debug && last_device_frame != 0 && device_.get_frame_number() != last_device_frame + 1)
	{
		static bool told = false;
		if (!told)
			fprintf(stderr, "[sync] the device advanced %llu frame contexts between two enqueued frames: published fences are re-recorded early (stricter waits)\n",
			        (unsigned long long)(device_.get_frame_number() - last_device_frame));
		told = true;
	}
	last_device_frame = device_.get_frame_number();
	int current_pass = -1;
	std::vector<void *> waited;
	const uint64_t device_completed = device_.get_completed_frame();
	auto wait_for = [&](hipStream_t stream, void *event, const char *kind = "", unsigned resource = 0, int src_pass = -1, uint64_t src_frame = 0,
	                    uint64_t src_device_frame = 0) {
		if (!event || std::find(waited.begin(), waited.end(), event) != waited.end())
			return;
		// recorded in a frame the host has already waited for (frame pacing: three frames back with the default lead): complete, no call at all
		// -- the write-after-read dependencies on the rotating copies' previous users are all of this kind
		if (src_device_frame != 0 && src_device_frame <= device_completed)
			return;
		waited.push_back(event);
		// The host runs one to two frames ahead of the GPU (Device::next_frame_context), so most cross-stream dependencies (anything on work of two
		// frames ago, usually the cluster build as well) are already complete when they are looked at: no barrier packet
		// is needed then, and each one costs the command processor several microseconds between two kernels.
		if (hipEventQuery(static_cast<hipEvent_t>(event)) == hipSuccess)
			return;
		if (sync_debug)
			fprintf(stderr, "[sync] frame %llu pass %s waits %s on %s: pass %s of frame %llu\n", (unsigned long long)this_frame,
			        current_pass >= 0 ? passes[current_pass]->get_name().c_str() : "blit", kind, physical_dimensions[resource].name.c_str(),
			        src_pass >= 0 && src_pass < int(passes.size()) ? passes[src_pass]->get_name().c_str() : "?", (unsigned long long)src_frame);
		if (hipStreamWaitEvent(stream, static_cast<hipEvent_t>(event), 0) != hipSuccess)
			throw std::runtime_error("cross-queue dependency failed");
	};
	// RAW / WAW / WAR against accesses recorded on the other stream.
	// (An access of the same stream type is ordered by the stream itself -- unless the type's stream alternates with the frame's parity, the
	// front's, and the access was recorded in a frame of the other parity: the device says, frame numbers being the device's.)
	const uint64_t device_frame = device_.get_frame_number();
	auto in_order_with = [&](int stream_index, int other_index, uint64_t other_device_frame) {
		return other_index == stream_index && device_.same_stream(HIP::CommandBuffer::Type(stream_index), device_frame, other_device_frame);
	};
	auto acquire = [&](hipStream_t stream, int stream_index, const std::vector<unsigned> &reads, const std::vector<unsigned> &writes) {
		for (unsigned r : reads)
			if (physical_sync[r].last_write && !in_order_with(stream_index, physical_sync[r].write_stream, physical_sync[r].write_device_frame))
				wait_for(stream, physical_sync[r].last_write, "RAW", r, physical_sync[r].write_pass, physical_sync[r].write_frame, physical_sync[r].write_device_frame);
		for (unsigned w : writes)
		{
			if (physical_sync[w].last_write && !in_order_with(stream_index, physical_sync[w].write_stream, physical_sync[w].write_device_frame))
				wait_for(stream, physical_sync[w].last_write, "WAW", w, physical_sync[w].write_pass, physical_sync[w].write_frame, physical_sync[w].write_device_frame);
			for (int other = 0; other < StreamCount; other++)
				if (!in_order_with(stream_index, other, physical_sync[w].read_device_frame[other]))
					wait_for(stream, physical_sync[w].last_read[other], "WAR", w, physical_sync[w].read_pass[other], physical_sync[w].read_frame[other],
					         physical_sync[w].read_device_frame[other]);
		}
	};
	// One event per RUN of consecutive passes on the same stream (not per pass): the accesses of every pass of the run are
	// published under the run's event, which is recorded once, after the run's last pass and before any pass of another
	// stream is enqueued.  Fewer packets between kernels: each event record / wait costs the command processor several
	// microseconds (measured: 23 us of a 283 us frame with one record per pass).
	auto ensure_event = [&](void *&event) {
		if (!event)
		{
			hipEvent_t e;
			// ordering between streams of this device only: no system-scope fence (GRANITE_SYNC_EVENT_SYSTEM_FENCE=1 restores it)
			static const unsigned flags = hipEventDisableTiming | (getenv("GRANITE_SYNC_EVENT_SYSTEM_FENCE") ? 0u : unsigned(hipEventDisableSystemFence));
			if (hipEventCreateWithFlags(&e, flags) != hipSuccess)
				throw std::runtime_error("hipEventCreate failed");
			event = e;
		}
	};
	auto release = [&](int stream_index, void *event, const std::vector<unsigned> &reads, const std::vector<unsigned> &writes) {
		for (unsigned r : reads)
		{
			physical_sync[r].last_read[stream_index] = event;
			physical_sync[r].read_pass[stream_index] = current_pass;
			physical_sync[r].read_frame[stream_index] = this_frame;
			physical_sync[r].read_device_frame[stream_index] = device_frame;
		}
		for (unsigned w : writes)
		{
			physical_sync[w].last_write = event;
			physical_sync[w].write_stream = stream_index;
			physical_sync[w].write_pass = current_pass;
			physical_sync[w].write_frame = this_frame;
			physical_sync[w].write_device_frame = device_frame;
			for (auto &read : physical_sync[w].last_read)
				read = nullptr;
		}
	};

	static const HIP::CommandBuffer::Type stream_types[StreamCount] = {HIP::CommandBuffer::Type::Generic, HIP::CommandBuffer::Type::AsyncCompute,
	                                                                    HIP::CommandBuffer::Type::Front, HIP::CommandBuffer::Type::Tail};
	static_assert(int(HIP::CommandBuffer::Type::Count) == StreamCount, "one hazard-tracking slot per executor stream");
	int run_stream = -1;        // stream of the run being enqueued
	unsigned run_slot = 0;      // index of the run within the frame (event ring row)
	bool run_published = false; // a pass of the run published accesses under the run's event
	static_assert(unsigned(EventRing) == HIP::Device::FrameFenceRing, "a run published under a device fence must stay named for as long as one under the graph's own events");
	// The last run a frame puts on a stream publishes under the DEVICE's fence of that stream and frame (the staging ring's, same depth as
	// EventRing) and records it here: the fence next_frame_context() would otherwise record right behind the run's own event.
	// Which run that is: a dry pass over the frame's passes (need_render_pass is asked once per pass and frame).
	std::vector<char> pass_runs(pass_stack.size(), 0);
	int last_run_of_stream[StreamCount] = {-1, -1, -1, -1};
	{
		int stream_of_run = -1, run = 0;
		for (size_t i = 0; i < pass_stack.size(); i++)
		{
			auto &pass = *passes[pass_stack[i]];
			pass_runs[i] = !(pass.may_not_need_render_pass() && !pass.need_render_pass());
			if (!pass_runs[i])
				continue;
			if (int(get_pass_stream(pass_stack[i])) != stream_of_run)
			{
				stream_of_run = int(get_pass_stream(pass_stack[i]));
				run++;
			}
			last_run_of_stream[stream_of_run] = run;
		}
	}
	const bool blit_follows = swapchain_attachment && swapchain_physical_index == RenderResource::Unused;
	auto run_event = [&]() -> void * {
		// (the final blit goes behind the generic stream's last run: that run keeps its own event, the fence is recorded behind the blit)
		if (int(run_slot) == last_run_of_stream[run_stream] && !(blit_follows && stream_types[run_stream] == HIP::CommandBuffer::Type::Generic))
			return device_.frame_fence(stream_types[run_stream]);
		void *&event = pass_done_event[run_slot * EventRing + ring_slot];
		ensure_event(event);
		return event;
	};
	auto close_run = [&]() {
		if (run_stream >= 0 && run_published)
		{
			void *event = run_event();
			if (event == device_.frame_fence(stream_types[run_stream]))
				device_.record_frame_fence(stream_types[run_stream]);
			else if (hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(device_.get_stream(stream_types[run_stream]))) != hipSuccess)
				throw std::runtime_error("hipEventRecord failed");
		}
		run_published = false;
	};

	for (size_t stack_index = 0; stack_index < pass_stack.size(); stack_index++)
	{
		const unsigned pass_index = pass_stack[stack_index];
		auto &pass = *passes[pass_index];
		if (!pass_runs[stack_index])
			continue;
		pass.prepare_render_pass(composer);

		auto type = stream_types[get_pass_stream(pass_index)];
		auto stream = static_cast<hipStream_t>(device_.get_stream(type));
		if (int(get_pass_stream(pass_index)) != run_stream)
		{
			close_run();
			run_stream = int(get_pass_stream(pass_index));
			run_slot++; // rows 1 .. (number of runs <= number of passes); row 0 belongs to the final blit
			waited.clear();
		}
		// Measurement hook (tests/test_gpu_graph_random.py shows that the waits matter): frames are WRONG with it, so
		// honouring it is announced on stderr instead of corrupting output silently through an inherited environment.
		static const bool no_cross_sync = []() {
			const bool off = getenv("GRANITE_UNSAFE_NO_CROSS_SYNC") != nullptr;
			if (off)
				fprintf(stderr, "[granite-hip] WARNING: GRANITE_UNSAFE_NO_CROSS_SYNC is set: cross-stream RAW/WAW/WAR waits are "
				                "DISABLED, rendered frames are not valid.\n");
			return off;
		}();
		const bool sync = uses_async_stream && pass_needs_sync[pass_index] && !no_cross_sync;
		current_pass = int(pass_index);
		if (sync)
			acquire(stream, int(type), pass_reads_physical[pass_index], pass_writes_physical[pass_index]);

		HIP::CommandBuffer cmd{device_, stream, type};

		PassTimestamp ts = {pass_index, nullptr, nullptr};
		if (enabled_timestamps)
		{
			ts.start = acquire_event();
			ts.stop = acquire_event();
			(void)hipEventRecord(static_cast<hipEvent_t>(ts.start), stream);
		}

		// LOAD_OP_CLEAR for colour outputs that have no input (render_graph.cpp:1871-1990): only on graphics passes.
		if ((pass.get_queue() & compute_queues) == 0)
		{
			for (unsigned i = 0; i < pass.get_color_outputs().size(); i++)
			{
				VkClearColorValue value = {};
				if (pass.get_color_inputs()[i] == nullptr && pass.get_clear_color(i, &value))
				{
					if (value.uint32[0] | value.uint32[1] | value.uint32[2] | value.uint32[3])
						throw std::logic_error("Only zero clear colours are supported by the HIP executor.");
					cmd.clear_image(get_physical_texture_resource(*pass.get_color_outputs()[i]));
				}
			}
		}

		{
			// one event per pass, named like the pass (render_graph.cpp:2261: cmd.begin_region(pass name))
			ScopedTimelineEvent pass_event{TimelineTrace::get().enabled() ? TimelineTrace::get().intern(pass.get_name()) : ""};
			pass.build_render_pass(cmd, 0);
		}

		if (enabled_timestamps)
		{
			(void)hipEventRecord(static_cast<hipEvent_t>(ts.stop), stream);
			pending_timestamps.push_back(ts);
		}
		if (sync)
		{
			release(int(type), run_event(), pass_reads_physical[pass_index], pass_writes_physical[pass_index]);
			run_published = true;
		}
	}
	close_run();

	// Backbuffer could not alias the swapchain image: final blit.  Same geometry only; R8G8B8A8 UNORM <-> SRGB is a byte
	// copy (the stored bytes are the gamma-space colour either way -- e.g. the un-sharpened FSR output, aa.cpp:80-84.  The
	// reference's scale pass, render_graph.cpp:2557-2566, would sample the UNORM view and let the sRGB store encode again).
	if (swapchain_attachment && swapchain_physical_index == RenderResource::Unused)
	{
		const unsigned src_index = resources[resource_to_index[backbuffer_source]]->get_physical_index();
		auto &src = get_physical_texture_resource(src_index);
		auto rgba8 = [](VkFormat f) { return f == VK_FORMAT_R8G8B8A8_UNORM || f == VK_FORMAT_R8G8B8A8_SRGB; };
		const bool same_texels = src.get_format() == swapchain_attachment->get_format() || (rgba8(src.get_format()) && rgba8(swapchain_attachment->get_format()));
		if (src.get_width() != swapchain_attachment->get_width() || src.get_height() != swapchain_attachment->get_height() || !same_texels)
			throw std::logic_error("Backbuffer source does not match the swapchain; scaling blits are not implemented.");
		HIP::CommandBuffer cmd{device_, device_.get_stream(HIP::CommandBuffer::Type::Generic), HIP::CommandBuffer::Type::Generic};
		auto stream = static_cast<hipStream_t>(cmd.get_stream());
		const std::vector<unsigned> reads = {src_index}, none;
		const bool sync = uses_async_stream && blit_needs_sync;
		current_pass = -1;
		if (sync)
		{
			waited.clear();
			acquire(stream, int(HIP::CommandBuffer::Type::Generic), reads, none);
		}
		cmd.copy_image(*swapchain_attachment, src);
		if (sync)
		{
			void *&event = pass_done_event[0 * EventRing + ring_slot]; // row 0 is reserved for the blit
			ensure_event(event);
			release(int(HIP::CommandBuffer::Type::Generic), event, reads, none);
			if (hipEventRecord(static_cast<hipEvent_t>(event), stream) != hipSuccess)
				throw std::runtime_error("hipEventRecord failed");
		}
	}

	device_.next_frame_context();
}

std::vector<RenderGraph::TimestampReport> RenderGraph::collect_timestamps()
{
	for (auto &ts : pending_timestamps)
	{
		auto start = static_cast<hipEvent_t>(ts.start), stop = static_cast<hipEvent_t>(ts.stop);
		float ms = 0.0f;
		if (hipEventSynchronize(stop) == hipSuccess && hipEventElapsedTime(&ms, start, stop) == hipSuccess)
		{
			// Tag = the physical pass, "a + b" when the reference would run several passes as subpasses of one
			// VkRenderPass (render_graph.cpp:2274-2289); one accumulation per physical pass instance.
			std::string name;
			bool first_of_group = true;
			if (ts.pass < pass_physical_pass.size())
			{
				unsigned group = pass_physical_pass[ts.pass];
				for (unsigned member : pass_stack)
				{
					if (pass_physical_pass[member] != group)
						continue;
					if (!name.empty())
						name += " + ";
					else
						first_of_group = member == ts.pass;
					name += passes[member]->get_name();
				}
			}
			else
				name = passes[ts.pass]->get_name();
			auto itr = timestamp_accum.find(name);
			if (itr == timestamp_accum.end())
			{
				timestamp_order.push_back(name);
				itr = timestamp_accum.emplace(name, std::make_pair(uint64_t(0), 0.0)).first;
			}
			if (first_of_group)
				itr->second.first++;
			itr->second.second += ms;
		}
		event_pool.push_back(ts.start);
		event_pool.push_back(ts.stop);
	}
	pending_timestamps.clear();

	std::vector<TimestampReport> out;
	for (auto &name : timestamp_order)
	{
		auto &acc = timestamp_accum[name];
		out.push_back({name, acc.first, acc.second});
	}
	return out;
}

void RenderGraph::reset_timestamps()
{
	collect_timestamps();
	timestamp_accum.clear();
	timestamp_order.clear();
}

// ---------------------------------------------------------------------------------------------------------------------
// log (render_graph.cpp:1394-1511)
// ---------------------------------------------------------------------------------------------------------------------
void RenderGraph::log()
{
	for (auto &dim : physical_dimensions)
	{
		unsigned i = unsigned(&dim - physical_dimensions.data());
		if (dim.buffer_info.size)
			fprintf(stderr, "Resource #%u (%s): size: %u\n", i, dim.name.c_str(), unsigned(dim.buffer_info.size));
		else
			fprintf(stderr, "Resource #%u (%s): %u x %u (fmt: %u), samples: %u%s\n", i, dim.name.c_str(), dim.width, dim.height,
			        unsigned(dim.format), dim.samples, physical_image_has_history[i] ? " (history)" : "");
	}
	unsigned order = 0;
	for (unsigned pass_index : pass_stack)
	{
		auto &pass = *passes[pass_index];
		fprintf(stderr, "Pass #%u: %s (queue %u)\n", order++, pass.get_name().c_str(), unsigned(pass.get_queue()));
		for (auto *o : pass.get_color_outputs())
			fprintf(stderr, "    ColorAttachment: %u (%s)\n", o->get_physical_index(), o->get_name().c_str());
		for (auto *o : pass.get_storage_texture_outputs())
			fprintf(stderr, "    StorageTexture: %u (%s)\n", o->get_physical_index(), o->get_name().c_str());
		for (auto *o : pass.get_storage_outputs())
			fprintf(stderr, "    StorageBuffer: %u (%s)\n", o->get_physical_index(), o->get_name().c_str());
		for (auto &in : pass.get_generic_texture_inputs())
			fprintf(stderr, "    Texture: %u (%s)\n", in.texture->get_physical_index(), in.texture->get_name().c_str());
		for (auto *in : pass.get_attachment_inputs())
			fprintf(stderr, "    InputAttachment: %u (%s)\n", in->get_physical_index(), in->get_name().c_str());
		for (auto &in : pass.get_generic_buffer_inputs())
			fprintf(stderr, "    Buffer: %u (%s)\n", in.buffer->get_physical_index(), in.buffer->get_name().c_str());
		for (auto *in : pass.get_history_inputs())
			fprintf(stderr, "    History: %u (%s)\n", in->get_physical_index(), in->get_name().c_str());
	}
}

std::string RenderGraph::dump_json() const
{
	std::ostringstream os;
	os << "{\"passes\":[";
	bool first = true;
	for (unsigned pass_index : pass_stack)
	{
		auto &pass = *passes[pass_index];
		if (!first)
			os << ",";
		first = false;
		os << "{\"name\":\"" << pass.get_name() << "\",\"queue\":" << unsigned(pass.get_queue())
		   << ",\"physical_pass\":" << int(get_physical_pass_index(pass_index))
		   << ",\"stream\":" << (get_pass_stream(pass_index) == 0 ? "\"generic\"" : get_pass_stream(pass_index) == 1 ? "\"async\"" : get_pass_stream(pass_index) == 2 ? "\"front\"" : "\"tail\"")
		   << ",\"writes\":[";
		bool f2 = true;
		auto emit = [&](const RenderResource *r) {
			if (!r)
				return;
			if (!f2)
				os << ",";
			f2 = false;
			os << "{\"name\":\"" << r->get_name() << "\",\"phys\":" << int(r->get_physical_index()) << "}";
		};
		for (auto *o : pass.get_color_outputs()) emit(o);
		for (auto *o : pass.get_storage_texture_outputs()) emit(o);
		for (auto *o : pass.get_storage_outputs()) emit(o);
		for (auto *o : pass.get_transfer_outputs()) emit(o);
		emit(pass.get_depth_stencil_output());
		os << "],\"reads\":[";
		f2 = true;
		for (auto &in : pass.get_generic_texture_inputs()) emit(in.texture);
		for (auto *in : pass.get_attachment_inputs()) emit(in);
		for (auto &in : pass.get_generic_buffer_inputs()) emit(in.buffer);
		for (auto *in : pass.get_color_inputs()) emit(in);
		for (auto *in : pass.get_storage_inputs()) emit(in);
		for (auto *in : pass.get_storage_texture_inputs()) emit(in);
		emit(pass.get_depth_stencil_input());
		os << "],\"history\":[";
		f2 = true;
		for (auto *in : pass.get_history_inputs()) emit(in);
		os << "]}";
	}
	os << "],\"resources\":[";
	for (size_t i = 0; i < physical_dimensions.size(); i++)
	{
		auto &d = physical_dimensions[i];
		if (i)
			os << ",";
		os << "{\"phys\":" << i << ",\"name\":\"" << d.name << "\",\"width\":" << d.width << ",\"height\":" << d.height
		   << ",\"format\":" << unsigned(d.format) << ",\"buffer_size\":" << d.buffer_info.size
		   << ",\"history\":" << (physical_image_has_history[i] ? "true" : "false")
		   << ",\"double_buffered\":" << (physical_buffer_is_double_buffered(unsigned(i)) ? "true" : "false")
		   << ",\"alias_of\":" << int(get_physical_alias(unsigned(i))) << "}";
	}
	os << "],\"swapchain_phys\":" << int(swapchain_physical_index) << "}";
	return os.str();
}
} // namespace Granite
