// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// Granite::RenderGraph — Vulkan-free restatement of the pass/attachment declaration API in
// renderer/render_graph.hpp:48-73,124-251,434-893 so that code written against Granite's graph (setup_hdr_postprocess,
// LightClusterer::add_render_passes, tests/render_graph_sandbox.cpp ...) declares its passes unchanged while the
// callbacks receive a HIP::CommandBuffer& instead of a Vulkan::CommandBuffer&.
//
// Kept: names, argument meaning, idempotent add_pass, resource read/write bookkeeping, validation + std::logic_error
// messages, back-to-front dependency walk, pass reordering, physical index assignment (RMW outputs alias inputs),
// history swap, InputRelative/SwapchainRelative size resolution (ceil(in * scale)), persistent-resource reuse, aliasing
// of attachment images with disjoint lifetimes (build_aliases).
// Dropped (no HIP analogue): image layouts, barriers/semaphores, subpass merging, transient attachments.  A pass is a
// sequence of kernel launches on an in-order stream, and every logical pass is its own physical pass.
#pragma once
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "hip_device.hpp"

namespace Granite
{
class RenderGraph;
class RenderPass;

// Granite hands a TaskComposer to enqueue_render_passes so callbacks can be recorded on worker threads
// (render_graph.cpp:2380-2403).  HIP launches are asynchronous already; the executor records inline and the type
// only keeps the signature.
class TaskComposer
{
};

class RenderPassInterface
{
public:
	virtual ~RenderPassInterface() = default;
	virtual bool render_pass_is_conditional() const { return false; }
	virtual bool render_pass_is_separate_layered() const { return false; }
	virtual bool need_render_pass() const { return true; }
	virtual bool get_clear_depth_stencil(VkClearDepthStencilValue *value) const
	{
		if (value)
			*value = {1.0f, 0u};
		return true;
	}
	virtual bool get_clear_color(unsigned, VkClearColorValue *value) const
	{
		if (value)
			*value = {};
		return true;
	}
	virtual void setup_dependencies(RenderPass &, RenderGraph &) {}
	virtual void setup(HIP::Device &) {}
	virtual void enqueue_prepare_render_pass(RenderGraph &, TaskComposer &) {}
	virtual void build_render_pass(HIP::CommandBuffer &) {}
	virtual void build_render_pass_separate_layer(HIP::CommandBuffer &, unsigned) {}
};
using RenderPassInterfaceHandle = std::shared_ptr<RenderPassInterface>;

enum SizeClass
{
	Absolute,
	SwapchainRelative,
	InputRelative
};

enum RenderGraphQueueFlagBits
{
	RENDER_GRAPH_QUEUE_GRAPHICS_BIT = 1 << 0,
	RENDER_GRAPH_QUEUE_COMPUTE_BIT = 1 << 1,
	RENDER_GRAPH_QUEUE_ASYNC_COMPUTE_BIT = 1 << 2
};
using RenderGraphQueueFlags = uint32_t;

enum AttachmentInfoFlagBits
{
	ATTACHMENT_INFO_PERSISTENT_BIT = 1 << 0,
	ATTACHMENT_INFO_UNORM_SRGB_ALIAS_BIT = 1 << 1,
	ATTACHMENT_INFO_SUPPORTS_PREROTATE_BIT = 1 << 2,
	ATTACHMENT_INFO_MIPGEN_BIT = 1 << 3,
	ATTACHMENT_INFO_INTERNAL_TRANSIENT_BIT = 1 << 16,
	ATTACHMENT_INFO_INTERNAL_PROXY_BIT = 1 << 17,
	// HIP executor extension: the producer keeps the attachment's contents from one frame to the next (a cached fill instead
	// of a per-frame render), so its memory is never shared with another image, whatever the stream assignment.
	ATTACHMENT_INFO_INTERNAL_RETAINED_BIT = 1 << 18
};
using AttachmentInfoFlags = uint32_t;

struct AttachmentInfo
{
	SizeClass size_class = SizeClass::SwapchainRelative;
	float size_x = 1.0f;
	float size_y = 1.0f;
	float size_z = 0.0f;
	VkFormat format = VK_FORMAT_UNDEFINED;
	std::string size_relative_name;
	unsigned samples = 1;
	unsigned levels = 1;
	unsigned layers = 1;
	VkImageUsageFlags aux_usage = 0;
	AttachmentInfoFlags flags = ATTACHMENT_INFO_PERSISTENT_BIT;
};

struct BufferInfo
{
	VkDeviceSize size = 0;
	VkBufferUsageFlags usage = 0;
	AttachmentInfoFlags flags = ATTACHMENT_INFO_PERSISTENT_BIT;
	bool operator==(const BufferInfo &other) const { return size == other.size && usage == other.usage && flags == other.flags; }
	bool operator!=(const BufferInfo &other) const { return !(*this == other); }
};

struct ResourceDimensions
{
	VkFormat format = VK_FORMAT_UNDEFINED;
	BufferInfo buffer_info;
	unsigned width = 0;
	unsigned height = 0;
	unsigned depth = 1;
	unsigned layers = 1;
	unsigned levels = 1;
	unsigned samples = 1;
	AttachmentInfoFlags flags = ATTACHMENT_INFO_PERSISTENT_BIT;
	RenderGraphQueueFlags queues = 0;
	VkImageUsageFlags image_usage = 0;
	std::string name;

	bool operator==(const ResourceDimensions &other) const
	{
		// image_usage and queues are deliberately not part of this test (render_graph.hpp:204-217).
		return format == other.format && width == other.width && height == other.height && depth == other.depth &&
		       layers == other.layers && levels == other.levels && buffer_info == other.buffer_info && flags == other.flags;
	}
	bool operator!=(const ResourceDimensions &other) const { return !(*this == other); }
	bool is_storage_image() const { return (image_usage & VK_IMAGE_USAGE_STORAGE_BIT) != 0; }
};

class RenderResource
{
public:
	enum class Type { Buffer, Texture, Proxy };
	enum { Unused = ~0u };

	RenderResource(Type type_, unsigned index_) : resource_type(type_), index(index_) {}
	virtual ~RenderResource() = default;

	Type get_type() const { return resource_type; }
	void written_in_pass(unsigned index_) { written_in_passes.insert(index_); }
	void read_in_pass(unsigned index_) { read_in_passes.insert(index_); }
	const std::unordered_set<unsigned> &get_read_passes() const { return read_in_passes; }
	const std::unordered_set<unsigned> &get_write_passes() const { return written_in_passes; }
	unsigned get_index() const { return index; }
	void set_physical_index(unsigned index_) { physical_index = index_; }
	unsigned get_physical_index() const { return physical_index; }
	void set_name(const std::string &name_) { name = name_; }
	const std::string &get_name() const { return name; }
	void add_queue(RenderGraphQueueFlagBits queue) { used_queues |= queue; }
	RenderGraphQueueFlags get_used_queues() const { return used_queues; }

protected:
	void reset_pass_sets(RenderGraphQueueFlags queues)
	{
		written_in_passes.clear();
		read_in_passes.clear();
		used_queues = queues;
	}

private:
	Type resource_type;
	unsigned index;
	unsigned physical_index = Unused;
	std::unordered_set<unsigned> written_in_passes;
	std::unordered_set<unsigned> read_in_passes;
	std::string name;
	RenderGraphQueueFlags used_queues = 0;
};

class RenderBufferResource : public RenderResource
{
public:
	explicit RenderBufferResource(unsigned index_) : RenderResource(RenderResource::Type::Buffer, index_) {}
	void set_buffer_info(const BufferInfo &info_) { info = info_; }
	const BufferInfo &get_buffer_info() const { return info; }
	void add_buffer_usage(VkBufferUsageFlags flags) { buffer_usage |= flags; }
	VkBufferUsageFlags get_buffer_usage() const { return buffer_usage; }

private:
	BufferInfo info;
	VkBufferUsageFlags buffer_usage = 0;
};

class RenderTextureResource : public RenderResource
{
public:
	explicit RenderTextureResource(unsigned index_) : RenderResource(RenderResource::Type::Texture, index_) {}
	void set_attachment_info(const AttachmentInfo &info_) { info = info_; }
	const AttachmentInfo &get_attachment_info() const { return info; }
	AttachmentInfo &get_attachment_info() { return info; }
	void add_image_usage(VkImageUsageFlags flags) { image_usage |= flags; }
	VkImageUsageFlags get_image_usage() const { return image_usage; }
	void become_write_alias_of(const RenderTextureResource &other, unsigned writer_pass);

private:
	AttachmentInfo info;
	VkImageUsageFlags image_usage = 0;
};

class RenderPass
{
public:
	RenderPass(RenderGraph &graph_, unsigned index_, RenderGraphQueueFlagBits queue_) : graph(graph_), index(index_), queue(queue_) {}
	enum { Unused = ~0u };

	struct AccessedTextureResource
	{
		VkPipelineStageFlags2 stages = 0;
		VkAccessFlags2 access = 0;
		RenderTextureResource *texture = nullptr;
	};
	struct AccessedBufferResource
	{
		VkPipelineStageFlags2 stages = 0;
		VkAccessFlags2 access = 0;
		RenderBufferResource *buffer = nullptr;
	};
	struct AccessedProxyResource
	{
		VkPipelineStageFlags2 stages = 0;
		VkAccessFlags2 access = 0;
		RenderResource *proxy = nullptr;
		RenderResource *alias_input = nullptr;
	};

	RenderGraphQueueFlagBits get_queue() const { return queue; }
	RenderGraph &get_graph() { return graph; }
	unsigned get_index() const { return index; }

	RenderTextureResource &set_depth_stencil_input(const std::string &name);
	RenderTextureResource &set_depth_stencil_output(const std::string &name, const AttachmentInfo &info);
	RenderTextureResource &add_color_output(const std::string &name, const AttachmentInfo &info, const std::string &input = "");
	RenderTextureResource &add_resolve_output(const std::string &name, const AttachmentInfo &info);
	RenderTextureResource &add_attachment_input(const std::string &name);
	RenderTextureResource &add_history_input(const std::string &name);
	RenderTextureResource &add_texture_input(const std::string &name, VkPipelineStageFlags2 stages = 0);
	RenderBufferResource &add_uniform_input(const std::string &name, VkPipelineStageFlags2 stages = 0);
	RenderBufferResource &add_storage_read_only_input(const std::string &name, VkPipelineStageFlags2 stages = 0);
	RenderBufferResource &add_storage_output(const std::string &name, const BufferInfo &info, const std::string &input = "");
	RenderBufferResource &add_transfer_output(const std::string &name, const BufferInfo &info);
	RenderTextureResource &add_storage_texture_output(const std::string &name, const AttachmentInfo &info, const std::string &input = "");
	// Proxies carry no memory: a pure ordering edge between a writer and its readers (render_graph.hpp:513-514), e.g.
	// to keep alive and order a pass whose output is consumed outside the graph or by the next frame.
	void add_proxy_output(const std::string &name, VkPipelineStageFlags2 stages, VkAccessFlags2 access, const std::string &input = "");
	void add_proxy_input(const std::string &name, VkPipelineStageFlags2 stages, VkAccessFlags2 access);
	RenderBufferResource &add_vertex_buffer_input(const std::string &name);
	RenderBufferResource &add_index_buffer_input(const std::string &name);
	RenderBufferResource &add_indirect_buffer_input(const std::string &name);
	// Declares that this pass "writes" `to` as a rename of `from` without touching it (render_graph.cpp:367-376):
	// later readers of `to` are ordered after this pass and share `from`'s physical image.
	void add_fake_resource_write_alias(const std::string &from, const std::string &to);

	void make_color_input_scaled(unsigned index_) { std::swap(color_scale_inputs[index_], color_inputs[index_]); }

	const std::vector<RenderTextureResource *> &get_color_outputs() const { return color_outputs; }
	const std::vector<RenderTextureResource *> &get_resolve_outputs() const { return resolve_outputs; }
	const std::vector<RenderTextureResource *> &get_color_inputs() const { return color_inputs; }
	const std::vector<RenderTextureResource *> &get_color_scale_inputs() const { return color_scale_inputs; }
	const std::vector<RenderTextureResource *> &get_storage_texture_outputs() const { return storage_texture_outputs; }
	const std::vector<RenderTextureResource *> &get_storage_texture_inputs() const { return storage_texture_inputs; }
	const std::vector<RenderTextureResource *> &get_attachment_inputs() const { return attachments_inputs; }
	const std::vector<RenderTextureResource *> &get_history_inputs() const { return history_inputs; }
	const std::vector<RenderBufferResource *> &get_storage_inputs() const { return storage_inputs; }
	const std::vector<RenderBufferResource *> &get_storage_outputs() const { return storage_outputs; }
	const std::vector<RenderBufferResource *> &get_transfer_outputs() const { return transfer_outputs; }
	const std::vector<AccessedTextureResource> &get_generic_texture_inputs() const { return generic_texture; }
	const std::vector<AccessedBufferResource> &get_generic_buffer_inputs() const { return generic_buffer; }
	const std::vector<AccessedProxyResource> &get_proxy_inputs() const { return proxy_inputs; }
	const std::vector<AccessedProxyResource> &get_proxy_outputs() const { return proxy_outputs; }
	const std::vector<std::pair<RenderTextureResource *, RenderTextureResource *>> &get_fake_resource_aliases() const
	{
		return fake_resource_alias;
	}
	RenderTextureResource *get_depth_stencil_input() const { return depth_stencil_input; }
	RenderTextureResource *get_depth_stencil_output() const { return depth_stencil_output; }

	bool need_render_pass() const { return render_pass_handle ? render_pass_handle->need_render_pass() : true; }
	bool may_not_need_render_pass() const { return render_pass_handle ? render_pass_handle->render_pass_is_conditional() : false; }
	bool get_clear_color(unsigned index_, VkClearColorValue *value = nullptr) const
	{
		if (render_pass_handle)
			return render_pass_handle->get_clear_color(index_, value);
		else if (get_clear_color_cb)
			return get_clear_color_cb(index_, value);
		return false;
	}
	bool get_clear_depth_stencil(VkClearDepthStencilValue *value = nullptr) const
	{
		if (render_pass_handle)
			return render_pass_handle->get_clear_depth_stencil(value);
		else if (get_clear_depth_stencil_cb)
			return get_clear_depth_stencil_cb(value);
		return false;
	}
	void prepare_render_pass(TaskComposer &composer)
	{
		if (render_pass_handle)
			render_pass_handle->enqueue_prepare_render_pass(graph, composer);
	}
	void setup(HIP::Device &device)
	{
		if (render_pass_handle)
			render_pass_handle->setup(device);
	}
	void setup_dependencies()
	{
		if (render_pass_handle)
			render_pass_handle->setup_dependencies(*this, graph);
	}
	void build_render_pass(HIP::CommandBuffer &cmd, unsigned layer)
	{
		if (render_pass_handle)
		{
			if (render_pass_handle->render_pass_is_separate_layered())
				render_pass_handle->build_render_pass_separate_layer(cmd, layer);
			else
				render_pass_handle->build_render_pass(cmd);
		}
		else if (build_render_pass_cb)
			build_render_pass_cb(cmd);
	}

	void set_render_pass_interface(RenderPassInterfaceHandle handle) { render_pass_handle = std::move(handle); }
	void set_build_render_pass(std::function<void(HIP::CommandBuffer &)> func) { build_render_pass_cb = std::move(func); }
	void set_get_clear_depth_stencil(std::function<bool(VkClearDepthStencilValue *)> func) { get_clear_depth_stencil_cb = std::move(func); }
	void set_get_clear_color(std::function<bool(unsigned, VkClearColorValue *)> func) { get_clear_color_cb = std::move(func); }
	void set_name(const std::string &name) { pass_name = name; }
	const std::string &get_name() const { return pass_name; }

private:
	RenderGraph &graph;
	unsigned index;
	RenderGraphQueueFlagBits queue;
	RenderPassInterfaceHandle render_pass_handle;
	std::function<void(HIP::CommandBuffer &)> build_render_pass_cb;
	std::function<bool(VkClearDepthStencilValue *)> get_clear_depth_stencil_cb;
	std::function<bool(unsigned, VkClearColorValue *)> get_clear_color_cb;

	std::vector<RenderTextureResource *> color_outputs;
	std::vector<RenderTextureResource *> resolve_outputs;
	std::vector<RenderTextureResource *> color_inputs;
	std::vector<RenderTextureResource *> color_scale_inputs;
	std::vector<RenderTextureResource *> storage_texture_inputs;
	std::vector<RenderTextureResource *> storage_texture_outputs;
	std::vector<RenderTextureResource *> attachments_inputs;
	std::vector<RenderTextureResource *> history_inputs;
	std::vector<RenderBufferResource *> storage_outputs;
	std::vector<RenderBufferResource *> storage_inputs;
	std::vector<RenderBufferResource *> transfer_outputs;
	std::vector<AccessedTextureResource> generic_texture;
	std::vector<AccessedBufferResource> generic_buffer;
	std::vector<AccessedProxyResource> proxy_inputs, proxy_outputs;
	RenderTextureResource *depth_stencil_input = nullptr;
	RenderTextureResource *depth_stencil_output = nullptr;
	std::vector<std::pair<RenderTextureResource *, RenderTextureResource *>> fake_resource_alias;
	std::string pass_name;

	RenderBufferResource &add_generic_buffer_input(const std::string &name, VkPipelineStageFlags2 stages, VkAccessFlags2 access,
	                                               VkBufferUsageFlags usage);
};

class RenderGraph
{
public:
	RenderGraph() = default;
	~RenderGraph();
	RenderGraph(const RenderGraph &) = delete;
	void operator=(const RenderGraph &) = delete;

	void set_device(HIP::Device *device_) { device = device_; }
	HIP::Device &get_device()
	{
		if (!device)
			throw std::logic_error("RenderGraph has no device.");
		return *device;
	}

	RenderPass &add_pass(const std::string &name, RenderGraphQueueFlagBits queue);
	RenderPass *find_pass(const std::string &name);
	void set_backbuffer_source(const std::string &name);
	void set_backbuffer_dimensions(const ResourceDimensions &dim) { swapchain_dimensions = dim; }
	const ResourceDimensions &get_backbuffer_dimensions() const { return swapchain_dimensions; }

	ResourceDimensions get_resource_dimensions(const RenderBufferResource &resource) const;
	ResourceDimensions get_resource_dimensions(const RenderTextureResource &resource) const;

	void enable_timestamps(bool enable) { enabled_timestamps = enable; }
	// HIP executor policy (no reference analogue; Granite decides queues at declaration time only): frame pipelining.
	// Passes that do not depend on anything carried over from the previous frame (the "front": cluster build, G-buffer,
	// lighting) leave the generic stream: input-free ones (cluster build) go to the async-compute stream, the others to a
	// third "front" stream.  Streams are ordered by per-resource events and whatever crosses from one stream to another
	// is double-buffered, so frame N+1's cluster build overlaps frame N's lighting and frame N+1's lighting overlaps
	// frame N's bloom / tonemap.  Default on.
	void set_hoist_independent_compute(bool enable) { hoist_independent_compute = enable; }
	// The end of the frame on a stream of its own (see split_tail below).  Default on; only has an effect while frames are pipelined.
	void set_split_tail(bool enable) { split_tail = enable; }
	// Share one allocation between attachment images of identical geometry whose lifetimes within the frame do not
	// overlap (build_aliases, render_graph.cpp:1548-1746).  On by default like the reference; off is for A/B tests.
	void set_alias_disjoint_images(bool enable) { alias_disjoint_images = enable; }
	unsigned get_physical_pass_index(unsigned pass_index) const { return pass_index < pass_physical_pass.size() ? pass_physical_pass[pass_index] : unsigned(RenderResource::Unused); }
	unsigned get_physical_pass_count() const { return physical_pass_count; }
	unsigned get_physical_alias(unsigned index) const { return index < physical_aliases.size() ? physical_aliases[index] : unsigned(RenderResource::Unused); }
	// 0 = generic stream (the back of the frame), 1 = async compute (passes the front does not wait for within a frame:
	// explicit ASYNC_COMPUTE passes and input-free front passes such as the cluster build), 2 = the rest of the front,
	// 3 = the tail (what follows the frame's last pass with a tie to the next frame and reads one image of it).
	unsigned get_pass_stream(unsigned pass_index) const { return pass_index < pass_stream.size() ? pass_stream[pass_index] : 0u; }
	bool physical_buffer_is_double_buffered(unsigned index) const { return index < physical_buffer_double.size() && physical_buffer_double[index]; }
	void bake();
	void reset();
	void log();
	std::string dump_json() const; // machine-readable twin of log(): passes in baked order, physical resources

	void setup_attachments(HIP::Device &device, HIP::ImageView *swapchain);
	void enqueue_render_passes(HIP::Device &device, TaskComposer &composer);

	RenderTextureResource &get_texture_resource(const std::string &name);
	RenderBufferResource &get_buffer_resource(const std::string &name);
	RenderResource &get_proxy_resource(const std::string &name);

	HIP::ImageView &get_physical_texture_resource(unsigned index);
	HIP::ImageView *get_physical_history_texture_resource(unsigned index);
	HIP::Buffer &get_physical_buffer_resource(unsigned index);
	HIP::ImageView &get_physical_texture_resource(const RenderTextureResource &resource)
	{
		return get_physical_texture_resource(resource.get_physical_index());
	}
	HIP::ImageView *maybe_get_physical_texture_resource(RenderTextureResource *resource)
	{
		if (resource && resource->get_physical_index() != RenderResource::Unused)
			return &get_physical_texture_resource(*resource);
		return nullptr;
	}
	HIP::ImageView *get_physical_history_texture_resource(const RenderTextureResource &resource)
	{
		return get_physical_history_texture_resource(resource.get_physical_index());
	}
	HIP::Buffer &get_physical_buffer_resource(const RenderBufferResource &resource)
	{
		return get_physical_buffer_resource(resource.get_physical_index());
	}
	HIP::Buffer *maybe_get_physical_buffer_resource(RenderBufferResource *resource)
	{
		if (resource && resource->get_physical_index() != RenderResource::Unused)
			return &get_physical_buffer_resource(*resource);
		return nullptr;
	}

	// For keeping feed-back resources alive during rebaking (render_graph.cpp:504-529).
	std::vector<HIP::BufferHandle> consume_physical_buffers() const { return physical_buffers; }
	void install_physical_buffers(std::vector<HIP::BufferHandle> buffers) { physical_buffers = std::move(buffers); }
	HIP::BufferHandle consume_persistent_physical_buffer_resource(unsigned index) const;
	void install_persistent_physical_buffer_resource(unsigned index, HIP::BufferHandle buffer);

	static RenderGraphQueueFlagBits get_default_post_graphics_queue() { return RENDER_GRAPH_QUEUE_GRAPHICS_BIT; }
	static RenderGraphQueueFlagBits get_default_compute_queue() { return RENDER_GRAPH_QUEUE_COMPUTE_BIT; }

	// Introspection used by the tests and by log().
	const std::vector<unsigned> &get_baked_pass_order() const { return pass_stack; }
	const RenderPass &get_pass(unsigned index) const { return *passes[index]; }
	RenderPass &get_pass(unsigned index) { return *passes[index]; }
	const std::vector<ResourceDimensions> &get_physical_dimensions() const { return physical_dimensions; }
	bool physical_resource_has_history(unsigned index) const { return physical_image_has_history[index]; }
	unsigned get_swapchain_physical_index() const { return swapchain_physical_index; }

	// Per-pass GPU time accumulated while enable_timestamps(true) (Device::timestamp_log analogue,
	// application_headless.cpp:616-654): {pass name -> (count, total ms)}.
	struct TimestampReport
	{
		std::string tag;
		uint64_t count;
		double total_ms;
	};
	std::vector<TimestampReport> collect_timestamps();
	// Device::timestamp_log_reset (application_headless.cpp:591): drop what was accumulated so far (the warm-up frame).
	void reset_timestamps();

private:
	HIP::Device *device = nullptr;
	std::vector<std::unique_ptr<RenderPass>> passes;
	std::vector<std::unique_ptr<RenderResource>> resources;
	std::unordered_map<std::string, unsigned> pass_to_index;
	std::unordered_map<std::string, unsigned> resource_to_index;
	std::string backbuffer_source;
	std::vector<unsigned> pass_stack;
	ResourceDimensions swapchain_dimensions;

	std::vector<std::unordered_set<unsigned>> pass_dependencies;
	std::vector<std::unordered_set<unsigned>> pass_merge_dependencies;

	std::vector<ResourceDimensions> physical_dimensions;
	std::vector<HIP::ImageView *> physical_attachments;
	std::vector<HIP::BufferHandle> physical_buffers;
	std::vector<HIP::ImageHandle> physical_image_attachments;
	std::vector<HIP::ImageHandle> physical_history_image_attachments;
	std::vector<bool> physical_image_has_history;
	HIP::ImageView *swapchain_attachment = nullptr;
	unsigned swapchain_physical_index = RenderResource::Unused;
	bool enabled_timestamps = false;

	struct PassTimestamp
	{
		unsigned pass;
		void *start, *stop; // hipEvent_t
	};
	std::vector<PassTimestamp> pending_timestamps;
	std::vector<void *> event_pool;

	// Cross-stream ordering.  Filled by bake(): the physical resources each pass reads / writes and the stream it runs
	// on.  At execution every pass that shares a resource with a pass on the OTHER stream waits on that pass's "done"
	// event (RAW, WAW and WAR); the state survives across frames, which is what lets frame N+1's hoisted passes start
	// as soon as frame N's readers of their outputs have finished.  With a single stream in use nothing is recorded.
	bool hoist_independent_compute = true;
	// The passes behind the last pass of the frame that leaves anything to the next frame (history, feedback) and that take ONE image from
	// what precedes them -- post-tonemap anti-aliasing reading `tonemapped` -- run on a stream of their own: frame N's tail beside frame
	// N + 1's resolve / bloom / tonemap (reference ordering kept: scene_viewer_application.cpp:1230-1261, smaa.cpp:95-208).  The image
	// handed over exists in HandOverCopies rotating copies like a front-to-back resource.
	bool split_tail = true;
	bool uses_async_stream = false;
	std::vector<uint8_t> pass_stream;
	std::vector<bool> pass_needs_sync; // touches a physical resource that the other stream also touches
	bool blit_needs_sync = false;
	std::vector<std::vector<unsigned>> pass_reads_physical, pass_writes_physical;
	// hipEvent_t ring per pass: a sync entry must keep naming the record of the frame it was made in (the alternate copy
	// of a double-buffered buffer was last read two frames ago), so the event of frame f is slot f % EventRing.  The
	// host never runs more than Device::StagingFrames - 1 frames ahead, so a slot is complete long before its reuse.
	enum { EventRing = 4 };
	std::vector<void *> pass_done_event;
	uint64_t frame_counter = 0;
	uint64_t last_device_frame = 0; // Device::get_frame_number() at the last enqueue (the two rings advance in lockstep)
	enum { StreamCount = 4 }; // generic (back), async compute, front, tail: HIP::CommandBuffer::Type
	struct PhysicalSync
	{
		void *last_write = nullptr;
		int write_stream = -1;
		void *last_read[StreamCount] = {};
		// who recorded those events (pass index, frame), for GRANITE_SYNC_DEBUG=1 traces
		int write_pass = -1, read_pass[StreamCount] = {-1, -1, -1, -1};
		uint64_t write_frame = 0, read_frame[StreamCount] = {};
		// the device's frame number at those records: which of a type's alternating streams they went to (HIP::Device::same_stream)
		uint64_t write_device_frame = 0, read_device_frame[StreamCount] = {};
	};
	std::vector<PhysicalSync> physical_sync;
	// Buffers written by a hoisted pass exist twice and alternate per frame (like an image with history), so the
	// hoisted pass of frame N+1 never waits for frame N's consumers: write-after-read across frames disappears.
	std::vector<bool> physical_buffer_double;
	// HandOverCopies - 1 spare copies per resource; every frame the current copy goes to the back of the ring and the oldest
	// spare becomes current.  Three copies: the producer of frame N+1 writes what the consumers of frame N-2 read last, so
	// a back-of-frame that runs late (it shares the chip with the next frame's lighting) never stalls the front.
	enum { HandOverCopies = 3 };
	std::vector<HIP::BufferHandle> physical_buffers_alternate[HandOverCopies - 1];
	std::vector<HIP::ImageHandle> physical_images_alternate[HandOverCopies - 1];
	std::vector<PhysicalSync> physical_sync_alternate[HandOverCopies - 1];
	void build_stream_assignment();
	void build_physical_passes();
	void build_aliases();
	// Which baked passes the reference would fold into one VkRenderPass as subpasses (build_physical_passes,
	// render_graph.cpp:1221-1392).  The HIP executor launches every pass on its own; the grouping is kept because the
	// reference measures attachment lifetimes for aliasing in physical passes, and for graph dumps.
	std::vector<unsigned> pass_physical_pass;
	unsigned physical_pass_count = 0;
	bool alias_disjoint_images = true;
	std::vector<unsigned> physical_aliases; // physical index whose image this one shares, or Unused
	std::unordered_map<std::string, std::pair<uint64_t, double>> timestamp_accum;
	std::vector<std::string> timestamp_order;

	void filter_passes(std::vector<unsigned> &list);
	void validate_passes();
	void build_physical_resources();
	void traverse_dependencies(const RenderPass &pass, unsigned stack_count);
	void depend_passes_recursive(const RenderPass &pass, const std::unordered_set<unsigned> &passes, unsigned stack_count, bool no_check,
	                             bool ignore_self, bool merge_dependency);
	bool depends_on_pass(unsigned dst_pass, unsigned src_pass);
	void reorder_passes(std::vector<unsigned> &passes);
	void setup_physical_buffer(HIP::Device &device, unsigned attachment);
	void setup_physical_image(HIP::Device &device, unsigned attachment);
	void *acquire_event();
};
} // namespace Granite
