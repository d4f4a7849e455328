// HIP:: — what Vulkan::{Device,ImageView,Buffer,CommandBuffer} are to Granite's pass callbacks
// (vulkan/device.hpp, vulkan/command_buffer.hpp), reduced to what the image-space chain uses:
// linear HBM images/buffers, an in-order HIP stream per queue, and the C-ABI kernel launchers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>
#include "vk_subset.hpp"
#include "../../../include/granite_hip.h"

namespace HIP
{
class Device;

// A linear row-major 2-D image in HBM.  Owns its memory unless wrapping an external pointer (swapchain image).
// With levels > 1 the allocation holds a mip chain, level after level, each level tightly packed
// (gr_mip_chain_offset); get_view() is level 0.
class Image
{
public:
	Image(Device &device, unsigned width, unsigned height, VkFormat format, const std::string &name, unsigned levels = 1);
	Image(unsigned width, unsigned height, VkFormat format, void *external_ptr);
	~Image();
	Image(const Image &) = delete;
	void operator=(const Image &) = delete;

	const gr_image &get_view() const { return view; }
	gr_image &get_view() { return view; }
	unsigned get_width() const { return view.width; }
	unsigned get_height() const { return view.height; }
	VkFormat get_format() const { return VkFormat(view.format); }
	size_t get_size_bytes() const { return levels > 1 ? chain_bytes : size_t(view.pitch_bytes) * view.height; }
	unsigned get_levels() const { return levels; }
	gr_image get_level_view(unsigned level) const; // Vulkan::ImageView with base_level = level, levels = 1
	void *get_device_pointer() const { return view.ptr; }
	const std::string &get_name() const { return name; }

private:
	Device *device = nullptr;
	gr_image view = {};
	std::string name;
	bool owned = false;
	unsigned levels = 1;
	size_t chain_bytes = 0;
};
using ImageHandle = std::shared_ptr<Image>;
using ImageView = Image; // callbacks receive HIP::ImageView& where the reference hands out Vulkan::ImageView&

class Buffer
{
public:
	Buffer(Device &device, size_t size, VkBufferUsageFlags usage, const std::string &name);
	~Buffer();
	Buffer(const Buffer &) = delete;
	void operator=(const Buffer &) = delete;
	void *get_device_pointer() const { return ptr; }
	size_t get_size() const { return size; }
	VkBufferUsageFlags get_usage() const { return usage; }

private:
	Device *device;
	void *ptr = nullptr;
	size_t size;
	VkBufferUsageFlags usage;
	std::string name;
};
using BufferHandle = std::shared_ptr<Buffer>;

// Records work for one pass.  "Recording" is an asynchronous launch on the pass's stream; there is nothing to submit.
class CommandBuffer
{
public:
	// Generic: graphics + compute queue of the reference; AsyncCompute: its async compute queue; Front: the executor's
	// frame-pipelining stream (render_graph.hpp, set_hoist_independent_compute); Tail: what follows the last pass of a frame
	// that hands anything to the next frame (post-tonemap anti-aliasing), so that it runs beside the next frame's back.
	enum class Type { Generic, AsyncCompute, Front, Tail, Count };
	CommandBuffer(Device &device_, void *stream_, Type type_) : device(device_), stream(stream_), type(type_) {}
	Device &get_device() { return device; }
	gr_ctx *get_context() const;
	gr_stream get_stream() const { return stream; }
	Type get_command_buffer_type() const { return type; }

	// cmd.barrier(...) between dispatches of one pass (e.g. hdr.cpp:356-378): a HIP stream is in-order and kernel
	// boundaries make writes visible, so this is a no-op kept for call-site parity.
	void barrier(VkPipelineStageFlags2, VkAccessFlags2, VkPipelineStageFlags2, VkAccessFlags2) {}

	// cmd.update_buffer analogue (clusterer.cpp:1178-1207): async H2D from a pinned staging ring owned by the device.
	void update_buffer(const Buffer &dst, size_t offset, size_t size, const void *data);
	// Several update_buffer calls as one launch: stage every range in the pinned ring, then one gr_upload_batch kernel
	// (no copy engine, no cross-engine signalling in the middle of a short pass).
	struct BufferUpdate
	{
		const Buffer *dst;
		size_t offset, size;
		const void *data;
	};
	void update_buffers(const BufferUpdate *updates, unsigned count);
	// Copies `size` bytes into this frame's slot of the pinned staging ring and returns the device-visible pointer: for a kernel
	// that consumes CPU-packed data where it lies instead of waiting for an upload launch.
	const void *stage(const void *data, size_t size);
	void fill_buffer(const Buffer &dst, size_t offset, size_t size);
	void copy_image(const Image &dst, const Image &src);
	void clear_image(const Image &dst);

	// Pre-recorded launch sequences (what VkCommandBuffer re-submission is to a Vulkan host).  `record` makes a fixed sequence of
	// launches on this command buffer's stream and nothing else -- no allocation, no event, no other stream -- and `key` holds
	// every value those launches depend on (device pointers, image geometry, push constants).  The second time a key is seen at a
	// call site the sequence is captured into a hipGraph; from then on the same key costs ONE graph launch instead of one API
	// call per kernel (measured on this runtime: 6 launches 19.2 us -> 5.6 us of host time).  A key that keeps changing (moving
	// camera) is never captured and costs a comparison.  Bypassed while the per-kernel timing brackets could touch one of
	// `kernel_names`.  OPT-IN (GRANITE_LAUNCH_GRAPHS=1): it takes the host from 0.078 to 0.063 ms per 4K frame, but the nodes of a
	// graph are dispatched with more latency than direct launches on this runtime and the FRAME gets slower (4K +2 %, 1080p
	// +5 %, tools: bench.py with / without the variable) -- the executor is GPU-bound, so direct launches stay the default.
	class LaunchKey
	{
	public:
		template <typename T> LaunchKey &add(const T &v)
		{
			static_assert(std::is_trivially_copyable<T>::value, "launch keys are raw bytes");
			const auto *p = reinterpret_cast<const uint8_t *>(&v);
			bytes.insert(bytes.end(), p, p + sizeof(T));
			return *this;
		}
		const std::vector<uint8_t> &get() const { return bytes; }
	private:
		std::vector<uint8_t> bytes;
	};
	void replayable(const char *site, const LaunchKey &key, std::initializer_list<const char *> kernel_names, const std::function<void()> &record);

	// Throws std::runtime_error with gr_last_error() when a launcher fails (the reference LOGEs and continues; a
	// silently wrong frame is worse for an executor that is being validated).
	void check(int status, const char *what);

private:
	Device &device;
	gr_stream stream;
	Type type;
};

class Device
{
public:
	explicit Device(int device_index = 0);
	~Device();
	Device(const Device &) = delete;
	void operator=(const Device &) = delete;

	gr_ctx *get_context() const { return ctx; }
	int get_device_index() const { return index; }
	void make_current() const; // hipSetDevice for the calling thread
	// Handing out a stream counts as enqueueing on it: the frame fence of that stream (below) is due again.
	gr_stream get_stream(CommandBuffer::Type type) const
	{
		stream_dirty[int(type)] = true;
		return physical_stream(int(type), frame_number);
	}
	// A fourth in-order stream for collectives that run beside the frame (the output all-gather of row-band tiling):
	// created on first use, drained by wait_idle() like the executor's own.
	gr_stream get_collective_stream();

	ImageHandle create_image(unsigned width, unsigned height, VkFormat format, const std::string &name, unsigned levels = 1);
	BufferHandle create_buffer(size_t size, VkBufferUsageFlags usage, const std::string &name);

	// Images are allocated with their row count rounded up to a multiple of this (row-band all-gathers write
	// rank_count * ceil(height / rank_count) rows).  The logical height is unchanged.
	void set_image_row_granularity(unsigned rows) { image_row_granularity = rows ? rows : 1; }
	unsigned get_image_row_granularity() const { return image_row_granularity; }

	// Pinned-host staging for update_buffer: N frames in flight, each with its own bump allocator.
	void *allocate_staging(size_t size);
	// The fence of the frame being enqueued for one stream (a hipEvent_t owned by the device, re-recorded StagingFrames frames later).
	// The executor records it itself behind the last run of passes it puts on that stream and publishes the run's accesses under it
	// (record_frame_fence): one record per stream and frame instead of the run's event plus the fence.  next_frame_context() records
	// the fences of the streams that were handed out since (get_stream) and of no others -- an idle stream costs nothing.
	void *frame_fence(CommandBuffer::Type type) const { return staging[staging_index].fence[int(type)]; }
	static constexpr unsigned FrameFenceRing = 4; // = StagingFrames: a frame's fences are re-recorded this many frames later
	void record_frame_fence(CommandBuffer::Type type);
	bool front_alternates() const { return front_alternate != nullptr; }
	// Whether two accesses on streams of the same type, recorded in frames a and b, were put on the same in-order stream (false only for
	// the front type, whose stream alternates with the frame's parity).
	bool same_stream(CommandBuffer::Type type, uint64_t frame_a, uint64_t frame_b) const
	{
		return physical_stream(int(type), frame_a) == physical_stream(int(type), frame_b);
	}
	// Every launch, copy and event record the executor's streams were given in frames up to this one (device frame numbers) has completed: what
	// frame pacing waited for on the host in next_frame_context().  An event recorded in such a frame needs neither a query nor a wait.
	uint64_t get_completed_frame() const { return completed_through; }
	// Number of the frame being enqueued (from 1; advanced by next_frame_context()).
	uint64_t get_frame_number() const { return frame_number; }
	void next_frame_context();
	void wait_idle();

	// The cache behind CommandBuffer::replayable(): dropped when the graph is re-baked (pointers change) and with the device.
	void reset_launch_cache();
	unsigned get_launch_graph_replays() const { return launch_graph_replays; }

	size_t get_allocated_bytes() const { return allocated_bytes; }
	// Seconds the host has spent blocked waiting for the GPU to release a staging slot (frame pacing back-pressure).
	double get_blocked_seconds() const { return blocked_seconds; }
	void account_alloc(ptrdiff_t delta) { allocated_bytes += delta; }

private:
	int index;
	gr_ctx *ctx = nullptr;
	gr_stream streams[int(CommandBuffer::Type::Count)] = {};
	// Experiment, off by default (GRANITE_ALTERNATE_FRONT=1): the front of a frame (G-buffer producer, lighting) alternates between TWO in-order
	// streams by frame parity.  Consecutive lighting launches share no resource (HDR-main, the G-buffer targets and the cluster buffers rotate
	// through three copies), so nothing but the stream makes launch N + 1 wait for the last workgroup of launch N.  Correct (hazards between
	// two front passes of different parity are ordered by events, frame pacing waits for both streams: the executor's GPU tests pass with it)
	// and 40 % slower: two lighting launches in flight take 250 us each and starve the back of the frame
	// (profiles/r06_front_stream_alternation.txt).
	gr_stream front_alternate = nullptr;
	gr_stream physical_stream(int type, uint64_t frame) const
	{
		return type == int(CommandBuffer::Type::Front) && front_alternate && (frame & 1u) != 0 ? front_alternate : streams[type];
	}
	gr_stream collective_stream = nullptr;
	struct StagingFrame
	{
		uint8_t *base = nullptr;
		size_t offset = 0;
		// hipEvent_t per executor stream, recorded when the frame was enqueued: staging memory is read by kernels on any of
		// the streams (the cluster pass's batched upload runs on the async-compute stream), so a slot is reusable only once
		// every stream has passed the frame that used it.
		void *fence[int(CommandBuffer::Type::Count)] = {};
		// the frame (Device::frame_number) at which each fence was last recorded; 0: never.  An event whose record is older than the
		// slot's current frame belongs to an earlier use of the slot and says nothing about this one.
		uint64_t fence_frame[int(CommandBuffer::Type::Count)] = {};
	};
	static constexpr size_t StagingBytes = 4u << 20;
	static constexpr unsigned StagingFrames = FrameFenceRing;
	StagingFrame staging[StagingFrames];
	unsigned staging_index = 0;
	uint64_t frame_number = 1; // of the frame being enqueued (from 1); its slot is staging[staging_index] = staging[(frame_number - 1) % StagingFrames]
	mutable bool stream_dirty[int(CommandBuffer::Type::Count)] = {};
	uint64_t completed_through = 0; // see get_completed_frame()
	uint64_t completed_frame[int(CommandBuffer::Type::Count)][2] = {}; // per stream (and parity of the front's two): the newest frame whose fence the pacing has seen complete
	size_t allocated_bytes = 0;
	double blocked_seconds = 0.0;
	unsigned image_row_granularity = 1;

	friend class CommandBuffer;
	struct ReplaySite
	{
		struct Entry
		{
			std::vector<uint8_t> key;
			void *exec = nullptr; // hipGraphExec_t
		};
		std::vector<Entry> captured;               // a handful of keys at most (ping-pong attachments, ring slots)
		std::vector<std::vector<uint8_t>> seen;    // keys met once and not captured yet
	};
	std::map<std::string, ReplaySite> replay_sites;
	bool launch_graphs = false; // opt-in (GRANITE_LAUNCH_GRAPHS): measured slower per frame than direct launches, see replayable()
	unsigned launch_graph_replays = 0;
};
} // namespace HIP
