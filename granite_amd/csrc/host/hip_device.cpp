#include "hip_device.hpp"
#include "timeline_trace.hpp"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace HIP
{
static void throw_hip(hipError_t err, const char *what)
{
	if (err != hipSuccess)
		throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(err));
}

Image::Image(Device &device_, unsigned width, unsigned height, VkFormat format, const std::string &name_, unsigned levels_)
    : device(&device_), name(name_), owned(true), levels(levels_ ? levels_ : 1)
{
	const unsigned bpp = vk_format_block_size(format);
	if (!bpp)
		throw std::logic_error("Unsupported image format for a linear HBM attachment: " + name);
	view.width = width;
	view.height = height;
	view.pitch_bytes = width * bpp;
	view.format = format;
	const unsigned gran = device->get_image_row_granularity();
	const size_t padded_rows = (size_t(height) + gran - 1) / gran * gran;
	size_t bytes = size_t(view.pitch_bytes) * padded_rows;
	if (levels > 1)
		bytes = chain_bytes = gr_mip_chain_size(width, height, bpp, levels);
	int ret = gr_alloc(device->get_context(), bytes, &view.ptr); // zero-initialised
	if (ret < 0)
		throw std::runtime_error(std::string("gr_alloc failed for ") + name + ": " + gr_last_error(device->get_context()));
	device->account_alloc(ptrdiff_t(get_size_bytes()));
}

Image::Image(unsigned width, unsigned height, VkFormat format, void *external_ptr) : name("external"), owned(false)
{
	view.ptr = external_ptr;
	view.width = width;
	view.height = height;
	view.pitch_bytes = width * vk_format_block_size(format);
	view.format = format;
}

gr_image Image::get_level_view(unsigned level) const
{
	if (level >= levels)
		throw std::logic_error("Image level out of range: " + name);
	const unsigned bpp = vk_format_block_size(VkFormat(view.format));
	gr_image v = view;
	v.width = std::max(view.width >> level, 1u);
	v.height = std::max(view.height >> level, 1u);
	v.pitch_bytes = v.width * bpp;
	v.ptr = static_cast<uint8_t *>(view.ptr) + gr_mip_chain_offset(view.width, view.height, bpp, level);
	return v;
}

Image::~Image()
{
	if (owned && view.ptr)
	{
		gr_free(device->get_context(), view.ptr);
		device->account_alloc(-ptrdiff_t(get_size_bytes()));
	}
}

Buffer::Buffer(Device &device_, size_t size_, VkBufferUsageFlags usage_, const std::string &name_)
    : device(&device_), size(size_), usage(usage_), name(name_)
{
	int ret = gr_alloc(device->get_context(), size, &ptr); // zero-initialised (render_graph.cpp:2586-2587)
	if (ret < 0)
		throw std::runtime_error(std::string("gr_alloc failed for ") + name + ": " + gr_last_error(device->get_context()));
	device->account_alloc(ptrdiff_t(size));
}

Buffer::~Buffer()
{
	if (ptr)
	{
		gr_free(device->get_context(), ptr);
		device->account_alloc(-ptrdiff_t(size));
	}
}

gr_ctx *CommandBuffer::get_context() const
{
	return device.get_context();
}

void CommandBuffer::check(int status, const char *what)
{
	if (status < 0)
		throw std::runtime_error(std::string(what) + ": " + gr_last_error(device.get_context()));
}

void CommandBuffer::update_buffer(const Buffer &dst, size_t offset, size_t size, const void *data)
{
	if (!size)
		return;
	if (offset + size > dst.get_size())
		throw std::logic_error("update_buffer out of range");
	void *staging = device.allocate_staging(size);
	memcpy(staging, data, size);
	check(gr_upload(get_context(), stream, static_cast<uint8_t *>(dst.get_device_pointer()) + offset, staging, size),
	      "update_buffer");
}

void CommandBuffer::update_buffers(const BufferUpdate *updates, unsigned count)
{
	gr_upload_range ranges[GR_MAX_UPLOAD_RANGES];
	unsigned n = 0;
	for (unsigned i = 0; i < count; i++)
	{
		auto &u = updates[i];
		if (!u.size)
			continue;
		if (u.offset + u.size > u.dst->get_size())
			throw std::logic_error("update_buffers out of range");
		if ((u.size & 3u) || (u.offset & 3u) || n == GR_MAX_UPLOAD_RANGES)
		{
			update_buffer(*u.dst, u.offset, u.size, u.data); // odd sizes take the copy-engine path
			continue;
		}
		void *staging = device.allocate_staging(u.size);
		memcpy(staging, u.data, u.size);
		ranges[n++] = {static_cast<uint8_t *>(u.dst->get_device_pointer()) + u.offset, staging, u.size};
	}
	if (n)
		check(gr_upload_batch(get_context(), stream, ranges, n), "update_buffers");
}

const void *CommandBuffer::stage(const void *data, size_t size)
{
	void *staging = device.allocate_staging(size);
	memcpy(staging, data, size);
	return staging;
}

void CommandBuffer::replayable(const char *site, const LaunchKey &key, std::initializer_list<const char *> kernel_names,
                               const std::function<void()> &record)
{
	bool direct = !device.launch_graphs;
	for (const char *name : kernel_names)
		direct = direct || gr_timing_brackets(get_context(), name) != 0; // a bracketed launch records events: not replayable
	if (direct)
	{
		record();
		return;
	}
	auto &entries = device.replay_sites[site];
	auto hip_stream = static_cast<hipStream_t>(stream);
	for (auto &e : entries.captured)
	{
		if (e.key == key.get())
		{
			throw_hip(hipGraphLaunch(static_cast<hipGraphExec_t>(e.exec), hip_stream), "hipGraphLaunch");
			device.launch_graph_replays++;
			return;
		}
	}
	bool seen_before = false;
	for (auto &k : entries.seen)
		seen_before = seen_before || k == key.get();
	if (!seen_before)
	{
		// first sight: launch directly and remember the key (a bounded memory: keys that never come back are forgotten)
		if (entries.seen.size() >= 8)
			entries.seen.erase(entries.seen.begin());
		entries.seen.push_back(key.get());
		record();
		return;
	}
	// second sight: capture the sequence, keep the instantiated graph, run it
	hipGraph_t graph = nullptr;
	throw_hip(hipStreamBeginCapture(hip_stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
	try
	{
		record();
	}
	catch (...)
	{
		(void)hipStreamEndCapture(hip_stream, &graph);
		if (graph)
			(void)hipGraphDestroy(graph);
		throw;
	}
	throw_hip(hipStreamEndCapture(hip_stream, &graph), "hipStreamEndCapture");
	hipGraphExec_t exec = nullptr;
	const hipError_t instantiated = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
	(void)hipGraphDestroy(graph);
	throw_hip(instantiated, "hipGraphInstantiate");
	if (entries.captured.size() >= 16)
	{
		(void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(entries.captured.front().exec));
		entries.captured.erase(entries.captured.begin());
	}
	entries.captured.push_back({key.get(), exec});
	for (auto itr = entries.seen.begin(); itr != entries.seen.end(); ++itr)
		if (*itr == key.get())
		{
			entries.seen.erase(itr);
			break;
		}
	throw_hip(hipGraphLaunch(exec, hip_stream), "hipGraphLaunch");
	device.launch_graph_replays++;
}

void Device::reset_launch_cache()
{
	for (auto &site : replay_sites)
		for (auto &e : site.second.captured)
			(void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(e.exec));
	replay_sites.clear();
}

void CommandBuffer::fill_buffer(const Buffer &dst, size_t offset, size_t size)
{
	check(gr_fill_zero(get_context(), stream, static_cast<uint8_t *>(dst.get_device_pointer()) + offset, size), "fill_buffer");
}

void CommandBuffer::copy_image(const Image &dst, const Image &src)
{
	if (dst.get_size_bytes() != src.get_size_bytes())
		throw std::logic_error("copy_image: size mismatch");
	check(gr_copy(get_context(), stream, dst.get_device_pointer(), src.get_device_pointer(), src.get_size_bytes()), "copy_image");
}

void CommandBuffer::clear_image(const Image &dst)
{
	check(gr_fill_zero(get_context(), stream, dst.get_device_pointer(), dst.get_size_bytes()), "clear_image");
}

Device::Device(int device_index) : index(device_index)
{
	ctx = gr_create(device_index);
	if (!ctx)
		throw std::runtime_error("gr_create failed: no usable HIP device " + std::to_string(device_index));
	// Queue priorities for the executor's frame pipelining: the back of a frame (bloom pyramid, tonemap: short, dependent,
	// bandwidth-bound launches) and the cluster build must not queue behind the next frame's lighting kernel, which fills
	// every wave slot of the chip for ~200 us; lighting is throughput work and runs at the lowest priority.
	int least = 0, greatest = 0;
	throw_hip(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
	const int middle = (least + greatest) / 2;
	int priorities[int(CommandBuffer::Type::Count)] = {greatest, middle, least, greatest}; // Generic, AsyncCompute, Front, Tail
	if (const char *env = getenv("GRANITE_STREAM_PRIORITIES")) // experiment: a letter per stream from {h, m, l}, e.g. "lmhh"
	{
		fprintf(stderr, "[granite-hip] note: stream priorities overridden by GRANITE_STREAM_PRIORITIES=%s (results unchanged, timing differs)\n", env);
		for (int i = 0; i < int(CommandBuffer::Type::Count) && env[i]; i++)
			priorities[i] = env[i] == 'h' ? greatest : env[i] == 'm' ? middle : least;
	}
	if (getenv("GRANITE_LAUNCH_GRAPHS"))
	{
		fprintf(stderr, "[granite-hip] note: GRANITE_LAUNCH_GRAPHS is set: repeating launch sequences are replayed as hipGraphs (results unchanged, timing differs)\n");
		launch_graphs = true;
	}
	for (int i = 0; i < int(CommandBuffer::Type::Count); i++)
	{
		hipStream_t stream;
		throw_hip(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, priorities[i]), "hipStreamCreateWithPriority");
		streams[i] = stream;
	}
	{
		// Opt-in experiment (GRANITE_ALTERNATE_FRONT=1): a second front stream, see front_alternate.  Measured and NOT the default: two lighting
		// launches in flight cost the 4K frame 0.199 -> 0.283-0.292 ms (profiles/r06_front_stream_alternation.txt).
		const char *env = getenv("GRANITE_ALTERNATE_FRONT");
		if (env && atoi(env) != 0)
		{
			hipStream_t stream;
			throw_hip(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, priorities[int(CommandBuffer::Type::Front)]), "hipStreamCreateWithPriority");
			front_alternate = stream;
		}
	}
	for (auto &frame : staging)
	{
		throw_hip(hipHostMalloc(reinterpret_cast<void **>(&frame.base), StagingBytes, hipHostMallocDefault), "hipHostMalloc");
		for (auto &fence : frame.fence)
		{
			// ordering between this device's streams and completion seen by the host: no system-scope release is needed for either (the
			// executor's run events are created the same way, render_graph.cpp; GRANITE_SYNC_EVENT_SYSTEM_FENCE=1 restores the fence)
			hipEvent_t e;
			throw_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming | (getenv("GRANITE_SYNC_EVENT_SYSTEM_FENCE") ? 0u : unsigned(hipEventDisableSystemFence))),
			          "hipEventCreate");
			fence = e;
		}
	}
}

Device::~Device()
{
	(void)hipSetDevice(index);
	(void)hipDeviceSynchronize();
	reset_launch_cache();
	for (auto &frame : staging)
	{
		if (frame.base)
			(void)hipHostFree(frame.base);
		for (auto &fence : frame.fence)
			if (fence)
				(void)hipEventDestroy(static_cast<hipEvent_t>(fence));
	}
	for (auto &s : streams)
		if (s)
			(void)hipStreamDestroy(static_cast<hipStream_t>(s));
	if (front_alternate)
		(void)hipStreamDestroy(static_cast<hipStream_t>(front_alternate));
	if (collective_stream)
		(void)hipStreamDestroy(static_cast<hipStream_t>(collective_stream));
	if (ctx)
		gr_destroy(ctx);
}

void Device::make_current() const
{
	throw_hip(hipSetDevice(index), "hipSetDevice");
}

ImageHandle Device::create_image(unsigned width, unsigned height, VkFormat format, const std::string &name, unsigned levels)
{
	return std::make_shared<Image>(*this, width, height, format, name, levels);
}

BufferHandle Device::create_buffer(size_t size, VkBufferUsageFlags usage, const std::string &name)
{
	return std::make_shared<Buffer>(*this, size, usage, name);
}

void *Device::allocate_staging(size_t size)
{
	auto &frame = staging[staging_index];
	size_t aligned = (frame.offset + 63) & ~size_t(63);
	if (aligned + size > StagingBytes)
		throw std::runtime_error("staging ring exhausted for this frame");
	frame.offset = aligned + size;
	return frame.base + aligned;
}

void Device::next_frame_context()
{
	// Mark the copies of the frame just recorded, then make sure the slot we are about to reuse has drained.
	// One fence per stream: a graph built through the public API may consume staging memory on a stream nothing on the
	// generic stream depends on in that frame, so the generic stream's progress alone does not bound the host's lead.
	// A stream nobody was handed since its last fence has nothing new to fence and records nothing: `fence_frame` says which frame each
	// event of a slot was last recorded for, so an older record left in a slot is never mistaken for this frame's.
	auto &done = staging[staging_index];
	for (int i = 0; i < int(CommandBuffer::Type::Count); i++)
		if (stream_dirty[i])
		{
			throw_hip(hipEventRecord(static_cast<hipEvent_t>(done.fence[i]), static_cast<hipStream_t>(physical_stream(i, frame_number))), "hipEventRecord");
			done.fence_frame[i] = frame_number;
			stream_dirty[i] = false;
		}
	staging_index = (staging_index + 1) % StagingFrames;
	frame_number++;
	// How far the host may run ahead.  The slot about to be reused belongs to the frame StagingFrames back: all the staging ring needs is
	// that frame complete on every stream.  The host waits for the frame `lead` = 2 back instead: while it enqueues frame N it then knows
	// frame N - 3 to be complete -- the frame whose ring copies (cluster buffers, HDR-main: three deep) frame N overwrites -- so the
	// write-after-read waits against it are found complete by hipEventQuery and never become barrier packets in front of the cluster
	// build and of the lighting kernel.  Two frames of queued work are 0.1 - 0.4 ms, several times what the host needs to enqueue one.
	// Measured (round 4, profiles/r04_host_lead_ab.txt): 4K / 4096 lights 0.2224 -> 0.2195 ms sustained, config 4 0.662 -> 0.634,
	// config 2 0.0703 -> 0.0668, config 1 0.0543 -> 0.0471 ms.  GRANITE_HOST_LEAD_FRAMES=3 restores the longer lead.
	static const unsigned lead = []() {
		const char *e = getenv("GRANITE_HOST_LEAD_FRAMES");
		const unsigned v = e ? unsigned(atoi(e)) : 2u;
		return v >= 1u && v <= StagingFrames - 1u ? v : 2u;
	}();
	// Per stream: the newest record that is at least `lead` frames old -- streams are in order, so it covers everything the stream was
	// given before it.  A stream that was not handed out in frame N - lead (a conditional pass, an API user's async stream) has no
	// record there; its last one may sit in an older slot, down to the slot about to be reused (frame N - StagingFrames), and that is
	// the one to wait for: the staging memory of that slot may still be read by the stream's copies.  Records older than
	// the ring were waited for when their slot came round.  One hipEventQuery per stream that has been used within the ring, none
	// for an idle stream.
	// (frame_number is the frame about to be enqueued; "lead frames old" counts from the frame just recorded, frame_number - 1.)
	// (The front type alternates between two streams: a record covers the frames of its own parity, so the newest old-enough record of
	// EACH parity is waited for.)
	for (int i = 0; i < int(CommandBuffer::Type::Count); i++)
	{
		const bool two_streams = i == int(CommandBuffer::Type::Front) && front_alternate != nullptr;
		bool covered[2] = {false, !two_streams};
		for (unsigned back = lead + 1u; back <= StagingFrames && back < frame_number; back++)
		{
			auto &slot = staging[(frame_number - back - 1u) % StagingFrames]; // frame f was enqueued into slot (f - 1) % StagingFrames
			if (slot.fence_frame[i] != frame_number - back)
				continue;
			const unsigned parity = two_streams ? unsigned((frame_number - back) & 1u) : 0u;
			if (covered[parity])
				continue;
			covered[parity] = true;
			// a record found complete (or waited for) at an earlier turn is not asked again: an intermittently used stream's last record
			// comes round once more as it ages through the ring
			if (slot.fence_frame[i] <= completed_frame[i][parity])
				continue;
			completed_frame[i][parity] = slot.fence_frame[i];
			const auto fence = static_cast<hipEvent_t>(slot.fence[i]);
			if (hipEventQuery(fence) != hipSuccess)
			{
				auto t0 = std::chrono::steady_clock::now();
				{
					GRANITE_SCOPED_TIMELINE_EVENT("wait-for-frame-in-flight"); // back-pressure: the host is `lead` frames ahead
					throw_hip(hipEventSynchronize(fence), "hipEventSynchronize");
				}
				blocked_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			}
			if (covered[0] && covered[1])
				break;
		}
	}
	// every stream's newest record of age lead + 1 or more has now been seen complete, and with it (the streams are in order, every stream
	// handed out in a frame is fenced at the end of it) all work of the frames up to frame_number - (lead + 1)
	completed_through = frame_number > lead + 1u ? frame_number - (lead + 1u) : 0u;
	staging[staging_index].offset = 0;
}

void Device::record_frame_fence(CommandBuffer::Type type)
{
	auto &frame = staging[staging_index];
	throw_hip(hipEventRecord(static_cast<hipEvent_t>(frame.fence[int(type)]), static_cast<hipStream_t>(physical_stream(int(type), frame_number))), "hipEventRecord");
	frame.fence_frame[int(type)] = frame_number;
	stream_dirty[int(type)] = false;
}

void Device::wait_idle()
{
	GRANITE_SCOPED_TIMELINE_EVENT("wait-idle");
	for (auto &s : streams)
		throw_hip(hipStreamSynchronize(static_cast<hipStream_t>(s)), "hipStreamSynchronize");
	if (front_alternate)
		throw_hip(hipStreamSynchronize(static_cast<hipStream_t>(front_alternate)), "hipStreamSynchronize");
	if (collective_stream)
		throw_hip(hipStreamSynchronize(static_cast<hipStream_t>(collective_stream)), "hipStreamSynchronize");
}

gr_stream Device::get_collective_stream()
{
	if (!collective_stream)
	{
		hipStream_t stream;
		throw_hip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
		collective_stream = stream;
	}
	return collective_stream;
}
} // namespace HIP
