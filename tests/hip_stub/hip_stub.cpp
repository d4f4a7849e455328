// TEST INFRASTRUCTURE ONLY -- a device-less stand-in for the HIP runtime entry points the host layer and the C ABI call, loaded with
// LD_PRELOAD in front of libamdhip64.so by CPU tests (tests/test_host_frame_loop_cpu.py).  It lets the whole frame loop of the executor
// run without a GPU -- render-graph scheduling, hazard events, the submission threads, the launchers' argument checks -- and counts what
// it was asked to do, so that a test can hold the number of launches / event records / waits per frame and the framework's own
// host time per frame.  Kernels do not run: "device memory" is host memory that nobody computes on.  Never linked into the product.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <hip/hip_runtime_api.h>

namespace
{
struct Counters
{
	std::atomic<uint64_t> launches{0}, event_records{0}, stream_waits{0}, event_queries{0}, memcpys{0}, memsets{0}, syncs{0}, graph_launches{0};
	std::atomic<uint64_t> waits_before_record{0}; // hipStreamWaitEvent on an event whose record of this use was not issued yet
} counters;

struct Event
{
	std::atomic<uint64_t> records{0};
	std::chrono::steady_clock::time_point at{};
};
struct Stream
{
	int priority = 0;
};
thread_local hipError_t last_error = hipSuccess;

// HIP_STUB_TRACE=1: one line per stream call on stderr -- "L s<stream> <kernel>", "R s<stream> e<event>", "W s<stream> e<event>", and the host's
// "Q s0 e<event>" (hipEventQuery) / "S s0 e<event>" (hipEventSynchronize) -- the order
// in which a frame hands its work to the runtime (streams and events numbered in creation order).
const bool tracing = getenv("HIP_STUB_TRACE") != nullptr;
std::mutex trace_lock;
std::map<const void *, int> stream_ids, event_ids;
std::map<const void *, std::string> kernel_names;
int id_of(std::map<const void *, int> &ids, const void *p)
{
	auto it = ids.find(p);
	if (it == ids.end())
		it = ids.emplace(p, int(ids.size())).first;
	return it->second;
}
void trace(const char *what, const void *stream, const void *event, const void *kernel)
{
	if (!tracing)
		return;
	std::lock_guard<std::mutex> holder{trace_lock};
	if (kernel)
		fprintf(stderr, "%s s%d %s\n", what, id_of(stream_ids, stream), kernel_names.count(kernel) ? kernel_names[kernel].c_str() : "?");
	else if (event)
		fprintf(stderr, "%s s%d e%d\n", what, id_of(stream_ids, stream), id_of(event_ids, event));
	else
		fprintf(stderr, "%s s%d\n", what, id_of(stream_ids, stream));
}
} // namespace

extern "C" {

// ---- what the tests read ----------------------------------------------------------------------------------------------------------
uint64_t hip_stub_count(const char *what)
{
	const std::string w = what;
	if (w == "launches") return counters.launches;
	if (w == "event_records") return counters.event_records;
	if (w == "stream_waits") return counters.stream_waits;
	if (w == "event_queries") return counters.event_queries;
	if (w == "memcpys") return counters.memcpys;
	if (w == "memsets") return counters.memsets;
	if (w == "syncs") return counters.syncs;
	if (w == "graph_launches") return counters.graph_launches;
	if (w == "waits_before_record") return counters.waits_before_record;
	return ~0ull;
}
// number of times this event has been recorded (a submission thread that must not wait before the record was issued checks its own
// bookkeeping against this)
uint64_t hip_stub_event_records(void *event) { return event ? static_cast<Event *>(event)->records.load() : 0; }

// ---- device ------------------------------------------------------------------------------------------------------------------------
hipError_t hipGetDeviceCount(int *count) { *count = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDriverGetVersion(int *v) { *v = 70200000; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t *p, int)
{
	memset(p, 0, sizeof(*p));
	strcpy(p->name, "hip_stub (no device)");
	strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
	p->multiProcessorCount = 256;
	p->totalGlobalMem = size_t(288) << 30;
	return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { counters.syncs++; return hipSuccess; }
hipError_t hipGetLastError(void) { hipError_t e = last_error; last_error = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hip_stub error"; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -2; return hipSuccess; }

// ---- memory: host memory -----------------------------------------------------------------------------------------------------------
hipError_t hipMalloc(void **p, size_t bytes) { *p = aligned_alloc(256, (bytes + 255) & ~size_t(255)); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { counters.memcpys++; memcpy(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { counters.memcpys++; memcpy(dst, src, n); return hipSuccess; }
hipError_t hipMemset(void *dst, int v, size_t n) { counters.memsets++; memset(dst, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { counters.memsets++; memset(dst, v, n); return hipSuccess; }
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int v, size_t count, hipStream_t)
{
	counters.memsets++;
	for (size_t i = 0; i < count; i++)
		static_cast<int *>(dst)[i] = v;
	return hipSuccess;
}

// ---- streams, events ---------------------------------------------------------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(new Stream); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int priority)
{
	auto *st = new Stream;
	st->priority = priority;
	*s = reinterpret_cast<hipStream_t>(st);
	return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<Stream *>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { counters.syncs++; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(new Event); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<Event *>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
	counters.event_records++;
	trace("R", s, e, nullptr);
	auto *ev = reinterpret_cast<Event *>(e);
	ev->at = std::chrono::steady_clock::now();
	ev->records++;
	return hipSuccess;
}
// HIP_STUB_EVENTS_PENDING=1: a recorded event never reads as complete, so every cross-stream dependency takes the wait path
hipError_t hipEventQuery(hipEvent_t e)
{
	counters.event_queries++;
	trace("Q", nullptr, e, nullptr);
	static const bool pending = getenv("HIP_STUB_EVENTS_PENDING") != nullptr;
	return pending ? hipErrorNotReady : hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
	counters.syncs++;
	trace("S", nullptr, e, nullptr);
	return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
	*ms = std::chrono::duration<float, std::milli>(reinterpret_cast<Event *>(b)->at - reinterpret_cast<Event *>(a)->at).count();
	return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
	counters.stream_waits++;
	trace("W", s, e, nullptr);
	if (reinterpret_cast<Event *>(e)->records.load() == 0)
		counters.waits_before_record++;
	return hipSuccess;
}

// ---- launches ----------------------------------------------------------------------------------------------------------------------
hipError_t hipLaunchKernel(const void *f, dim3, dim3, void **, size_t, hipStream_t s)
{
	counters.launches++;
	trace("L", s, nullptr, f);
	return hipSuccess;
}
struct CallConfig { dim3 grid, block; size_t shared; hipStream_t stream; };
static thread_local CallConfig pushed;
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shared, hipStream_t stream) { pushed = {grid, block, shared, stream}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shared, hipStream_t *stream)
{
	*grid = pushed.grid; *block = pushed.block; *shared = pushed.shared; *stream = pushed.stream;
	return hipSuccess;
}
void **__hipRegisterFatBinary(const void *) { static void *handle[4]; return handle; }
void __hipRegisterFunction(void **, const void *host_function, char *, const char *device_name, unsigned, void *, void *, void *, void *, int *)
{
	if (tracing)
		kernel_names[host_function] = device_name;
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}

// ---- graphs (the parked hipGraph replay path of hip_device.cpp) ---------------------------------------------------------------------
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = reinterpret_cast<hipGraph_t>(new int(0)); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, hipGraphNode_t *, char *, size_t) { *e = reinterpret_cast<hipGraphExec_t>(new int(0)); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { counters.graph_launches++; return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete reinterpret_cast<int *>(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete reinterpret_cast<int *>(e); return hipSuccess; }
}
