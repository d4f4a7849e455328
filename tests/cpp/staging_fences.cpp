// CPU-only exercise of HIP::Device's frame pacing (host/hip_device.cpp: next_frame_context) against tests/hip_stub with HIP_STUB_TRACE=1
// and HIP_STUB_EVENTS_PENDING=1 (every query says "not ready", so whatever the host looks at it also waits for, and the trace names it).
// The case of ADVICE r4: a stream that takes staging memory in one frame and is not handed out in the frames after it.  Its copies may
// still be reading the pinned slot when the slot comes round again four frames later; the host must have waited for THAT stream's
// fence of THAT frame by then, not for whatever older record the newer slots still hold for the stream.
// Prints "frame <n>" markers on stderr between the frames; tests/test_host_frame_loop_cpu.py reads the trace.
#include <cstdio>
#include "host/hip_device.hpp"

using namespace HIP;

int main()
{
	Device device(0);
	auto frame = [&](int number, bool async_takes_staging) {
		fprintf(stderr, "=== frame %d\n", number);
		(void)device.get_stream(CommandBuffer::Type::Generic); // the generic stream works every frame
		if (async_takes_staging)
		{
			(void)device.get_stream(CommandBuffer::Type::AsyncCompute);
			(void)device.allocate_staging(4096);
		}
		device.next_frame_context();
	};
	for (int n = 1; n <= 3; n++)
		frame(n, false);
	frame(4, true); // the async stream's only frame
	for (int n = 5; n <= 12; n++)
		frame(n, false);
	fprintf(stderr, "=== end\n");
	return 0;
}
