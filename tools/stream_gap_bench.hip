// What sits between two long kernels on one in-order stream: period of a ~200 us kernel launched back to back, (1) plain, (2) with an
// event record behind each launch, (3) with a wait on another stream's (already complete) event in front of each, (4) both -- the
// executor's lighting stream --, (5) the record folded into the launch (hipExtLaunchKernelGGL's stop event), (6) the same work
// alternating between two streams.  Decides where the 17-35 us between consecutive lighting kernels go.
// build: hipcc -O2 --offload-arch=gfx950 tools/stream_gap_bench.hip -o /tmp/stream_gap_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_busy(float *p, int iters)
{
	float a = p[threadIdx.x], b = a * 0.5f + 1.0f, c = a + 2.0f, d = b - c;
	for (int i = 0; i < iters; i++)
	{
		a = fmaf(a, b, c); b = fmaf(b, c, d); c = fmaf(c, d, a); d = fmaf(d, a, b);
	}
	if (a + b + c + d == 12345.678f)
		p[blockIdx.x] = a;
}
// the same kernel announcing its own completion: the last workgroup to finish stores `seq` to a flag the other queue's packet processor polls
// (hipStreamWaitValue32) -- no packet behind the launch on its own queue
__global__ __launch_bounds__(256) void k_busy_flag(float *p, int iters, unsigned *counter, unsigned *flag, unsigned seq)
{
	float a = p[threadIdx.x], b = a * 0.5f + 1.0f, c = a + 2.0f, d = b - c;
	for (int i = 0; i < iters; i++)
	{
		a = fmaf(a, b, c); b = fmaf(b, c, d); c = fmaf(c, d, a); d = fmaf(d, a, b);
	}
	if (a + b + c + d == 12345.678f)
		p[blockIdx.x] = a;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		__threadfence_system();
		if (atomicAdd(counter, 1u) == gridDim.x - 1u)
		{
			*counter = 0u;
			__hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}
__global__ void k_tiny(float *p) { if (threadIdx.x == 9999) p[0] = 1.0f; }

int main()
{
	float *buf; hipMalloc(&buf, 1 << 22); hipMemset(buf, 0, 1 << 22);
	hipStream_t s, s2, other; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	hipStreamCreateWithFlags(&other, hipStreamNonBlocking);
	const int blocks = 256 * 8 * 4, iters_kernel = 6000, n = 60;
	const unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
	std::vector<hipEvent_t> rec(n), dep(n);
	for (auto &e : rec) hipEventCreateWithFlags(&e, flags);
	for (auto &e : dep) hipEventCreateWithFlags(&e, flags);
	auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
	auto run = [&](const char *name, auto body) {
		body(); hipDeviceSynchronize(); // warm
		auto t0 = std::chrono::steady_clock::now();
		body();
		hipDeviceSynchronize();
		auto t1 = std::chrono::steady_clock::now();
		printf("%-60s %.1f us per launch\n", name, us(t0, t1) / n);
	};
	auto launch = [&](hipStream_t st) { hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, st, buf, iters_kernel); };
	run("plain back-to-back", [&]() { for (int i = 0; i < n; i++) launch(s); });
	run("+ event record behind each", [&]() { for (int i = 0; i < n; i++) { launch(s); hipEventRecord(rec[i], s); } });
	run("+ wait on another stream's event in front of each", [&]() {
		for (int i = 0; i < n; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); hipEventRecord(dep[i], other); hipStreamWaitEvent(s, dep[i], 0); launch(s); }
	});
	run("+ both (the lighting stream of the executor)", [&]() {
		for (int i = 0; i < n; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); hipEventRecord(dep[i], other); hipStreamWaitEvent(s, dep[i], 0); launch(s); hipEventRecord(rec[i], s); }
	});
	run("+ both, and a consumer on the other stream waiting for each record", [&]() {
		for (int i = 0; i < n; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); hipEventRecord(dep[i], other); hipStreamWaitEvent(s, dep[i], 0); launch(s); hipEventRecord(rec[i], s);
			hipStreamWaitEvent(other, rec[i], 0); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); }
	});
	run("wait in front + record folded into the launch (hipExtLaunchKernelGGL stop event)", [&]() {
		for (int i = 0; i < n; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); hipEventRecord(dep[i], other); hipStreamWaitEvent(s, dep[i], 0);
			hipExtLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s, nullptr, rec[i], 0, buf, iters_kernel); }
	});
	run("both, alternating between two streams", [&]() {
		for (int i = 0; i < n; i++) { hipStream_t st = (i & 1) ? s2 : s; hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); hipEventRecord(dep[i], other); hipStreamWaitEvent(st, dep[i], 0); launch(st); hipEventRecord(rec[i], st); }
	});
	unsigned *flags_mem = nullptr, *counter = nullptr;
	int can = 0;
	hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
	printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
	unsigned *flags_mem2 = nullptr; // signal memory comes in 8-byte allocations
	if (hipExtMallocWithFlags(reinterpret_cast<void **>(&flags_mem), 8, hipMallocSignalMemory) != hipSuccess ||
	    hipExtMallocWithFlags(reinterpret_cast<void **>(&flags_mem2), 8, hipMallocSignalMemory) != hipSuccess)
	{
		printf("no signal memory: %s\n", hipGetErrorString(hipGetLastError()));
		flags_mem = nullptr;
	}
	hipMalloc(&counter, 4); hipMemset(counter, 0, 4);
	unsigned *f_dep = flags_mem, *f_rec = flags_mem2;
	unsigned seq_base = 0;
	auto zero = [&]() { hipDeviceSynchronize(); seq_base += 1000; };
	if (can && flags_mem)
	{
		hipMemset(flags_mem, 0, 8); hipMemset(flags_mem2, 0, 8);
		run("stream write / wait values instead of events: front + behind + consumer", [&]() {
			zero();
			for (int i = 0; i < n; i++) { const unsigned q = seq_base + i + 1;
				hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); hipStreamWriteValue32(other, f_dep, q, 0); hipStreamWaitValue32(s, f_dep, q, hipStreamWaitValueGte, 0xffffffffu);
				launch(s); hipStreamWriteValue32(s, f_rec, q, 0);
				hipStreamWaitValue32(other, f_rec, q, hipStreamWaitValueGte, 0xffffffffu); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); }
		});
		run("kernel stores its completion flag, consumer waits on the value; event wait in front", [&]() {
			zero();
			for (int i = 0; i < n; i++) { const unsigned q = seq_base + i + 1;
				hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); hipEventRecord(dep[i], other); hipStreamWaitEvent(s, dep[i], 0);
				hipLaunchKernelGGL(k_busy_flag, dim3(blocks), dim3(256), 0, s, buf, iters_kernel, counter, f_rec, q);
				hipStreamWaitValue32(other, f_rec, q, hipStreamWaitValueGte, 0xffffffffu); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); }
		});
		run("kernel stores its completion flag, consumer waits on the value; value wait in front", [&]() {
			zero();
			for (int i = 0; i < n; i++) { const unsigned q = seq_base + i + 1;
				hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); hipStreamWriteValue32(other, f_dep, q, 0); hipStreamWaitValue32(s, f_dep, q, hipStreamWaitValueGte, 0xffffffffu);
				hipLaunchKernelGGL(k_busy_flag, dim3(blocks), dim3(256), 0, s, buf, iters_kernel, counter, f_rec, q);
				hipStreamWaitValue32(other, f_rec, q, hipStreamWaitValueGte, 0xffffffffu); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); }
		});
		run("kernel stores its completion flag, consumer waits on the value; nothing in front", [&]() {
			zero();
			for (int i = 0; i < n; i++) { const unsigned q = seq_base + i + 1;
				hipLaunchKernelGGL(k_busy_flag, dim3(blocks), dim3(256), 0, s, buf, iters_kernel, counter, f_rec, q);
				hipStreamWaitValue32(other, f_rec, q, hipStreamWaitValueGte, 0xffffffffu); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, buf); }
		});
		run("plain back-to-back, the flag-storing kernel (cost of the counter)", [&]() {
			zero();
			for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_busy_flag, dim3(blocks), dim3(256), 0, s, buf, iters_kernel, counter, f_rec, seq_base + i + 1);
		});
	}
	run("plain, alternating between two streams", [&]() { for (int i = 0; i < n; i++) launch((i & 1) ? s2 : s); });
	return 0;
}
