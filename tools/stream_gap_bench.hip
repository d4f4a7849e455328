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
	run("plain, alternating between two streams", [&]() { for (int i = 0; i < n; i++) launch((i & 1) ? s2 : s); });
	return 0;
}
