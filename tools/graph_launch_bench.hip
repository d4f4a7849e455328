// Host cost of N small dependent kernel launches on one stream: direct (hipLaunchKernelGGL) vs one pre-instantiated hipGraph
// captured from the same launches.  Decides whether pre-recorded launch sequences pay on this runtime.
// build: hipcc -O2 --offload-arch=gfx950 tools/graph_launch_bench.hip -o /tmp/graph_launch_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Args { float *p; float a[24]; int n; };
__global__ void k_small(Args x) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < x.n) x.p[i] = x.p[i] * x.a[3] + x.a[7]; }

int main()
{
	float *buf; hipMalloc(&buf, 1 << 22);
	hipStream_t s; hipStreamCreate(&s);
	Args a{}; a.p = buf; a.n = 1 << 18; a.a[3] = 1.0f; a.a[7] = 0.5f;
	auto record = [&](int kernels) { for (int k = 0; k < kernels; k++) hipLaunchKernelGGL(k_small, dim3(1024), dim3(256), 0, s, a); };
	for (int kernels : {4, 6})
	{
		const int iters = 2000;
		record(kernels); hipStreamSynchronize(s);
		auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < iters; i++) record(kernels);
		auto t1 = std::chrono::steady_clock::now();
		hipStreamSynchronize(s);
		auto t1s = std::chrono::steady_clock::now();
		hipGraph_t g; hipGraphExec_t e;
		hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); record(kernels); hipStreamEndCapture(s, &g);
		hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
		hipGraphLaunch(e, s); hipStreamSynchronize(s);
		auto t2 = std::chrono::steady_clock::now();
		for (int i = 0; i < iters; i++) hipGraphLaunch(e, s);
		auto t3 = std::chrono::steady_clock::now();
		hipStreamSynchronize(s);
		auto t3s = std::chrono::steady_clock::now();
		auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
		printf("%d kernels: direct %.2f us host / iteration (%.2f incl. drain), graph %.2f us host / iteration (%.2f incl. drain)\n", kernels,
		       us(t0, t1) / iters, us(t0, t1s) / iters, us(t2, t3) / iters, us(t2, t3s) / iters);
		hipGraphExecDestroy(e); hipGraphDestroy(g);
	}
	return 0;
}
