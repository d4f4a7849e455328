// Ticket-counter microbenchmark for gfx950 (one 8-XCD partition): what does an agent-scope fetch-add cost when N waves share a
// counter?  Decides how the persistent lighting kernel may deal tiles (lighting.hip).  Each wave's lane 0 issues ROUNDS dependent
// fetch-adds (the next one waits for the previous one's return, as a ticket loop does), 4096 waves resident (1024 x 256 threads).
//   mode 0: one counter for the chip          mode 1: one counter per XCD (XCC_ID)        mode 2: one per workgroup
//   mode 3: one per wave                      mode 4: one per XCD, but `spin` VALU work between two tickets (a tile being shaded)
// Reports the launch time, ns per fetch-add as a wave sees it (latency) and per counter (service time of the hottest line).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ROUNDS 64
__global__ __launch_bounds__(256) void k(unsigned long long *counters, unsigned *sink, int mode, int spin)
{
	unsigned xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
	unsigned index = 0;
	if (mode == 1 || mode == 4) index = xcc & 7;
	if (mode == 2) index = blockIdx.x;
	if (mode == 3) index = wave;
	unsigned long long *p = counters + size_t(index) * 16; // 128 B apart
	unsigned acc = 0;
	float f = float(threadIdx.x);
	for (int r = 0; r < ROUNDS; r++)
	{
		unsigned long long v = 0;
		if ((threadIdx.x & 63) == 0)
			v = __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		acc += unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(v))));
		for (int i = 0; i < spin; i++)
			f = __builtin_fmaf(f, 1.0001f, 0.5f);
	}
	if (acc == 0x12345678u || f == 3.25f)
		sink[0] = acc;
}
int main()
{
	unsigned long long *counters; unsigned *sink;
	hipMalloc(&counters, 4096 * 128); hipMalloc(&sink, 64);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const char *names[] = {"one counter for the chip", "one counter per XCD", "one counter per workgroup", "one counter per wave", "one per XCD + 8 us of VALU work between tickets"};
	for (int mode = 0; mode < 5; mode++)
	{
		const int spin = mode == 4 ? 4000 : 0;
		hipMemset(counters, 0, 4096 * 128);
		hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, counters, sink, mode, spin);
		hipDeviceSynchronize();
		float best = 1e9f;
		for (int rep = 0; rep < 5; rep++)
		{
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, counters, sink, mode, spin);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			best = ms < best ? ms : best;
		}
		const double per_counter = mode == 0 ? 4096.0 * ROUNDS : (mode == 1 || mode == 4) ? 512.0 * ROUNDS : mode == 2 ? 4.0 * ROUNDS : ROUNDS;
		printf("mode %d  %-48s launch %8.1f us   per fetch-add as a wave sees it %8.1f ns   per fetch-add on one counter %7.1f ns\n", mode, names[mode], best * 1e3,
		       best * 1e6 / ROUNDS, best * 1e6 / per_counter);
	}
	return 0;
}
