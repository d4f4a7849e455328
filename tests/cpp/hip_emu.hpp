// TEST INFRASTRUCTURE.  A host stand-in for the part of the HIP device environment the kernels of
// granite_amd/csrc/aa_fast_kernels.hpp use, so that the SAME kernel text runs on the CPU under pytest -m "not gpu" and is
// compared with the oracle before a GPU sees it: one std::thread per lane of a workgroup (workgroups run one after another),
// __shared__ = function-local static storage, __syncthreads* = a std::barrier over the lanes that have not returned, wave votes
// (__ballot / __all / __any) = a rendezvous of the 64 lanes of a wave.  Nothing here is built into the product.
#pragma once
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3
{
	unsigned x, y, z;
	dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { return {x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }
using std::max;
using std::min;

namespace emu
{
struct Block
{
	unsigned threads = 0;
	std::unique_ptr<std::barrier<>> barrier;
	std::atomic<int> vote_or{0}, vote_count{0};
	// wave rendezvous: one slot per wave
	struct Wave
	{
		std::unique_ptr<std::barrier<>> barrier;
		std::atomic<uint64_t> mask{0};
		uint32_t exchange[64];
	};
	std::vector<Wave> waves;
};
inline Block *&current_block()
{
	static Block *b = nullptr;
	return b;
}
inline thread_local dim3 t_threadIdx, t_blockIdx;
inline dim3 g_blockDim, g_gridDim;
inline thread_local unsigned t_flat = 0;
} // namespace emu
#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

inline void __syncthreads() { emu::current_block()->barrier->arrive_and_wait(); }
inline int __syncthreads_or(int p)
{
	emu::Block *b = emu::current_block();
	if (p)
		b->vote_or.store(1);
	b->barrier->arrive_and_wait();
	const int r = b->vote_or.load();
	b->barrier->arrive_and_wait();
	b->vote_or.store(0); // every lane (lane 0 may have returned already); all have read r
	b->barrier->arrive_and_wait();
	return r;
}
inline int __syncthreads_count(int p)
{
	emu::Block *b = emu::current_block();
	if (p)
		b->vote_count.fetch_add(1);
	b->barrier->arrive_and_wait();
	const int r = b->vote_count.load();
	b->barrier->arrive_and_wait();
	b->vote_count.store(0);
	b->barrier->arrive_and_wait();
	return r;
}
inline uint64_t __ballot(int p)
{
	emu::Block::Wave &w = emu::current_block()->waves[emu::t_flat / 64];
	if (p)
		w.mask.fetch_or(uint64_t(1) << (emu::t_flat % 64));
	w.barrier->arrive_and_wait();
	const uint64_t r = w.mask.load();
	w.barrier->arrive_and_wait();
	w.mask.store(0);
	w.barrier->arrive_and_wait();
	return r;
}
inline int __any(int p) { return __ballot(p) != 0; }
// lanes that already returned do not vote: __all is over the lanes still present, like EXEC
inline int __all(int p) { return __ballot(!p) == 0; }
inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
inline int __popc(uint32_t v) { return __builtin_popcount(v); }
inline int __ffsll(uint64_t v) { return __builtin_ffsll((long long)v); }
inline int __clzll(uint64_t v) { return v ? __builtin_clzll(v) : 64; }

namespace emu
{
// Runs kernel(args...) for every workgroup of `grid`, one after another, with block.x * block.y lanes as threads.
template <typename Kernel, typename... Args>
void launch(Kernel kernel, dim3 grid, dim3 block, Args... args)
{
	g_blockDim = block;
	g_gridDim = grid;
	const unsigned n = block.x * block.y * block.z;
	for (unsigned bz = 0; bz < grid.z; bz++)
		for (unsigned by = 0; by < grid.y; by++)
			for (unsigned bx = 0; bx < grid.x; bx++)
			{
				Block blk;
				blk.threads = n;
				blk.barrier = std::make_unique<std::barrier<>>(n);
				blk.waves = std::vector<Block::Wave>((n + 63) / 64);
				for (unsigned w = 0; w < blk.waves.size(); w++)
					blk.waves[w].barrier = std::make_unique<std::barrier<>>(std::min(64u, n - 64 * w));
				current_block() = &blk;
				std::vector<std::thread> lanes;
				lanes.reserve(n);
				for (unsigned t = 0; t < n; t++)
					lanes.emplace_back([&, t]() {
						t_flat = t;
						t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
						t_blockIdx = dim3(bx, by, bz);
						kernel(args...);
						// a lane that has returned no longer takes part in barriers or votes
						blk.waves[t / 64].barrier->arrive_and_drop();
						blk.barrier->arrive_and_drop();
					});
				for (auto &l : lanes)
					l.join();
				current_block() = nullptr;
			}
}
} // namespace emu
