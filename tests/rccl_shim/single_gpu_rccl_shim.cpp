// TEST INFRASTRUCTURE ONLY -- never loaded unless GRANITE_RCCL_LIBRARY points at it.
//
// A stand-in for the five RCCL entry points the executor uses (host/collective.cpp), for the one situation the real library
// refuses: several ranks = several processes on ONE GPU.  The GPU box of this build has a single MI355X, RCCL rejects two
// ranks on one device, and so the multi-process path of bench.py (rendezvous, two communicators, the in-frame gather of the 1/8
// level, the output gather beside the frame, the RGB888 transport buffers) could otherwise only ever run for the first time on
// the driver's 8-GPU node.  With this shim the same processes, command line and code path run here.
//
// ncclAllGather(send, recv, count, ncclUint8, comm, stream), in-place form only (send == recv + rank * count):
//   wait for `stream` (own chunk complete) -> publish the IPC handle of recv's allocation in a POSIX shared-memory segment
//   named after the unique id -> barrier -> copy every peer's chunk out of its (IPC-mapped) buffer on `stream` -> wait ->
//   barrier (nobody rewrites its chunk while a peer still reads it).
// Blocking, slow, and only correct for ranks that issue the same sequence of gathers -- which is what the executor does.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <string>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace
{
constexpr int MaxRanks = 16;
struct Published
{
	hipIpcMemHandle_t handle;
	uint64_t offset; // of recv inside its allocation
	uint64_t base;   // the owner's base address (cache key on the reading side)
};
struct Shared
{
	std::atomic<uint32_t> ready;       // set by the creator once the segment is zeroed
	std::atomic<uint32_t> joined;      // ranks that called ncclCommInitRank
	std::atomic<uint32_t> arrived;     // barrier: arrivals of the current generation
	std::atomic<uint32_t> generation;  // barrier: generation counter
	Published slot[MaxRanks];
};
struct Comm
{
	Shared *shared = nullptr;
	std::string name;
	int rank = 0, ranks = 1;
	std::map<std::pair<int, uint64_t>, void *> opened; // (peer, peer base) -> mapping in this process
};
struct UniqueId { char internal[128]; };

void barrier(Comm *c)
{
	const uint32_t gen = c->shared->generation.load();
	if (c->shared->arrived.fetch_add(1) + 1 == uint32_t(c->ranks))
	{
		c->shared->arrived.store(0);
		c->shared->generation.fetch_add(1);
		return;
	}
	while (c->shared->generation.load() == gen)
		usleep(20);
}

std::string segment_name(const UniqueId &id)
{
	char name[64];
	uint64_t a, b;
	memcpy(&a, id.internal, 8);
	memcpy(&b, id.internal + 8, 8);
	snprintf(name, sizeof(name), "/granite_rccl_shim_%016llx%016llx", (unsigned long long)a, (unsigned long long)b);
	return name;
}
const char *last_error = "ok";
int fail(const char *what)
{
	last_error = what;
	fprintf(stderr, "[rccl shim] %s\n", what);
	return 1;
}
} // namespace

extern "C" {
int ncclGetUniqueId(UniqueId *id)
{
	memset(id, 0, sizeof(*id));
	timespec ts;
	clock_gettime(CLOCK_REALTIME, &ts);
	static std::atomic<uint32_t> counter{0};
	const uint64_t a = (uint64_t(getpid()) << 32) ^ uint64_t(ts.tv_nsec) ^ (uint64_t(counter.fetch_add(1)) << 20), b = uint64_t(ts.tv_sec);
	memcpy(id->internal, &a, 8);
	memcpy(id->internal + 8, &b, 8);
	return 0;
}

int ncclCommInitRank(void **comm, int nranks, UniqueId id, int rank)
{
	if (nranks < 1 || nranks > MaxRanks || rank < 0 || rank >= nranks)
		return fail("ncclCommInitRank: bad rank / size");
	auto *c = new Comm;
	c->rank = rank;
	c->ranks = nranks;
	c->name = segment_name(id);
	bool creator = true;
	int fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
	if (fd < 0)
	{
		creator = false;
		fd = shm_open(c->name.c_str(), O_RDWR, 0600);
	}
	if (fd < 0)
		return fail("shm_open failed");
	if (creator && ftruncate(fd, sizeof(Shared)) != 0)
		return fail("ftruncate failed");
	for (int i = 0; !creator && i < 5000; i++) // the creator may not have sized the segment yet
	{
		off_t size = lseek(fd, 0, SEEK_END);
		if (size >= off_t(sizeof(Shared)))
			break;
		usleep(200);
	}
	void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED)
		return fail("mmap failed");
	c->shared = static_cast<Shared *>(p);
	if (creator)
		c->shared->ready.store(1); // ftruncate zero-fills: counters start at 0
	while (c->shared->ready.load() == 0)
		usleep(50);
	c->shared->joined.fetch_add(1);
	while (c->shared->joined.load() < uint32_t(nranks))
		usleep(50);
	*comm = c;
	return 0;
}

int ncclCommDestroy(void *comm)
{
	auto *c = static_cast<Comm *>(comm);
	if (!c)
		return 0;
	for (auto &e : c->opened)
		(void)hipIpcCloseMemHandle(e.second);
	shm_unlink(c->name.c_str());
	munmap(c->shared, sizeof(Shared));
	delete c;
	return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int datatype, void *comm, void *stream)
{
	auto *c = static_cast<Comm *>(comm);
	if (!c || datatype != 1 /* ncclUint8 */)
		return fail("ncclAllGather: only ncclUint8 is provided");
	if (static_cast<const uint8_t *>(send) != static_cast<uint8_t *>(recv) + size_t(c->rank) * count)
		return fail("ncclAllGather: only the in-place form is provided");
	auto s = static_cast<hipStream_t>(stream);
	if (hipStreamSynchronize(s) != hipSuccess)
		return fail("hipStreamSynchronize failed");
	if (c->ranks == 1)
		return 0;
	hipDeviceptr_t base = nullptr;
	size_t size = 0;
	if (hipMemGetAddressRange(&base, &size, recv) != hipSuccess)
		return fail("hipMemGetAddressRange failed");
	Published mine;
	memset(&mine, 0, sizeof(mine));
	if (hipIpcGetMemHandle(&mine.handle, base) != hipSuccess)
		return fail("hipIpcGetMemHandle failed");
	mine.offset = uint64_t(static_cast<uint8_t *>(recv) - static_cast<uint8_t *>(base));
	mine.base = uint64_t(reinterpret_cast<uintptr_t>(base));
	c->shared->slot[c->rank] = mine;
	barrier(c);
	for (int peer = 0; peer < c->ranks; peer++)
	{
		if (peer == c->rank)
			continue;
		const Published theirs = c->shared->slot[peer];
		void *&mapped = c->opened[{peer, theirs.base}];
		if (!mapped && hipIpcOpenMemHandle(&mapped, theirs.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
			return fail("hipIpcOpenMemHandle failed");
		const uint8_t *src = static_cast<const uint8_t *>(mapped) + theirs.offset + size_t(peer) * count;
		if (hipMemcpyAsync(static_cast<uint8_t *>(recv) + size_t(peer) * count, src, count, hipMemcpyDeviceToDevice, s) != hipSuccess)
			return fail("hipMemcpyAsync failed");
	}
	if (hipStreamSynchronize(s) != hipSuccess)
		return fail("hipStreamSynchronize failed");
	barrier(c);
	return 0;
}

const char *ncclGetErrorString(int) { return last_error; }
}
