#include "collective.hpp"
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <mutex>
#include <stdexcept>

namespace HIP
{
namespace
{
// The slice of rccl.h this file uses (rccl/rccl.h:36-52,187,220,260,339,460,678).
struct UniqueId { char internal[Collective::UniqueIdBytes]; };
using GetUniqueIdFn = int (*)(UniqueId *);
using CommInitRankFn = int (*)(void **comm, int nranks, UniqueId id, int rank);
using CommDestroyFn = int (*)(void *comm);
using AllGatherFn = int (*)(const void *send, void *recv, size_t count, int datatype, void *comm, void *stream);
using GetErrorStringFn = const char *(*)(int);
constexpr int kUint8 = 1; // ncclUint8

struct Api
{
	GetUniqueIdFn get_unique_id = nullptr;
	CommInitRankFn comm_init_rank = nullptr;
	CommDestroyFn comm_destroy = nullptr;
	AllGatherFn all_gather = nullptr;
	GetErrorStringFn get_error_string = nullptr;
};

const Api &api()
{
	static Api table;
	static std::once_flag once;
	std::call_once(once, []() {
		void *lib = nullptr;
		if (const char *other = getenv("GRANITE_RCCL_LIBRARY"))
		{
			// Another implementation of the same five entry points (tests/rccl_shim: several ranks on ONE GPU, which RCCL refuses).
			fprintf(stderr, "[granite-hip] note: collectives go through %s (GRANITE_RCCL_LIBRARY), not librccl.so.1\n", other);
			lib = dlopen(other, RTLD_NOW | RTLD_LOCAL);
			if (!lib)
				throw std::runtime_error(std::string("cannot load GRANITE_RCCL_LIBRARY: ") + dlerror());
		}
		if (!lib)
			lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!lib)
			lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!lib)
			throw std::runtime_error(std::string("cannot load librccl.so.1: ") + dlerror());
		table.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(lib, "ncclGetUniqueId"));
		table.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(lib, "ncclCommInitRank"));
		table.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(lib, "ncclCommDestroy"));
		table.all_gather = reinterpret_cast<AllGatherFn>(dlsym(lib, "ncclAllGather"));
		table.get_error_string = reinterpret_cast<GetErrorStringFn>(dlsym(lib, "ncclGetErrorString"));
		if (!table.get_unique_id || !table.comm_init_rank || !table.comm_destroy || !table.all_gather || !table.get_error_string)
			throw std::runtime_error("librccl.so.1 lacks an expected entry point");
	});
	return table;
}

void check(int result, const char *what)
{
	if (result != 0)
		throw std::runtime_error(std::string(what) + ": " + api().get_error_string(result));
}
} // namespace

Collective::~Collective()
{
	if (comm)
		(void)api().comm_destroy(comm);
}

void Collective::create_unique_id(uint8_t id[UniqueIdBytes])
{
	UniqueId uid;
	check(api().get_unique_id(&uid), "ncclGetUniqueId");
	for (int i = 0; i < UniqueIdBytes; i++)
		id[i] = uint8_t(uid.internal[i]);
}

void Collective::init(const uint8_t id[UniqueIdBytes], int rank_, int ranks_)
{
	if (comm)
		throw std::logic_error("Collective is already initialised.");
	if (ranks_ < 1 || rank_ < 0 || rank_ >= ranks_)
		throw std::logic_error("Collective: rank out of range.");
	UniqueId uid;
	for (int i = 0; i < UniqueIdBytes; i++)
		uid.internal[i] = char(id[i]);
	check(api().comm_init_rank(&comm, ranks_, uid, rank_), "ncclCommInitRank");
	rank = rank_;
	ranks = ranks_;
}

void Collective::all_gather_in_place(void *base, size_t chunk_bytes, void *stream)
{
	if (!comm)
		throw std::logic_error("Collective is not initialised.");
	auto *bytes = static_cast<uint8_t *>(base);
	// In-place form: sendbuff == recvbuff + rank * sendcount (rccl.h, ncclAllGather).
	check(api().all_gather(bytes + size_t(rank) * chunk_bytes, bytes, chunk_bytes, kUint8, comm, stream), "ncclAllGather");
}
} // namespace HIP
