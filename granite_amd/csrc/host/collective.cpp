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
using CommCountFn = int (*)(void *comm, int *count);
using GetVersionFn = int (*)(int *version);
constexpr int kUint8 = 1; // ncclUint8

struct Api
{
	GetUniqueIdFn get_unique_id = nullptr;
	CommInitRankFn comm_init_rank = nullptr;
	CommDestroyFn comm_destroy = nullptr;
	AllGatherFn all_gather = nullptr;
	GetErrorStringFn get_error_string = nullptr;
	CommCountFn comm_count = nullptr;   // optional: reporting only
	GetVersionFn get_version = nullptr; // optional: reporting only
	bool stand_in = false;              // GRANITE_RCCL_LIBRARY: not RCCL
};

const Api &api()
{
	static Api table;
	static std::once_flag once;
	std::call_once(once, []() {
		void *lib = nullptr;
#ifdef GRANITE_TEST_HOOKS
		// Test builds only (make testhooks -> granite_amd/lib_testhooks, -DGRANITE_TEST_HOOKS): the product library does not contain this
		// branch and cannot be pointed at anything but librccl.so.1.
		if (const char *other = getenv("GRANITE_RCCL_LIBRARY"))
		{
			// Another implementation of the same five entry points (tests/rccl_shim: several ranks on ONE GPU, which RCCL refuses).
			// A test facility: a stale variable in a production environment must not silently replace RCCL, so it only counts
			// together with GRANITE_RCCL_LIBRARY_IS_A_TEST_STAND_IN=1, and every record made through it says so (is_stand_in()).
			const char *ack = getenv("GRANITE_RCCL_LIBRARY_IS_A_TEST_STAND_IN");
			if (!ack || ack[0] != '1')
				throw std::runtime_error("GRANITE_RCCL_LIBRARY is set without GRANITE_RCCL_LIBRARY_IS_A_TEST_STAND_IN=1: refusing to replace librccl.so.1");
			table.stand_in = true;
			fprintf(stderr, "[granite-hip] note: collectives go through %s (GRANITE_RCCL_LIBRARY), not librccl.so.1\n", other);
			lib = dlopen(other, RTLD_NOW | RTLD_LOCAL);
			if (!lib)
				throw std::runtime_error(std::string("cannot load GRANITE_RCCL_LIBRARY: ") + dlerror());
		}
#else
		if (getenv("GRANITE_RCCL_LIBRARY"))
			fprintf(stderr, "[granite-hip] note: GRANITE_RCCL_LIBRARY is ignored: this library was built without GRANITE_TEST_HOOKS\n");
#endif
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
		table.comm_count = reinterpret_cast<CommCountFn>(dlsym(lib, "ncclCommCount"));
		table.get_version = reinterpret_cast<GetVersionFn>(dlsym(lib, "ncclGetVersion"));
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

int Collective::communicator_ranks() const
{
	int count = -1;
	if (comm && api().comm_count && api().comm_count(comm, &count) == 0)
		return count;
	return -1;
}

int Collective::library_version()
{
	int version = -1;
	if (api().get_version && api().get_version(&version) == 0)
		return version;
	return -1;
}

bool Collective::is_stand_in() { return api().stand_in; }

void Collective::all_gather_in_place(void *base, size_t chunk_bytes, void *stream)
{
	if (!comm)
		throw std::logic_error("Collective is not initialised.");
	auto *bytes = static_cast<uint8_t *>(base);
	// In-place form: sendbuff == recvbuff + rank * sendcount (rccl.h, ncclAllGather).
	check(api().all_gather(bytes + size_t(rank) * chunk_bytes, bytes, chunk_bytes, kUint8, comm, stream), "ncclAllGather");
}
} // namespace HIP
