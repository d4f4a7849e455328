// HIP::Collective — the one collective the path needs, an in-place all-gather of row bands, on RCCL.
// librccl.so is opened on first use (dlopen) so that single-device runs and CPU-only tooling never load it; it is the
// ROCm build of the same runtime this library's kernels run on, and it is driven on the executor's own HIP stream, so a
// band exchange is ordered between the kernels that produce and consume it without any host synchronisation.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace HIP
{
class Collective
{
public:
	enum { UniqueIdBytes = 128 }; // NCCL_UNIQUE_ID_BYTES
	Collective() = default;
	~Collective();
	Collective(const Collective &) = delete;
	void operator=(const Collective &) = delete;

	// Rank 0 creates the id and ships it to the other ranks out of band (bench.py: torch.distributed broadcast).
	static void create_unique_id(uint8_t id[UniqueIdBytes]);
	void init(const uint8_t id[UniqueIdBytes], int rank, int ranks);
	bool is_initialized() const { return comm != nullptr; }
	int get_rank() const { return rank; }
	int get_ranks() const { return ranks; }
	// For run records: what the communicator itself says (ncclCommCount; -1 if the library has no such entry point), the
	// library's version code (ncclGetVersion; -1 likewise), and whether the library is a test stand-in (GRANITE_RCCL_LIBRARY).
	int communicator_ranks() const;
	static int library_version();
	static bool is_stand_in();

	// base points at ranks * chunk_bytes bytes; this rank's chunk is already at base + rank * chunk_bytes.
	void all_gather_in_place(void *base, size_t chunk_bytes, void *stream);

private:
	void *comm = nullptr; // ncclComm_t
	int rank = 0, ranks = 1;
};
} // namespace HIP
