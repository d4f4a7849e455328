#!/bin/bash
# bench.py exactly as the driver launches it for N ranks, all on GPU 0, collectives through the test shim (tests/rccl_shim):
# (the -DGRANITE_TEST_HOOKS build of the host layer, granite_amd/lib_testhooks: the product library has no stand-in loader):
# a functional run of the multi-process path (the numbers mean nothing: N processes share one GPU and the shim blocks).
N=${1:-8}; shift
export GRANITE_LIB_DIR=${MULTIRANK_LIB_DIR:-lib_testhooks} GRANITE_RCCL_LIBRARY_IS_A_TEST_STAND_IN=1 GRANITE_RCCL_LIBRARY=$PWD/tests/rccl_shim/libgranite_rccl_shim.so GRANITE_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29871 bench.py --gpus $N "$@"
