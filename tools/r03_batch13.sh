#!/bin/bash
# Round 3, thirteenth GPU batch: TAA history as a boundary-row exchange (emulated ranks, separate processes through the stand-in).
O=gpurun_out/r03m; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_aa.py tests/test_gpu_strips.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $O/pytest_gpu.txt; head -5 $O/pytest_gpu.txt | cut -c1-300; tail -12 $O/pytest_gpu.txt | cut -c1-300
