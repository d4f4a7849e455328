#!/bin/bash
O=gpurun_out/r03t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_multiprocess.py -q -m gpu -k "bench_runs" 2>&1 | tail -3
timeout 300 bash tools/multirank_one_gpu.sh 2 > $O/multirank_2.json 2> $O/multirank_2.err; head -c 300 $O/multirank_2.json; echo; wc -l $O/multirank_2.json
