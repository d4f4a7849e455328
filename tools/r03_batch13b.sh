#!/bin/bash
O=gpurun_out/r03m; mkdir -p $O
run() { timeout 300 python -m pytest "$@" -q -m gpu -x -k "anti_aliasing and FXAA and 480" > $O/x.txt 2>&1; echo "rc=$? $*"; grep -m3 "fault\|passed\|failed\|error" $O/x.txt | cut -c1-200; }
run tests/test_gpu_strips.py tests/test_gpu_multiprocess.py tests/test_gpu_aa.py
run tests/test_gpu_strips.py tests/test_gpu_aa.py
run tests/test_gpu_strips.py tests/test_gpu_multiprocess.py
run tests/test_gpu_strips.py
run tests/test_gpu_strips.py tests/test_gpu_multiprocess.py tests/test_gpu_aa.py
