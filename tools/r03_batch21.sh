#!/bin/bash
O=gpurun_out/r03u; mkdir -p $O
GR_TIMING_DUMP=$O/spans.txt timeout 200 python tools/gpu_timeline.py > /dev/null 2>&1; tail -33 $O/spans.txt > $O/gpu_timeline.txt; rm -f $O/spans.txt; cat $O/gpu_timeline.txt
