#!/bin/bash
# Round 6: two front streams in turn with the launches kept in series (GRANITE_ALTERNATE_FRONT=2) against one front stream.
O=gpurun_out/r06ad; mkdir -p $O
bash tools/frame_ab.sh r06ad/ab "config3_4k_4096lights" one series:GRANITE_ALTERNATE_FRONT=2 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
bash tools/frame_ab.sh r06ad/ab2 "config3_4k_4096lights" one series:GRANITE_ALTERNATE_FRONT=2 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee -a $O/ab.txt
