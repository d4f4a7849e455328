#!/bin/bash
# What would the frame gain if no cross-stream wait / record sat between kernels?  (GRANITE_UNSAFE_NO_CROSS_SYNC: frames are invalid, timing only.)
O=gpurun_out/r06e; mkdir -p $O
bash tools/frame_ab.sh r06e/ab "config3_4k_4096lights config4_4k_smaa_taa config2_1080p_256lights" base nosync:GRANITE_UNSAFE_NO_CROSS_SYNC=1 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
