#!/bin/bash
# Round 6: are the executor's streams sharing hardware queues?  GPU_MAX_HW_QUEUES (HIP runtime, default 4) with one and with two front streams.
O=gpurun_out/r06ae; mkdir -p $O
bash tools/frame_ab.sh r06ae/ab "config3_4k_4096lights" one one_q8:GPU_MAX_HW_QUEUES=8 alt_q8:GRANITE_ALTERNATE_FRONT=1,GPU_MAX_HW_QUEUES=8 series_q8:GRANITE_ALTERNATE_FRONT=2,GPU_MAX_HW_QUEUES=8 alt:GRANITE_ALTERNATE_FRONT=1 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
bash tools/frame_ab.sh r06ae/ab4 "config4_4k_smaa_taa" one one_q8:GPU_MAX_HW_QUEUES=8 -- --steps 100 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab4.txt
