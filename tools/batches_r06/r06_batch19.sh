#!/bin/bash
# Round 6: the frame with this round's lighting kernel against round 5's (lib_r5light = the same library with lighting.o from commit a2c263f), one box.
O=gpurun_out/r06t; mkdir -p $O
bash tools/frame_ab.sh r06t/ab "config3_4k_4096lights" r6 r5:GRANITE_LIB_DIR=lib_r5light -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
bash tools/frame_ab.sh r06t/ab20 "config3_4k_4096lights" r6 r5:GRANITE_LIB_DIR=lib_r5light -- --steps 20 --warmup 5 --sustain-seconds 1 2>&1 | tee $O/ab20.txt
bash tools/frame_ab.sh r06t/ab20b "config3_4k_4096lights" r6 r5:GRANITE_LIB_DIR=lib_r5light -- --steps 20 --warmup 5 --sustain-seconds 1 2>&1 | tee $O/ab20b.txt
