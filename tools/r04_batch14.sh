O=gpurun_out/r04o; mkdir -p $O
GRANITE_SELF_FED_HISTORY_ON_BACK=1 GR_TIMING_DUMP=$O/timeline_config4_back.txt timeout 200 python tools/gpu_timeline.py config4 > /dev/null 2>&1
GR_TIMING_DUMP=$O/timeline_config4_front.txt timeout 200 python tools/gpu_timeline.py config4 > /dev/null 2>&1
tail -45 $O/timeline_config4_back.txt
