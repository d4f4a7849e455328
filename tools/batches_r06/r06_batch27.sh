#!/bin/bash
# Round 6: kernel-trace timelines of one steady frame at HEAD: config 3, config 4 (tail stream), config 1 (two launches).
O=$PWD/gpurun_out/r06ab; mkdir -p $O; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for wl in config3_4k_4096lights config4_4k_smaa_taa config1_256_post_only; do
  timeout 300 rocprofv3 --kernel-trace -d $O/trace_$wl -o t --output-format csv -- python $ROOT/bench.py --workload $wl --steps 60 --warmup 10 --no-cpu-baseline --sustain-seconds 0 > $O/bench_$wl.json 2>/dev/null
  marker=k_lighting; [ $wl = config1_256_post_only ] && marker=k_bloom_pyramid
  python $ROOT/tools/trace_timeline.py $O/trace_$wl $marker > $O/timeline_$wl.txt 2>&1
  rm -rf $O/trace_$wl
  echo "== $wl"; cat $O/timeline_$wl.txt
done
