#!/bin/bash
# Round 6, final kernel sources: PMC passes -> pmc_traffic.json, kernel stats under rocprofv3, the default line and the driver's line with roofline.traffic quoted.
TAG=r06final; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 bash tools/pmc_passes.sh pmc_$TAG > $O/pmc_passes.log 2>&1; python tools/pmc_to_traffic.py gpurun_out/pmc_$TAG/summary.json $O/pmc_traffic.json 6 | tail -2
cp gpurun_out/pmc_$TAG/summary.txt $O/pmc_counters_per_kernel.txt; rm -rf gpurun_out/pmc_$TAG
cp $O/pmc_traffic.json profiles/pmc_traffic.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kstats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null)
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; rm -rf $O/kstats; head -3 $O/bench_kernel_stats.csv | cut -c1-140
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_brief.py $O/bench_default.json
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_driver_line.$i.json; done
for sc in depth_split hot_spot; do timeout 300 python bench.py --scene $sc --steps 100 --warmup 10 --sustain-seconds 1 --no-cpu-baseline > $O/bench_scene_$sc.json 2>/dev/null; python tools/bench_brief.py $O/bench_scene_$sc.json; done
python -c "
import json
d=json.loads(open('$O/bench_driver_line.1.json').read().strip().splitlines()[-1]); r=d['roofline']
print({k:r.get(k) for k in ('bound','achieved','frac','traffic','avg_launch_us')})"
