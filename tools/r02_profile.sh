#!/bin/bash
# Round-2 evidence: rocprofv3 kernel stats of the bench command, PMC passes, the bench line itself.
O=gpurun_out/r02prof; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py > $O/bench.json 2>$O/bench.err; python tools/bench_brief.py $O/bench.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>$GRAFT_REPO_ROOT/$O/rocprof.err )
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
bash tools/pmc_passes.sh r02prof/pmc --sustain-seconds 0 > $O/pmc.log 2>&1; tail -3 $O/pmc.log
python tools/pmc_to_traffic.py $O/pmc/summary.json $O/pmc_traffic.json 2
