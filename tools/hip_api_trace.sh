#!/bin/bash
# Host-side HIP API calls per frame (counts and time): rocprofv3 --hip-trace --stats over a short bench run.
O=$GRAFT_REPO_ROOT/gpurun_out/hiptrace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --stats -d $O/out -o api --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > $O/bench.json 2> $O/err.txt
f=$(find $O/out -name "*hip_api_stats.csv" | head -1); head -25 "$f" | cut -c1-150
