#!/bin/bash
# Round 4, fourth GPU batch: whole -m gpu suite at HEAD, bench default + driver line + small configs, 2-rank functional run with the new
# rccl fields and the config5 sub-record, TAA ulp histogram at 4K, FETCH/WRITE passes over the AA kernels.
O=gpurun_out/r04d; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt | cut -c1-400
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_brief.py $O/bench_default.json; tail -3 $O/bench_default.err | cut -c1-300
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_line.json 2>/dev/null; python tools/bench_brief.py $O/bench_driver_line.json
for wl in config1_256_post_only config2_1080p_256lights config4_4k_smaa_taa; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2>/dev/null; python tools/bench_brief.py $O/bench_$wl.json
done
timeout 400 bash tools/multirank_one_gpu.sh 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/multirank_2.json 2> $O/multirank_2.err; python - $O/multirank_2.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('2 ranks: rccl', json.dumps(j.get('rccl'))[:900]); print('config5', j.get('config5_8k'))
except Exception as e: print('multirank parse failed', e)
PY
tail -3 $O/multirank_2.err | cut -c1-300
timeout 600 python tools/taa_ulp_hist.py > $O/taa_ulp_hist.json 2> $O/taa_ulp_hist.err; cut -c1-1500 $O/taa_ulp_hist.json; tail -2 $O/taa_ulp_hist.err
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  AA_TIME_REPS=2 AA_TIME_INPUT=card timeout 400 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/$O/aa_$c -o aa --output-format csv -- python $GRAFT_REPO_ROOT/tools/aa_time.py 3840 2160 > $GRAFT_REPO_ROOT/$O/aa_$c.log 2>&1; echo "aa $c rc=$?"
done
cd $GRAFT_REPO_ROOT; python tools/pmc_summary.py $O > $O/aa_traffic_summary.txt 2>&1; grep -E "kernel|fxaa|smaa|taa" $O/aa_traffic_summary.txt | cut -c1-200 | head -30
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; du -sh $O
