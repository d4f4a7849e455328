#!/bin/bash
# Round 3, ninth GPU batch: up-tail as 256-thread workgroups (parity + in-frame time), and where the fixed cost of a short timed
# region goes (K = 20 / 40 / 80, active-wait timeout of the runtime's host waits).
O=gpurun_out/r03i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_post.py tests/test_gpu_app.py tests/test_gpu_golden.py -q -m gpu 2>&1 | tail -3
brief() { python - "$1" "$2" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
r=j['roofline']
s=j.get('sustained') or {}
print(sys.argv[2], 'K', j['steps'], 'ms/step %.4f' % j['ms_per_step'], 'total_ms %.3f' % (j['ms_per_step']*j['steps']), 'sustained %.4f' % s.get('ms_per_step',0), 'light_us %.1f' % (r.get('avg_launch_us') or 0), 'host %.3f' % j.get('host_busy_ms_per_step',0))
PY
}
for k in 20 40 80; do for i in 1 2; do
  timeout 200 python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline --sustain-seconds 0.3 > $O/k$k.$i.json 2>/dev/null; brief $O/k$k.$i.json plain
done; done
for k in 20 80; do for i in 1 2; do
  ROC_ACTIVE_WAIT_TIMEOUT=5000 timeout 200 python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline --sustain-seconds 0.3 > $O/spin_k$k.$i.json 2>/dev/null; brief $O/spin_k$k.$i.json active_wait
done; done
for wl in config2_1080p_256lights config3_4k_4096lights; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2>/dev/null; python tools/bench_brief.py $O/bench_$wl.json
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kstats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null)
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; rm -rf $O/kstats; head -12 $O/bench_kernel_stats.csv | cut -c1-150
