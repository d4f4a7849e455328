O=gpurun_out/r05y; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kstats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null)
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; rm -rf $O/kstats; head -4 $O/bench_kernel_stats.csv | cut -c1-150
python -c "
import json; j=json.load(open('$O/bench_under_rocprof.json')); print('bench under rocprof', j['ms_per_step'], j['roofline']['avg_launch_us'], j['roofline']['launches'])"
bash tools/frame_ab.sh r05_c4res "config4_4k_smaa_taa" wg4:GR_LIGHTING_WGS_PER_CU=4 wg3:GR_LIGHTING_WGS_PER_CU=3 wg2:GR_LIGHTING_WGS_PER_CU=2 -- --steps 100 --warmup 10 --sustain-seconds 1
