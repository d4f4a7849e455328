O=gpurun_out/r04n; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_app.py tests/test_gpu_aa.py tests/test_gpu_graph_random.py tests/test_gpu_strips.py tests/test_gpu_multiprocess.py tests/test_gpu_fullsize.py tests/test_gpu_headless.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6 | cut -c1-600
for v in front back front back; do
  unset GRANITE_SELF_FED_HISTORY_ON_BACK; [ $v = back ] && export GRANITE_SELF_FED_HISTORY_ON_BACK=1
  timeout 300 python bench.py --workload config4_4k_smaa_taa --no-cpu-baseline > $O/bench_config4_$v.json 2>/dev/null; python tools/bench_brief.py $O/bench_config4_$v.json | sed "s/^/taa on $v: config4 /"
done
