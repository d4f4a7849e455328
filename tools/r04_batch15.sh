O=gpurun_out/r04p; mkdir -p $O
for lib in lib lib_prio lib lib_prio; do
  GRANITE_LIB_DIR=$lib timeout 300 python bench.py --workload config4_4k_smaa_taa --no-cpu-baseline > $O/bench_config4_$lib.json 2>/dev/null; python tools/bench_brief.py $O/bench_config4_$lib.json | sed "s/^/$lib config4 /"
  GRANITE_LIB_DIR=$lib timeout 300 python bench.py --no-cpu-baseline > $O/bench_config3_$lib.json 2>/dev/null; python tools/bench_brief.py $O/bench_config3_$lib.json | sed "s/^/$lib config3 /"
done
