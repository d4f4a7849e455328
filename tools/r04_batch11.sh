O=gpurun_out/r04l; mkdir -p $O
for v in ring single split ring single split; do
  unset GRANITE_HANDOVER_SINGLE GRANITE_RUN_SPLIT_AFTER; [ $v != ring ] && export GRANITE_RUN_SPLIT_AFTER=taa-resolve; [ $v = single ] && export GRANITE_HANDOVER_SINGLE=HDR-main
  timeout 300 python bench.py --workload config4_4k_smaa_taa --no-cpu-baseline > $O/bench_config4_$v.json 2>/dev/null; python tools/bench_brief.py $O/bench_config4_$v.json | sed "s/^/$v config4 /"
done
