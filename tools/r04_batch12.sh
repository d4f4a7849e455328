O=gpurun_out/r04m; mkdir -p $O
for v in 0 1 0 1; do
  export HIP_FORCE_DEV_KERNARG=$v
  for wl in config3_4k_4096lights config4_4k_smaa_taa config2_1080p_256lights config1_256_post_only; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/bench_${wl}_$v.json 2>/dev/null; python tools/bench_brief.py $O/bench_${wl}_$v.json | sed "s/^/devkernarg=$v $wl /"
  done
done
