O=gpurun_out/r04k; mkdir -p $O
for v in 3 2 3 2; do
  export GRANITE_HOST_LEAD_FRAMES=$v
  timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > $O/bench200_lead$v.json 2>/dev/null; python tools/bench_brief.py $O/bench200_lead$v.json | sed "s/^/lead $v 200 steps /"
  for i in 1 2; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_lead$v.json 2>/dev/null; python tools/bench_brief.py $O/bench_lead$v.json | sed "s/^/lead $v /"; done
done
for v in 3 2; do export GRANITE_HOST_LEAD_FRAMES=$v; for wl in config1_256_post_only config2_1080p_256lights config4_4k_smaa_taa; do timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/bench_${wl}_lead$v.json 2>/dev/null; python tools/bench_brief.py $O/bench_${wl}_lead$v.json | sed "s/^/lead $v $wl /"; done; done
