#!/bin/bash
O=gpurun_out/r03q; mkdir -p $O
brief() { python - "$1" "$2" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
s=j.get('sustained') or {}
print(sys.argv[2], 'K', j['steps'], 'ms/step %.4f' % j['ms_per_step'], 'sustained %.4f' % s.get('ms_per_step',0), 'host %.3f' % j['host_busy_ms_per_step'], 'settle', j['clock_settle']['frames'])
PY
}
for i in 1 2 3; do
  timeout 300 python bench.py --workload config3_4k_4096lights_b10g11r11 --no-cpu-baseline > $O/packed.$i.json 2>/dev/null; brief $O/packed.$i.json packed
done
GRANITE_BENCH_SETTLE_MS=0 timeout 300 python bench.py --workload config3_4k_4096lights_b10g11r11 --no-cpu-baseline > $O/packed_nosettle.json 2>/dev/null; brief $O/packed_nosettle.json packed_nosettle
for i in 1 2; do
  timeout 300 python bench.py --workload config4_4k_smaa_taa --no-cpu-baseline > $O/c4.$i.json 2>/dev/null; brief $O/c4.$i.json config4
done
GRANITE_BENCH_SETTLE_MS=0 timeout 300 python bench.py --workload config4_4k_smaa_taa --no-cpu-baseline > $O/c4_nosettle.json 2>/dev/null; brief $O/c4_nosettle.json config4_nosettle
timeout 300 python bench.py --no-cpu-baseline > $O/def.json 2>/dev/null; brief $O/def.json default
