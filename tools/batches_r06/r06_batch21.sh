#!/bin/bash
# Round 6, at HEAD after pmc_traffic.json was regenerated: the default line and the driver's line with roofline.traffic quoted.
O=gpurun_out/r06z; mkdir -p $O
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_brief.py $O/bench_default.json
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_driver_line.$i.json; done
python -c "
import json
d=json.loads(open('$O/bench_driver_line.1.json').read().strip().splitlines()[-1]); r=d['roofline']
print({k:r.get(k) for k in ('bound','achieved','peak','unit','frac','traffic','avg_launch_us')}); print(r.get('valu_issue',{}).get('one_per_quad_cycle_us'), d.get('parity_checked'), d['cpu_baseline'])"
