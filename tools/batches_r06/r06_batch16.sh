#!/bin/bash
O=gpurun_out/r06q; mkdir -p $O
bash tools/frame_ab.sh r06q/ab "config3_4k_4096lights" one alt5:GRANITE_ALTERNATE_FRONT=1 alt4:GRANITE_ALTERNATE_FRONT=1,GR_LIGHTING_WGS_PER_CU=4 alt3:GRANITE_ALTERNATE_FRONT=1,GR_LIGHTING_WGS_PER_CU=3 one4:GR_LIGHTING_WGS_PER_CU=4 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
python - <<'PY' | tee $O/kernels.txt
import json,glob
for p in sorted(glob.glob("gpurun_out/r06q/ab/*.1.json")):
    d=json.loads(open(p).read().strip().splitlines()[-1])
    print(p.split("lights_")[1], "%.4f"%d["ms_per_step"], {k:round(v["avg_us"],1) for k,v in d["kernels_warmup"].items()})
PY
