#!/bin/bash
O=gpurun_out/r06r; mkdir -p $O
bash tools/frame_ab.sh r06r/ab "config3_4k_4096lights" pad13 pad8:GR_LIGHTING_PAD_KIB=8 pad4:GR_LIGHTING_PAD_KIB=4 pad2:GR_LIGHTING_PAD_KIB=2 alt_pad4:GRANITE_ALTERNATE_FRONT=1,GR_LIGHTING_PAD_KIB=4 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
python - <<'PY' | tee $O/kernels.txt
import json,glob
for p in sorted(glob.glob("gpurun_out/r06r/ab/*.json")):
    d=json.loads(open(p).read().strip().splitlines()[-1])
    print(p.split("lights_")[1], "%.4f"%d["ms_per_step"], {k:round(v["avg_us"],1) for k,v in d["kernels_warmup"].items()})
PY
