#!/bin/bash
# Round 6: the whole bloom pass of a small frame as one launch (gr_bloom_pyramid): configs 1 / 2 A/B (and 720p / 540p through the headless runner's sizes).
O=gpurun_out/r06m; mkdir -p $O
bash tools/frame_ab.sh r06m/ab "config1_256_post_only config2_1080p_256lights" whole parts:GR_NO_PYRAMID_FUSION=1 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
python - <<'PY' | tee $O/kernels.txt
import json,glob
for p in sorted(glob.glob("gpurun_out/r06m/ab/*.json")):
    d=json.loads(open(p).read().strip().splitlines()[-1])
    print(p.split("bench_")[1], "%.4f"%d["ms_per_step"], {k:round(v["avg_us"],1) for k,v in d["kernels_warmup"].items()})
PY
