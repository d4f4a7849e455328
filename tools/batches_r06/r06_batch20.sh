#!/bin/bash
O=gpurun_out/r06u; mkdir -p $O
for rep in 1 2; do
bash tools/frame_ab.sh r06u/ab$rep "config3_4k_4096lights" r6 r5:GRANITE_LIB_DIR=lib_r5light r6narrow:GR_LIGHTING_NARROW_ONLY=1 -- --steps 200 --warmup 20 --sustain-seconds 2 2>&1
done | tee $O/ab.txt
python - <<'PY' | tee $O/sustained.txt
import json,glob
for p in sorted(glob.glob("gpurun_out/r06u/ab*/*.json")):
    d=json.loads(open(p).read().strip().splitlines()[-1])
    print(p.split("r06u/")[1], "timed %.4f sustained %.4f"%(d["ms_per_step"], d["sustained"]["ms_per_step"]))
PY
