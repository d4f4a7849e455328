#!/bin/bash
O=gpurun_out/r06s; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_post.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
bash tools/frame_ab.sh r06s/ab "config3_4k_4096lights" up256 up1024:GR_UP_ALL_THREADS=1024 alt256:GRANITE_ALTERNATE_FRONT=1 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
bash tools/frame_ab.sh r06s/ab4 "config4_4k_smaa_taa config2_1080p_256lights" up256 up1024:GR_UP_ALL_THREADS=1024 -- --steps 100 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab4.txt
python - <<'PY' | tee $O/kernels.txt
import json,glob
for p in sorted(glob.glob("gpurun_out/r06s/ab/*.json")):
    d=json.loads(open(p).read().strip().splitlines()[-1])
    print(p.split("lights_")[1], "%.4f"%d["ms_per_step"], {k:round(v["avg_us"],1) for k,v in d["kernels_warmup"].items()})
PY
