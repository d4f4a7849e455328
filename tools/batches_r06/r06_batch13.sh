#!/bin/bash
# Round 6: one-launch bloom pass up to 640 x 384 + four-row tonemap workgroups for tiny frames: tests, config 1 / 2 lines, config 3 check.
O=gpurun_out/r06n; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_post.py tests/test_gpu_app.py tests/test_gpu_golden.py tests/test_gpu_graph_random.py tests/test_gpu_headless.py tests/test_gpu_packed_hdr.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest.txt
bash tools/frame_ab.sh r06n/ab "config1_256_post_only config2_1080p_256lights" whole parts:GR_NO_PYRAMID_FUSION=1 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
bash tools/frame_ab.sh r06n/ab20 "config1_256_post_only" whole parts:GR_NO_PYRAMID_FUSION=1 -- --steps 20 --warmup 5 --sustain-seconds 1 2>&1 | tee $O/ab20.txt
python - <<'PY' | tee $O/kernels.txt
import json,glob
for p in sorted(glob.glob("gpurun_out/r06n/ab*/*.json")):
    d=json.loads(open(p).read().strip().splitlines()[-1])
    print(p.split("r06n/")[1], "%.4f"%d["ms_per_step"], "host %.4f"%d["host_busy_ms_per_step"], {k:round(v["avg_us"],1) for k,v in d["kernels_warmup"].items()})
PY
